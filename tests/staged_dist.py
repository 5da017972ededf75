"""tests/staged_dist.py -- TEST INFRASTRUCTURE.  A `torch.distributed`-shaped adapter that stages CUDA tensors through host memory, so that
parallel.py's orchestration can run as real processes with the real per-rank GPU operations (HipOps) over the gloo backend -- several ranks on
ONE GPU, where RCCL refuses to run (it wants one device per rank).  The calls parallel.py's two commits (relay, all-to-all) and the sharded open make."""
import torch
import torch.distributed as dist


class _RecvWork:
    def __init__(self, work, host, dev):
        self.work, self.host, self.dev = work, host, dev

    def wait(self):
        self.work.wait()
        self.dev.copy_(self.host)
        torch.cuda.synchronize()


class _SendWork:
    def __init__(self, work, host):
        self.work, self.host = work, host          # the host copy lives until the send has completed

    def wait(self):
        self.work.wait()


class StagedDist:
    ReduceOp = dist.ReduceOp

    @staticmethod
    def _host(t):
        torch.cuda.synchronize()
        return t.detach().to("cpu").contiguous()

    def isend(self, t, dst):
        h = self._host(t)
        return _SendWork(dist.isend(h, dst), h)

    def irecv(self, t, src):
        h = torch.empty(t.shape, dtype=t.dtype)
        return _RecvWork(dist.irecv(h, src), h, t)

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def batch_isend_irecv(self, ops):
        # sends first (each is already on its way when the receives are posted), as torch's own batch does not order them either
        works = [o.op(o.tensor, o.peer) for o in ops if o.op == self.isend]
        works += [o.op(o.tensor, o.peer) for o in ops if o.op == self.irecv]
        return works

    def broadcast(self, t, src):
        h = self._host(t)
        dist.broadcast(h, src)
        t.copy_(h); torch.cuda.synchronize()

    def all_reduce(self, t, op=dist.ReduceOp.SUM):
        h = self._host(t)
        dist.all_reduce(h, op=op)
        t.copy_(h); torch.cuda.synchronize()

    def all_gather(self, outs, t):
        h = self._host(t)
        hs = [torch.empty_like(h) for _ in outs]
        dist.all_gather(hs, h)
        for o, x in zip(outs, hs):
            o.copy_(x)
        torch.cuda.synchronize()

    def barrier(self):
        dist.barrier()
