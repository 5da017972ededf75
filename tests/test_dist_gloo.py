"""world_size-2 (and 4) CPU test of the chunk-sharded commit orchestration
(<package>/parallel.py): gloo backend, the per-rank compute supplied by the oracle, the collective
pattern (one digest exchange + one all-gather of subtree roots) exactly the one the GPU path
uses.  The assembled tree must equal the single-process commit_standard bit for bit."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """CPU stand-in for HipOps (tests only): same three operations, computed by the oracle."""
    device = torch.device("cpu")

    def __init__(self, orc, lib):
        self.orc, self.lib = orc, lib

    def inner_digests(self, local_chunks, plan):
        out = np.zeros((len(local_chunks), plan.M, 32), np.uint8)
        for li, msg in enumerate(local_chunks):
            t = self.orc.compute_tensorcode(msg, plan.trs, 1)            # (2trs, cols, 2) row-major
            for j in range(plan.trs // 2):
                blk = np.ascontiguousarray(t[4 * j:4 * j + 4].transpose(1, 0, 2)).view(np.uint8).reshape(plan.cols, 64)
                out[li, j * plan.cols:(j + 1) * plan.cols] = self.orc.blake3_64(blk)
        return torch.from_numpy(out)

    def chain_and_tree(self, mine, plan):
        d = mine.numpy()
        leaves = np.zeros((plan.m_local, 32), np.uint8)
        for i in range(plan.K):
            leaves = self.orc.blake3_64(np.concatenate([d[i], leaves], axis=1))
        return torch.from_numpy(self.orc.create_tree_blake(leaves))

    def tree_top(self, roots):
        from __graft_entry__ import load_package
        return load_package().parallel_tree_top(self.lib, roots)

    def after_collective(self):
        pass


def _worker(rank, world, port, N, K, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from __graft_entry__ import load_package
    mod = load_package()
    orc = pyoracle.Oracle()
    trs = N // (K << 11)
    orc.rng_reset(); poly = orc.generate_randomness(N); orc.expander_init_store(trs)     # same inputs on every rank
    plan = mod.parallel.ShardPlan(N, K, trs, world)
    M = plan.M
    local = [poly[i * M:(i + 1) * M] for i in plan.chunks_of(rank)]
    res = mod.parallel.sharded_commit(OracleOps(orc, mod.load_library()), dist, plan, rank, local)
    q.put((rank, res["subtree"].numpy(), res["top"], res["root"]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_commit_matches_single_process(oracle, world):
    from __graft_entry__ import load_package, build_hip
    build_hip()
    mod = load_package()
    N, K = 1 << 18, 32
    trs = N // (K << 11)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    import queue as _q
    import time as _t
    deadline = _t.time() + 240
    while len(got) < world:
        try:
            got.append(q.get(timeout=2))
        except _q.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: exit codes %s" % [p.exitcode for p in procs]
            assert _t.time() < deadline, "timeout waiting for ranks"
    got.sort(key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    want, _ = oracle.commit_standard(poly, K, trs, 1)
    plan = mod.parallel.ShardPlan(N, K, trs, world)
    levels = mod.parallel.assemble_levels(plan, [g[1] for g in got], got[0][2])
    assert np.array_equal(levels, want)
    for g in got:                                   # every rank ends with the same top levels / root
        assert np.array_equal(g[2], got[0][2]) and np.array_equal(g[3], want[-1])


def test_shard_plan():
    from __graft_entry__ import load_package
    P = load_package().parallel.ShardPlan
    p = P(1 << 28, 32, 4096, 8)
    assert p.M == 1 << 23 and p.cols == 4096 and p.m_local == 1 << 20
    assert p.chunks_of(3) == [3, 11, 19, 27] and p.owner(19) == (3, 2)
    assert p.leaf_range(7) == (7 << 20, 8 << 20)
    with pytest.raises(AssertionError):
        P(1 << 20, 32, 16, 3)
