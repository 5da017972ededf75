"""Chunk-sharded Our_PC commit across the GPUs of one node.

The reference is single-process; this is new work, not a port.  One process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

  1. rank g computes the tensor codes of chunks i = g, g+G, g+2G, ... (independent, no exchange)
     and their inner leaf digests H(t[4j..4j+3][c])      -- half of all BLAKE3 compressions;
  2. ONE exchange: rank g owns the leaf range [g*M/G, (g+1)*M/G) and receives every other rank's
     inner digests for that range (all-to-all of 32-byte records; at N = 2^28 that is 8 GiB in
     total, ~0.9 GiB out of each GPU, spread over the 7 xGMI links);
  3. rank g runs the Merkle-Damgard chain leaf = H(inner_i | leaf) over i = 0..K-1 for its range
     and builds the subtree over it;
  4. ONE all-gather of the G subtree roots (32 B each); every rank computes the top log2 G levels.

The orchestration below is backend-agnostic: `ops` supplies the per-rank compute (the HIP library
on the GPU, see HipOps; the tests pass a CPU implementation), `dist` is torch.distributed.  The
exchange uses isend/irecv pairs, which both RCCL and gloo implement.
"""
import numpy as np


class ShardPlan:
    def __init__(self, N, K, trs, world):
        assert N % K == 0 and K % world == 0, "world size must divide the number of chunks K"
        self.N, self.K, self.trs, self.world = N, K, trs, world
        self.M = N // K
        assert self.M % world == 0
        self.cols = 2 * self.M // trs
        self.m_local = self.M // world

    def chunks_of(self, rank):
        return list(range(rank, self.K, self.world))

    def owner(self, chunk):
        return chunk % self.world, chunk // self.world        # (rank, local index)

    def leaf_range(self, rank):
        return rank * self.m_local, (rank + 1) * self.m_local


def sharded_commit(ops, dist, plan, rank, local_chunks):
    """local_chunks: this rank's chunk messages in `ops`' native form.  Returns a dict with this
    rank's leaf range, its subtree levels (flat: m_local leaves ... subtree root), the top levels
    (flat: G subtree roots ... root) and the root."""
    import torch
    G = plan.world
    own = plan.chunks_of(rank)
    # 1. local tensor codes + inner digests: uint8 tensor [n_own, M, 32] on ops.device
    digests = ops.inner_digests(local_chunks, plan)
    assert tuple(digests.shape) == (len(own), plan.M, 32)
    # 2. the one exchange
    lo, hi = plan.leaf_range(rank)
    recv = {h: torch.empty((len(plan.chunks_of(h)), plan.m_local, 32), dtype=torch.uint8, device=digests.device) for h in range(G) if h != rank}
    reqs = []
    for h in range(G):
        if h == rank:
            continue
        l2, h2 = plan.leaf_range(h)
        reqs.append(dist.P2POp(dist.isend, digests[:, l2:h2].contiguous(), h))
        reqs.append(dist.P2POp(dist.irecv, recv[h], h))
    if reqs:
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
    ops.after_collective()        # RCCL: wait() only orders torch's stream; the library has its own
    # assemble the K digests of my leaf range in global chunk order
    parts = []
    for i in range(plan.K):
        owner, li = plan.owner(i)
        parts.append(digests[li, lo:hi] if owner == rank else recv[owner][li])
    mine = torch.stack(parts).contiguous()                     # [K, m_local, 32]
    # 3. chain + subtree
    subtree = ops.chain_and_tree(mine, plan)                   # flat uint8 [2*m_local-1, 32]
    # 4. all-gather of subtree roots, top of the tree everywhere
    my_root = subtree[-1:].contiguous()
    roots = [torch.empty_like(my_root) for _ in range(G)]
    if G > 1:
        dist.all_gather(roots, my_root)
    else:
        roots = [my_root]
    top = ops.tree_top(torch.cat(roots).cpu().numpy())         # flat [2G-1, 32] (numpy, host)
    return dict(leaf_range=(lo, hi), subtree=subtree, top=top, root=top[-1])


def assemble_levels(plan, subtrees, top):
    """Rebuild the reference's flat level list (M leaves ... root) from every rank's subtree
    (host arrays, rank order) and the top levels -- used by tests and by rank 0 when a caller
    wants MT_hashes as the reference returns it."""
    G, m = plan.world, plan.m_local
    out, off, sz = [], 0, m
    while sz >= 1:
        out.append(np.concatenate([np.asarray(s)[off:off + sz] for s in subtrees]))
        off += sz; sz //= 2
    # top[0:G] are the subtree roots = the level just emitted; append the levels above it
    t = np.asarray(top)
    out.append(t[G:])
    return np.concatenate(out)


class HipOps:
    """Per-rank compute on the GPU through the C ABI (torch only provides the device buffers the
    collectives need)."""

    def __init__(self, hb, torch_device):
        self.hb = hb
        self.device = torch_device

    def inner_digests(self, local_chunks, plan):
        """local_chunks: (device_ptr, n_own): n_own messages of M F each, contiguous, resident."""
        import torch
        from ctypes import c_void_p
        ptr, n_own = local_chunks
        hb = self.hb
        self._tensor = hb.alloc(16 * 4 * plan.M * n_own)       # retained: the commitment's tensor shard
        hb._chk(hb.lib.hobbit_tensorcode_chunks(hb.ctx, ptr, plan.M, n_own, plan.trs, 1, self._tensor.ptr))
        out = torch.empty((n_own, plan.M, 32), dtype=torch.uint8, device=self.device)
        hb._chk(hb.lib.hobbit_inner_digests(hb.ctx, self._tensor.ptr, plan.M, n_own, plan.trs, out.data_ptr()))
        hb.sync()
        return out

    def chain_and_tree(self, mine, plan):
        import torch
        hb = self.hb
        m = plan.m_local
        levels = torch.zeros((2 * m, 32), dtype=torch.uint8, device=self.device)
        hb._chk(hb.lib.hobbit_chain_digests(hb.ctx, mine.data_ptr(), 32 * m, plan.K, m, levels.data_ptr()))
        hb._chk(hb.lib.hobbit_merkle_levels(hb.ctx, levels.data_ptr(), m, 1))
        hb.sync()
        return levels[:2 * m - 1]

    def tree_top(self, roots):
        return tree_top_host(self.hb.lib, roots)

    def after_collective(self):
        import torch
        torch.cuda.synchronize(self.device)


def tree_top_host(lib, roots):
    """levels above the G subtree roots, with the reference's left|left parent rule
    (src/merkle_tree.cpp:275-280), on the host (a handful of hashes)."""
    import ctypes
    lv = [np.ascontiguousarray(roots, np.uint8).reshape(-1, 32)]
    while lv[-1].shape[0] > 1:
        prev = lv[-1]
        blk = np.ascontiguousarray(np.concatenate([prev[0::2], prev[0::2]], axis=1))
        out = np.zeros((blk.shape[0], 32), np.uint8)
        lib.hobbit_blake3_64_host(blk.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), blk.shape[0])
        lv.append(out)
    return np.concatenate(lv)
