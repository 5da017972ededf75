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

That all-to-all moves K*M*32 bytes in total (8 GiB at 2^28) whatever the world size: fine over the seven links of an 8-GPU node
(128 MiB per link), but 2 GiB over ONE link at world size 2 -- longer than the whole single-GPU commit.  `sharded_commit_relay` is
the default for Our_PC instead (the all-to-all stays for the streaming Elastic commit, whose chain state only exists after a whole
group pass, and as `bench.py --exchange alltoall`):

  1'. rank g owns the CONTIGUOUS chunks [g*K/G, (g+1)*K/G) and computes their tensor codes;
  2'. the leaf chain is relayed: rank 0 chains its chunks from the zero state, hands the 32-byte running state of every leaf to rank 1,
      which chains its chunks on top, ... -- M*32 bytes (256 MiB at 2^28) per hop, each hop on its own link, cut into blocks of leaves so
      that rank g works on block b while rank g-1 works on block b+1.  Nothing is hashed twice and the tensor shard is read once
      (the all-to-all path reads it a second time for the inner digests);
  3'. the last rank ends up with the leaves, builds the tree (1 ms at 2^28) and serves the Merkle paths of the opening; the root is
      broadcast.

The orchestration below is backend-agnostic: `ops` supplies the per-rank compute (the HIP library
on the GPU, see HipOps; the tests pass a CPU implementation), `dist` is torch.distributed.  The
exchanges use isend/irecv, which both RCCL and gloo implement.
"""
import numpy as np


class ShardPlan:
    """contiguous=False: rank g owns chunks g, g+G, ... (all-to-all commit); True: chunks [g*K/G, (g+1)*K/G) (chain relay)"""

    def __init__(self, N, K, trs, world, contiguous=False):
        assert N % K == 0 and K % world == 0, "world size must divide the number of chunks K"
        self.N, self.K, self.trs, self.world, self.contiguous = N, K, trs, world, contiguous
        self.M = N // K
        assert self.M % world == 0
        self.cols = 2 * self.M // trs
        self.m_local = self.M // world
        self.k_local = K // world

    def chunks_of(self, rank):
        if self.contiguous:
            return list(range(rank * self.k_local, (rank + 1) * self.k_local))
        return list(range(rank, self.K, self.world))

    def owner(self, chunk):
        if self.contiguous:
            return chunk // self.k_local, chunk % self.k_local
        return chunk % self.world, chunk // self.world        # (rank, local index)

    def leaf_range(self, rank):
        return rank * self.m_local, (rank + 1) * self.m_local


def _generation(ops):
    """An ops object that keeps the commitment's tensor shard / tree in RETAINED buffers (HipOps: no allocation per commit) holds ONE live
    commitment: the next commit overwrites them.  Such an object counts its commits (`generation`); a commit result carries the count it was
    made under and sharded_open refuses a stale one instead of silently serving paths and replies of a later commitment against this root."""
    return getattr(ops, "generation", None)


def _check_live(ops, commit_res):
    g = getattr(ops, "generation", None)
    if g is not None and commit_res.get("gen") is not None and commit_res["gen"] != g:
        raise RuntimeError("sharded_open: this commitment's retained tensor shard and tree were overwritten by a later commit on the same ops object "
                           "(commitment generation %r, current %r): an ops object holds one live commitment" % (commit_res["gen"], g))


def sharded_commit(ops, dist, plan, rank, local_chunks):
    """local_chunks: this rank's chunk messages in `ops`' native form.  Returns a dict with this
    rank's leaf range, its subtree levels (flat: m_local leaves ... subtree root), the top levels
    (flat: G subtree roots ... root) and the root.

    The exchange is posted chunk by chunk: as soon as local chunk li's tensor code and inner digests exist, its leaf ranges go out to
    their owners (and the matching receives are posted) while the next chunk is being encoded; everything is received straight into
    the [K, m_local, 32] buffer the chain kernel reads -- no staging copy of the 8 GiB / G of digests."""
    import torch
    G = plan.world
    own = plan.chunks_of(rank)
    lo, hi = plan.leaf_range(rank)
    mine = ops.empty_digests(plan.K, plan.m_local)             # uint8 [K, m_local, 32] on ops.device, global chunk order
    works, keep = [], []
    for li, i in enumerate(own):
        d = ops.inner_digests_one(local_chunks, li, plan)      # uint8 [M, 32], complete (the library's stream has been drained)
        keep.append(d)                                         # alive until the sends have completed
        mine[i].copy_(d[lo:hi])
        reqs = []
        for h in range(G):
            if h == rank:
                continue
            l2, h2 = plan.leaf_range(h)
            reqs.append(dist.P2POp(dist.isend, d[l2:h2], h))
            reqs.append(dist.P2POp(dist.irecv, mine[plan.chunks_of(h)[li]], h))      # every rank owns the same number of chunks
        if reqs:
            works += dist.batch_isend_irecv(reqs)              # asynchronous: overlaps the next chunk's tensor code
    for w in works:
        w.wait()
    ops.after_collective()        # RCCL: wait() only orders torch's stream; the library has its own
    del keep
    # chain + subtree
    subtree = ops.chain_and_tree(mine, plan)                   # flat uint8 [2*m_local-1, 32]
    # all-gather of subtree roots, top of the tree everywhere
    my_root = subtree[-1:].contiguous()
    roots = [torch.empty_like(my_root) for _ in range(G)]
    if G > 1:
        dist.all_gather(roots, my_root)
        ops.after_collective()
    else:
        roots = [my_root]
    top = ops.tree_top(ops.to_host("roots", torch.cat(roots)).copy())   # flat [2G-1, 32] (numpy, host)
    return dict(leaf_range=(lo, hi), subtree=subtree, top=top, root=top[-1], gen=_generation(ops))


def relay_blocks(world):
    """Blocks the M leaf states are cut into for the relay.  A hop moves M*32 bytes over ONE link whatever the world size, so the relay takes
    about (blocks + world - 1) block transfers: the fill grows with the world size and shrinks with the block count, while every block
    costs each rank a receive, a launch and a send (~50 us of host time).  16 blocks up to two ranks, 8 per rank beyond, at most 64."""
    return max(16, min(64, 8 * world))


def sharded_commit_relay(ops, dist, plan, rank, local_chunks, blocks=None):
    """Chain-relay commit (module docstring, 1'-3').  plan must be contiguous.  Returns dict(root, owner = the last rank, levels = the
    whole tree flat [2M-1, 32] on the owner (None elsewhere)).  Leaves travel in SLOT order (slot = col * trs/2 + j, the order the
    shard is read in); the last rank writes them out in the reference's leaf order."""
    import torch
    assert plan.contiguous, "the relay needs contiguous chunk ownership (ShardPlan(..., contiguous=True))"
    G, M = plan.world, plan.M
    last = G - 1
    if blocks is None:
        blocks = relay_blocks(G)
    if G == 1:
        blocks = 1                                             # nothing to relay: one chain launch
    while blocks > 1 and M % blocks:
        blocks //= 2
    per = M // blocks
    ops.encode_local(local_chunks, plan)                       # tensor codes of this rank's chunks (no exchange)
    st_in = ops.empty_state(M, "in") if rank > 0 else None     # uint8 [M, 32], slot order
    st_out = ops.empty_state(M, "out") if rank < last else None
    levels = ops.empty_state(2 * M, "levels") if rank == last else None
    # Receives are posted ONE block ahead, not all up front: RCCL runs a communicator's point-to-point operations in the order they were
    # posted, so this rank's send of block b would otherwise queue behind the receives of every later block and the ranks downstream
    # would start only when rank 0 has finished.  Order on the wire: recv b0, recv b1, send b0, recv b2, send b1, ...
    nxt = dist.irecv(st_in[0:per], rank - 1) if rank > 0 else None
    sends = []
    for b in range(blocks):
        lo = b * per
        if rank > 0:
            ops.wait_recv(nxt)                                 # block b of the running state has arrived
            nxt = dist.irecv(st_in[lo + per:lo + 2 * per], rank - 1) if b + 1 < blocks else None
        ops.chain_block(plan, lo, per, st_in[lo:lo + per] if rank > 0 else None, st_out[lo:lo + per] if rank < last else None, levels)
        if rank < last:
            sends.append(dist.isend(st_out[lo:lo + per], rank + 1))      # overlaps this rank's next block and the next rank's chain
    for w in sends:
        w.wait()
    if sends:
        ops.after_collective()
    root = torch.zeros(32, dtype=torch.uint8, device=ops.device)
    if rank == last:
        levels = ops.tree_full(levels, M)                      # flat [2M-1, 32]
        root.copy_(levels[-1])
    if G > 1:
        dist.broadcast(root, last)
        ops.after_collective()
    return dict(root=ops.to_host("root", root).copy(), owner=last, levels=levels, gen=_generation(ops))


class ElasticPlan:
    """Elastic_PC streaming commit sharded over GPUs: a leaf hashes 4 consecutive chunks together
    (src/Elastic_PC.cpp:228-243), so the unit of work is a GROUP of 4 chunks; rank g encodes groups g, g+G, ... and owns the leaf range
    [g*4B/G, (g+1)*4B/G) of the 4B running leaves.  Same exchange as the Our_PC commit: `sharded_commit(ops, dist, ElasticPlan(...), rank,
    source)` with ops.inner_digests_one producing a group's 4B inner digests."""

    def __init__(self, N, B, opt, world):
        assert N % (4 * B) == 0, "N must be a whole number of 4-chunk groups"
        self.N, self.B, self.opt, self.world = N, B, opt, world
        self.lin, self.trs = (0, B >> 11) if opt == 1 else (1, B >> 14)
        self.K = N // (4 * B)                                  # groups: the chain length of every leaf
        assert self.K % world == 0, "world size must divide the number of 4-chunk groups"
        self.M = 4 * B                                         # leaves
        assert self.M % world == 0
        self.m_local = self.M // world

    def chunks_of(self, rank):
        return list(range(rank, self.K, self.world))           # group indices

    def owner(self, group):
        return group % self.world, group // self.world

    def leaf_range(self, rank):
        return rank * self.m_local, (rank + 1) * self.m_local


def sharded_open(ops, dist, plan, rank, commit_res, x, queries=5900):
    """The open of a chunk-sharded commitment (the reference's open_standard, src/Our_PC.cpp:604-661, on one process):
      1. rank g aggregates ITS chunks with their eq-table coefficients: partial_g = sum_{i = g (mod G)} beta[i] * chunk_i;
      2. ONE 64-bit integer all-reduce of the partials (M F each; up to 8 ranks: sums of canonical components fit 64 bits) and a local
         reduction mod p -> every rank holds the aggregate (more than 8 ranks: all-gather + field sums);
      3. every rank runs the rest of the open from the aggregate (tensor code of the aggregate, inner commitments, five sumchecks,
         both shockwave_prove / WHIR proofs): it depends on the aggregate alone, its rounds are sequential and short, so it is
         replicated, not sharded -- and because every rank draws the same libc sequence, every rank holds the same queries;
      4. replies reply[q][i] = tensor_i[row_q][col_q] are gathered where chunk i's tensor lives, ONE all-gather;
      5. the Merkle path of query q comes from the rank that owns its leaf (subtree part, ONE all-gather of padded paths) followed by
         the top log2 G levels every rank has.
    `x`: (log2 N, 2) uint64 host array; commit_res: what sharded_commit returned on this rank.  Returns the transcript (as
    Hobbit.open_standard / open_from_aggregate) with "reply" (queries, K, 2) and "paths" (queries, log2 M, 32) filled in."""
    import torch
    _check_live(ops, commit_res)
    G = plan.world
    own = plan.chunks_of(rank)
    logK = plan.K.bit_length() - 1
    beta = ops.eq_table_host(np.asarray(x)[:logK])                       # (K, 2) uint64, host
    # 1-2. aggregate
    partial = ops.aggregate_local(np.ascontiguousarray(beta[own]), plan)          # int64 tensor (M, 2) on ops.device
    if G == 1:
        aggr = partial
    elif G <= 8:
        # ONE integer all-reduce instead of an all-gather of G partials and G - 1 field sums: canonical components are < 2^61, so eight of
        # them add up inside 64 bits; shifted by -2^60 each, the SIGNED sum the collective computes cannot overflow either.  Per rank the
        # links carry 2 (G-1)/G x 128 MiB instead of (G-1) x 128 MiB (N = 2^28); the sum is brought back into the field locally.
        ops.bias_words(partial, -(1 << 60))
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)
        ops.after_collective()
        aggr = ops.fold_words(partial, G << 60)
    else:
        parts = [torch.empty_like(partial) for _ in range(G)]
        dist.all_gather(parts, partial)
        ops.after_collective()
        aggr = ops.sum_vectors(parts)
    # 3. the rest of the open, replicated.  Its challenges and queries are libc draws, so every rank must hold the same generator state:
    #    rank 0 draws one value, everybody seeds with it (whatever a rank's runtime, RCCL or loader drew from libc before is forgotten)
    if G > 1:
        import ctypes
        libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
        seed = ops.upload("seed", np.array([libc.random() if rank == 0 else 0], np.int64))      # (pinned staging both ways: see HipOps._upload)
        dist.broadcast(seed, 0)
        ops.after_collective()
        libc.srandom(ctypes.c_uint(int(ops.to_host("seed", seed)[0]) & 0xFFFFFFFF))
    res = ops.open_from_aggregate(aggr, plan, queries)
    cols = np.asarray(res["cols"], np.int64); rows = np.asarray(res["rows"], np.int64)
    if G > 1:
        # every rank must have drawn the same libc sequence inside the replicated open: a divergence (a library on one rank drawing
        # from libc, a different call history) would silently answer different queries -- compare with rank 0's and fail loudly
        q_mine = ops.upload("queries", np.stack([cols, rows]))
        q0 = q_mine.clone()
        dist.broadcast(q0, 0)
        ops.after_collective()
        if not np.array_equal(ops.to_host("queries0", q0), np.stack([cols, rows])):
            raise RuntimeError("sharded_open: rank %d drew different queries than rank 0 (the libc generator streams diverged)" % rank)
    # 4. replies
    mine = ops.gather_local(rows, cols, plan)                                        # int64 tensor (queries, n_own, 2)
    if G > 1:
        rep = [torch.empty_like(mine) for _ in range(G)]
        dist.all_gather(rep, mine)
        ops.after_collective()
    else:
        rep = [mine]
    reply = np.empty((queries, plan.K, 2), np.uint64)
    for h in range(G):
        r_h = ops.to_host("reply", rep[h]).view(np.uint64)          # (consumed before the next to_host("reply"))
        ch = list(plan.chunks_of(h))
        reply[:, ch] = r_h[:, :len(ch)]                             # one strided copy per rank (a Python loop over the K chunks cost 0.8 ms)
    res["reply"] = reply
    # 5. paths: leaf position (row/4) * cols + col (src/merkle_tree.cpp:309)
    pos = (rows // 4) * plan.cols + cols
    if "levels" in commit_res:                                                        # relay commit: the whole tree lives on one rank
        depth = plan.M.bit_length() - 1
        owner = commit_res["owner"]
        host = ops.tree_paths(commit_res["levels"], pos, plan.M) if rank == owner else None      # numpy (queries, log2 M, 32)
        if G > 1:
            pt = ops.paths_buffer(queries, depth, host)
            dist.broadcast(pt, owner)
            ops.after_collective()
            host = ops.to_host("paths", pt).copy()
        res["paths"] = host
        return res
    # all-to-all commit: owner of a leaf = pos // m_local
    depth_l = plan.m_local.bit_length() - 1
    lo, hi = plan.leaf_range(rank)
    sel = np.nonzero((pos >= lo) & (pos < hi))[0]
    local_paths = torch.zeros((queries, depth_l, 32), dtype=torch.uint8, device=commit_res["subtree"].device)
    if len(sel):
        local_paths[torch.from_numpy(sel).to(local_paths.device)] = ops.subtree_paths(commit_res["subtree"], pos[sel] - lo, plan)
    allp = [torch.empty_like(local_paths) for _ in range(G)]
    if G > 1:
        dist.all_gather(allp, local_paths)
        ops.after_collective()
    else:
        allp = [local_paths]
    top = np.asarray(commit_res["top"]).reshape(-1, 32)
    depth_t = G.bit_length() - 1
    paths = np.zeros((queries, depth_l + depth_t, 32), np.uint8)
    owner = pos // plan.m_local
    stacked = np.stack([ops.to_host("subpaths", a).copy() for a in allp])               # (G, queries, depth_l, 32)
    paths[:, :depth_l] = stacked[owner, np.arange(queries)]
    off, sz, p = 0, G, owner.copy()
    for l in range(depth_t):                                                          # siblings in the top levels
        paths[:, depth_l + l] = top[off + (p ^ 1)]
        off += sz; sz //= 2; p //= 2
    res["paths"] = paths
    return res


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
        self.generation = 0            # commits made through this object; its retained tensor shard and tree belong to the latest one only

    def empty_digests(self, K, m_local):
        import torch
        return torch.empty((K, m_local, 32), dtype=torch.uint8, device=self.device)

    def upload(self, role, host):
        return self._upload(role, np.ascontiguousarray(host))

    def _upload(self, role, host):
        """host numpy array -> retained device tensor through a retained PINNED host tensor.  torch.from_numpy(x).to(device) copies from
        pageable memory; on this runtime that now and then takes ~27 ms (a staging buffer being set up) -- measured in the open of the
        sharded bench as two outlier steps in ten (scripts/relay_phases.py)."""
        import torch
        cache = self.__dict__.setdefault("_up", {})
        ent = cache.get(role)
        if ent is None or ent[0].shape != host.shape or ent[0].dtype != torch.from_numpy(host[:0]).dtype:
            pin = torch.empty(host.shape, dtype=torch.from_numpy(host[:0]).dtype).pin_memory()
            ent = cache[role] = (pin, torch.empty(host.shape, dtype=pin.dtype, device=self.device))
        ent[0].numpy()[...] = host
        ent[1].copy_(ent[0], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return ent[1]

    def to_host(self, role, t):
        """device tensor -> numpy through a retained PINNED host tensor (one per role and shape): `t.cpu()` copies into pageable memory,
        which on this runtime stalls for 30-40 ms now and then (the same effect as in _upload; one step in ~40 of the sharded bench).
        The returned array is a view of the retained buffer: valid until the next to_host of the same role."""
        import torch
        if isinstance(t, np.ndarray):
            return t
        cache = self.__dict__.setdefault("_down", {})
        pin = cache.get(role)
        if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
            pin = cache[role] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        pin.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin.numpy()

    def inner_digests_one(self, local_chunks, li, plan):
        """local_chunks: (device_ptr, n_own): n_own messages of M F each, contiguous, resident.  Tensor code of local chunk li into the
        retained shard, its inner digests into a fresh torch buffer; returns once the library's stream has produced them (the
        collective that follows runs on RCCL's stream, which only orders itself after torch's)."""
        import torch
        ptr, n_own = local_chunks
        hb = self.hb
        if li == 0:
            self.generation += 1
        if li == 0 and (getattr(self, "_tensor", None) is None or self._tensor.ptr is None or self._tensor.nbytes != 16 * 4 * plan.M * n_own):
            self._tensor = hb.alloc(16 * 4 * plan.M * n_own)   # retained: the commitment's tensor shard (re-used by the next commit of the same shape:
                                                               # allocating and freeing 16 GiB / G per commit stalls the device queue for seconds now and then)
        t = self._tensor.ptr + 16 * 4 * plan.M * li
        hb._chk(hb.lib.hobbit_tensorcode_chunks(hb.ctx, ptr + 16 * plan.M * li, plan.M, 1, plan.trs, 1, t))
        out = torch.empty((plan.M, 32), dtype=torch.uint8, device=self.device)
        hb._chk(hb.lib.hobbit_inner_digests(hb.ctx, t, plan.M, 1, plan.trs, out.data_ptr()))
        hb.sync()
        return out

    def inner_digests(self, local_chunks, plan):
        import torch
        return torch.stack([self.inner_digests_one(local_chunks, li, plan) for li in range(local_chunks[1])])

    def chain_and_tree(self, mine, plan):
        import torch
        hb = self.hb
        m = plan.m_local
        levels = torch.empty((2 * m, 32), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)                    # (allocation only; nothing of torch's is pending on `levels`)
        hb._chk(hb.lib.hobbit_memset(hb.ctx, levels.data_ptr(), 0, 32 * m))     # the chain starts from zero leaves; zeroed on the LIBRARY's stream
        hb._chk(hb.lib.hobbit_chain_digests(hb.ctx, mine.data_ptr(), 32 * m, plan.K, m, levels.data_ptr()))
        hb._chk(hb.lib.hobbit_merkle_levels(hb.ctx, levels.data_ptr(), m, 1))
        hb.sync()
        return levels[:2 * m - 1]

    def tree_top(self, roots):
        return tree_top_host(self.hb.lib, roots)

    def after_collective(self):
        import torch
        torch.cuda.synchronize(self.device)

    # ---- chain relay
    def empty_state(self, n, role="tmp"):
        """one retained buffer per role and size: a commit per step must not allocate (a 512 MiB allocation stalls the queue now and then)"""
        import torch
        cache = self.__dict__.setdefault("_state", {})
        t = cache.get(role)
        if t is None or t.shape[0] != n:
            t = cache[role] = torch.empty((n, 32), dtype=torch.uint8, device=self.device)
        return t

    def encode_local(self, local_chunks, plan):
        """tensor codes of all local chunks into the retained shard, one library call"""
        ptr, n_own = local_chunks
        hb = self.hb
        self.generation += 1
        if getattr(self, "_tensor", None) is None or self._tensor.ptr is None or self._tensor.nbytes != 16 * 4 * plan.M * n_own:
            self._tensor = hb.alloc(16 * 4 * plan.M * n_own)
        hb._chk(hb.lib.hobbit_tensorcode_chunks(hb.ctx, ptr, plan.M, n_own, plan.trs, 1, self._tensor.ptr))
        self._n_own = n_own

    def wait_recv(self, work):
        """the received block must be visible to the LIBRARY's stream: wait() orders torch's current stream behind the transfer, and
        only that stream is drained -- a device-wide synchronize would also wait for every later block's receive"""
        import torch
        work.wait()
        torch.cuda.current_stream(self.device).synchronize()

    def chain_block(self, plan, slot_begin, slot_count, st_in, st_out, levels):
        hb = self.hb
        hb._chk(hb.lib.hobbit_leaf_chain_relay(hb.ctx, self._tensor.ptr, plan.M, self._n_own, plan.trs, 1, slot_begin, slot_count,
                                               st_in.data_ptr() if st_in is not None else None, st_out.data_ptr() if st_out is not None else None,
                                               levels.data_ptr() if levels is not None else None))
        hb.sync()                                              # the block is complete before its send is posted

    def tree_full(self, levels, M):
        hb = self.hb
        hb._chk(hb.lib.hobbit_merkle_levels(hb.ctx, levels.data_ptr(), M, 1))
        hb.sync()
        return levels[:2 * M - 1]

    def tree_paths(self, levels, pos, M):
        import torch
        hb = self.hb
        p = np.ascontiguousarray(pos, np.uint64)
        depth = M.bit_length() - 1
        out = np.zeros((len(p), depth, 32), np.uint8)
        hb._chk(hb.lib.hobbit_merkle_paths(hb.ctx, levels.data_ptr(), M, p.ctypes.data, len(p), out.ctypes.data))
        return out                                             # host array: only a multi-rank open needs it on the device (to broadcast it)

    def paths_buffer(self, queries, depth, host=None):
        """retained device tensor for the broadcast of the paths; filled from `host` on the rank that computed them"""
        import torch
        if host is not None:
            return self._upload("paths", host)
        cache = self.__dict__.setdefault("_up", {})
        ent = cache.get("paths_rx")
        if ent is None or ent.shape != (queries, depth, 32):
            ent = cache["paths_rx"] = torch.empty((queries, depth, 32), dtype=torch.uint8, device=self.device)
        return ent

    # ---- open
    def set_local_chunks(self, local_chunks):
        self._chunks = local_chunks                                  # (device_ptr, n_own)

    def eq_table_host(self, r):
        return self.hb.precompute_beta(r)

    def aggregate_local(self, coeffs, plan):
        import torch
        hb = self.hb
        ptr, n_own = self._chunks
        out = torch.empty((plan.M, 2), dtype=torch.int64, device=self.device)
        hb._chk(hb.lib.hobbit_aggregate(hb.ctx, ptr, plan.M * n_own, coeffs.ctypes.data, n_own, out.data_ptr()))
        hb.sync()
        return out

    def bias_words(self, t, bias):
        """every 64-bit word of the int64 tensor t += bias (mod 2^64), on the library's stream; complete on return"""
        hb = self.hb
        hb._chk(hb.lib.hobbit_u64_bias_fold(hb.ctx, t.data_ptr(), t.numel(), bias & 0xFFFFFFFFFFFFFFFF, 0))
        hb.sync()

    def fold_words(self, t, bias):
        """every word w of t -> (w + bias) mod p, canonical: the integer sum of the ranks' partials back in the field"""
        hb = self.hb
        hb._chk(hb.lib.hobbit_u64_bias_fold(hb.ctx, t.data_ptr(), t.numel(), bias & 0xFFFFFFFFFFFFFFFF, 1))
        hb.sync()
        return t

    def sum_vectors(self, parts):
        hb = self.hb
        import torch
        if len(parts) == 1:
            return parts[0]                                        # one rank: the partial is the aggregate
        acc = parts[0].clone()
        torch.cuda.current_stream(self.device).synchronize()   # the clone ran on torch's stream; the sums run on the library's
        for p in parts[1:]:
            hb._chk(hb.lib.hobbit_f_binop(hb.ctx, 0, acc.data_ptr(), p.data_ptr(), acc.data_ptr(), acc.shape[0]))
        hb.sync()
        return acc

    def open_from_aggregate(self, aggr, plan, queries):
        return self.hb.open_from_aggregate((aggr.data_ptr(), plan.M), plan.K, plan.trs, queries)

    def gather_local(self, rows, cols, plan):
        """reply[q][li] = tensor shard of local chunk li at (row_q, col_q): the shard is codeword-major, chunk stride 4M"""
        import torch
        hb = self.hb
        n_own = self._chunks[1]
        r = np.ascontiguousarray(rows, np.uint32); c = np.ascontiguousarray(cols, np.uint32)
        out = np.zeros((len(r), n_own, 2), np.uint64)
        hb._chk(hb.lib.hobbit_tensor_gather(hb.ctx, self._tensor.ptr, plan.M, n_own, plan.trs, r.ctypes.data, c.ctypes.data, len(r), out.ctypes.data))
        if plan.world == 1:
            return out.view(np.int64)                              # nobody to gather with: the answers stay on the host (to_host passes arrays through)
        return self._upload("reply", out.view(np.int64))

    def subtree_paths(self, subtree, local_pos, plan):
        import torch
        hb = self.hb
        pos = np.ascontiguousarray(local_pos, np.uint64)
        depth = plan.m_local.bit_length() - 1
        out = np.zeros((len(pos), depth, 32), np.uint8)
        hb._chk(hb.lib.hobbit_merkle_paths(hb.ctx, subtree.data_ptr(), plan.m_local, pos.ctypes.data, len(pos), out.ctypes.data))
        return self._upload("subpaths", out)


class ElasticHipOps(HipOps):
    """Per-rank compute of the sharded streaming commit: a group's four chunks go through hobbit_elastic_push_inner.
    `source(chunk_index)` returns the device pointer of that B-element chunk (valid until the next call)."""

    def inner_digests_one(self, source, li, plan):
        import ctypes
        import torch
        hb = self.hb
        if li == 0:
            e = ctypes.c_void_p()
            hb._chk(hb.lib.hobbit_elastic_begin(hb.ctx, plan.B, plan.trs, plan.lin, 1, ctypes.byref(e)))
            self._e = e
        g = plan.chunks_of(self.rank)[li]
        out = torch.empty((plan.M, 32), dtype=torch.uint8, device=self.device)
        for q in range(4):
            hb._chk(hb.lib.hobbit_elastic_push_inner(hb.ctx, self._e, source(4 * g + q), out.data_ptr()))
        hb.sync()
        if li == len(plan.chunks_of(self.rank)) - 1:
            hb.lib.hobbit_elastic_free(self._e); self._e = None
        return out

    def __init__(self, hb, torch_device, rank):
        super().__init__(hb, torch_device)
        self.rank = rank


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
