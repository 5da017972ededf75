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

    def empty_digests(self, K, m_local):
        return torch.empty((K, m_local, 32), dtype=torch.uint8)

    def inner_digests_one(self, local_chunks, li, plan):
        out = np.zeros((plan.M, 32), np.uint8)
        t = self.orc.compute_tensorcode(local_chunks[li], plan.trs, 1)    # (2trs, cols, 2) row-major
        for j in range(plan.trs // 2):
            blk = np.ascontiguousarray(t[4 * j:4 * j + 4].transpose(1, 0, 2)).view(np.uint8).reshape(plan.cols, 64)
            out[j * plan.cols:(j + 1) * plan.cols] = self.orc.blake3_64(blk)
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

    def to_host(self, role, t):
        return t.cpu().numpy()

    def upload(self, role, host):
        return torch.from_numpy(np.ascontiguousarray(host))

    def after_collective(self):
        pass

    # ---- chain relay (CPU stand-ins)
    def empty_state(self, n, role="tmp"):
        return torch.zeros((n, 32), dtype=torch.uint8)

    def encode_local(self, local_chunks, plan):
        half = plan.trs // 2
        self._slots = []                                       # per local chunk: the 64-byte leaf groups in SLOT order (col * trs/2 + j)
        for m in local_chunks:
            t = self.orc.compute_tensorcode(m, plan.trs, 1)    # (2trs, cols, 2) row-major
            g = np.ascontiguousarray(t.reshape(half, 4, plan.cols, 2).transpose(2, 0, 1, 3)).view(np.uint8).reshape(plan.cols * half, 64)
            self._slots.append(g)

    def wait_recv(self, work):
        work.wait()

    def chain_block(self, plan, lo, cnt, st_in, st_out, levels):
        st = st_in.numpy().copy() if st_in is not None else np.zeros((cnt, 32), np.uint8)
        for g in self._slots:
            st = self.orc.blake3_64(np.concatenate([self.orc.blake3_64(g[lo:lo + cnt]), st], axis=1))
        if st_out is not None:
            st_out.copy_(torch.from_numpy(st))
        if levels is not None:                                 # leaf order j * cols + col
            half = plan.trs // 2
            slot = np.arange(lo, lo + cnt)
            levels.numpy()[(slot % half) * plan.cols + slot // half] = st

    def tree_full(self, levels, M):
        return torch.from_numpy(self.orc.create_tree_blake(levels.numpy()[:M].copy()))

    def tree_paths(self, levels, pos, M):
        lv = levels.numpy()
        return np.stack([self.orc.open_tree_blake(lv, M, int(p), 0, 0) for p in pos])

    def paths_buffer(self, queries, depth, host=None):
        return torch.from_numpy(host.copy()) if host is not None else torch.zeros((queries, depth, 32), dtype=torch.uint8)

    # ---- open (CPU stand-ins for HipOps' open methods)
    def set_local_chunks(self, local_chunks):
        self._chunks = local_chunks

    def eq_table_host(self, r):
        return self.orc.precompute_beta(r)

    def aggregate_local(self, coeffs, plan):
        flat = np.concatenate(self._chunks)
        return torch.from_numpy(self.orc.aggregate(flat, coeffs).view(np.int64))

    def bias_words(self, t, bias):
        v = t.numpy().view(np.uint64); v += np.uint64(bias & 0xFFFFFFFFFFFFFFFF)

    def fold_words(self, t, bias):
        P = np.uint64((1 << 61) - 1)
        v = t.numpy().view(np.uint64); v += np.uint64(bias & 0xFFFFFFFFFFFFFFFF)
        v[...] = (v & P) + (v >> np.uint64(61))
        v[v >= P] -= P
        return t

    def sum_vectors(self, parts):
        acc = parts[0].numpy().view(np.uint64)
        for p in parts[1:]:
            acc = self.orc.f_add(acc, p.numpy().view(np.uint64))
        return torch.from_numpy(np.ascontiguousarray(acc).view(np.int64))

    def open_from_aggregate(self, aggr, plan, queries):
        res = self.orc.open_standard_from_aggregate(aggr.numpy().view(np.uint64), plan.K, plan.trs, queries)
        res["cols"] = res["I"][:, 0].copy(); res["rows"] = res["I"][:, 1].copy()
        return res

    def gather_local(self, rows, cols, plan):
        ts = [self.orc.compute_tensorcode(m, plan.trs, 1) for m in self._chunks]          # (2trs, cols, 2) row-major each
        out = np.stack([t[rows, cols] for t in ts], axis=1)                                # (queries, n_own, 2)
        return torch.from_numpy(np.ascontiguousarray(out).view(np.int64))

    def subtree_paths(self, subtree, local_pos, plan):
        lv = subtree.numpy()
        return torch.from_numpy(np.stack([self.orc.open_tree_blake(lv, plan.m_local, int(p), 0, 0) for p in local_pos]))


class OracleElasticOps(OracleOps):
    """CPU stand-in for ElasticHipOps: a group's inner digests from the oracle (stream model set by the worker)"""

    def __init__(self, orc, lib, rank):
        super().__init__(orc, lib); self.rank = rank

    def inner_digests_one(self, source, li, plan):
        import ctypes
        g = plan.chunks_of(self.rank)[li]
        out = np.zeros((plan.M, 32), np.uint8)
        self.orc.lib.orc_elastic_group_digests(ctypes.c_size_t(plan.B), ctypes.c_int(plan.opt), ctypes.c_size_t(g), out.ctypes.data_as(ctypes.c_void_p))
        return torch.from_numpy(out)


def _elastic_worker(rank, world, port, N, B, opt, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from __graft_entry__ import load_package
    mod = load_package()
    orc = pyoracle.Oracle()
    plan = mod.parallel.ElasticPlan(N, B, opt, world)
    orc.rng_reset()
    if opt == 2:
        orc.expander_init_store(plan.trs)
    orc.stream_config(1, 9000)                       # chunk c = splitmix_field(B, 9000 + c): every group differs
    ops = OracleElasticOps(orc, mod.load_library(), rank)
    res = mod.parallel.sharded_commit(ops, dist, plan, rank, None)
    q.put((rank, res["subtree"].numpy(), res["top"], res["root"], None))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, N, K, q, do_open=False):
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
    ops = OracleOps(orc, mod.load_library())
    res = mod.parallel.sharded_commit(ops, dist, plan, rank, local)
    out = None
    if do_open:
        import ctypes
        x = orc.generate_randomness(N.bit_length() - 1)                  # same point on every rank
        ctypes.CDLL(None).srandom(2024)                                  # same libc stream on every rank
        ops.set_local_chunks(local)
        o = mod.parallel.sharded_open(ops, dist, plan, rank, res, x, 300)
        out = {k: o[k] for k in ("I", "poly", "r", "vr", "fin", "scalars", "roots", "reply", "paths")}
        out["sp_c_wq"] = o["sp_c"]["wq"]; out["sp_f_q1"] = o["sp_f"]["q1"]
    q.put((rank, res["subtree"].numpy(), res["top"], res["root"], out))
    dist.barrier()
    dist.destroy_process_group()


def _relay_worker(rank, world, port, N, K, q, do_open=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from __graft_entry__ import load_package
    mod = load_package()
    orc = pyoracle.Oracle()
    trs = N // (K << 11)
    orc.rng_reset(); poly = orc.generate_randomness(N); orc.expander_init_store(trs)     # same inputs on every rank
    plan = mod.parallel.ShardPlan(N, K, trs, world, contiguous=True)
    M = plan.M
    local = [poly[i * M:(i + 1) * M] for i in plan.chunks_of(rank)]
    ops = OracleOps(orc, mod.load_library())
    res = mod.parallel.sharded_commit_relay(ops, dist, plan, rank, local, blocks=8)
    out = None
    if do_open:
        import ctypes
        x = orc.generate_randomness(N.bit_length() - 1)
        ctypes.CDLL(None).srandom(2024)
        ops.set_local_chunks(local)
        o = mod.parallel.sharded_open(ops, dist, plan, rank, res, x, 300)
        out = {k: o[k] for k in ("I", "poly", "r", "vr", "fin", "scalars", "roots", "reply", "paths")}
        out["sp_c_wq"] = o["sp_c"]["wq"]; out["sp_f_q1"] = o["sp_f"]["q1"]
    q.put((rank, res["levels"].numpy() if res["levels"] is not None else None, res["owner"], res["root"], out))
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


def test_sharded_open_matches_single_process(oracle):
    """world 2: per-rank partial aggregates + one all-gather, replicated open from the aggregate, replies gathered from the tensor
    shards, Merkle paths stitched from the owner's subtree and the shared top levels -- everything equal to the single-process
    open_standard of the same polynomial with the same libc stream."""
    import ctypes
    from __graft_entry__ import load_package, build_hip
    build_hip()
    world, N, K, queries = 2, 1 << 20, 32, 300
    trs = N // (K << 11)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    import queue as _q
    import time as _t
    deadline = _t.time() + 300
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
    lv, T = oracle.commit_standard(poly, K, trs, 1, want_tensor=True)
    x = oracle.generate_randomness(N.bit_length() - 1)
    libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
    libc.srandom(2024)
    libc.srandom(ctypes.c_uint(libc.random() & 0xFFFFFFFF))            # sharded_open re-seeds every rank with one value drawn by rank 0
    want = oracle.open_standard(poly, K, trs, x, queries, tensor=T)
    M = N // K
    for g in got:
        o = g[4]
        for k in ("I", "poly", "r", "vr", "fin", "scalars", "roots", "reply"):
            assert np.array_equal(o[k], want[k]), (g[0], k)
        assert np.array_equal(o["sp_c_wq"], want["sp_c"]["wq"]) and np.array_equal(o["sp_f_q1"], want["sp_f"]["q1"])
        for qi in (0, 7, 150, queries - 1):
            assert np.array_equal(o["paths"][qi], oracle.open_tree_blake(lv, M, int(want["I"][qi, 0]), int(want["I"][qi, 1]), 2 * M // trs)), (g[0], qi)


@pytest.mark.parametrize("opt", [1, 2])
def test_sharded_elastic_commit_matches_single_process(oracle, opt):
    """config 5's split at world 2: groups of 4 consecutive chunks per rank, one exchange of inner digests, per-rank chain + subtree, one
    all-gather of subtree roots -- the assembled tree equals the single-process streaming commit on the same (varying) stream"""
    import ctypes
    from __graft_entry__ import load_package, build_hip
    build_hip()
    mod = load_package()
    world, N, B = 2, 1 << 19, 1 << 14                 # 8 groups of 4 chunks
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_elastic_worker, args=(r, world, port, N, B, opt, q)) for r in range(world)]
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
    plan = mod.parallel.ElasticPlan(N, B, opt, world)
    oracle.rng_reset()
    if opt == 2:
        oracle.expander_init_store(plan.trs)
    oracle.stream_config(1, 9000)
    try:
        want = np.zeros((8 * B, 32), np.uint8)
        oracle.lib.orc_elastic_commit_model.restype = ctypes.c_size_t
        cnt = oracle.lib.orc_elastic_commit_model(ctypes.c_size_t(N), ctypes.c_size_t(B), ctypes.c_int(opt), want.ctypes.data_as(ctypes.c_void_p))
    finally:
        oracle.stream_config(0, 0)
    levels = mod.parallel.assemble_levels(plan, [g[1] for g in got], got[0][2])
    T = 4 * B
    assert np.array_equal(levels[:T - 1], want[:T - 1]) and np.array_equal(levels[T:], want[T:cnt])     # leaf 4B-1 is undefined in the reference


def test_shard_plan():
    from __graft_entry__ import load_package
    P = load_package().parallel.ShardPlan
    p = P(1 << 28, 32, 4096, 8)
    assert p.M == 1 << 23 and p.cols == 4096 and p.m_local == 1 << 20
    assert p.chunks_of(3) == [3, 11, 19, 27] and p.owner(19) == (3, 2)
    assert p.leaf_range(7) == (7 << 20, 8 << 20)
    with pytest.raises(AssertionError):
        P(1 << 20, 32, 16, 3)


@pytest.mark.parametrize("world", [2, 4])
def test_relay_commit_matches_single_process(oracle, world):
    """chain-relay commit (contiguous chunks per rank, the 32-byte running leaf states handed from rank to rank in blocks, the tree on
    the last rank, root broadcast): every level equals the single-process commit_standard"""
    from __graft_entry__ import build_hip
    build_hip()
    N, K = 1 << 18, 32
    trs = N // (K << 11)
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_relay_worker, args=(r, world, port, N, K, q)) for r in range(world)]
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
    for g in got:
        assert g[2] == world - 1 and np.array_equal(g[3], want[-1])          # every rank knows the owner and the root
        assert (g[1] is not None) == (g[0] == world - 1)
    assert np.array_equal(got[-1][1], want)


@pytest.mark.parametrize("world", [2, 4])
def test_relay_open_matches_single_process(oracle, world):
    """world 2 and 4 on a relay commitment: partial aggregates + all-gather, replicated open, replies from the (contiguous) tensor shards, all
    Merkle paths from the rank that holds the tree, broadcast -- equal to the single-process open_standard with the same libc stream"""
    import ctypes
    from __graft_entry__ import build_hip
    build_hip()
    N, K, queries = 1 << 20, 32, 300
    trs = N // (K << 11)
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_relay_worker, args=(r, world, port, N, K, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    import queue as _q
    import time as _t
    deadline = _t.time() + 300
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
    lv, T = oracle.commit_standard(poly, K, trs, 1, want_tensor=True)
    x = oracle.generate_randomness(N.bit_length() - 1)
    libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
    libc.srandom(2024)
    libc.srandom(ctypes.c_uint(libc.random() & 0xFFFFFFFF))
    want = oracle.open_standard(poly, K, trs, x, queries, tensor=T)
    M = N // K
    for g in got:
        o = g[4]
        for k in ("I", "poly", "r", "vr", "fin", "scalars", "roots", "reply"):
            assert np.array_equal(o[k], want[k]), (g[0], k)
        assert np.array_equal(o["sp_c_wq"], want["sp_c"]["wq"]) and np.array_equal(o["sp_f_q1"], want["sp_f"]["q1"])
        for qi in range(0, queries, 17):
            assert np.array_equal(o["paths"][qi], oracle.open_tree_blake(lv, M, int(want["I"][qi, 0]), int(want["I"][qi, 1]), 2 * M // trs)), (g[0], qi)
