"""GPU parity tests: every call goes through the C ABI (libhobbit_hip.so) and is compared
BIT-EXACTLY (integer / byte work, no tolerance) with the oracle on the same seeded inputs, with
the committed golden vectors the real reference produced, and -- at sizes the oracle cannot
finish in seconds -- through size-independent properties."""
import os
import sys
import numpy as np
import pytest
import golden_cases
from golden_cases import dg, samp
from oracle.pyoracle import splitmix_field, P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hb():
    from __graft_entry__ import load_package
    mod = load_package()
    h = mod.Hobbit(0)          # raises if the HIP library or the GPU is missing: no fallback
    yield h
    h.close()


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rehash_path(hb, node, pos, path):
    """Walk one open_tree_blake sibling path (src/merkle_tree.cpp:308-324) up to the root with create_tree_blake's left|left rule
    (:275-280): a parent is H(L | L) with L the EVEN node of the pair -- the running node when its position is even, the sibling the
    path carries when it is odd.  (So an odd leaf never reaches the root: that is the reference's tree, kept bit for bit.)"""
    import ctypes
    node = np.ascontiguousarray(node, np.uint8).copy()
    for lvl in range(path.shape[0]):
        L = node if pos % 2 == 0 else path[lvl]
        blk = np.concatenate([L, L]).astype(np.uint8); out = np.zeros(32, np.uint8)
        hb.lib.hobbit_blake3_64_host(blk.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), 1)
        node = out; pos //= 2
    return node


def leaf_from_tensor(hb, c, col, row):
    """the commitment's leaf for query (col,row): hash_double_field_element_merkle_damgard_blake chained over the K chunks on rows
    4j..4j+3 of the column (src/Our_PC.cpp:162-166), recomputed on the HOST from gathered tensor entries"""
    import ctypes
    j = row // 4
    t = c.gather(np.arange(4 * j, 4 * j + 4), np.full(4, col))          # (4, K, 2)
    prev = np.zeros(32, np.uint8)
    for i in range(c.K):
        blk = np.ascontiguousarray(t[:, i, :]).view(np.uint8).reshape(64)
        inner = np.zeros(32, np.uint8)
        hb.lib.hobbit_blake3_64_host(blk.ctypes.data_as(ctypes.c_void_p), inner.ctypes.data_as(ctypes.c_void_p), 1)
        blk2 = np.concatenate([inner, prev]); out = np.zeros(32, np.uint8)
        hb.lib.hobbit_blake3_64_host(blk2.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), 1)
        prev = out
    return prev


def graphs_from(oracle, n):
    lv, dep, m = {}, 0, n
    while m > 13:
        for kind in (0, 1):
            lv[(dep, kind)] = oracle.graph(dep, kind)
        m = int(0.211 * m); dep += 1
    return lv


# ---- field / hashes ------------------------------------------------------------------------
def test_field_ops_vs_golden(hb):
    g = gold("field")
    a, b = golden_cases.field_inputs(1024)
    assert np.array_equal(hb.f_binop(0, a, b), g["add"])
    assert np.array_equal(hb.f_binop(1, a, b), g["sub"])
    assert np.array_equal(hb.f_binop(2, a, b), g["mul"])
    A, B = golden_cases.field_inputs(4096)
    assert np.array_equal(dg(hb.f_binop(2, A, B)), g["mul4k"])


def test_field_mul_large_vs_oracle(hb, oracle):
    a = splitmix_field(1 << 16, 21); b = splitmix_field(1 << 16, 22)
    assert np.array_equal(hb.f_binop(2, a, b), oracle.f_mul(a, b))
    assert np.array_equal(hb.f_binop(0, a, b), oracle.f_add(a, b))
    assert np.array_equal(hb.f_binop(1, a, b), oracle.f_sub(a, b))


def test_device_splitmix_matches_host(hb):
    d = hb.fill_splitmix(5000, 7)
    assert np.array_equal(hb.to_host(d, (5000, 2), np.uint64), splitmix_field(5000, 7))


def test_blake_vs_golden(hb):
    g = gold("blake")
    blk = np.random.default_rng(0).integers(0, 256, (256, 64), dtype=np.uint8)
    blk[0] = [(7 * i + 3) % 256 for i in range(64)]
    assert np.array_equal(hb.blake3_64(blk), g["h64"])
    a, _ = golden_cases.field_inputs(256)
    prev = np.random.default_rng(1).integers(0, 256, (64, 32), dtype=np.uint8); prev[:8] = 0
    assert np.array_equal(hb.hash_md(a, prev), g["md"])


def test_merkle_vs_golden(hb):
    g = gold("merkle")
    a, _ = golden_cases.field_inputs(1024)
    l0 = np.random.default_rng(2).integers(0, 256, (1024, 32), dtype=np.uint8)
    assert np.array_equal(hb.mt_commit_blake(a[:4]), g["mt4"])
    assert np.array_equal(hb.mt_commit_blake(a[:32]), g["mt32"])
    assert np.array_equal(hb.mt_commit_blake(a), g["mt1024"])
    assert np.array_equal(hb.create_tree_blake(l0), g["tree1024"])       # pins the left|left quirk
    assert np.array_equal(hb.create_tree_blake(l0[:2]), g["tree2"])
    assert np.array_equal(hb.create_tree_blake(l0[:1]), g["tree1"])


def test_merkle_conventional_tree_differs_and_verifies(hb, oracle):
    # quirk off: parent = H(left|right); check against the oracle's 64-byte hash directly
    l0 = np.random.default_rng(3).integers(0, 256, (64, 32), dtype=np.uint8)
    lv = hb.create_tree_blake(l0, quirk=0)
    want = oracle.blake3_64(l0.reshape(32, 64))
    assert np.array_equal(lv[64:96], want)
    assert not np.array_equal(lv, hb.create_tree_blake(l0, quirk=1))


# ---- FFT / eq table ------------------------------------------------------------------------
def test_fft_vs_golden(hb):
    g = gold("fft")
    for logn in (1, 4, 8, 12):
        x = splitmix_field(1 << logn, 7 + logn)
        for inv in (0, 1):
            y = hb.fft(x, bool(inv))
            if logn <= 8:
                assert np.array_equal(y, g["fft_%d_%d" % (logn, inv)]), (logn, inv)
            else:
                assert np.array_equal(dg(y), g["fft_%d_%d_dg" % (logn, inv)]), (logn, inv)
    x = splitmix_field(4096, 3); x[2048:] = 0
    assert np.array_equal(dg(hb.fft(x)), g["fft_pad_dg"])
    for k in (1, 5, 10):
        b = hb.precompute_beta(splitmix_field(k, 3))
        assert np.array_equal(b if k <= 5 else dg(b), g["beta_%d" % k])
    assert np.array_equal(hb.evaluate_vector(splitmix_field(1024, 4), splitmix_field(12, 5)), g["eval"])


@pytest.mark.gpu
def test_commit_unwritten_zero_tail_on_a_dirty_buffer(hb):
    """commit_standard does not write the zero tail of the RS x expander codewords (rows past the codeword length rounded up to the leaf group:
    1144 of 8192 rows at trs = 4096); the leaf chain, the gathers and the row reads answer those rows as zeros, and hobbit_commitment_tensor_dev
    fills them in before the raw pointer leaves the library.  Here the tensor buffer the commit re-uses is full of garbage: the commitment, every
    accessor and the opening must be those of HOBBIT_COMMIT_SKIP_TAIL=0 (everything written)."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, K, trs = 1 << 24, 2, 4096
    hb.rng_reset(); code_len = hb.expander_init_store(trs)
    rv = (code_len + 3) & ~3
    assert 4096 < code_len < rv < 8192
    d = hb.fill_splitmix(N, 5151)
    os.environ["HOBBIT_COMMIT_SKIP_TAIL"] = "0"
    try:
        c0 = hb.commit_standard((d, N), K, trs, 1)
        want_levels = c0.levels()
        probe_rows = [0, code_len - 1, code_len, rv - 1, rv, rv + 1, 8191]
        want_rows = {r: c0.tensor_row(1, r) for r in probe_rows}
        x = splitmix_field(24, 5)
        libc.srandom(11); want_open = hb.open_standard((d, N), c0, x, 5900)
        # dirty the whole buffer through the raw pointer, then park it for the next commit
        ptr = hb.lib.hobbit_commitment_tensor_dev(c0.h)
        hb._chk(hb.lib.hobbit_fill_splitmix(hb.ctx, ptr, 4 * N, 99)); hb.sync()
        c0.free()
    finally:
        del os.environ["HOBBIT_COMMIT_SKIP_TAIL"]
    c = hb.commit_standard((d, N), K, trs, 1)                # re-uses the dirty buffer; the tail stays unwritten
    assert np.array_equal(c.levels(), want_levels)
    for r in probe_rows:
        assert np.array_equal(c.tensor_row(1, r), want_rows[r]), r
    assert not want_rows[rv].any() and want_rows[code_len - 1].any()
    libc.srandom(11); got = hb.open_standard((d, N), c, x, 5900)
    assert (got["rows"] >= rv).sum() > 500                    # plenty of queries land in the unwritten rows
    for k in ("cols", "rows", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots", "checks"):
        assert np.array_equal(got[k], want_open[k]), k
    # the raw pointer: the tail is real zeros from here on
    ptr = hb.lib.hobbit_commitment_tensor_dev(c.h)
    col = np.zeros((8192, 2), np.uint64)
    hb._chk(hb.lib.hobbit_memcpy_d2h(hb.ctx, col.ctypes.data, ptr + 16 * 8192 * (4096 + 77), 16 * 8192))      # column 77 of chunk 1
    assert not col[rv:].any() and col[code_len - 1].any()
    c.free()


def test_evaluate_vector_all_fold_paths_vs_oracle(hb, oracle):
    """evaluate_vector folds two levels per launch above 4096 elements and the last (up to 12) levels in one workgroup: every split of the
    levels between the two kernels, against the oracle's level-by-level fold (src/utils.cpp:789-802)"""
    for logn in (1, 2, 5, 11, 12, 13, 14, 15, 17, 20):
        v = splitmix_field(1 << logn, 40 + logn); r = splitmix_field(logn, 90 + logn)
        assert np.array_equal(hb.evaluate_vector(v, r), oracle.evaluate_vector(v, r)), logn


@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_fft_all_sizes_batched_vs_oracle(hb, oracle, logn):
    """every LDS-resident length (64 ... 2048: the radix-8 kernel with its radix-1/2/4 tail; 4096: its own kernel; below 64: the generic
    one), a row count that is not a multiple of the rows a workgroup owns"""
    nb = 5 if logn >= 9 else 70
    x = splitmix_field(nb << logn, 100 + logn).reshape(nb, 1 << logn, 2)
    y = hb.fft(x)
    for b in range(0, nb, max(1, nb // 7)):
        assert np.array_equal(y[b], oracle.fft(x[b])), (logn, b)
    assert np.array_equal(y[nb - 1], oracle.fft(x[nb - 1]))
    z = hb.fft(y, inverse=True)                     # round trip: property at every size
    assert np.array_equal(z, x)


def test_eq_table_large_vs_oracle_and_sum(hb, oracle):
    r = splitmix_field(16, 55)
    b = hb.precompute_beta(r)
    assert np.array_equal(b, oracle.precompute_beta(r))
    # sum of the eq table is 1 (size-independent property)
    s = np.array([int(b[:, 0].astype(object).sum() % P), int(b[:, 1].astype(object).sum() % P)])
    assert s.tolist() == [1, 0]


# ---- error behaviour ---------------------------------------------------------------------------
def test_invalid_arguments_are_errors_not_crashes(hb, oracle):
    """C ABI calls return negative HOBBIT_E* codes with a message (the mirror turns them into the reference's printf + exit(-1));
    nothing is launched on shapes a kernel or its grid does not assume."""
    import ctypes
    from __graft_entry__ import load_package
    E = load_package().HobbitError
    poly = splitmix_field(1 << 12, 1)
    d = hb.to_device(poly)
    h = ctypes.c_void_p()
    lib, ctx = hb.lib, hb.ctx
    assert lib.hobbit_commit_standard(ctx, d.ptr, 1 << 12, 3, 4, 1, ctypes.byref(h)) < 0            # K does not divide N
    assert lib.hobbit_commit_standard(ctx, d.ptr, 1 << 12, 4, 6, 1, ctypes.byref(h)) < 0            # trs not a multiple of 4 dividing N/K
    assert b"trs" in lib.hobbit_last_error(ctx)
    assert lib.hobbit_fft_batch(ctx, d.ptr, 25, 1, 1 << 25, 0) < 0                                   # longer than 2^24
    assert lib.hobbit_fft_batch(ctx, d.ptr, 13, 1, 1 << 13, 1) < 0                                   # long inverse transforms are not built
    assert lib.hobbit_mt_commit_blake(ctx, d.ptr, 24, d.ptr) < 0                                     # N/4 not a power of two
    hb.rng_reset(); hb.expander_init_store(64)
    assert lib.hobbit_encode_batch(ctx, d.ptr, d.ptr, 100, 1, 100, 200) < 0                          # graphs are finalized for n = 64, not 100
    assert lib.hobbit_encode_batch(ctx, d.ptr, d.ptr, 64, 1, 32, 128) < 0                            # leading dimension shorter than n
    assert lib.hobbit_sumcheck2(ctx, d.ptr, d.ptr, 1000, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr) < 0      # n not a power of two
    pos = np.array([1 << 40], np.uint64); out = np.zeros((1, 10, 32), np.uint8)
    assert lib.hobbit_merkle_paths(ctx, d.ptr, 1 << 10, pos.ctypes.data, 1, out.ctypes.data) < 0     # src/merkle_tree.cpp:310-313: position out of range
    with pytest.raises(E):
        hb.shockwave_commit(poly, 3)                                                                 # k must be a power of two in [4, 64]
    with pytest.raises(E):
        hb.open_from_aggregate(poly, 32, 6, 10)                                                      # trs must give 4096-point rows
    # the context is still usable afterwards
    assert np.array_equal(hb.precompute_beta(splitmix_field(6, 2)), oracle.precompute_beta(splitmix_field(6, 2)))


# ---- expander code -------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 13, 14, 16, 64, 100, 256, 1024, 4096])
def test_encode_vs_golden(hb, n):
    g = gold("graph_encode")
    hb.rng_reset()
    assert hb.expander_init_store(n) == g["len_%d" % n][0]          # graphs drawn on the host, libc order
    for tag, src in (("small", (splitmix_field(n, 40 + n) % np.uint64(1 << 32)) * np.array([1, 0], dtype=np.uint64)),
                     ("full", splitmix_field(n, 41 + n))):
        d = hb.encode_monolithic(src)
        if n <= 256 and tag == "full":
            assert np.array_equal(d, g["enc_%s_%d" % (tag, n)])
        else:
            assert np.array_equal(dg(d), g["enc_%s_%d_dg" % (tag, n)])


@pytest.mark.parametrize("n", [2427, 3000, 3333, 4000])
def test_encode_odd_sizes_vs_oracle(hb, oracle, n):
    """message lengths that are not powers of two, large enough for the 64-wide slices (>= 512 outputs) with ragged last slices;
    single-pass (n <= 2560: codeword under 80 KB) and two-pass launches"""
    oracle.rng_reset(); oracle.expander_init_store(n)
    hb.upload_graphs(n, graphs_from(oracle, n))
    x = splitmix_field(2 * n, 70 + n).reshape(2, n, 2)
    got = hb.encode_monolithic(x)
    for b in range(2):
        want, ln = oracle.encode_monolithic(x[b])
        assert np.array_equal(got[b][:ln], want[:ln]), (n, b)
        assert not got[b][ln:].any()


def test_encode_full_range_weights_vs_golden(hb, oracle):
    g = gold("graph_encode")
    oracle.rng_reset(); oracle.expander_init_store(64)
    lv = graphs_from(oracle, 64)
    for kind in (0, 1):
        lv[(0, kind)]["w"] = splitmix_field(lv[(0, kind)]["L"] * lv[(0, kind)]["degree"], 100 + kind)
    hb.upload_graphs(64, lv)
    assert np.array_equal(hb.encode_monolithic(splitmix_field(64, 5)), g["enc_fullw_64"])


def test_encode_batch_linearity_4096(hb, oracle):
    hb.rng_reset(); hb.expander_init_store(4096)
    x = splitmix_field(3 * 4096, 9).reshape(3, 4096, 2)
    x[2] = oracle.f_add(x[0], x[1])
    y = hb.encode_monolithic(x)
    assert np.array_equal(y[2], oracle.f_add(y[0], y[1]))           # linear code
    assert np.array_equal(y[:, :4096], x)                            # systematic
    assert not y[:, 7045:].any()                                     # tail past the codeword stays 0
    oracle.rng_reset(); oracle.expander_init_store(4096)
    assert np.array_equal(y[0], oracle.encode_monolithic(x[0])[0])


@pytest.mark.parametrize("batch", [1, 255, 700])
def test_encode_in_place_4096_persistent_kernels(hb, oracle, batch, monkeypatch):
    """n = 4096 in place -- the commit's form, served by the persistent register-resident kernels (k_enc_fat for C_0 and D_0, LDS-DMA
    double-buffered windows): fewer columns than workgroups, one short of a full round, and several rounds with a ragged last one; against
    the oracle on sampled columns and against the one-workgroup-per-column kernels (HOBBIT_ENC_FAT=0) and every mix of the two on all of them."""
    oracle.rng_reset(); oracle.expander_init_store(4096)
    hb.upload_graphs(4096, graphs_from(oracle, 4096))
    x = splitmix_field(batch * 4096, 77 + batch).reshape(batch, 4096, 2)
    got = {}
    for mode in ("0", "1", "3", "7"):
        monkeypatch.setenv("HOBBIT_ENC_FAT", mode)
        got[mode] = hb.encode_monolithic(x, in_place=True)
    monkeypatch.delenv("HOBBIT_ENC_FAT")
    for m2, c1 in (("0", "1"), ("1", "1"), ("2", "0"), ("4", "1")):                   # the middle steps: one launch / C_1 + narrow remainder, C_1 generic or fat
        monkeypatch.setenv("HOBBIT_ENC_M2", m2); monkeypatch.setenv("HOBBIT_ENC_FAT_C1", c1)
        got["m2_%s_%s" % (m2, c1)] = hb.encode_monolithic(x, in_place=True)
    monkeypatch.delenv("HOBBIT_ENC_M2"); monkeypatch.delenv("HOBBIT_ENC_FAT_C1")
    got["default"] = hb.encode_monolithic(x, in_place=True)
    for mode in got:
        assert np.array_equal(got[mode], got["0"]), mode
    assert np.array_equal(got["0"], hb.encode_monolithic(x))                          # the out-of-place form
    for b in sorted({0, batch // 2, batch - 1}):
        want, ln = oracle.encode_monolithic(x[b])
        assert ln == 7045 and np.array_equal(got["default"][b][:ln], want[:ln]) and not got["default"][b][ln:].any()


@pytest.mark.parametrize("logN,K,lin", [(18, 32, 1), (22, 32, 1), (24, 32, 1), (24, 16, 0), (23, 2, 0)])
def test_commit_standard_host_streaming_upload(hb, logN, K, lin):
    """hobbit_commit_standard_host: the polynomial comes from pageable host memory in 64 MiB pieces through two pinned buffers, chunk group g + 1
    in flight while group g's row FFT runs (the reference-shaped commit_standard(vector<F> &) of the mirror uses it).  Same tree, same tensor,
    and the device copy it leaves behind is the polynomial: small tensors (no pipeline), the piped shape, RS x RS, long rows."""
    N = 1 << logN
    trs = N // (K << 11) if lin else 128
    poly = splitmix_field(N, 300 + logN)
    if lin:
        hb.rng_reset(); hb.expander_init_store(trs)
    c0 = hb.commit_standard(poly, K, trs, lin)
    c1, d = hb.commit_standard_host(poly, K, trs, lin)
    assert np.array_equal(c1.levels(), c0.levels())
    assert np.array_equal(c1.tensor_row(K - 1, 2 * trs - 1), c0.tensor_row(K - 1, 2 * trs - 1)) and np.array_equal(c1.tensor_row(0, 1), c0.tensor_row(0, 1))
    assert np.array_equal(hb.to_host(d, (N, 2), np.uint64), poly)
    c0.free(); c1.free()


# ---- tensor code / commit ------------------------------------------------------------------
def test_tensorcode_vs_golden(hb):
    g = gold("tensorcode")
    for (M, trs) in ((1 << 13, 4), (1 << 15, 16), (1 << 17, 64)):
        for lin in (0, 1):
            hb.rng_reset(); hb.expander_init_store(trs)
            t = hb.compute_tensorcode(splitmix_field(M, 11), trs, lin)
            key = "tc_%d_%d_%d" % (M, trs, lin)
            assert np.array_equal(dg(t), g[key + "_dg"]), key


@pytest.mark.parametrize("N,K", golden_cases.COMMIT_CASES)
def test_commit_standard_vs_golden(hb, N, K):
    g = gold("commit")
    trs = N // (K << 11)
    hb.rng_reset()
    poly = hb.generate_randomness(N)                 # test_PC's input sequence (src/Our_PC.cpp:757-813)
    hb.expander_init_store(trs)
    c = hb.commit_standard(poly, K, trs, 1)
    key = "c_%d_%d_" % (N, K)
    lv = c.levels()
    assert np.array_equal(lv[-1], g[key + "root"])
    off, sz, dgs = 0, N // K, []
    while sz >= 1:
        dgs.append(dg(lv[off:off + sz])); off += sz; sz //= 2
    assert np.array_equal(np.stack(dgs), g[key + "level_dg"])
    T = c.tensor()
    assert np.array_equal(dg(T), g[key + "tensor_dg"])
    paths = []
    for (col, rw) in golden_cases.QUERIES:
        row = rw if rw >= 0 else (2 * trs - 1 if rw == -1 else trs)
        row = min(row, 2 * trs - 1)
        paths.append(c.open_tree_blake(col, row))
    assert np.array_equal(np.stack(paths), g[key + "paths"])
    # _compute_aggregation_reply gather
    rows = np.array([0, 3, 2 * trs - 1, trs], np.uint32); cols = np.array([0, 5, 4095, 100], np.uint32)
    rep = c.gather(rows, cols)
    for q in range(4):
        assert np.array_equal(rep[q], T[:, rows[q], cols[q]])
    c.free()


@pytest.mark.parametrize("N,K", golden_cases.COMMIT_RS_CASES)
def test_commit_standard_rs_vs_golden(hb, N, K):
    """test_PC(N, 1, K)'s commitment (linear_time == false, tensor_row_size = 128) against the REAL reference: every level, the whole tensor, paths"""
    g = gold("commit_rs")
    hb.rng_reset()
    poly = hb.generate_randomness(N)
    c = hb.commit_standard(poly, K, 128, 0)
    key = "crs_%d_%d_" % (N, K)
    lv = c.levels(); M = N // K; cols = 2 * M // 128
    assert np.array_equal(lv[-1], g[key + "root"])
    off, sz, dgs = 0, M, []
    while sz >= 1:
        dgs.append(dg(lv[off:off + sz])); off += sz; sz //= 2
    assert np.array_equal(np.stack(dgs), g[key + "level_dg"])
    if N <= 1 << 20:
        assert np.array_equal(dg(c.tensor()), g[key + "tensor_dg"])
    paths = [c.open_tree_blake(col, row) for (col, row) in ((0, 0), (5, 3), (cols - 1, 255), (cols // 2, 128), (7, 127))]
    assert np.array_equal(np.stack(paths), g[key + "paths"])
    c.free()


def test_commit_full_range_poly_vs_golden(hb):
    g = gold("commit")
    poly = splitmix_field(1 << 18, 77)
    for lin in (0, 1):
        c = hb.commit_standard(poly, 32, 4, lin)
        lv = c.levels()
        assert np.array_equal(lv[-1], g["cfull_%d_root" % lin])
        assert np.array_equal(dg(lv), g["cfull_%d_lv_dg" % lin])
        assert np.array_equal(dg(c.tensor()), g["cfull_%d_t_dg" % lin])
        c.free()
    assert np.array_equal(hb.aggregate(splitmix_field(1 << 14, 77), splitmix_field(16, 78)), g["aggr"])


def test_commit_2e22_vs_oracle(hb, oracle):
    N, K = 1 << 22, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    want, _ = oracle.commit_standard(poly, K, trs, 1)
    c = hb.commit_standard(poly, K, trs, 1)
    assert np.array_equal(c.levels(), want)
    c.free()


def test_commit_linearity_2e24(hb, oracle):
    """full-size-style property: the tensor code is linear, so tensor(a+b) = tensor(a)+tensor(b);
    checked on sampled entries of device-resident commitments of 2^24-coefficient polynomials."""
    N, K = 1 << 24, 32
    trs = N // (K << 11)
    hb.rng_reset(); hb.expander_init_store(trs)
    da, db = hb.fill_splitmix(N, 1), hb.fill_splitmix(N, 2)
    ds = hb.alloc(16 * N)
    hb._chk(hb.lib.hobbit_f_binop(hb.ctx, 0, da.ptr, db.ptr, ds.ptr, N))   # argtypes are declared: 64-bit safe
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 2 * trs, 512).astype(np.uint32); cols = rng.integers(0, 4096, 512).astype(np.uint32)
    reps = []
    for d in (da, db, ds):
        c = hb.commit_standard((d, N), K, trs, 1)
        reps.append(c.gather(rows, cols)); root = c.root(); c.free()
    assert np.array_equal(oracle.f_add(reps[0].reshape(-1, 2), reps[1].reshape(-1, 2)), reps[2].reshape(-1, 2))
    assert reps[2].any()


# ---- sumchecks -----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 4, 32, 1024, 1 << 16])
def test_sumchecks_vs_golden(hb, n):
    g = gold("sumcheck")
    pr = np.array([33, 0], dtype=np.uint64)
    v1, v2, v3, v1z, v2z = golden_cases.sumcheck_inputs(n)
    for tag, res in (("s2", hb.generate_2product_sumcheck_proof(v1, v2, pr)),
                     ("s3", hb.generate_3product_sumcheck_proof(v1, v2, v3, pr)),
                     ("s3z", hb.generate_3product_sumcheck_proof(v1z, v2z, v3, pr))):
        for k, v in res.items():
            assert np.array_equal(v, g["%s_%d_%s" % (tag, n, k)]), (tag, n, k)


def test_sumcheck2_beta_table_vs_golden(hb):
    g = gold("sumcheck")
    v1 = splitmix_field(1 << 12, 1)
    v2 = hb.precompute_beta(splitmix_field(12, 9))
    for k, v in hb.generate_2product_sumcheck_proof(v1, v2, np.array([33, 0], np.uint64)).items():
        assert np.array_equal(v, g["s2beta_%s" % k]), k


def test_sumcheck2_2e20_vs_oracle(hb, oracle):
    n = 1 << 20
    v1 = splitmix_field(n, 1); v2 = oracle.precompute_beta(splitmix_field(20, 9)); pr = np.array([33, 0], np.uint64)
    a = hb.generate_2product_sumcheck_proof(v1, v2, pr); b = oracle.sumcheck2(v1, v2, pr)
    for k in b:
        assert np.array_equal(a[k], b[k]), k


def test_sumcheck2_2e24_claim_consistency(hb, oracle):
    """C2 at its stated size (2^24), bit-exact against the REAL reference's generate_2product_sumcheck_proof on the same inputs
    (tests/golden/sumcheck2_2e24.npz: every round polynomial, challenge, vr and the final value; oracle/gen_sumcheck_2e24.py ran
    oracle/_ref once), plus the protocol's own invariants (what the reference's inline self-checks verify): q_0(0)+q_0(1) = <v1,v2>,
    q_{i+1}(0)+q_{i+1}(1) = q_i(r_i), and q_last(r_last) = vr[0]*vr[1]."""
    n = 1 << 24
    d1 = hb.fill_splitmix(n, 1)
    r = splitmix_field(24, 9)
    d2 = hb.precompute_beta(r, keep_on_device=True)
    pr = np.array([33, 0], np.uint64)
    res = hb.generate_2product_sumcheck_proof((d1, n), (d2, n), pr)
    g = gold("sumcheck2_2e24")
    for k in ("poly", "r", "vr", "fin"):
        assert np.array_equal(res[k], g[k]), k

    def ev(q, x):   # ((a x) + b) x + c
        t = oracle.f_add(oracle.f_mul(q[0:1], x), q[1:2])
        return oracle.f_add(oracle.f_mul(t, x), q[2:3])
    one = np.array([[1, 0]], np.uint64); zero = np.array([[0, 0]], np.uint64)
    # <v1, eq(r)> equals the multilinear evaluation of v1 at r with the reference's variable order
    v1 = hb.to_host(d1, (n, 2), np.uint64)
    claim = hb.evaluate_vector(v1, r).reshape(1, 2)
    q = res["poly"]
    assert np.array_equal(oracle.f_add(ev(q[0], zero), ev(q[0], one)), claim)
    for i in range(23):
        assert np.array_equal(oracle.f_add(ev(q[i + 1], zero), ev(q[i + 1], one)), ev(q[i], res["r"][i:i + 1])), i
    assert np.array_equal(ev(q[23], res["r"][23:24]), oracle.f_mul(res["vr"][0:1], res["vr"][1:2]))
    # transcript chain: r_i = mimc(mimc(mimc(r_{i-1}, a), b), c)
    rr = pr.reshape(1, 2)
    for i in range(24):
        for t in range(3):
            rr = oracle.mimc(rr, q[i, t:t + 1])
        assert np.array_equal(rr[0], res["r"][i])


# ---- the C++ host mirror (reference signatures over the C ABI) -------------------------------
def test_cpp_host_mirror_test_pc_and_sumcheck(oracle):
    """libhobbit_host.so keeps the reference's C++ signatures (commit_standard, expander_init_store,
    generate_2product_sumcheck_proof, ...); drive them through its extern-C hooks."""
    import ctypes
    from __graft_entry__ import PKG, build_host
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    g = gold("commit")
    root = np.zeros(32, np.uint8)
    lib.hobbit_host_test_pc_root.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    levels = lib.hobbit_host_test_pc_root(1 << 20, 32, root.ctypes.data_as(ctypes.c_void_p))   # test_PC(2^20,4,32) input sequence
    assert levels == 16 and np.array_equal(root, g["c_1048576_32_root"])
    lib.hobbit_host_graph_reupload_check.argtypes = [ctypes.c_size_t, ctypes.c_int]
    assert lib.hobbit_host_graph_reupload_check(1 << 20, 32) == 1          # graphs re-uploaded from the host arrays _C / D give the same commitment
    n = 1 << 10
    v1, v2, _, _, _ = golden_cases.sumcheck_inputs(n)
    pr = np.array([33, 0], np.uint64)
    q = np.zeros((10, 3, 2), np.uint64); r = np.zeros((10, 2), np.uint64); vr = np.zeros((2, 2), np.uint64); fin = np.zeros(2, np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_sumcheck2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 5
    assert lib.hobbit_host_sumcheck2(P(v1), P(v2), n, P(pr), P(q), P(r), P(vr), P(fin)) == 10
    gs = gold("sumcheck")
    assert np.array_equal(q, gs["s2_1024_poly"]) and np.array_equal(r, gs["s2_1024_r"])
    assert np.array_equal(vr, gs["s2_1024_vr"]) and np.array_equal(fin, gs["s2_1024_fin"])
    lib.hobbit_host_close()


@pytest.mark.parametrize("logN,K", [(18, 32), (20, 32), (20, 16), (22, 32), (24, 32), (26, 32), (28, 32)])
def test_test_pc_driver_prints_the_reference_proof_size(logN, K):
    """End to end against the REAL reference binary, up to the north-star size: `./pigeon <logN> 4 <K>` (test_PC, src/Our_PC.cpp:757-826)
    prints the proof size `ps`, which depends on every query index drawn from libc anywhere in the open (Merkle-path de-duplication in
    the commitment tree, both shockwave trees and every WHIR layer).  The device-backed C++ mirror's driver must print the same number
    (fixtures: tests/golden/ps_fingerprints.json, recorded from the reference's stdout)."""
    import json, subprocess
    from __graft_entry__ import PKG, build_host
    build_host()
    fp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ps_fingerprints.json")))["test_PC_ps_KB"]
    r = subprocess.run([os.path.join(PKG, "host", "test_pc"), str(logN), "4", str(K)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    last = r.stdout.strip().splitlines()[-1]                       # "ps,vt" (src/Our_PC.cpp:822)
    assert float(last.split(",")[0]) == fp["%d,%d" % (logN, K)], last
    assert ">>OK" in r.stdout
    # the commitment root the driver prints, against the REAL reference's commit_standard on the same inputs (oracle/_ref fixtures:
    # tests/golden/commit.npz for 2^18 / 2^20, tests/golden/bigroot_2e<logN>.npz from oracle/gen_big_roots.py for 2^22 ... 2^28)
    if logN <= 20 or os.path.exists(os.path.join(GOLD, "bigroot_2e%d.npz" % logN)):
        want = gold("commit")["c_%d_%d_root" % (1 << logN, K)] if logN <= 20 else gold("bigroot_2e%d" % logN)["root"]
        line = [l for l in r.stdout.splitlines() if l.startswith("root ")]
        assert line and line[0].split()[1] == want.tobytes().hex(), line


def test_ctx_create_leaves_libc_rng_alone():
    """The reference's transcript is a function of the process-wide libc generator; initialising the HIP runtime draws from it
    (measured).  hobbit_ctx_create must hand the caller's stream back untouched -- checked in a fresh process, where context
    creation is the first HIP call, including the first allocation and the first kernel launch."""
    import subprocess, sys
    code = (
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "libc = ctypes.CDLL(None)\n"
        "libc.srandom(1); a = [libc.rand() for _ in range(4)] + [libc.random()]\n"
        "libc.srandom(1)\n"
        "from __graft_entry__ import load_package\n"
        "hb = load_package().Hobbit(0)\n"
        "d = hb.fill_splitmix(1 << 12, 3); hb.sync()\n"
        "b = [libc.rand() for _ in range(4)] + [libc.random()]\n"
        "hb.close()\n"
        "assert a == b and a[0] == 1804289383, (a, b)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_host_mirror_open_standard(oracle):
    """open_standard(poly, x, MT, _tensor, K, vt, ps) of the C++ mirror (reference signature, src/Our_PC.hpp:13) on test_PC's own
    input sequence: transcripts equal the oracle's open_standard, all of the reference's exit(-1) checks pass, ps is the
    reference's proof-size accounting."""
    import ctypes
    from __graft_entry__ import PKG, build_host
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    libc = ctypes.CDLL(None)
    N, K, queries, seed = 1 << 20, 32, 5900, 1234
    trs = N // (K << 11)
    R1, logc = (2 * trs).bit_length() - 1, 12
    rounds = R1 + logc + 2 * (R1 + logc) + logc
    q = np.zeros((rounds, 3, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); I = np.zeros((queries, 2), np.uint32)
    roots = np.zeros((4, 32), np.uint8); checks = np.zeros(5, np.int32); ps = ctypes.c_double(0)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_test_pc_open.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_uint] + [ctypes.c_void_p] * 6
    assert lib.hobbit_host_test_pc_open(N, K, seed, P(q), P(r), P(I), P(roots), P(checks), ctypes.byref(ps)) == rounds
    # the proof on the wire: serialise -> parse -> serialise is the identity, a truncated or mis-tagged buffer is rejected; it carries at
    # least the replies and all (not de-duplicated) Merkle paths
    lib.hobbit_host_wire_roundtrip.restype = ctypes.c_long
    wire = lib.hobbit_host_wire_roundtrip(1)
    assert wire > queries * K * 16 + queries * ((N // K).bit_length() - 1) * 32, wire
    lib.hobbit_host_close()
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(N.bit_length() - 1)
    libc.srandom(seed); want = oracle.open_standard(poly, K, trs, x, queries)
    assert checks.tolist() == [1, 1, 1, 1, 1]
    assert np.array_equal(q, want["poly"]) and np.array_equal(r, want["r"]) and np.array_equal(I, want["I"])
    assert np.array_equal(roots[:2], want["roots"])
    assert np.array_equal(roots[2], want["sp_c"]["whir_root"]) and np.array_equal(roots[3], want["sp_f"]["whir_root"])
    # proof size: at least the replies (5900 x 32 F) plus something for every other message, and well under the tensor itself
    assert queries * K * 16 / 1024.0 < ps.value < 4 * queries * K * 16 / 1024.0


def test_host_mirror_test_pc_option1(oracle):
    """test_PC(N, 1, K) through the C++ mirror (commit_standard + open_standard with linear_time == false, reference signatures): the root
    equals the REAL reference's (commit_rs.npz), the opening's transcript equals the oracle's, the reference's exit(-1) checks pass."""
    import ctypes
    from __graft_entry__ import PKG, build_host
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    libc = ctypes.CDLL(None)
    N, K, queries, seed, trs = 1 << 20, 16, 790, 77, 128
    cols = 2 * (N // K) // trs; logc = cols.bit_length() - 1; logr = 8; logt = 7
    maxr = 11 + logr + logr + (logt + logc) + logc
    q = np.zeros((maxr, 3, 2), np.uint64); r = np.zeros((maxr, 2), np.uint64); I = np.zeros((queries, 2), np.uint32)
    root = np.zeros(32, np.uint8); roots = np.zeros((2, 32), np.uint8); checks = np.zeros(3, np.int32); ps = ctypes.c_double(0)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_test_pc_rs_open.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_uint] + [ctypes.c_void_p] * 7
    rounds = lib.hobbit_host_test_pc_rs_open(N, K, seed, P(root), P(q), P(r), P(I), P(roots), P(checks), ctypes.byref(ps))
    lib.hobbit_host_wire_roundtrip.restype = ctypes.c_long
    assert lib.hobbit_host_wire_roundtrip(2) > queries * K * 16
    lib.hobbit_host_close()
    assert np.array_equal(root, gold("commit_rs")["crs_%d_%d_root" % (N, K)])
    oracle.rng_reset(); poly = oracle.generate_randomness(N)
    x = oracle.generate_randomness(N.bit_length() - 1)
    libc.srandom(seed); want = oracle.open_standard_rs(poly, K, trs, x, queries)
    assert checks.tolist() == [1, 1, 1] and rounds == want["poly"].shape[0]
    assert np.array_equal(q[:rounds], want["poly"]) and np.array_equal(r[:rounds], want["r"]) and np.array_equal(I, want["I"])
    assert np.array_equal(roots[0], want["cf_root"]) and np.array_equal(roots[1], want["sp_f"]["whir_root"])
    assert queries * K * 16 / 1024.0 < ps.value


@pytest.mark.parametrize("lookups", [0, 1])
def test_host_mirror_gate_consistency_stream(oracle, lookups):
    """prove_gate_consistency / prove_gate_consistency_lookups with the reference's signatures (stream_descriptor, r, vt, ps) through the C++
    mirror, the trace served by hobbit_read_trace_hook chunk by chunk (uploaded per read, reset and re-read for the Peval pass): challenges,
    folded values and the closing sumcheck equal the oracle's; the reference's exit(-1) checks pass; ps is the reference's accounting."""
    import ctypes
    from __graft_entry__ import PKG, build_host
    from oracle.pyoracle import gate_standard_inputs
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    libc = ctypes.CDLL(None)
    B, nch, seed = 1 << 12, 8, 99
    n = B * nch; logB = 12; lR = 3
    if lookups:
        L, R, O, S = lookup_trace(B, nch, 40)
    else:
        parts = [gate_standard_inputs(B, 50 + c) for c in range(nch)]
        L, R, O = [np.concatenate([p[i] for p in parts]) for i in range(3)]
        S = np.concatenate([p[3][:, 0] for p in parts]).astype(np.int32)
    r = splitmix_field(logB, 3); lr = splitmix_field(2, 77)
    nt = 9 if lookups else 6
    Rout = np.zeros((nch, 2), np.uint64); fin = np.zeros((nt, 2), np.uint64); q2 = np.zeros((lR, 3, 2), np.uint64); checks = np.zeros(5, np.int32); ps = ctypes.c_double(0)
    Pv = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_gate_stream.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint] + [ctypes.c_void_p] * 11
    assert lib.hobbit_host_gate_stream(n, B, lookups, seed, Pv(L), Pv(R), Pv(O), Pv(S), Pv(r), Pv(lr), Pv(Rout), Pv(fin), Pv(q2), Pv(checks), ctypes.byref(ps)) == nt
    lib.hobbit_host_close()
    libc.srandom(seed)
    want = oracle.gate_consistency_lookups_stream(L, R, O, S, B, r, lr) if lookups else oracle.gate_consistency_stream(L, R, O, S, B, r)
    assert checks.tolist() == [1] * 5
    assert np.array_equal(Rout, want["R"]) and np.array_equal(fin, want["fin9" if lookups else "fin6"]) and np.array_equal(q2, want["q2"])
    per_chunk = 15 if lookups else 12
    want_ps = ((5 if lookups else 4) + (nch - 1) * per_chunk + logB * 5 + (3 * lR + 2) + 5) * 16 / 1024.0
    assert abs(ps.value - want_ps) < 1e-9


# ---- multi-GPU building blocks on one GPU ------------------------------------------------------
def test_sharded_commit_hip_ops_world1(hb, oracle):
    """The per-rank GPU operations of the chunk-sharded commit (tensor codes of the local chunks,
    inner digests in leaf order, Merkle-Damgard chain, subtree) at world size 1 must reproduce
    commit_standard; the collective pattern itself is covered by tests/test_dist_gloo.py."""
    import torch
    from __graft_entry__ import load_package
    mod = load_package()
    N, K = 1 << 20, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    want, _ = oracle.commit_standard(poly, K, trs, 1)
    plan = mod.parallel.ShardPlan(N, K, trs, 1)
    d = hb.to_device(poly)
    res = mod.parallel.sharded_commit(mod.parallel.HipOps(hb, torch.device("cuda", 0)), None, plan, 0, (d.ptr, K))
    levels = mod.parallel.assemble_levels(plan, [res["subtree"].cpu().numpy()], res["top"])
    assert np.array_equal(levels, want)
    # emulate two ranks on the one GPU: digests of the odd/even chunks chained in global order
    plan2 = mod.parallel.ShardPlan(N, K, trs, 2)
    M = plan2.M
    digs = {}
    for r in range(2):
        loc = np.concatenate([poly[i * M:(i + 1) * M] for i in plan2.chunks_of(r)])
        dl = hb.to_device(loc)
        digs[r] = mod.parallel.HipOps(hb, torch.device("cuda", 0)).inner_digests((dl.ptr, K // 2), plan2).cpu().numpy()
    full = np.stack([digs[i % 2][i // 2] for i in range(K)])          # [K, M, 32] in global chunk order
    leaves = np.zeros((M, 32), np.uint8)
    for i in range(K):
        leaves = oracle.blake3_64(np.concatenate([full[i], leaves], axis=1))
    assert np.array_equal(leaves, want[:M])



def test_relay_commit_hip_ops(hb, oracle):
    """GPU operations of the chain-relay commit (hobbit_tensorcode_chunks on a contiguous chunk range, hobbit_leaf_chain_relay per block of
    leaf slots with the running state in / out, tree on the last rank): world size 1 through sharded_commit_relay, then four emulated ranks
    on the one GPU -- each rank's shard chained on top of the previous rank's state, block by block -- against commit_standard.  The
    hand-over protocol itself runs under gloo at world sizes 2 and 4 in tests/test_dist_gloo.py."""
    import torch
    from __graft_entry__ import load_package
    mod = load_package()
    N, K = 1 << 20, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    want, _ = oracle.commit_standard(poly, K, trs, 1)
    dev = torch.device("cuda", 0)
    plan = mod.parallel.ShardPlan(N, K, trs, 1, contiguous=True)
    d = hb.to_device(poly)
    res = mod.parallel.sharded_commit_relay(mod.parallel.HipOps(hb, dev), None, plan, 0, (d.ptr, K))
    assert np.array_equal(res["levels"].cpu().numpy(), want) and np.array_equal(res["root"], want[-1])
    # emulated worlds: 4 ranks x 8 blocks, and 8 ranks x relay_blocks(8) = 64 blocks -- the shape the driver's SCALE run launches (bench.py
    # --gpus 8: K/G = 4 chunks per rank, sharded_commit_relay's default block count)
    assert mod.parallel.relay_blocks(8) == 64 and mod.parallel.relay_blocks(2) == 16 and mod.parallel.relay_blocks(4) == 32
    for G, blocks in ((4, 8), (8, mod.parallel.relay_blocks(8))):
        planG = mod.parallel.ShardPlan(N, K, trs, G, contiguous=True)
        M = planG.M; per = M // blocks
        state = None
        for r in range(G):
            ops = mod.parallel.HipOps(hb, dev)
            loc = np.concatenate([poly[i * M:(i + 1) * M] for i in planG.chunks_of(r)])
            dl = hb.to_device(loc)
            ops.encode_local((dl.ptr, K // G), planG)
            out = ops.empty_state(M) if r < G - 1 else None
            lv = ops.empty_state(2 * M) if r == G - 1 else None
            for b in range(blocks):
                ops.chain_block(planG, b * per, per, state[b * per:(b + 1) * per] if state is not None else None, out[b * per:(b + 1) * per] if out is not None else None, lv)
            state = out
        assert np.array_equal(ops.tree_full(lv, M).cpu().numpy(), want), G
        pos = np.array([0, 5, M - 1, M // 2 + 3], np.uint64)
        got = ops.tree_paths(lv, pos, M)
        for k, p in enumerate(pos):
            assert np.array_equal(got[k], oracle.open_tree_blake(want, M, int(p), 0, 0))


def test_sharded_commit_result_goes_stale(hb, oracle):
    """A HipOps object keeps the tensor shard and the tree in retained buffers: a second commit overwrites them, and sharded_open must refuse
    the first commit's result instead of serving the second commitment's replies and paths against the first root."""
    import torch
    from __graft_entry__ import load_package
    mod = load_package()
    N, K = 1 << 18, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    dev = torch.device("cuda", 0)
    plan = mod.parallel.ShardPlan(N, K, trs, 1, contiguous=True)
    ops = mod.parallel.HipOps(hb, dev)
    d = hb.to_device(poly); d2 = hb.to_device(poly[::-1].copy())
    ops.set_local_chunks((d.ptr, K))
    res_a = mod.parallel.sharded_commit_relay(ops, None, plan, 0, (d.ptr, K))
    res_b = mod.parallel.sharded_commit_relay(ops, None, plan, 0, (d2.ptr, K))
    assert res_a["gen"] + 1 == res_b["gen"] and not np.array_equal(res_a["root"], res_b["root"])
    x = oracle.generate_randomness(18)
    with pytest.raises(RuntimeError, match="one live commitment"):
        mod.parallel.sharded_open(ops, None, plan, 0, res_a, x, 64)

@pytest.mark.parametrize("logn,K", [(20, 32), (22, 32), (21, 16), (24, 32)])
def test_open_standard_transcript_vs_reference(hb, logn, K, monkeypatch):
    """The orchestration of open_standard / recursive_prover_Spielman against the REAL reference, hash for hash: oracle/gen_open_transcript.py ran the
    reference's own commit_standard + open_standard (test_PC's inputs and libc stream) under the call-through mimc_hash recorder until the process
    died on the first SHA3 call (the prebuilt library is not linked, nothing stands in for it): 237-267 transcript hashes -- every round of P1..P4,
    shockwave_prove(C_c)'s sumcheck and prove_fft, the first hashes of its _whir_prove (tests/golden/open_transcripts.json).  The library records its
    own transcript of the same run here.  The reference runs shockwave_prove(C_c) BEFORE P5, the library (one thread) after it: the reference's
    sequence must be the library's with exactly P5's block cut out."""
    import json
    fix = json.load(open(os.path.join(GOLD, "open_transcripts.json")))["test_pc_2e%d_K%d" % (logn, K)]
    ref = np.array(fix["records"], np.uint64).reshape(-1, 6)
    assert ref.shape[0] == fix["count"] > 200
    monkeypatch.setenv("HOBBIT_OPEN_THREADS", "0")               # one thread: a deterministic order of the recorded hashes
    N = 1 << logn; trs = N // (K << 11)
    hb.rng_reset()
    poly = hb.generate_randomness(N)                             # src/Our_PC.cpp:758
    hb.expander_init_store(trs)                                  # :813
    c = hb.commit_standard(poly, K, trs, 1)
    x = hb.generate_randomness(logn)                             # :818
    hb.lib.hobbit_transcript_record(1)
    res = hb.open_standard(poly, c, x, 5900, want_paths=False)
    hb.lib.hobbit_transcript_record(0)
    n = hb.lib.hobbit_transcript_count()
    mine = np.zeros((n, 6), np.uint64)
    hb.lib.hobbit_transcript_read(mine.ctypes.data, n)
    c.free()
    assert res["checks"].tolist() == [1, 1, 1]
    cols = 2 * (N // K) // trs
    R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1; R3 = R1 + logc
    head = 3 * (R1 + logc + 2 * R3) + 2 * 4                      # P1..P4: three coefficients per round, two closing hashes per sumcheck
    assert np.array_equal(mine[:head], ref[:head]), "P1..P4 differ from the reference at record %d" % int(np.nonzero((mine[:head] != ref[:head]).any(axis=1))[0][0])
    rest = ref[head:]                                            # shockwave_prove(C_c) in the reference
    hits = [j for j in range(head, n - len(rest) + 1) if np.array_equal(mine[j], rest[0])]
    assert hits, "the reference's shockwave_prove(C_c) transcript does not start anywhere in the library's"
    j0 = hits[0]
    assert np.array_equal(mine[j0:j0 + len(rest)], rest), "shockwave_prove(C_c) differs from the reference"
    assert j0 - head == 3 * logc + 2, "between P4 and shockwave_prove(C_c) the library hashes exactly P5 (prove_fft_matrix): %d records" % (j0 - head)


@pytest.mark.parametrize("logn,K", [(20, 32), (22, 16)])
def test_open_standard_rs_transcript_vs_reference(hb, logn, K, monkeypatch):
    """The same pin for test_PC option 1 (RS x RS, tensor_row_size = 128; open_standard's !linear_time branch -> recursive_prover_RS): the REAL
    reference's own test_PC(N, 1, K), called as it is, recorded until it died on its first SHA3 call -- 228 / 267 transcript hashes
    (tests/golden/open_transcripts.json, driver_test_pc1_*).  Every one of them must appear, in order, in the library's own transcript of the same
    run; the library may hash blocks in between that the reference only reaches later (at most one such block)."""
    import json
    fix = json.load(open(os.path.join(GOLD, "open_transcripts.json")))["driver_test_pc1_2e%d_K%d" % (logn, K)]
    ref = np.array(fix["records"], np.uint64).reshape(-1, 6)
    assert ref.shape[0] == fix["count"] > 200
    monkeypatch.setenv("HOBBIT_OPEN_THREADS", "0")
    N = 1 << logn
    hb.rng_reset()
    poly = hb.generate_randomness(N)                             # src/Our_PC.cpp:758
    c = hb.commit_standard(poly, K, 128, 0)                      # :765-767
    x = hb.generate_randomness(logn)                             # :775
    hb.lib.hobbit_transcript_record(1)
    hb.open_standard_rs(poly, c, x, 790, want_paths=False)
    hb.lib.hobbit_transcript_record(0)
    n = hb.lib.hobbit_transcript_count()
    mine = np.zeros((n, 6), np.uint64)
    hb.lib.hobbit_transcript_read(mine.ctypes.data, n)
    c.free()
    i = j = 0; gaps = []
    while i < len(ref) and j < n:
        if np.array_equal(ref[i], mine[j]):
            i += 1; j += 1
            continue
        hit = [t for t in range(j + 1, n) if np.array_equal(mine[t], ref[i])]
        assert hit, "reference record %d of %d is nowhere in the library's transcript (library record %d of %d)" % (i, len(ref), j, n)
        gaps.append((j, hit[0] - j)); j = hit[0]
    assert i == len(ref), "the library's transcript ends before the reference's recorded prefix (%d of %d matched)" % (i, len(ref))
    assert len(gaps) <= 1, "more than one block out of order: %s" % gaps


def _two_rank_gpu_worker(rank, world, port, N, K, queries, seed, q, exchange="relay"):
    """one rank of test_two_process_relay_commit_and_open_on_one_gpu: real HipOps on cuda:0, gloo transport staged through the host"""
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from staged_dist import StagedDist
    from __graft_entry__ import load_package
    mod = load_package()
    orc = pyoracle.Oracle()
    trs = N // (K << 11)
    orc.rng_reset(); poly = orc.generate_randomness(N); orc.expander_init_store(trs)
    x = orc.generate_randomness(N.bit_length() - 1)
    hb = mod.Hobbit(0)
    hb.upload_graphs(trs, graphs_from(orc, trs))
    plan = mod.parallel.ShardPlan(N, K, trs, world, contiguous=(exchange == "relay"))
    M = plan.M
    own = plan.chunks_of(rank)
    d_local = hb.to_device(np.concatenate([poly[i * M:(i + 1) * M] for i in own]))
    ops = mod.parallel.HipOps(hb, torch.device("cuda", 0))
    sd = StagedDist()
    commit = mod.parallel.sharded_commit_relay if exchange == "relay" else mod.parallel.sharded_commit
    out = []
    for it in range(2):                                   # twice: the retained buffers of HipOps are re-used by the second commit
        res = commit(ops, sd, plan, rank, (d_local.ptr, len(own)))
        ops.set_local_chunks((d_local.ptr, len(own)))
        ctypes.CDLL(None).srandom(seed)                   # (only rank 0's state matters: it draws the value every rank re-seeds with)
        o = mod.parallel.sharded_open(ops, sd, plan, rank, res, x, queries)
        keep = {k: o[k] for k in ("cols", "rows", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots", "checks")}
        keep["sp_c_wq"] = o["sp_c"]["wq"]; keep["sp_f_q1"] = o["sp_f"]["q1"]
        out.append((res["root"], keep))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    hb.close()


@pytest.mark.parametrize("world,exchange", [(2, "relay"), (4, "relay"), (2, "alltoall"), (4, "alltoall")])
def test_two_process_relay_commit_and_open_on_one_gpu(hb, oracle, world, exchange):
    """parallel.py end to end as two / four real processes with the real per-rank GPU operations (HipOps, all on this one GPU) -- the relay commit
    or (exchange = "alltoall") SURVEY 8e's digest all-to-all with per-rank subtrees and the all-gather of their roots, then: relay commit
    (16 / 32 blocks handed from rank to rank), the integer all-reduce of the partial aggregates with its shift and fold kernels, the replicated
    open, replies all-gathered from the two tensor shards, paths broadcast from the tree's owner -- over gloo with the device tensors staged
    through the host (tests/staged_dist.py: RCCL wants one device per rank).  Equal, twice in a row, to the single-process commit_standard +
    open_standard on the same inputs and libc stream."""
    import ctypes
    import multiprocessing as mp
    import queue as _q
    import socket
    import time as _t
    N, K, queries, seed = 1 << 20, 32, 700, 4242          # (4 ranks + this process = 5 processes on the card: the box allows 6)
    trs = N // (K << 11)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_gpu_worker, args=(r, world, port, N, K, queries, seed, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    deadline = _t.time() + 400
    while len(got) < world:
        try:
            got.append(q.get(timeout=2))
        except _q.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: exit codes %s" % [p.exitcode for p in procs]
            assert _t.time() < deadline, "timeout waiting for the ranks"
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(N.bit_length() - 1)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    c = hb.commit_standard(poly, K, trs, 1)
    libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
    libc.srandom(seed); libc.srandom(ctypes.c_uint(libc.random() & 0xFFFFFFFF))
    want = hb.open_standard(poly, c, x, queries, want_paths=True)
    root = c.root(); c.free()
    for rank, outs in got:
        for it, (r, o) in enumerate(outs):
            assert np.array_equal(r, root), (rank, it)
            for k in ("cols", "rows", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots", "checks"):
                assert np.array_equal(o[k], want[k]), (rank, it, k)
            assert np.array_equal(o["sp_c_wq"], want["sp_c"]["wq"]) and np.array_equal(o["sp_f_q1"], want["sp_f"]["q1"]), (rank, it)


def _elastic_rank_gpu_worker(rank, world, port, N, B, opt, q):
    """one rank of test_two_process_sharded_elastic_commit_on_one_gpu"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from staged_dist import StagedDist
    from __graft_entry__ import load_package
    mod = load_package()
    orc = pyoracle.Oracle()
    plan = mod.parallel.ElasticPlan(N, B, opt, world)
    hb = mod.Hobbit(0)
    orc.rng_reset()
    if opt == 2:
        orc.expander_init_store(plan.trs)
        hb.upload_graphs(plan.trs, graphs_from(orc, plan.trs)) if plan.trs > 13 else hb.expander_init_store(plan.trs)
    live = []

    def source(c):
        buf = hb.to_device(splitmix_field(B, 9000 + c)); live.append(buf); del live[:-2]
        return buf.ptr
    ops = mod.parallel.ElasticHipOps(hb, torch.device("cuda", 0), rank)
    res = mod.parallel.sharded_commit(ops, StagedDist(), plan, rank, source)
    q.put((rank, res["subtree"].cpu().numpy(), np.asarray(res["top"]), np.asarray(res["root"])))
    dist.barrier()
    dist.destroy_process_group()
    hb.close()


@pytest.mark.parametrize("opt", [1, 2])
def test_two_process_sharded_elastic_commit_on_one_gpu(hb, oracle, opt):
    """BASELINE config 5's shape in small: the streaming Elastic_PC commit sharded by groups of four chunks over TWO real processes with the real
    per-rank GPU operations (ElasticHipOps: hobbit_elastic_push_inner per group, digest all-to-all, chain, per-rank subtree, all-gather of the
    subtree roots) on a stream whose chunks all differ -- gloo transport staged through the host -- against the oracle's streaming commit."""
    import ctypes
    import multiprocessing as mp
    import queue as _q
    import socket
    import time as _t
    from __graft_entry__ import load_package
    mod = load_package()
    N, B, world = 1 << 20, 1 << 14, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_elastic_rank_gpu_worker, args=(r, world, port, N, B, opt, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    deadline = _t.time() + 400
    while len(got) < world:
        try:
            got.append(q.get(timeout=2))
        except _q.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: exit codes %s" % [p.exitcode for p in procs]
            assert _t.time() < deadline, "timeout waiting for the ranks"
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
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
    assert np.array_equal(levels[:T - 1], want[:T - 1]) and np.array_equal(levels[T:], want[T:cnt])
    for g in got:
        assert np.array_equal(g[3], want[cnt - 1]) and np.array_equal(g[2], got[0][2])


def test_sharded_open_hip_ops_world1(hb, oracle):
    """The per-rank GPU operations of the multi-GPU open (local aggregate, field sum of partials, open from the aggregate, replies
    from the tensor shard, subtree paths) at world size 1 against the single-process hobbit_open_standard and the oracle; the
    collective pattern is covered by tests/test_dist_gloo.py.  Also emulates two ranks' partial aggregates on the one GPU."""
    import ctypes
    import torch
    from __graft_entry__ import load_package
    mod = load_package()
    libc = ctypes.CDLL(None)
    N, K, queries = 1 << 20, 32, 5900
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(N.bit_length() - 1)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    plan = mod.parallel.ShardPlan(N, K, trs, 1)
    d = hb.to_device(poly)
    ops = mod.parallel.HipOps(hb, torch.device("cuda", 0))
    cres = mod.parallel.sharded_commit(ops, None, plan, 0, (d.ptr, K))
    ops.set_local_chunks((d.ptr, K))
    libc.srandom(31); got = mod.parallel.sharded_open(ops, None, plan, 0, cres, x, queries)
    c = hb.commit_standard(poly, K, trs, 1)
    libc.srandom(31); want = hb.open_standard(poly, c, x, queries, want_paths=True)
    for k in ("cols", "rows", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots", "checks"):
        assert np.array_equal(got[k], want[k]), k
    for sp in ("sp_c", "sp_f"):
        for k in want[sp]:
            assert np.array_equal(got[sp][k], want[sp][k]), (sp, k)
    c.free()
    # two ranks' partial aggregates, summed, equal the aggregate
    plan2 = mod.parallel.ShardPlan(N, K, trs, 2)
    M = plan2.M
    beta = hb.precompute_beta(x[:5])
    parts = []
    for r in range(2):
        own = plan2.chunks_of(r)
        dl = hb.to_device(np.concatenate([poly[i * M:(i + 1) * M] for i in own]))
        o2 = mod.parallel.HipOps(hb, torch.device("cuda", 0)); o2.set_local_chunks((dl.ptr, len(own)))
        parts.append(o2.aggregate_local(np.ascontiguousarray(beta[own]), plan2))
    total = ops.sum_vectors(parts).cpu().numpy().view(np.uint64)
    assert np.array_equal(total, oracle.aggregate(poly, beta))
    # ... and so does the exchange the open uses up to 8 ranks: partials shifted by -2^60, a plain int64 sum (what the all-reduce computes),
    # folded back into the field -- here with eight "ranks" of extreme values as well (8 (p - 1) is the largest sum there can be)
    a, b = parts[0].clone(), parts[1].clone()
    for t in (a, b):
        ops.bias_words(t, -(1 << 60))
    summed = a + b                                          # torch int64 addition: two's complement, as RCCL's sum
    assert np.array_equal(ops.fold_words(summed, 2 << 60).cpu().numpy().view(np.uint64), oracle.aggregate(poly, beta))
    P = (1 << 61) - 1
    edge = np.array([P - 1, 0, 1, P - 1, 12345, P - 2], np.uint64)
    acc = torch.zeros(len(edge), dtype=torch.int64, device="cuda")
    for _ in range(8):
        t = torch.from_numpy(edge.view(np.int64)).to("cuda"); ops.bias_words(t, -(1 << 60)); acc += t
    want = np.array([(8 * int(v)) % P for v in edge], np.uint64)
    assert np.array_equal(ops.fold_words(acc, 8 << 60).cpu().numpy().view(np.uint64), want)


# ---- Elastic_PC streaming commit + long-row tensor codes ---------------------------------------
@pytest.mark.parametrize("opt", [1, 2])
def test_elastic_commit_vs_golden(hb, opt):
    g = gold("elastic")
    B = 1 << 14
    hb.rng_reset()
    lv = hb.elastic_commit(1 << 18, B, opt)
    T = 4 * B
    assert np.array_equal(lv[-1], g["el_%d_root" % opt])
    assert np.array_equal(dg(lv[:T - 1]), g["el_%d_leaves_dg" % opt])      # leaf T-1 is undefined in the reference
    assert np.array_equal(dg(lv[T:]), g["el_%d_upper_dg" % opt])


def test_elastic_stream_generator_matches_oracle(hb, oracle):
    assert np.array_equal(hb.read_stream_PC(4096), oracle.read_stream_pc(4096))


@pytest.mark.parametrize("logc,trs,lin", [(13, 16, 1), (14, 16, 0), (15, 64, 1), (15, 4, 1)])
def test_tensorcode_long_rows_vs_oracle(hb, oracle, logc, trs, lin):
    """row codes longer than 4096 (Elastic_PC opt 2 uses 32768): split FFT = R strided FFT-4096 + combine"""
    M = trs << (logc - 1)
    oracle.rng_reset(); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs)) if trs > 13 else hb.expander_init_store(trs)
    msg = splitmix_field(M, 500 + logc)
    assert np.array_equal(hb.compute_tensorcode(msg, trs, lin), oracle.compute_tensorcode(msg, trs, lin))


def test_elastic_commit_2e22_opt2_vs_oracle(hb, oracle):
    """C5-shaped case at reduced size: N = 2^22, B = 2^20 (trs = 64, 32768-point rows), one 4-chunk group"""
    B = 1 << 20
    oracle.rng_reset()
    want = oracle.elastic_commit(1 << 22, B, 2)
    hb.upload_graphs(64, graphs_from(oracle, 64))
    e = hb.lib.hobbit_elastic_begin
    import ctypes
    h = ctypes.c_void_p()
    hb._chk(hb.lib.hobbit_elastic_begin(hb.ctx, B, 64, 1, 1, ctypes.byref(h)))
    chunk = hb.to_device(oracle.read_stream_pc(B))
    for _ in range(4):
        hb._chk(hb.lib.hobbit_elastic_push(hb.ctx, h, chunk.ptr))
    lv = hb.alloc(32 * 8 * B)
    hb._chk(hb.lib.hobbit_elastic_finish(hb.ctx, h, lv.ptr))
    got = hb.to_host(lv, (8 * B - 1, 32), np.uint8)
    hb.lib.hobbit_elastic_free(h)
    T = 4 * B
    assert np.array_equal(got[:T - 1], want[:T - 1]) and np.array_equal(got[T:], want[T:])


# ---- code-membership / FFT-as-sumcheck proofs ----------------------------------------------------
def test_codeproofs_vs_golden(hb):
    import ctypes
    g = gold("codeproofs")
    for n in (16, 64, 256, 1024):
        hb.rng_reset(); ln = hb.expander_init_store(n)
        k = (2 * n).bit_length() - 1
        beta = hb.precompute_beta(splitmix_field(k, 300 + n))
        A = hb.evaluate_parity_matrix(beta, n)
        assert np.array_equal(A if n <= 64 else dg(A), g["pm_%d" % n]), n
        cw = hb.encode_monolithic(splitmix_field(n, 310 + n))
        r1 = g["plc_%d_r1" % n]                        # the challenge vector the reference drew (libc, seed 777+n)
        ctypes.CDLL(None).srandom(777 + n)
        assert np.array_equal(hb.generate_randomness(k), r1)
        res = hb.prove_linear_code(cw, n, r1)
        for kk, v in res.items():
            assert np.array_equal(v, g["plc_%d_%s" % (n, kk)]), (n, kk)
        # a codeword satisfies the parity check: claimed sum q0(0)+q0(1) = 0
        q0 = res["poly"][0].astype(object)
        assert [(int(q0[0][i]) + int(q0[1][i]) + 2 * int(q0[2][i])) % P for i in range(2)] == [0, 0]
    for nn in (1, 2, 5, 10):
        rx = splitmix_field(nn, 320 + nn)
        for ifft in (0, 1):
            t = hb.phiGInit(rx, (7, 3) if ifft else (1, 0), bool(ifft))
            assert np.array_equal(t if nn <= 5 else dg(t), g["phig_%d_%d" % (nn, ifft)]), (nn, ifft)
    M = splitmix_field(64 * 256, 330).reshape(64, 256, 2)
    # prepare_matrix(M, r) folds along each ROW of M; the device op folds the row index of columns: feed M^T
    assert np.array_equal(hb.prepare_matrix_cols(np.ascontiguousarray(M.transpose(1, 0, 2)), splitmix_field(8, 331)), g["prepmat"])
    res = hb.prove_fft(splitmix_field(512, 340), splitmix_field(10, 341))
    for kk, v in res.items():
        assert np.array_equal(v, g["pfft_%s" % kk]), kk
    gm = gold("codeproofs_matrix")
    res = hb.prove_fft_matrix(splitmix_field(16 * 256, 350).reshape(16, 256, 2), splitmix_field(9 + 4, 351))
    for kk, v in res.items():
        assert np.array_equal(v, gm["pfm_%s" % kk]), kk


def test_phi_g_large_vs_oracle(hb, oracle):
    rx = splitmix_field(16, 77)
    assert np.array_equal(hb.phiGInit(rx), oracle.phi_g_init(rx))
    assert np.array_equal(hb.phiGInit(rx, (5, 9), True), oracle.phi_g_init(rx, (5, 9), True))


# ---- Our_PC open without the inner shockwave/WHIR PCS ---------------------------------------------
@pytest.mark.parametrize("N,K", [(1 << 20, 32), (1 << 22, 32)])
def test_open_core_vs_oracle(hb, oracle, N, K):
    """open_standard + recursive_prover_Spielman minus shockwave/WHIR: five sumcheck transcripts, the queries
    (libc order), replies and Merkle paths, bit-exact against the oracle's restatement; the reference's own
    consistency checks must hold."""
    import ctypes
    libc = ctypes.CDLL(None)
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(N.bit_length() - 1)
    lv, T = oracle.commit_standard(poly, K, trs, 1, want_tensor=True)
    queries = 5900
    libc.srandom(4242); want = oracle.open_core(poly, K, trs, x, queries, tensor=T)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    c = hb.commit_standard(poly, K, trs, 1)
    libc.srandom(4242); got = hb.open_core(poly, c, x, queries)
    assert want["checks"].tolist() == [1, 1, 1] and got["checks"].tolist() == [1, 1, 1]
    assert np.array_equal(got["I"], want["I"])
    assert np.array_equal(got["reply"], want["reply"])
    for k in ("scalars", "poly", "r", "vr", "fin", "roots"):
        assert np.array_equal(got[k], want[k]), k
    M = N // K
    for q in (0, 17, queries - 1):
        assert np.array_equal(got["paths"][q], oracle.open_tree_blake(lv, M, int(got["I"][q, 0]), int(got["I"][q, 1]), 4096))
    c.free()


SP_KEYS = ("I", "q1", "r1", "vr1", "fin1", "q2", "r2", "vr2", "fin2", "iters", "wq", "wa", "wroots", "wscal", "wchecks", "whir_root",
           "reply", "paths", "qn", "qidx", "qreply", "qpaths", "final_pb")


@pytest.mark.parametrize("N,K", [(1 << 18, 32), (1 << 20, 32), (1 << 22, 32), (1 << 24, 32), (1 << 22, 16), (1 << 23, 64), (1 << 21, 4)])
def test_open_standard_vs_oracle(hb, oracle, N, K):
    """The whole prover side of open_standard (src/Our_PC.cpp:604-661): the core above followed by
    shockwave_prove(C_c, .) and shockwave_prove(C_f, .) (src/PC_utils.cpp:368,385) with their WHIR proofs; every transcript
    bit-exact against the oracle, libc draws in the reference's order across all three stages."""
    import ctypes
    libc = ctypes.CDLL(None)
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(N.bit_length() - 1)
    queries = 5900
    small = N <= (1 << 20)                                      # replies and paths too where the oracle's commit takes a second or two
    lv, T = oracle.commit_standard(poly, K, trs, 1, want_tensor=True) if small else (None, None)
    libc.srandom(777); want = oracle.open_standard(poly, K, trs, x, queries, tensor=T)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    c = hb.commit_standard(poly, K, trs, 1)
    libc.srandom(777); got = hb.open_standard(poly, c, x, queries, want_paths=small)
    assert want["checks"].tolist() == [1, 1, 1] and got["checks"].tolist() == [1, 1, 1]
    for k in ("I", "scalars", "poly", "r", "vr", "fin", "roots"):
        assert np.array_equal(got[k], want[k]), k
    if small:
        assert np.array_equal(got["reply"], want["reply"])
        M = N // K
        for q in range(0, queries, 97):
            assert np.array_equal(got["paths"][q], oracle.open_tree_blake(lv, M, int(got["I"][q, 0]), int(got["I"][q, 1]), 4096)), q
    for sp in ("sp_c", "sp_f"):
        has_whir = int(want[sp]["iters"][0]) > 0                 # none for a width of 256 (C_f at 2^18): src/Virgo.cpp:479-483
        assert want[sp]["wchecks"].tolist() == ([1, 1] if has_whir else [0, 0]), sp
        for k in SP_KEYS:
            assert np.array_equal(got[sp][k], want[sp][k]), (sp, k)
    c.free()


@pytest.mark.gpu
def test_execution_modes_bit_identical(hb, monkeypatch):
    """The step as a dependency graph -- chunk groups' layout changes on a side stream beside the next group's row FFT
    (HOBBIT_COMMIT_PIPE), shockwave_prove(C_c) on a helper context from a second host thread with its libc draws taken first
    (HOBBIT_OPEN_THREADS), the inner commitments and the query answers on a third stream (HOBBIT_OPEN_COMMITS_SIDE, HOBBIT_OPEN_QUERIES_SIDE), P3 beside
    P1/P2 (HOBBIT_OPEN_P3_THREAD), P3 against its second table as a sparse list (HOBBIT_OPEN_SPARSE_P3) --
    and the plain list on one stream and one thread must produce the same commitment and the same transcript, message for message, and
    leave the libc generator in the same state."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, K = 1 << 24, 32
    trs = N // (K << 11)
    d = hb.fill_splitmix(N, 99)
    hb.rng_reset(); hb.expander_init_store(trs)
    x = splitmix_field(24, 5)
    runs = []
    for env in ({}, {"HOBBIT_COMMIT_PIPE": "0", "HOBBIT_OPEN_THREADS": "0", "HOBBIT_OPEN_SPARSE_P3": "0"}, {"HOBBIT_COMMIT_PIPE": "4", "HOBBIT_OPEN_COMMITS_SIDE": "0"},
                {"HOBBIT_COMMIT_PIPE": "16", "HOBBIT_OPEN_P3_THREAD": "1", "HOBBIT_OPEN_QUERIES_SIDE": "0"}):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            c = hb.commit_standard((d, N), K, trs, 1)
            root = c.root().tobytes()
            libc.srandom(31); a = hb.open_standard((d, N), c, x, 5900, want_paths=True)
            after = int(libc.random())
            c.free()
        runs.append((root, a, after))
    root0, a0, after0 = runs[0]
    assert a0["checks"].tolist() == [1, 1, 1]
    for root, a, after in runs[1:]:
        assert root == root0 and after == after0
        for k in ("I", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots", "checks"):
            assert np.array_equal(a[k], a0[k]), k
        for sp in ("sp_c", "sp_f"):
            for k in a0[sp]:
                assert np.array_equal(a[sp][k], a0[sp][k]), (sp, k)


def test_open_standard_2e26_selfchecks(hb):
    """Full-size property check (no CPU oracle at this size): commit + open of a 2^26-coefficient polynomial generated on the
    device; every consistency check the reference would exit(-1) on must hold, the query replies must be the committed tensor's
    entries, and two runs with the same libc seed must give the same transcript (no race anywhere in ~600 launches)."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, K = 1 << 26, 32
    trs = N // (K << 11)
    d = hb.fill_splitmix(N, 4242)
    hb.rng_reset(); hb.expander_init_store(trs)
    c = hb.commit_standard((d, N), K, trs, 1)
    x = splitmix_field(26, 77)
    libc.srandom(5); a = hb.open_standard((d, N), c, x, 5900, want_paths=True)
    libc.srandom(5); b = hb.open_standard((d, N), c, x, 5900, want_paths=True)
    assert a["checks"].tolist() == [1, 1, 1]
    for sp in ("sp_c", "sp_f"):
        assert a[sp]["wchecks"].tolist() == [1, 1] and int(a[sp]["iters"][0]) >= 3
    for k in ("I", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "roots"):
        assert np.array_equal(a[k], b[k]), k
    for sp in ("sp_c", "sp_f"):
        for k in a[sp]:
            assert np.array_equal(a[sp][k], b[sp][k]), (sp, k)
    # replies are tensor entries: row I[q,1] of chunk i, column I[q,0]
    for q in (0, 1234, 5899):
        col, row = int(a["I"][q, 0]), int(a["I"][q, 1])
        for i in (0, K - 1):
            assert np.array_equal(a["reply"][q, i], c.tensor_row(i, row)[col])
    # Merkle paths really re-hash to the root: the leaf is recomputed on the host from the four tensor rows of all K chunks, then
    # walked up with the reference's left|left rule (even positions only: an odd node never feeds its parent in that tree)
    root = c.root()
    assert a["paths"].shape == (5900, (N // K).bit_length() - 1, 32) and root.shape == (32,)
    done = 0
    for q in range(5900):
        col, row = int(a["I"][q, 0]), int(a["I"][q, 1])
        pos = (row // 4) * 4096 + col
        leaf = leaf_from_tensor(hb, c, col, row)
        if pos % 2 == 0:
            assert np.array_equal(rehash_path(hb, leaf, pos, a["paths"][q]), root), q
            done += 1
        else:
            # odd leaf: the path's first sibling is the even neighbour, which does chain to the root; the leaf itself must still be the stored one
            assert np.array_equal(rehash_path(hb, a["paths"][q][0], pos - 1, a["paths"][q]), root), q
            assert np.array_equal(c.open_tree_blake(col - 1, row)[0], leaf), q          # sibling path of the even neighbour starts with this leaf
        if q >= 40 and done >= 3:
            break
    assert done >= 3
    c.free()


def test_gate_standard_vs_golden(hb):
    """prove_gate_consistency_standard (src/sumcheck.cpp:434-501) against what the real reference produced: pins hobbit_gate_sumcheck"""
    g = gold("gate")
    got = golden_cases.case_gate(hb)
    assert set(got) == set(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k]), k


# ---- degree-4 gate-consistency sumcheck (a22) -------------------------------------------------------
@pytest.mark.parametrize("logn", [4, 10, 11, 14, 18])
def test_gate_sumcheck_vs_oracle(hb, oracle, logn):
    """src/sumcheck.cpp:875-929 over six folded tables; B = 2^18 is the MLP config's chunk size.  The oracle's loop is the
    restatement (not runnable in oracle/_ref: inline in prove_gate_consistency); the reference's own round check must hold."""
    n = 1 << logn
    sel = (np.arange(n) % 3 == 0)
    add = splitmix_field(n, 801); mul = splitmix_field(n, 802)          # folded selectors are full-range after the streaming phase
    tabs = [add, splitmix_field(n, 803), splitmix_field(n, 804), splitmix_field(n, 805), splitmix_field(n, 806), mul]
    a = splitmix_field(4, 807); rand0 = splitmix_field(1, 808)[0]
    claim = oracle.gate_claim(tabs, a)
    want = oracle.gate_sumcheck(tabs, a, rand0, claim)
    got = hb.gate_sumcheck(tabs, a, rand0, claim)
    assert want["check"].tolist() == [1] and got["check"].tolist() == [1]
    for k in ("poly", "r", "fin", "rand", "sum"):
        assert np.array_equal(got[k], want[k]), k
    bad = claim.copy(); bad[0] ^= np.uint64(1)
    assert hb.gate_sumcheck(tabs, a, rand0, bad)["check"].tolist() == [0]


# ---- streaming-sumcheck error terms and folds ----------------------------------------------------
def test_streamfold_vs_golden(hb):
    g = gold("streamfold")
    got = golden_cases.case_streamfold(hb)
    assert set(got) == set(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k]), k


def test_survey_named_exports(hb, oracle):
    """hobbit_leaf_chain (in-out, leaf order), hobbit_axpy_aggregate, hobbit_stream_fold: SURVEY 8(b)'s names over the same kernels"""
    import ctypes
    N, K = 1 << 18, 32; trs = N // (K << 11); M = N // K
    poly, _ = golden_cases.test_pc_inputs(hb, N, K)
    c = hb.commit_standard(poly, K, trs, 1)
    want = c.levels()[:M]
    d = hb.to_device(poly)
    t = hb.alloc(16 * 4 * M * K); lv = hb.alloc(32 * M)
    hb._chk(hb.lib.hobbit_tensorcode_chunks(hb.ctx, d.ptr, M, K, trs, 1, t.ptr))
    hb._chk(hb.lib.hobbit_memset(hb.ctx, lv.ptr, 0, 32 * M))
    hb._chk(hb.lib.hobbit_leaf_chain(hb.ctx, t.ptr, M, K // 2, trs, 1, lv.ptr))                              # first half of the chunks,
    hb._chk(hb.lib.hobbit_leaf_chain(hb.ctx, t.ptr + 16 * 4 * M * (K // 2), M, K // 2, trs, 1, lv.ptr))      # then the rest on top
    got = np.zeros((M, 32), np.uint8); hb._chk(hb.lib.hobbit_memcpy_d2h(hb.ctx, got.ctypes.data, lv.ptr, 32 * M))
    assert np.array_equal(got, want)
    c.free()
    n = 4096
    tabs = [splitmix_field(n, 400 + i) for i in range(8)]
    dt = [hb.to_device(x) for x in tabs]
    coeff = splitmix_field(1, 9)
    hb._chk(hb.lib.hobbit_axpy_aggregate(hb.ctx, dt[0].ptr, coeff.ctypes.data, dt[1].ptr, n))
    acc = np.zeros((n, 2), np.uint64); hb._chk(hb.lib.hobbit_memcpy_d2h(hb.ctx, acc.ctypes.data, dt[1].ptr, 16 * n))
    assert np.array_equal(acc, oracle.f_add(tabs[1], oracle.f_mul(np.repeat(coeff, n, 0), tabs[0])))
    gate = (np.arange(n) * 7 % 5 < 2).astype(np.int32); dg_ = hb.to_device(gate)
    ptrs = (ctypes.c_void_p * 8)(*[dt[i].ptr for i in (0, 2, 3, 4, 5, 6, 7, 7)])
    K3 = np.zeros((3, 2), np.uint64)
    hb._chk(hb.lib.hobbit_stream_fold(hb.ctx, 3, ptrs, dg_.ptr, n, K3.ctypes.data))
    assert np.array_equal(K3, hb.err3p(tabs[0], gate, tabs[2], tabs[3], tabs[4], tabs[5]))


def test_fingerprint_map(hb, oracle):
    """addr + 1 + a value + b freq (the wiring-consistency fingerprints) on the device against the oracle's field operations"""
    n = 5000
    addr, value, freq = [splitmix_field(n, 610 + i) for i in range(3)]
    a, b = splitmix_field(1, 620), splitmix_field(1, 621)
    da, dv, df = hb.to_device(addr), hb.to_device(value), hb.to_device(freq)
    out = hb.alloc(16 * n)
    one = np.zeros_like(addr); one[:, 0] = 1
    base = oracle.f_add(oracle.f_add(addr, one), oracle.f_mul(np.repeat(a, n, 0), value))
    for with_freq in (1, 0):
        hb._chk(hb.lib.hobbit_fingerprint_map(hb.ctx, da.ptr, dv.ptr, df.ptr if with_freq else None, a.ctypes.data, b.ctypes.data, out.ptr, n))
        got = np.zeros((n, 2), np.uint64); hb._chk(hb.lib.hobbit_memcpy_d2h(hb.ctx, got.ctypes.data, out.ptr, 16 * n))
        want = oracle.f_add(base, oracle.f_mul(np.repeat(b, n, 0), freq)) if with_freq else base
        assert np.array_equal(got, want)


def test_lkpfold_vs_golden(hb):
    """compute{3,4}p_error_terms under has_lookups (hobbit_set_lookups), every selector value of the lookup prover, against the REAL reference"""
    g = gold("lkpfold")
    got = golden_cases.case_lkpfold(hb)
    assert set(got) == set(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k]), k


def test_streamfold_large_vs_oracle(hb, oracle):
    n = 1 << 18                                     # B = 2^18, the MLP config's chunk size
    t = [splitmix_field(n, 600 + i) for i in range(8)]
    gate = (np.arange(n) % 3 == 0).astype(np.int32)
    assert np.array_equal(hb.err2p(t[0], t[1], t[2], t[3]), oracle.err2p(t[0], t[1], t[2], t[3]))
    assert np.array_equal(hb.err3p(t[0], gate, t[1], t[2], t[3], t[4]), oracle.err3p(t[0], gate, t[1], t[2], t[3], t[4]))
    assert np.array_equal(hb.err4p(t[0], t[1], t[2], gate, t[3], t[4], t[5], t[6]), oracle.err4p(t[0], t[1], t[2], gate, t[3], t[4], t[5], t[6]))


# ---- batched cubic sumcheck and the multiplication-tree prover -------------------------------------
def test_multree_vs_golden(hb):
    g = gold("multree")
    got = golden_cases.case_multree(hb)
    assert set(got) == set(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k]), k


def test_mul_tree_2e20_vs_oracle(hb, oracle):
    x = splitmix_field(8 << 17, 700).reshape(8, 1 << 17, 2)          # C4's shape scaled: 8 vectors
    pr = np.array([17, 5], np.uint64); px = splitmix_field(3, 701)
    a = hb.mul_tree(x, pr, px); b = oracle.mul_tree(x, pr, px)
    for k in b:
        assert np.array_equal(a[k], b[k]), k


# ---- long FFTs and the inner PCS commitments of the opening ----------------------------------------
@pytest.mark.parametrize("logn", [13, 14, 16, 17, 18, 19, 20, 22])       # 17..20: the radix-8 second half; 22: beyond the 2-D twiddle table, the five-pass form
def test_fft_long_vs_oracle(hb, oracle, logn):
    x = splitmix_field(1 << logn, 900 + logn)
    assert np.array_equal(hb.fft(x), oracle.fft(x))


@pytest.mark.parametrize("logn", [14, 15, 17, 18, 19, 20])
def test_fft_long_batched_vs_oracle(hb, oracle, logn):
    """Long transforms up to 2^21 take the one-pass second half (k_fft_cols, R = 4 ... 256; R = 128 and 256 are the inner
    commitments of the 2^28 opening).  Rows 0, 3 and 7 of a batch against the oracle, in place."""
    x = splitmix_field(8 << logn, 930 + logn).reshape(8, 1 << logn, 2)
    y = hb.fft(x)
    for b in (0, 3, 7):
        assert np.array_equal(y[b], oracle.fft(x[b])), (logn, b)


def test_innerpcs_vs_golden(hb):
    g = gold("innerpcs")
    got = golden_cases.case_innerpcs(hb)
    assert set(got) == set(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k]), k


def test_shockwave_commit_2e21_vs_oracle(hb, oracle):
    """shape of C_f at N = 2^26: aggregate of 2^21 elements in 32 rows -> 32 FFTs of 2^17"""
    p = splitmix_field(1 << 21, 950)
    e1, l1 = hb.shockwave_commit(p, 32); e2, l2 = oracle.shockwave_commit(p, 32)
    assert np.array_equal(e1, e2) and np.array_equal(l1, l2)


# ---- inner PCS provers of the opening (prover side) --------------------------------------------------
@pytest.mark.parametrize("logN", [9, 10, 13, 16, 18])
def test_whir_prove_vs_oracle(hb, oracle, logN):
    import ctypes
    libc = ctypes.CDLL(None)
    p = splitmix_field(1 << logN, 1); x = splitmix_field(logN, 2)
    libc.srandom(7); want = oracle.whir_prove(p, x)
    libc.srandom(7); got = hb.whir_prove(p, x)
    assert want["checks"].tolist() == [1, 1] and got["checks"].tolist() == [1, 1]      # the reference's exit(-1) checks hold
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert libc.rand() == (libc.srandom(7), oracle.whir_prove(p, x), libc.rand())[2]    # both leave the libc stream at the same point


@pytest.mark.parametrize("N,k", [(1 << 16, 32), (1 << 19, 32), (1 << 12, 8), (1 << 13, 32)])     # widths 2048, 16384, 512 (smallest with a WHIR proof), 256 (none)
def test_shockwave_prove_vs_oracle(hb, oracle, N, k):
    import ctypes
    libc = ctypes.CDLL(None)
    p = splitmix_field(N, 3)
    enc, lv = oracle.shockwave_commit(p, k)
    x = splitmix_field(N.bit_length() - 1, 4)
    libc.srandom(9); want = oracle.shockwave_prove(p, enc, k, x, lv)
    libc.srandom(9); got = hb.shockwave_prove(p, enc, k, x, lv)
    for kk in want:
        assert np.array_equal(got[kk], want[kk]), kk


# ---- reference-anchored values at the C3 / north-star sizes -------------------------------------------------------------
@pytest.mark.parametrize("logN", [22, 24, 26, 28])
def test_commit_standard_vs_reference_big(hb, logN):
    """commit_standard on test_PC(2^logN, 4, 32)'s exact inputs (libc-drawn polynomial and graphs) against the REAL reference's
    output at that size (oracle/gen_big_roots.py ran oracle/_ref's commit_standard once: 167 s at 2^26, ~11 min at 2^28 on one core):
    root, sha256 of every Merkle level, 64 sampled leaves, 64 sampled tensor entries, five open_tree_blake paths."""
    from oracle.gen_big_roots import sample_plan
    import hashlib
    if not os.path.exists(os.path.join(GOLD, "bigroot_2e%d.npz" % logN)):
        pytest.skip("fixture bigroot_2e%d not generated" % logN)
    g = gold("bigroot_2e%d" % logN)
    N, K = 1 << logN, 32
    M, trs, leaves, chunk, row, col, qs = sample_plan(logN)
    hb.rng_reset()
    poly = hb.generate_randomness(N)                          # src/Our_PC.cpp:758
    hb.expander_init_store(trs)                               # :813
    d = hb.to_device(poly); del poly
    c = hb.commit_standard((d, N), K, trs, 1)
    assert np.array_equal(c.root(), g["root"])
    lv = c.levels()
    off, sz, dgs = 0, M, []
    while sz >= 1:
        dgs.append(np.frombuffer(hashlib.sha256(lv[off:off + sz].tobytes()).digest(), np.uint8)); off += sz; sz //= 2
    assert np.array_equal(np.stack(dgs), g["level_dg"])
    assert np.array_equal(lv[leaves], g["leaves_s"])
    del lv
    for i in range(len(chunk)):
        assert np.array_equal(c.gather([row[i]], [col[i]])[0, chunk[i]], g["tensor_s"][i]), i
    for i, (cc, rr) in enumerate(qs):
        assert np.array_equal(c.open_tree_blake(cc, rr), g["paths"][i]), i
    c.free(); d.free()


def test_aggregate_roots_vs_golden(hb):
    """_aggregate's inner commitments C_f / C_c (src/Our_PC.cpp:274-287) on test_PC's own inputs and evaluation point: the `roots` the
    open returns against the roots the REAL reference's _aggregate produced (tests/golden/commit.npz, c_*_cfcc)"""
    g = gold("commit")
    for N in (1 << 18, 1 << 20):
        K = 32; trs = N // (K << 11)
        hb.rng_reset(); poly = hb.generate_randomness(N); hb.expander_init_store(trs)
        c = hb.commit_standard(poly, K, trs, 1)
        x = hb.generate_randomness(N.bit_length() - 1)
        res = hb.open_core(poly, c, x, 5900, want_paths=False)
        assert np.array_equal(res["roots"], g["c_%d_32_cfcc" % N]), N
        c.free()


# ---- Elastic_PC open, option 1 (RS x RS) ----------------------------------------------------------------------------------
@pytest.mark.parametrize("N,B", golden_cases.ELASTIC_OPEN_CASES)
def test_elastic_open_passes_vs_golden(hb, N, B):
    """aggregate + C_f and compute_aggregation_reply / update_reply of Elastic_PC::open against what the REAL reference's functions
    returned (tests/golden/elastic_open.npz): the queries open() draws, the root of C_f (a function of the whole aggregate), the replies"""
    import ctypes
    g = gold("elastic_open"); key = "eo_%d_%d_" % (N, B)
    x, I = golden_cases.elastic_open_inputs(N, B)
    ctypes.CDLL(None).srandom(901)
    res = hb.elastic_open(N, B, x, 700, shockwave=False)
    assert np.array_equal(dg(res["I"].astype(np.uint64)), g[key + "I_dg"])
    assert np.array_equal(res["cf_root"], g[key + "cf_root"])
    assert res["reply"].shape == (700, N // B, 2)
    assert np.array_equal(dg(res["reply"]), g[key + "reply_dg"])
    assert res["checks"].tolist() == [1, 1]


@pytest.mark.parametrize("logN,logB", [(18, 14), (20, 16), (22, 18), (22, 20), (21, 13)])
def test_elastic_open_vs_oracle(hb, oracle, logN, logB):
    """The whole prover side of Elastic_PC::open option 1 on test_Elastic_PC's sequence (commit, x = generate_randomness(log N), open):
    queries, replies, Merkle paths of the commitment, the four sumcheck transcripts of recursive_prover_RS and shockwave_prove(C_f, r_x)
    with its WHIR proof, bit-exact against the oracle; the reference's exit(-1) checks hold on both."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, B = 1 << logN, 1 << logB
    oracle.rng_reset(); lv = oracle.elastic_commit(N, B, 1)
    x = oracle.generate_randomness(logN)
    libc.srandom(99); want = oracle.elastic_open(N, B, x, 700, lv)
    got_lv, dlv = hb.elastic_commit(N, B, 1, keep_levels=True)
    T = 4 * B
    assert np.array_equal(got_lv[:T - 1], lv[:T - 1]) and np.array_equal(got_lv[T:], lv[T:])
    libc.srandom(99); got = hb.elastic_open(N, B, x, 700, commit_levels=dlv)
    assert want["checks"].tolist() == [1, 1] and got["checks"].tolist() == [1, 1]
    assert int(got["ncols"][0]) == int(want["ncols"][0]) and int(got["reply_len"][0]) == N // B
    for k in ("I", "rv0", "cf_root", "reply", "poly", "r", "vr", "fin", "rx"):
        assert np.array_equal(got[k], want[k]), k
    # leaf 4B-1 is undefined in the reference; no path of the opening reaches it as a sibling unless position 4B-2 is queried (it is not:
    # positions are (row/4)*cols + col < B)
    assert np.array_equal(got["paths"], want["paths"])
    has_whir = int(want["sp_f"]["iters"][0]) > 0
    assert want["sp_f"]["wchecks"].tolist() == ([1, 1] if has_whir else [0, 0])
    for k in SP_KEYS:
        assert np.array_equal(got["sp_f"][k], want["sp_f"][k]), k


# ---- Elastic_PC open, option 2 (RS x expander, linear_time) ------------------------------------------------------------------
@pytest.mark.parametrize("N,B", golden_cases.ELASTIC_OPEN2_CASES)
def test_elastic_open2_passes_vs_golden(hb, N, B):
    """aggregate()'s linear_time branch (aggregate, C_f, aux_commit -> C_c) and compute_aggregation_reply / update_reply_spielman of
    Elastic_PC::open option 2 against what the REAL reference's functions returned AS BUILT (tests/golden/elastic_open2.npz; B = 2^20 is
    test_OurPC.sh's own shape): the queries, both inner roots (functions of the whole aggregate and of every remaining column's expander
    codeword), the number of remaining columns, the replies in the reference's order -- including the stale-parity reads."""
    import ctypes
    g = gold("elastic_open2"); key = "eo2_%d_%d_" % (N, B)
    x, I = golden_cases.elastic_open2_inputs(N, B)
    hb.rng_reset(); hb.expander_init_store(B >> 14)
    ctypes.CDLL(None).srandom(902)
    res = hb.elastic_open2(N, B, x, 5900)
    assert np.array_equal(dg(res["I"].astype(np.uint64)), g[key + "I_dg"])
    assert np.array_equal(res["cf_root"], g[key + "cf_root"])
    assert np.array_equal(res["cc_root"], g[key + "cc_root"])
    assert int(res["nr"][0]) == int(g[key + "nr"][0])
    assert res["reply"].shape == (5900, N // B, 2)
    assert np.array_equal(dg(res["reply"]), g[key + "reply_dg"])
    assert res["checks"].tolist() == [1]


@pytest.mark.parametrize("logN,logB,kind", [(20, 16, 0), (22, 18, 1), (23, 20, 1), (24, 20, 0)])
def test_elastic_open2_vs_oracle(hb, oracle, logN, logB, kind):
    """The whole prover side of Elastic_PC::open option 2 on test_Elastic_PC(N, 2)'s sequence (graphs, commit, x = generate_randomness(log N),
    open): queries, replies, Merkle paths of the commitment, aux_commit's commitment, the four sumcheck transcripts of
    recursive_prover_Spielman_stream and both shockwave_prove calls with their WHIR proofs, bit-exact against the oracle (whose two stream passes
    are pinned by the real reference: elastic_open2.npz); prove_fft_matrix's exit(-1) check holds on both.  kind 1: a stream whose chunks all
    differ (the reference's default stream repeats one chunk), so that a chunk-order or reply-order mistake cannot hide."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, B = 1 << logN, 1 << logB
    oracle.rng_reset(); lv = oracle.elastic_commit(N, B, 2)
    x = oracle.generate_randomness(logN)
    oracle.stream_config(kind, 4242)
    try:
        libc.srandom(77); want = oracle.elastic_open2(N, B, x, 5900, lv)
    finally:
        oracle.stream_config(0, 0)
    hb.rng_reset()
    got_lv, dlv = hb.elastic_commit(N, B, 2, keep_levels=True)
    T = 4 * B
    assert np.array_equal(got_lv[:T - 1], lv[:T - 1]) and np.array_equal(got_lv[T:], lv[T:])
    chunks = (lambda i: splitmix_field(B, 4242 + i)) if kind else None
    libc.srandom(77); got = hb.elastic_open2(N, B, x, 5900, commit_levels=dlv, chunks=chunks)
    assert want["checks"].tolist() == [1] and got["checks"].tolist() == [1]
    assert int(got["nr"][0]) == int(want["nr"][0]) and int(got["reply_len"][0]) == N // B
    for k in ("I", "rv0", "cf_root", "cc_root", "reply", "scal", "poly", "r", "vr", "fin", "rx"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["paths"], want["paths"])
    for sp in ("sp_c", "sp_f"):
        has_whir = int(want[sp]["iters"][0]) > 0
        assert want[sp]["wchecks"].tolist() == ([1, 1] if has_whir else [0, 0])
        for k in SP_KEYS:
            assert np.array_equal(got[sp][k], want[sp][k]), (sp, k)


@pytest.mark.parametrize("logN,K", [(18, 32), (20, 16), (22, 32), (24, 32), (23, 2)])
def test_open_standard_rs_vs_oracle(hb, oracle, logN, K):
    """Our_PC with linear_time == false on test_PC(N, 1, K)'s sequence (src/Our_PC.cpp:764-777: poly = generate_randomness(N), tensor_row_size = 128,
    commit_standard, x = generate_randomness(log N), open_standard with 790 queries -> recursive_prover_RS): commitment levels, queries, replies,
    paths, the four sumcheck transcripts and shockwave_prove(C_f, r_x) with its WHIR proof, bit-exact against the oracle; both exit(-1) checks
    hold.  Row codes of 128 .. 8192 points (2^24: the long-row FFT path) and, with K = 2 at 2^23, of 65536 points (the chunk-at-a-time long
    transform test_PC option 1 needs from 2^27 on).  The commit for this shape is pinned by oracle/_ref in commit.npz (trs=4)
    and at trs = 128 in test_oracle_vs_ref."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, trs = 1 << logN, 128
    oracle.rng_reset(); poly = oracle.generate_randomness(N)
    lv, T = oracle.commit_standard(poly, K, trs, 0, want_tensor=True)
    x = oracle.generate_randomness(logN)
    libc.srandom(41); want = oracle.open_standard_rs(poly, K, trs, x, 790, lv, T)
    del T
    c = hb.commit_standard(poly, K, trs, 0)
    assert np.array_equal(c.levels(), lv)
    libc.srandom(41); got = hb.open_standard_rs(poly, c, x, 790)
    c.free()
    assert want["checks"].tolist() == [1, 1] and got["checks"].tolist() == [1, 1]
    assert int(got["ncols"][0]) == int(want["ncols"][0]) and int(got["reply_len"][0]) == K
    for k in ("I", "rv0", "cf_root", "reply", "paths", "poly", "r", "vr", "fin", "rx"):
        assert np.array_equal(got[k], want[k]), k
    has_whir = int(want["sp_f"]["iters"][0]) > 0
    assert want["sp_f"]["wchecks"].tolist() == ([1, 1] if has_whir else [0, 0])
    for k in SP_KEYS:
        assert np.array_equal(got["sp_f"][k], want["sp_f"][k]), k


def test_open_standard_rs_2e28_selfchecks(hb):
    """test_PC option 1 at the north-star size (131072-point row codes through the chunk-at-a-time long transform; no CPU oracle at this
    size): both exit(-1) checks and the WHIR checks hold, two runs with the same libc seed give the same transcript, replies are the
    committed tensor's entries, and queried leaves -- recomputed on the host from the four tensor rows of all K chunks -- walk up their
    Merkle paths to the root (hobbit_verify_path_host, the reference's left|left parents)."""
    import ctypes
    libc = ctypes.CDLL(None)
    N, K, trs = 1 << 28, 32, 128
    cols = 2 * (N // K) // trs
    d = hb.fill_splitmix(N, 777)
    c = hb.commit_standard((d, N), K, trs, 0)
    x = splitmix_field(28, 78)
    libc.srandom(6); a = hb.open_standard_rs((d, N), c, x, 790)
    libc.srandom(6); b = hb.open_standard_rs((d, N), c, x, 790)
    assert a["checks"].tolist() == [1, 1] and a["sp_f"]["wchecks"].tolist() == [1, 1]
    for k in ("I", "reply", "paths", "poly", "r", "vr", "fin", "rx", "cf_root"):
        assert np.array_equal(a[k], b[k]), k
    root = c.root()
    V = lambda v: v.ctypes.data_as(ctypes.c_void_p)
    hb.lib.hobbit_verify_path_host.restype = ctypes.c_int
    for q in (0, 100, 789):
        col, row = int(a["I"][q, 0]), int(a["I"][q, 1])
        for i in (0, K - 1):
            assert np.array_equal(a["reply"][q, i], c.tensor_row(i, row)[col])
        leaf = leaf_from_tensor(hb, c, col, row)
        pos = (row // 4) * cols + col
        path = np.ascontiguousarray(a["paths"][q])
        assert hb.lib.hobbit_verify_path_host(V(leaf), ctypes.c_uint64(pos), V(path), path.shape[0], V(root), 1) == 1
    c.free()


def test_open_standard_rs_rejects_expander_commitment(hb):
    """the RS x RS opening refuses an RS x expander commitment (and vice versa hobbit_open_standard refuses an RS x RS one) instead of opening it wrongly"""
    mod = __import__("__graft_entry__").load_package()
    N, K = 1 << 18, 32
    poly, trs = golden_cases.test_pc_inputs(hb, N, K)
    c = hb.commit_standard(poly, K, trs, 1)
    with pytest.raises(mod.HobbitError, match="RS x expander"):
        hb.open_standard_rs(poly, c, splitmix_field(18, 1), 790)
    c.free()
    c = hb.commit_standard(poly, K, 128, 0)
    with pytest.raises(mod.HobbitError, match="bad arguments"):
        hb.open_standard_rs(poly, c, splitmix_field(18, 1), 0)
    c.free()


def test_elastic_open_skips_zero_chunks(hb, oracle):
    """compute_aggregation_reply appends nothing for an all-zero chunk (src/Elastic_PC.cpp:510-517): push zero chunks 1 and 2 of 4"""
    import ctypes
    N, B = 1 << 16, 1 << 14
    x = splitmix_field(16, 5)
    ch = hb.to_device(hb.read_stream(B)); z = hb.to_device(np.zeros((B, 2), np.uint64))
    e = ctypes.c_void_p()
    ctypes.CDLL(None).srandom(7)
    hb._chk(hb.lib.hobbit_elastic_open_begin(hb.ctx, N, B, B >> 11, x.ctypes.data_as(ctypes.c_void_p), 64, ctypes.byref(e)))
    for b in (ch, z, z, ch):
        hb._chk(hb.lib.hobbit_elastic_open_aggregate_push(hb.ctx, e, b.ptr))
    hb._chk(hb.lib.hobbit_elastic_open_aggregate_finish(hb.ctx, e))
    for b in (ch, z, z, ch):
        hb._chk(hb.lib.hobbit_elastic_open_reply_push(hb.ctx, e, b.ptr))
    rl = np.zeros(1, np.int32); rep = np.zeros((64, 4, 2), np.uint64)
    q = np.zeros((64, 3, 2), np.uint64); r = np.zeros((64, 2), np.uint64); vr = np.zeros((4, 2, 2), np.uint64); fin = np.zeros((4, 2), np.uint64); chk = np.zeros(2, np.int32)
    names = ("cols", "rows", "rv0", "reply", "reply_len", "paths", "cf_root", "ncols", "poly", "r", "vr", "fin", "checks", "rx", "sp_f")

    class Out(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in names]
    o = Out(None, None, None, rep.ctypes.data, rl.ctypes.data, None, None, None, q.ctypes.data, r.ctypes.data, vr.ctypes.data, fin.ctypes.data, chk.ctypes.data, None, None)
    hb._chk(hb.lib.hobbit_elastic_open_finish(hb.ctx, e, None, ctypes.byref(o)))
    hb.lib.hobbit_elastic_open_free(e)
    assert rl[0] == 2 and chk.tolist() == [1, 1]
    rep = rep.reshape(-1, 2)[:128].reshape(64, 2, 2)
    assert np.array_equal(rep[:, 0], rep[:, 1]) and rep.any()


# ---- C5: Elastic_PC streaming commit at 2^30 on one GPU (and 2^26), against the REAL reference ---------------------------------
@pytest.mark.parametrize("logN,opt", [(26, 1), (26, 2), (30, 1), (30, 2)])
def test_elastic_commit_vs_reference_big(hb, logN, opt):
    """test_Elastic_PC's commit with B = 2^20 (opt 1: RS x RS, trs = 512; opt 2: 32768-point rows x expander, trs = 64) against the
    REAL reference's root and level digests (oracle/gen_elastic_roots.py ran oracle/_ref once per case; 2^30 took ~35 min per option).
    The stream's chunk is generated once on the host and stays resident (the reference's "test" stream repeats one chunk)."""
    import glob
    name = "elastic_root_%d_20_%d" % (logN, opt)
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("fixture %s not generated" % name)
    g = gold(name)
    N, B = 1 << logN, 1 << 20
    hb.rng_reset()
    lv = hb.elastic_commit(N, B, opt)
    T = 4 * B
    assert np.array_equal(lv[-1], g["root"])
    assert np.array_equal(dg(lv[:T - 1]), g["leaves_dg"]) and np.array_equal(dg(lv[T:]), g["upper_dg"])


def test_host_mirror_test_elastic_pc(oracle):
    """test_Elastic_PC(2^18, 1) through the C++ mirror (commit(stream_descriptor), open(stream_descriptor), read_stream, the reference's
    signatures): root, queries, replies, sumcheck transcripts and the proof size against the oracle run on the same libc sequence"""
    import ctypes
    from __graft_entry__ import PKG, build_host
    from oracle.pyoracle import elastic_open_proof_size
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    N, B = 1 << 18, 1 << 14
    oracle.rng_reset(); lv = oracle.elastic_commit(N, B, 1)
    x = oracle.generate_randomness(18)
    want = oracle.elastic_open(N, B, x, 700, lv)
    root = np.zeros(32, np.uint8); I = np.zeros((700, 2), np.uint32); reply = np.zeros((700, N // B, 2), np.uint64)
    q = np.zeros((64, 3, 2), np.uint64); r = np.zeros((64, 2), np.uint64); chk = np.zeros(3, np.int32); ps = ctypes.c_double()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_elastic_open.argtypes = [ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_void_p] * 7
    rounds = lib.hobbit_host_elastic_open(N, B, P(root), P(I), P(reply), P(q), P(r), P(chk), ctypes.byref(ps))
    lib.hobbit_host_close()
    assert np.array_equal(root, lv[-1]) and chk.tolist() == [1, 1, 1]
    assert rounds == want["poly"].shape[0]
    assert np.array_equal(I, want["I"]) and np.array_equal(reply, want["reply"])
    assert np.array_equal(q[:rounds], want["poly"]) and np.array_equal(r[:rounds], want["r"])
    assert ps.value == elastic_open_proof_size(want, N, B)


def test_host_mirror_test_elastic_pc_option2(oracle):
    """test_Elastic_PC(2^22, 2) with BUFFER_SPACE = 2^18 through the C++ mirror (expander_init_store, commit(stream_descriptor),
    open(stream_descriptor) under linear_time, the reference's signatures): root, queries, replies, C_c's root, the sumcheck transcripts of
    recursive_prover_Spielman_stream and the proof size against the oracle run on the same libc sequence"""
    import ctypes
    from __graft_entry__ import PKG, build_host
    from oracle.pyoracle import elastic_open2_proof_size
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    N, B = 1 << 22, 1 << 18
    oracle.rng_reset(); lv = oracle.elastic_commit(N, B, 2)
    x = oracle.generate_randomness(22)
    want = oracle.elastic_open2(N, B, x, 5900, lv)
    root = np.zeros(32, np.uint8); I = np.zeros((5900, 2), np.uint32); reply = np.zeros((5900, N // B, 2), np.uint64)
    q = np.zeros((96, 3, 2), np.uint64); r = np.zeros((96, 2), np.uint64); chk = np.zeros(3, np.int32); ps = ctypes.c_double(); nrem = ctypes.c_int()
    ccr = np.zeros(32, np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hobbit_host_elastic_open2.argtypes = [ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_void_p] * 9
    rounds = lib.hobbit_host_elastic_open2(N, B, P(root), P(I), P(reply), P(q), P(r), P(chk), ctypes.byref(ps), ctypes.byref(nrem), P(ccr))
    lib.hobbit_host_close()
    assert np.array_equal(root, lv[-1]) and chk.tolist() == [1, 1, 1]
    assert rounds == want["poly"].shape[0] and nrem.value == int(want["nr"][0])
    assert np.array_equal(I, want["I"]) and np.array_equal(reply, want["reply"]) and np.array_equal(ccr, want["cc_root"])
    assert np.array_equal(q[:rounds], want["poly"]) and np.array_equal(r[:rounds], want["r"])
    assert ps.value == elastic_open2_proof_size(want, N, B)


# ---- streaming provers (BASELINE config 4's math phases) ---------------------------------------------------------------------
def test_stream_product_layers_vs_golden(hb):
    """read_mul_tree_layer / read_mul_tree_data on the reference's default stream against the REAL reference (tests/golden/streamdrv.npz)"""
    g = gold("streamdrv"); src = hb.chunk_source(0)
    for layer in (1, 2, 4):
        assert np.array_equal(dg(hb.read_mul_tree_layer(src, 1 << 11, layer)), g["layer_%d" % layer])
    for (layer, dist, bt) in ((0, 1, 1), (2, 1, 1), (1, 2, 2), (0, 3, 3)):
        assert np.array_equal(dg(hb.read_mul_tree_data(src, 1 << 12, layer, dist, bt)), g["data_%d_%d_%d" % (layer, dist, bt)])


@pytest.mark.parametrize("shape", golden_cases.STREAM_SHAPES)
def test_stream_sumcheck3_vs_golden(hb, shape):
    """generate_3product_sumcheck_beta_stream_batch_optimized against the REAL reference: the new claims and challenge rows are functions
    of every message of the streaming pass, of batch_3product_sumcheck, of the Partial_Evals pass and of the closing 2-product sumcheck,
    and of every libc draw in between"""
    import ctypes
    fd, B, layer_id, batches, dist = shape
    g = gold("streamdrv"); key = "gsb_%d_%d_%d_%d_%d_" % shape
    rr, oc = golden_cases.stream_batch_inputs(fd, B, layer_id, batches, dist)
    ctypes.CDLL(None).srandom(5)
    res = hb.sumcheck3_stream_batch(hb.chunk_source(0), fd, B, rr, batches, dist, layer_id, oc)
    assert np.array_equal(res["new_claims"], g[key + "claims"])
    for i, row in enumerate(res["new_r"]):
        assert np.array_equal(row, g[key + "r%d" % i]), i
    assert res["checks"].tolist()[1:] == [1, 1]


@pytest.mark.parametrize("fd,B,layer_id,batches,dist", [(1 << 16, 1 << 10, 0, 1, 1), (1 << 17, 1 << 10, 1, 3, 2), (1 << 22, 1 << 16, 1, 2, 3)])
def test_stream_sumcheck3_varying_stream_vs_oracle(hb, oracle, fd, B, layer_id, batches, dist):
    """the same on a stream whose reads all differ (the reference's default stream repeats one chunk, which would hide a chunk-order
    mistake): every transcript piece against the oracle"""
    import ctypes
    libc = ctypes.CDLL(None)
    rr, oc = golden_cases.stream_batch_inputs(fd, B, layer_id, batches, dist)
    oracle.stream_config(1, 777)
    try:
        libc.srandom(8); want = oracle.sumcheck3_stream_batch(fd, B, rr, batches, dist, layer_id, oc, full=True)
    finally:
        oracle.stream_config(0, 0)
    libc.srandom(8); got = hb.sumcheck3_stream_batch(hb.chunk_source(1, 777), fd, B, rr, batches, dist, layer_id, oc)
    for k in ("new_claims", "cpoly1", "r1", "vr1", "qpoly2", "r2", "vr2", "fin2", "R", "checks"):
        assert np.array_equal(got[k], want[k]), k
    for a, b in zip(got["new_r"], want["new_r"]):
        assert np.array_equal(a, b)


def test_generate_claims_opt_vs_golden(hb):
    g = gold("streamdrv")
    assert np.array_equal(hb.generate_claims_opt(hb.chunk_source(0), 1 << 16, 1 << 10, splitmix_field(16, 70), 2, 1, 2), g["claims_opt"])


@pytest.mark.parametrize("distance,naive,kind", [(5, True, 0), (5, True, 1), (2, False, 1), (1, True, 1)])
def test_mul_tree_stream_shallow_vs_oracle(hb, oracle, distance, naive, kind):
    """prove_multiplication_tree_stream_shallow (config 4's shape scaled down: 8 vectors, layers = 4 <= distance = 5; then the batched and
    the naive paths): product-layer read, in-memory tree, and every streaming sumcheck, bit-exact against the oracle; on the default
    stream the output also against the REAL reference.  Each step's K_partial equals the claim handed down by the previous layer."""
    import ctypes
    libc = ctypes.CDLL(None)
    B, vectors, size = 1 << 10, 8, 1 << 12
    pr = np.array([32, 0], np.uint64); px = splitmix_field(3, 9)
    oracle.stream_config(kind, 4000)
    try:
        libc.srandom(11); want = oracle.mul_tree_stream_shallow(vectors * size, B, vectors, size, pr, distance, px, naive=naive)
    finally:
        oracle.stream_config(0, 0)
    libc.srandom(11); got = hb.mul_tree_stream_shallow(hb.chunk_source(kind, 4000), vectors * size, B, vectors, size, pr, distance, px, naive=naive)
    assert got["layers"] == want["layers"] and len(got["steps"]) == len(want["steps"]) > 0
    assert np.array_equal(got["output"], want["output"])
    if kind == 0 and distance == 5:
        assert np.array_equal(got["output"], gold("streamdrv")["shallow_out"])
    for k in ("final_r", "final_eval", "out_eval", "vr", "fin"):
        assert np.array_equal(got["tree"][k], want["tree"][k]), k
    for a, b in zip(got["steps"], want["steps"]):
        assert a["checks"].tolist() == [1, 1, 1] and b["checks"].tolist() == [1, 1, 1]
        for k in ("new_claims", "cpoly1", "r1", "vr1", "qpoly2", "r2", "vr2", "fin2", "R"):
            assert np.array_equal(a[k], b[k]), k
        for x, y in zip(a["new_r"], b["new_r"]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("logB,nch", [(8, 8), (14, 4), (18, 4)])
def test_gate_consistency_stream_vs_oracle(hb, oracle, logB, nch):
    """prove_gate_consistency's chunk loop, degree-4 sumcheck, Peval pass and closing sumcheck over a synthetic consistent trace
    (logB = 18, 4 chunks = config 4's transcript_stream shape): every message against the oracle; the three exit(-1) checks hold"""
    import ctypes
    from oracle.pyoracle import gate_standard_inputs
    libc = ctypes.CDLL(None)
    B = 1 << logB
    if logB <= 14:
        parts = [gate_standard_inputs(B, 100 + c) for c in range(nch)]
    else:                                                   # (gate_standard_inputs multiplies in Python: too slow at 2^18; vectorised here)
        parts = []
        for c in range(nch):
            g = np.random.default_rng(200 + c); z = np.zeros(B, np.uint64)
            sel = g.integers(0, 2, B).astype(np.uint64); Lr = g.integers(0, 1 << 30, B).astype(np.uint64); Rr = g.integers(0, 1 << 30, B).astype(np.uint64)
            prod = (Lr * Rr) % np.uint64(P)
            parts.append((np.stack([Lr, z], 1), np.stack([Rr, z], 1), np.stack([np.where(sel == 1, Lr + Rr, prod), z], 1), np.stack([sel, z], 1)))
    L, R, O = [np.concatenate([p[i] for p in parts]) for i in range(3)]
    S = np.concatenate([p[3][:, 0] for p in parts]).astype(np.int32)
    r = splitmix_field(logB, 3)
    libc.srandom(21); want = oracle.gate_consistency_stream(L, R, O, S, B, r)
    libc.srandom(21); got = hb.gate_consistency_stream(hb.trace_source(L, R, O, S, B), nch, B, r)
    assert want["checks"].tolist() == [1, 1, 1] and got["checks"].tolist() == [1, 1, 1]
    for k in ("R", "a", "poly", "gr", "fin6", "Peval", "b", "q2", "r2", "vr2", "fin2"):
        assert np.array_equal(got[k], want[k]), k


def lookup_trace(B, nch, seed):
    """consistent trace with lookup rows (selectors 0 add, 1 mul, 2 lookup), vectorised; lookup outputs are full-range elements"""
    parts = []
    for c in range(nch):
        g = np.random.default_rng(seed + c); z = np.zeros(B, np.uint64)
        S = g.integers(0, 3, B).astype(np.int32); Lr = g.integers(0, 1 << 30, B).astype(np.uint64); Rr = g.integers(0, 1 << 30, B).astype(np.uint64)
        prod = (Lr * Rr) % np.uint64(P)
        Ore = np.where(S == 0, Lr + Rr, np.where(S == 1, prod, g.integers(0, P, B).astype(np.uint64)))
        Oim = np.where(S == 2, g.integers(0, P, B).astype(np.uint64), z)
        parts.append((np.stack([Lr, z], 1), np.stack([Rr, z], 1), np.stack([Ore, Oim], 1), S))
    return [np.concatenate([p[i] for p in parts]) for i in range(4)]


@pytest.mark.parametrize("logB,nch", [(8, 8), (14, 4), (18, 4)])
def test_gate_consistency_lookups_stream_vs_oracle(hb, oracle, logB, nch):
    """prove_gate_consistency_lookups (src/sumcheck.cpp:503-795): chunk loop with the lookup gate maps of compute{3,4}p_error_terms, the 9-table
    degree-4 sumcheck, the 8-column Peval pass and the closing sumcheck, every message against the oracle; all five reference self-checks hold.
    The oracle restates the reference's in-place selector rewrites literally, the device derives its selector columns once per chunk."""
    import ctypes
    libc = ctypes.CDLL(None)
    B = 1 << logB
    L, R, O, S = lookup_trace(B, nch, 700)
    r = splitmix_field(logB, 3); lr = splitmix_field(2, 77)
    libc.srandom(23); want = oracle.gate_consistency_lookups_stream(L, R, O, S, B, r, lr)
    libc.srandom(23); got = hb.gate_consistency_lookups_stream(hb.trace_source(L, R, O, S, B), nch, B, r, lr)
    assert want["checks"].tolist() == [1] * 5 and got["checks"].tolist() == [1] * 5
    for k in ("R", "a", "poly", "gr", "fin9", "Peval", "b", "q2", "r2", "vr2", "fin2"):
        assert np.array_equal(got[k], want[k]), k
    # a broken addition gate must trip "Error in gate consistency 1" on both sides, with identical messages up to there
    O2 = O.copy(); O2[2 * B + int(np.nonzero(S[2 * B:3 * B] == 0)[0][0]), 0] ^= np.uint64(1)
    libc.srandom(23); w2 = oracle.gate_consistency_lookups_stream(L, R, O2, S, B, r, lr)
    libc.srandom(23); g2 = hb.gate_consistency_lookups_stream(hb.trace_source(L, R, O2, S, B), nch, B, r, lr)
    assert w2["checks"][0] == 0 and g2["checks"][0] == 0 and np.array_equal(g2["R"], w2["R"]) and np.array_equal(g2["poly"], w2["poly"])


def test_lookups_flag_is_required_and_scoped(hb):
    """hobbit_gate_consistency_lookups_stream refuses to run with has_lookups unset (the reference would silently use the plain gate maps);
    the flag set by hobbit_set_lookups changes compute3p/4p only while it is on"""
    n = 4096
    t = [splitmix_field(n, 900 + i) for i in range(8)]
    gate = (np.arange(n) % 5 - 1).astype(np.int32)                                           # -1 .. 3
    plain = hb.err3p(t[0], gate, t[1], t[2], t[3], t[4])
    hb.set_lookups(splitmix_field(2, 5))
    lk = hb.err3p(t[0], gate, t[1], t[2], t[3], t[4])
    hb.set_lookups(None)
    assert not np.array_equal(plain, lk) and np.array_equal(plain, hb.err3p(t[0], gate, t[1], t[2], t[3], t[4]))
    L, R, O, S = lookup_trace(256, 2, 1)
    mod = __import__("__graft_entry__").load_package()
    with pytest.raises(mod.HobbitError, match="has_lookups"):
        hb.gate_consistency_lookups_stream(hb.trace_source(L, R, O, S, 256), 2, 256, splitmix_field(8, 3), None)


@pytest.mark.parametrize("opt", [1, 2])
def test_sharded_elastic_commit_hip_ops_world1(hb, oracle, opt):
    """Per-rank GPU operations of the sharded streaming commit (groups of 4 consecutive chunks; hobbit_elastic_push_inner,
    chain of the group digests, subtree) at world size 1 on a stream whose chunks all differ, against the oracle's streaming commit on the
    same stream; the collective pattern is covered by tests/test_dist_gloo.py at world 2."""
    import ctypes
    import torch
    from __graft_entry__ import load_package
    mod = load_package()
    N, B = 1 << 19, 1 << 14
    plan = mod.parallel.ElasticPlan(N, B, opt, 1)
    oracle.rng_reset()
    if opt == 2:
        oracle.expander_init_store(plan.trs)
        hb.upload_graphs(plan.trs, graphs_from(oracle, plan.trs)) if plan.trs > 13 else hb.expander_init_store(plan.trs)
    oracle.stream_config(1, 9000)
    try:
        want = np.zeros((8 * B, 32), np.uint8)
        oracle.lib.orc_elastic_commit_model.restype = ctypes.c_size_t
        cnt = oracle.lib.orc_elastic_commit_model(ctypes.c_size_t(N), ctypes.c_size_t(B), ctypes.c_int(opt), want.ctypes.data_as(ctypes.c_void_p))
    finally:
        oracle.stream_config(0, 0)
    live = []

    def source(c):
        buf = hb.to_device(splitmix_field(B, 9000 + c)); live.append(buf); del live[:-2]
        return buf.ptr
    ops = mod.parallel.ElasticHipOps(hb, torch.device("cuda", 0), 0)
    res = mod.parallel.sharded_commit(ops, None, plan, 0, source)
    levels = mod.parallel.assemble_levels(plan, [res["subtree"].cpu().numpy()], res["top"])
    T = 4 * B
    assert np.array_equal(levels[:T - 1], want[:T - 1]) and np.array_equal(levels[T:], want[T:cnt])


def test_host_mirror_remaining_wrappers(oracle):
    """The reference-named C++ entry points SURVEY.md 8(b) lists beyond the drivers -- prove_multiplication_tree_new, batch_3product_sumcheck,
    _compute_tensorcode(F*, F**, int), shockwave_commit / shockwave_prove with the globals C_f / C_c, and the streaming
    prove_multiplication_tree_stream_shallow over read_stream -- each called once through libhobbit_host.so and compared with the oracle."""
    import ctypes
    from __graft_entry__ import PKG, build_host
    build_host()
    lib = ctypes.CDLL(os.path.join(PKG, "libhobbit_host.so"))
    libc = ctypes.CDLL(None)
    P_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    tree = splitmix_field(256, 1201); b3 = splitmix_field(960, 1202); b3a = splitmix_field(2, 1203)
    out = np.zeros((64, 2), np.uint64); roots = np.zeros((2, 32), np.uint8); ps = ctypes.c_double()
    lib.hobbit_host_mirror_check.argtypes = [ctypes.c_void_p] * 6
    n = lib.hobbit_host_mirror_check(P_(tree), P_(b3), P_(b3a), P_(out), P_(roots), ctypes.byref(ps))
    want = []
    libc.srandom(77); mt = oracle.mul_tree(tree.reshape(4, 64, 2), np.array([17, 5], np.uint64), None)
    want += [mt["out_eval"], mt["final_eval"]] + [oracle.field_prod(tree.reshape(4, 64, 2)[j]) for j in range(4)] + [mt["final_r"][0], mt["final_r"][7], np.array([int(mt["layers"][0]), 0], np.uint64)]
    lens = [256, 64]
    t = [np.concatenate([b3[320 * k:320 * k + 256], b3[320 * k + 256:320 * k + 320]]) for k in range(3)]
    bs = oracle.batch_3product_sumcheck(t[0], t[1], t[2], lens, b3a)
    want += [bs["poly"][0][0], bs["poly"][-1][3], bs["r"][-1]] + list(bs["vr"].reshape(-1, 2))
    tc = oracle.compute_tensorcode(tree, 4, 0)
    want += [tc[0][0], tc[7][127], tc[3][64]]
    big = np.concatenate([oracle.f_mul(tree, np.tile(np.array([[i + 1, 0]], np.uint64), (256, 1))) for i in range(32)])
    enc, lv = oracle.shockwave_commit(big, 32)
    want += [enc[31][511], big[5 * 256 + 7]]
    _, lv8 = oracle.shockwave_commit(tree, 8)
    x = np.concatenate([b3a, b3[2:13]])
    libc.srandom(78); sp = oracle.shockwave_prove(big, enc, 32, x, lv)
    want += [sp["q1"][0][0], sp["fin1"], sp["fin2"], np.array([int(sp["iters"][0]), 0], np.uint64), np.array([int(sp["wchecks"][0]) + 2 * int(sp["wchecks"][1]), 0], np.uint64)]
    want = np.stack([np.asarray(w, np.uint64).reshape(2) for w in want])
    assert n == want.shape[0]
    assert np.array_equal(out[:n], want), np.nonzero((out[:n] != want).any(axis=1))
    assert np.array_equal(roots[0], lv[-1]) and np.array_equal(roots[1], lv8[-1])
    # streaming multiplication tree through the mirror on the default stream: config 4's shape scaled down; output vs the REAL reference
    o2 = np.zeros((8, 2), np.uint64); px = splitmix_field(3, 9)
    lib.hobbit_host_mul_tree_stream.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    m = lib.hobbit_host_mul_tree_stream(1 << 10, 8, 1 << 12, 5, P_(px), 3, P_(o2), ctypes.byref(ps), 1)
    assert m == 8 and np.array_equal(o2, gold("streamdrv")["shallow_out"]) and ps.value > 0
    # commit_layers (src/sumcheck.cpp:983-1003): Elastic commitments to the PC_layer streams, roots against the REAL reference
    g = gold("streamdrv"); roots = np.zeros((2, 32), np.uint8)
    lib.hobbit_host_commit_layers.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert lib.hobbit_host_commit_layers(1 << 20, 1 << 13, 3, 1, 2, P_(roots)) == 2
    assert np.array_equal(roots, g["cl_roots"])
    # the batched path end to end through the mirror (layers 6 > distance 2, naive = false): commit_layers, the in-memory tree,
    # generate_claims_opt, two batched streaming sumchecks, open_layers (two Elastic opens); every exit(-1) check of the way holds
    # (the mirror would exit), the products equal the oracle's
    o3 = np.zeros((8, 2), np.uint64)
    m = lib.hobbit_host_mul_tree_stream(1 << 13, 8, 1 << 17, 2, P_(px), 3, P_(o3), ctypes.byref(ps), 0)
    lib.hobbit_host_close()
    libc.srandom(11); want3 = oracle.mul_tree_stream_shallow(1 << 20, 1 << 13, 8, 1 << 17, np.array([32, 0], np.uint64), 2, px, naive=False)
    assert m == 8 and np.array_equal(o3, want3["output"]) and all(st["checks"].tolist() == [1, 1, 1] for st in want3["steps"])


# ---- host pointers of the C ABI are 8-byte aligned, not 16 (regression for the one crash of round 2) ----------------------------
class _MisalignedNumpy:
    """stands in for `np` inside the binding module: every uint64 / int64 buffer it hands out starts at an address = 8 (mod 16) -- what a
    C caller's `new hobbit_F[n]`, a std::vector<F> or a struct member may legitimately be (hobbit_F is two uint64_t: alignment 8)"""

    def __init__(self):
        self.count = 0

    def __getattr__(self, name):
        return getattr(np, name)

    def _mis(self, shape, dtype):
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if not isinstance(shape, int) else shape
        raw = np.zeros(n * dt.itemsize + 32, np.uint8)
        off = (8 - raw.ctypes.data) % 16
        v = raw[off:off + n * dt.itemsize].view(dt).reshape(shape)
        assert v.ctypes.data % 16 == 8 or n == 0
        self.count += 1
        return v

    def zeros(self, shape, dtype=float, **kw):
        return self._mis(shape, dtype) if np.dtype(dtype).itemsize == 8 else np.zeros(shape, dtype, **kw)

    def ascontiguousarray(self, a, dtype=None):
        b = np.ascontiguousarray(a, dtype)
        if b.dtype.itemsize == 8 and b.size and b.ctypes.data % 16 == 0:
            v = self._mis(b.shape, b.dtype); v[...] = b
            return v
        return b


def test_abi_host_pointers_8_mod_16(hb, oracle):
    """Every host-pointer argument of include/hobbit_hip.h is `hobbit_F *` / `uint64_t *` = 8-byte aligned.  Round 2's one crash was a
    16-byte-aligned vector load (movaps) on such a buffer inside hobbit_elastic_open_finish; the fix reads host field elements through an
    8-byte-aligned view (csrc/hobbit_field.hpp HF).  Here every field-element buffer the binding hands to the library -- inputs and outputs,
    also the nested shockwave / WHIR output structs -- starts at an address = 8 (mod 16), across the openings, the sumchecks, the multiplication
    tree, the code-membership provers and the streaming drivers; results must equal the aligned run's."""
    import ctypes
    import sys
    mod = sys.modules["hobbit_amd"]
    libc = ctypes.CDLL(None)
    N, K = 1 << 18, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    hb.upload_graphs(trs, graphs_from(oracle, trs))
    x = splitmix_field(18, 3)
    v1 = splitmix_field(1 << 12, 1); v2 = splitmix_field(1 << 12, 2); v3 = splitmix_field(1 << 12, 3); pr = np.array([33, 0], np.uint64)
    tree_in = splitmix_field(4 * 64, 9).reshape(4, 64, 2)

    def run_all():
        out = {}
        c = hb.commit_standard(poly, K, trs, 1)
        libc.srandom(5); out["open"] = hb.open_standard(poly, c, x, 5900)
        c.free()
        out["sc2"] = hb.generate_2product_sumcheck_proof(v1, v2, pr)
        out["sc3"] = hb.generate_3product_sumcheck_proof(v1, v2, v3, pr) if hasattr(hb, "generate_3product_sumcheck_proof") else {}
        libc.srandom(6); out["tree"] = hb.mul_tree(tree_in, np.array([17, 5], np.uint64))
        out["beta"] = {"b": hb.precompute_beta(splitmix_field(10, 4))}
        out["eval"] = {"e": hb.evaluate_vector(v1, splitmix_field(12, 5))}
        libc.srandom(7); out["eo1"] = hb.elastic_open(1 << 18, 1 << 14, x, 700)
        hb.rng_reset(); hb.expander_init_store(4)
        libc.srandom(8); out["eo2"] = hb.elastic_open2(1 << 20, 1 << 16, splitmix_field(20, 6), 5900)
        hb.upload_graphs(trs, graphs_from(oracle, trs))
        return out

    def flat(d, pre=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, pre + k + ".")
            elif isinstance(v, np.ndarray):
                yield pre + k, v
    want = dict(flat(run_all()))
    shim = _MisalignedNumpy()
    mod.np = shim
    try:
        got = dict(flat(run_all()))
    finally:
        mod.np = np
    assert shim.count > 100, "the shim did not see the binding's buffers"
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
