"""Golden-vector case definitions (shared by oracle/gen_golden.py, which runs them against the
REAL reference build oracle/_ref and writes tests/golden/*.npz, and by
tests/test_oracle_golden.py, which runs them against the C restatement oracle/hobbit_oracle.c).

Each case is ``fn(lib) -> dict[str, np.ndarray]``; ``lib`` is oracle.pyoracle.Ref or .Oracle (same
method names).  Large outputs are stored as a SHA-256 digest plus sampled entries so that every
fixture stays small (SURVEY.md 8c list, items 1-13).  Inputs are re-derived deterministically
(splitmix64 full-range elements, or the libc generator from its default seed as the reference
itself does), so fixtures hold expected OUTPUTS only.
"""
import hashlib
import numpy as np
from oracle.pyoracle import splitmix_field, P


def dg(a):
    """sha256 of the raw bytes -> uint8[32]"""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def samp(a, k=64, seed=5):
    flat = np.ascontiguousarray(a).reshape(-1, a.shape[-1])
    idx = np.random.default_rng(seed).integers(0, flat.shape[0], k)
    return flat[idx].copy()


EDGE = np.array([[0, 0], [1, 0], [P - 1, 0], [0, P - 1], [P - 1, P - 1], [0, 1], [1, 1], [2, P - 2]], dtype=np.uint64)


def field_inputs(n):
    a = splitmix_field(n, 1); b = splitmix_field(n, 2)
    a[:8] = EDGE; b[:8] = EDGE[::-1]
    a[8:16] = EDGE; b[8:16] = EDGE           # squares of the edge set
    return a, b


def case_field(lib):
    a, b = field_inputs(1024)
    A, B = field_inputs(4096)
    return dict(add=lib.f_add(a, b), sub=lib.f_sub(a, b), mul=lib.f_mul(a, b), neg=lib.f_neg(a),
                inv=lib.f_inv(a[1:65]), mul4k=dg(lib.f_mul(A, B)), add4k=dg(lib.f_add(A, B)), sub4k=dg(lib.f_sub(A, B)),
                rou=np.stack([lib.root_of_unity(l) for l in (1, 2, 4, 12, 13, 20, 61)]))


def case_mimc(lib):
    a, b = field_inputs(256)
    out = lib.mimc(a, b)
    # 64-step chained transcript: r <- mimc(r, x_i)
    r = np.array([[33, 0]], dtype=np.uint64); chain = []
    for i in range(64):
        r = lib.mimc(r, a[i:i + 1]); chain.append(r[0].copy())
    kat = lib.mimc(np.array([[5, 7]], dtype=np.uint64), np.array([[11, 13]], dtype=np.uint64))
    return dict(out=out, chain=np.stack(chain), kat=kat)


def case_blake(lib):
    blk = np.random.default_rng(0).integers(0, 256, (256, 64), dtype=np.uint8)
    blk[0] = [(7 * i + 3) % 256 for i in range(64)]
    a, _ = field_inputs(256)
    prev = np.random.default_rng(1).integers(0, 256, (64, 32), dtype=np.uint8)
    prev[:8] = 0
    return dict(h64=lib.blake3_64(blk), md=lib.hash_md(a, prev))


def case_merkle(lib):
    a, _ = field_inputs(1024)
    l0 = np.random.default_rng(2).integers(0, 256, (1024, 32), dtype=np.uint8)
    return dict(mt4=lib.mt_commit_blake(a[:4]), mt32=lib.mt_commit_blake(a[:32]), mt1024=lib.mt_commit_blake(a),
                tree1024=lib.create_tree_blake(l0), tree2=lib.create_tree_blake(l0[:2]), tree1=lib.create_tree_blake(l0[:1]))


def _graph_dump(lib, n, full):
    out = {}
    dep, m = 0, n
    while m > 13:
        for kind in (0, 1):
            g = lib.graph(dep, kind)
            key = "n%d_d%d_k%d_" % (n, dep, kind)
            out[key + "dims"] = np.array([g["L"], g["R"], g["degree"]], dtype=np.int64)
            if full:
                out[key + "nbr"] = g["nbr"]; out[key + "w"] = g["w"]
            else:
                out[key + "nbr_dg"] = dg(g["nbr"]); out[key + "w_dg"] = dg(g["w"])
        m = int(0.211 * m); dep += 1
    return out


def case_graph_encode(lib):
    out = {}
    for n in (4, 13, 14, 16, 64, 100, 256, 1024, 4096):
        lib.rng_reset()
        out["len_%d" % n] = np.array([lib.expander_init_store(n)], dtype=np.int64)
        out.update(_graph_dump(lib, n, full=(n <= 64)))
        if hasattr(lib, "encode_reset_scratch"):
            lib.encode_reset_scratch()
        for tag, src in (("small", (splitmix_field(n, 40 + n) % np.uint64(1 << 32)) * np.array([1, 0], dtype=np.uint64)),
                         ("full", splitmix_field(n, 41 + n))):
            d, ln = lib.encode_monolithic(src)
            out["enc_%s_%d_len" % (tag, n)] = np.array([ln], dtype=np.int64)
            if n <= 256 and tag == "full":
                out["enc_%s_%d" % (tag, n)] = d
            else:
                out["enc_%s_%d_dg" % (tag, n)] = dg(d)
    # full-range F_{p^2} weights on the n=64 graph (the reference only ever draws 31-bit weights)
    lib.rng_reset(); lib.expander_init_store(64)
    if hasattr(lib, "encode_reset_scratch"):
        lib.encode_reset_scratch()
    for kind in (0, 1):
        g = lib.graph(0, kind)
        lib.graph_set_weights(0, kind, splitmix_field(g["L"] * g["degree"], 100 + kind))
    d, ln = lib.encode_monolithic(splitmix_field(64, 5))
    out["enc_fullw_64"] = d
    return out


def case_fft(lib):
    out = {}
    for logn in (1, 4, 8, 12):
        x = splitmix_field(1 << logn, 7 + logn)
        for inv in (0, 1):
            y = lib.fft(x, bool(inv))
            if logn <= 8:
                out["fft_%d_%d" % (logn, inv)] = y
            else:
                out["fft_%d_%d_dg" % (logn, inv)] = dg(y); out["fft_%d_%d_s" % (logn, inv)] = samp(y)
    # zero-padded message rows, as compute_tensorcode feeds them
    x = splitmix_field(4096, 3); x[2048:] = 0
    out["fft_pad_dg"] = dg(lib.fft(x, False)); out["fft_pad_s"] = samp(lib.fft(x, False))
    for k in (1, 5, 10):
        b = lib.precompute_beta(splitmix_field(k, 3))
        out["beta_%d" % k] = b if k <= 5 else dg(b)
    out["eval"] = lib.evaluate_vector(splitmix_field(1024, 4), splitmix_field(12, 5))
    lib.rng_reset()
    out["genrand"] = lib.generate_randomness(1000)
    return out


def case_tensorcode(lib):
    out = {}
    for (M, trs) in ((1 << 13, 4), (1 << 15, 16), (1 << 17, 64)):
        for lin in (0, 1):
            lib.rng_reset(); lib.expander_init_store(trs)
            if hasattr(lib, "encode_reset_scratch"):
                lib.encode_reset_scratch()
            t = lib.compute_tensorcode(splitmix_field(M, 11), trs, lin)
            key = "tc_%d_%d_%d" % (M, trs, lin)
            out[key + "_dg"] = dg(t); out[key + "_s"] = samp(t)
    return out


COMMIT_CASES = ((1 << 18, 32), (1 << 20, 32), (1 << 20, 16))
QUERIES = ((0, 0), (5, 3), (4095, -1), (100, -2), (2048, 7))   # (col,row); row<0 -> 2*trs+row, -2 -> trs


def test_pc_inputs(lib, N, K):
    """test_PC(N,4,K) input sequence (src/Our_PC.cpp:757-813): poly, then graphs, from a fresh RNG."""
    trs = N // (K << 11)
    lib.rng_reset()
    poly = lib.generate_randomness(N)
    lib.expander_init_store(trs)
    if hasattr(lib, "encode_reset_scratch"):
        lib.encode_reset_scratch()
    return poly, trs


def case_commit(lib):
    out = {}
    for (N, K) in COMMIT_CASES:
        poly, trs = test_pc_inputs(lib, N, K)
        M = N // K
        lv, T = lib.commit_standard(poly, K, trs, 1, want_tensor=True)
        key = "c_%d_%d_" % (N, K)
        out[key + "root"] = lv[-1].copy()
        off, sz, lvl = 0, M, 0
        dgs = []
        while sz >= 1:
            dgs.append(dg(lv[off:off + sz])); off += sz; sz //= 2; lvl += 1
        out[key + "level_dg"] = np.stack(dgs)
        out[key + "tensor_dg"] = dg(T); out[key + "tensor_s"] = samp(T)
        paths = []
        for (c, rw) in QUERIES:
            row = rw if rw >= 0 else (2 * trs - 1 if rw == -1 else trs)
            row = min(row, 2 * trs - 1)
            if lib.__class__.__name__ == "Ref":
                paths.append(lib.open_tree_blake(c, row, 4096))
            else:
                paths.append(lib.open_tree_blake(lv, M, c, row, 4096))
        out[key + "paths"] = np.stack(paths)
        if lib.__class__.__name__ == "Ref":
            lib.release_commit()
        if K == 32:
            # _aggregate (src/Our_PC.cpp:258-289) on test_PC's own evaluation point (drawn after the commit, :822): the aggregate and
            # the roots of its two inner commitments C_f, C_c
            x = lib.generate_randomness(N.bit_length() - 1)
            aggr, roots = lib.aggregate_roots(poly, lib.precompute_beta(x[:5]), trs)
            out[key + "aggr_dg"] = dg(aggr); out[key + "cfcc"] = roots
    # full-range polynomial (img != 0 everywhere), 2^18 / K=32 and RSxRS (linear_time=false, trs=4)
    poly = splitmix_field(1 << 18, 77)
    for lin in (0, 1):
        lv, T = lib.commit_standard(poly, 32, 4, lin, want_tensor=True)
        out["cfull_%d_root" % lin] = lv[-1].copy(); out["cfull_%d_lv_dg" % lin] = dg(lv); out["cfull_%d_t_dg" % lin] = dg(T)
    if lib.__class__.__name__ == "Ref":
        lib.release_commit()
    pb = splitmix_field(1 << 14, 77); beta = splitmix_field(16, 78)
    out["aggr"] = lib.aggregate(pb, beta)
    return out


COMMIT_RS_CASES = ((1 << 18, 32), (1 << 20, 16), (1 << 22, 32))


def case_commit_rs(lib):
    """test_PC(N, 1, K) (src/Our_PC.cpp:764-777): poly = generate_randomness(N) from a fresh generator, linear_time == false, tensor_row_size = 128,
    commit_standard -- the RS x RS Our_PC commitment of option 1 and of prove_circuit_standard's circuit polynomial; rows of 128 / 1024 / 2048 points"""
    out = {}
    for (N, K) in COMMIT_RS_CASES:
        lib.rng_reset()
        poly = lib.generate_randomness(N)
        M = N // K; cols = 2 * M // 128
        lv, T = lib.commit_standard(poly, K, 128, 0, want_tensor=True)
        key = "crs_%d_%d_" % (N, K)
        out[key + "root"] = lv[-1].copy()
        off, sz, dgs = 0, M, []
        while sz >= 1:
            dgs.append(dg(lv[off:off + sz])); off += sz; sz //= 2
        out[key + "level_dg"] = np.stack(dgs)
        out[key + "tensor_dg"] = dg(T); out[key + "tensor_s"] = samp(T)
        paths = []
        for (c, row) in ((0, 0), (5, 3), (cols - 1, 255), (cols // 2, 128), (7, 127)):
            if lib.__class__.__name__ == "Ref":
                paths.append(lib.open_tree_blake(c, row, cols))
            else:
                paths.append(lib.open_tree_blake(lv, M, c, row, cols))
        out[key + "paths"] = np.stack(paths)
        if lib.__class__.__name__ == "Ref":
            lib.release_commit()
    return out


def sumcheck_inputs(n):
    v1 = splitmix_field(n, 1); v2 = splitmix_field(n, 2); v3 = splitmix_field(n, 3)
    v2z = v2.copy(); v2z[0:n // 2] = 0
    v1z = v1.copy(); v1z[n // 4:n // 2] = 0
    return v1, v2, v3, v1z, v2z


def case_sumcheck(lib):
    out = {}
    pr = np.array([33, 0], dtype=np.uint64)
    for n in (2, 4, 32, 1024, 1 << 16):
        v1, v2, v3, v1z, v2z = sumcheck_inputs(n)
        for tag, res in (("s2", lib.sumcheck2(v1, v2, pr)), ("s3", lib.sumcheck3(v1, v2, v3, pr)), ("s3z", lib.sumcheck3(v1z, v2z, v3, pr))):
            for k, v in res.items():
                out["%s_%d_%s" % (tag, n, k)] = v
    # table 2 = eq-table of full-range challenges (the C2 benchmark shape, at 2^12)
    v1 = splitmix_field(1 << 12, 1); v2 = lib.precompute_beta(splitmix_field(12, 9))
    for k, v in lib.sumcheck2(v1, v2, pr).items():
        out["s2beta_%s" % k] = v
    return out


def case_elastic(lib):
    out = {}
    B = 1 << 14
    for opt in (1, 2):
        lib.rng_reset()
        if hasattr(lib, "encode_reset_scratch"):
            lib.encode_reset_scratch()
        lv = lib.elastic_commit(1 << 18, B, opt)
        T = 4 * B
        # leaf T-1 is undefined in the reference (out-of-bounds read, see hobbit_oracle.c); it is an
        # odd leaf and never feeds a parent (left|left quirk), so everything else is pinned.
        out["el_%d_root" % opt] = lv[-1].copy()
        out["el_%d_leaves_dg" % opt] = dg(lv[:T - 1])
        out["el_%d_upper_dg" % opt] = dg(lv[T:])
    return out


ELASTIC_OPEN_CASES = ((1 << 18, 1 << 14), (1 << 19, 1 << 16))


def elastic_open_inputs(N, B, nq=700, seed=901):
    """Elastic_PC::open's own draws (src/Elastic_PC.cpp:645-655) from a seeded libc generator: x (splitmix), then r_v[0] =
    generate_randomness(1) (one random(), one rand()), then nq x (rand() % cols, rand() % 2trs).  Returns (x, I)."""
    import ctypes
    libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
    trs = B >> 11; cols = 2 * B // trs
    x = splitmix_field(N.bit_length() - 1, 900)
    libc.srandom(seed); libc.random(); libc.rand()
    I = np.zeros((nq, 2), np.uint64)
    for q in range(nq):
        I[q, 0] = libc.rand() % cols; I[q, 1] = libc.rand() % (2 * trs)
    return x, I


ELASTIC_OPEN2_CASES = ((1 << 20, 1 << 16), (1 << 22, 1 << 18), (1 << 24, 1 << 20))


def elastic_open2_inputs(N, B, nq=5900, seed=902):
    """Elastic_PC::open's own draws under linear_time (src/Elastic_PC.cpp:626-655; option 2: 5900 queries, tensor_row_size = B / 2^14) from a
    seeded libc generator, as elastic_open_inputs does for option 1.  Returns (x, I)."""
    import ctypes
    libc = ctypes.CDLL(None); libc.random.restype = ctypes.c_long
    trs = B >> 14; cols = 2 * B // trs
    x = splitmix_field(N.bit_length() - 1, 903)
    libc.srandom(seed); libc.random(); libc.rand()
    I = np.zeros((nq, 2), np.uint64)
    for q in range(nq):
        I[q, 0] = libc.rand() % cols; I[q, 1] = libc.rand() % (2 * trs)
    return x, I


def elastic_open_queries(B, nq, seed):
    trs = B >> 11
    return np.random.default_rng(seed).integers(0, [2 * B // trs, 2 * trs], (nq, 2)).astype(np.uint64)


def case_elastic_open(lib):
    """Elastic_PC::open option 1 (RS x RS), the two stream passes that do not reach SHA3: aggregate (src/Elastic_PC.cpp:316-347: axpy over
    the read_stream default "test" stream + shockwave_commit) and compute_aggregation_reply / update_reply (:487-533, 59-111), on the
    chunk coefficients and queries open() itself derives (:638-655)"""
    out = {}
    for (N, B) in ELASTIC_OPEN_CASES:
        key = "eo_%d_%d_" % (N, B)
        x, I = elastic_open_inputs(N, B)
        aggr, root = lib.elastic_aggregate(N, B, lib.precompute_beta(x[:(N // B).bit_length() - 1]))
        out[key + "aggr_dg"] = dg(aggr); out[key + "aggr_s"] = samp(aggr); out[key + "cf_root"] = root
        rep = lib.elastic_reply(N, B, I)
        out[key + "I_dg"] = dg(I); out[key + "reply_dg"] = dg(rep); out[key + "reply_s"] = samp(rep.reshape(-1, 2))
    out["read_stream"] = dg(lib.read_stream(4096))
    return out


def case_elastic_open2(lib):
    """Elastic_PC::open option 2 (RS x expander, test_Elastic_PC(N, 2)), the stream passes that do not reach SHA3, run in the real reference AS
    BUILT: aggregate()'s linear_time branch (src/Elastic_PC.cpp:348-413: the aggregate, C_f, the expander codewords aux_commit of the queried
    columns that have a parity-row query, C_c over them) and compute_aggregation_reply -> update_reply_spielman (:431-533, including its
    read of buff2 past size(): oracle/check_elastic_open2_determinism.py), on the coefficients and queries open() itself derives.
    B = 2^16: tensor_row_size 4 (the expander code is the identity, parity rows are zero); B = 2^18: 16; B = 2^20: 64, test_OurPC.sh's own shape."""
    out = {}
    for (N, B) in ELASTIC_OPEN2_CASES:
        key = "eo2_%d_%d_" % (N, B)
        lib.rng_reset()
        if hasattr(lib, "encode_reset_scratch"):
            lib.encode_reset_scratch()
        lib.expander_init_store(B >> 14)
        x, I = elastic_open2_inputs(N, B)
        a = lib.elastic_aggregate2(N, B, lib.precompute_beta(x[:(N // B).bit_length() - 1]), I)
        out[key + "aggr_dg"] = dg(a["aggr"]); out[key + "cf_root"] = a["cf_root"]; out[key + "cc_root"] = a["cc_root"]
        out[key + "nr"] = np.array([a["aux"].shape[0]]); out[key + "aux_dg"] = dg(a["aux"]); out[key + "aux_s"] = samp(a["aux"].reshape(-1, 2))
        rep = lib.elastic_reply2(N, B, I)
        out[key + "I_dg"] = dg(I); out[key + "reply_dg"] = dg(rep); out[key + "reply_s"] = samp(rep.reshape(-1, 2))
    return out


STREAM_SHAPES = ((1 << 15, 1 << 10, 0, 1, 1), (1 << 15, 1 << 10, 2, 1, 1), (1 << 15, 1 << 10, 1, 2, 2), (1 << 17, 1 << 10, 1, 3, 2), (1 << 18, 1 << 12, 0, 1, 1))


def stream_batch_inputs(fd_size, B, layer_id, batches, distance):
    size = fd_size >> layer_id
    rlen = (size // 2).bit_length() - 1
    return splitmix_field(batches * rlen, 50 + layer_id).reshape(batches, rlen, 2), splitmix_field(batches, 60)


def case_streamdrv(lib):
    """Streaming multiplication-tree prover on the reference's default "test" stream: read_mul_tree_layer / read_mul_tree_data
    (src/witness_stream.cpp:2413-2510), generate_claims_opt (src/sumcheck.cpp:1014-1054),
    generate_3product_sumcheck_beta_stream_batch_optimized (:1150-1393: new claims and challenge rows, functions of every transcript
    message and libc draw in it) and prove_multiplication_tree_stream_shallow's output (:1746-1915)"""
    import ctypes
    libc = ctypes.CDLL(None)
    out = {}
    for layer in (1, 2, 4):
        out["layer_%d" % layer] = dg(lib.read_mul_tree_layer(1 << 16, 1 << 11, layer))
    for (layer, dist, bt) in ((0, 1, 1), (2, 1, 1), (1, 2, 2), (0, 3, 3)):
        out["data_%d_%d_%d" % (layer, dist, bt)] = dg(lib.read_mul_tree_data(1 << 16, 1 << 12, layer, dist, bt))
    for (fd, B, layer_id, batches, dist) in STREAM_SHAPES:
        rr, oc = stream_batch_inputs(fd, B, layer_id, batches, dist)
        libc.srandom(5)
        res = lib.sumcheck3_stream_batch(fd, B, rr, batches, dist, layer_id, oc)
        key = "gsb_%d_%d_%d_%d_%d_" % (fd, B, layer_id, batches, dist)
        out[key + "claims"] = res["new_claims"]
        for i, row in enumerate(res["new_r"]):
            out[key + "r%d" % i] = row
    out["claims_opt"] = lib.generate_claims_opt(1 << 16, 1 << 10, splitmix_field(16, 70), 2, 1, 2)
    # commit_layers (src/sumcheck.cpp:983-1003) of the batched path: vectors*size = 2^20, B = 2^13, distance 2 -> layers 6, batches 3
    sz, ly, roots = lib.commit_layers(1 << 20, 1 << 13, 3, 1, 2)
    out["cl_sizes"] = sz; out["cl_layers"] = ly; out["cl_roots"] = roots
    libc.srandom(11)
    o = lib.mul_tree_stream_shallow(1 << 15, 1 << 10, 8, 1 << 12, np.array([32, 0], np.uint64), 5, splitmix_field(3, 9))
    out["shallow_out"] = o["output"] if isinstance(o, dict) else o
    return out


def case_codeproofs(lib):
    """evaluate_parity_matrix / prove_linear_code / phiGInit / prepare_matrix / prove_fft / prove_fft_matrix
    (src/sumcheck.cpp:2888-2929, 3223-3235, 2975-3027; src/utils.cpp:694-775)"""
    is_ref = lib.__class__.__name__ == "Ref"
    out = {}
    for n in (16, 64, 256, 1024):
        lib.rng_reset(); ln = lib.expander_init_store(n)
        if hasattr(lib, "encode_reset_scratch"):
            lib.encode_reset_scratch()
        k = (2 * n).bit_length() - 1
        beta = lib.precompute_beta(splitmix_field(k, 300 + n))
        A, l2 = lib.evaluate_parity_matrix(beta, n)
        out["pm_%d" % n] = A if n <= 64 else dg(A)
        out["pm_%d_len" % n] = np.array([l2, ln], np.int64)
        # prove_linear_code on a real codeword: the parity check A . codeword must vanish
        cw, _ = lib.encode_monolithic(splitmix_field(n, 310 + n))
        if is_ref:
            r1, res = lib.prove_linear_code(cw, n, 777 + n)
            out["plc_%d_r1" % n] = r1
        else:
            # the reference draws r1 itself (generate_randomness): reproduce the draw from the same seed
            import ctypes
            libc = ctypes.CDLL(None); libc.srandom(777 + n)
            r1 = lib.generate_randomness(k)
            out["plc_%d_r1" % n] = r1
            res = lib.prove_linear_code(cw, n, r1)
        for kk, v in res.items():
            out["plc_%d_%s" % (n, kk)] = v
    for nn in (1, 2, 5, 10):
        rx = splitmix_field(nn, 320 + nn)
        for ifft in (0, 1):
            g = lib.phi_g_init(rx, (7, 3) if ifft else (1, 0), bool(ifft))
            out["phig_%d_%d" % (nn, ifft)] = g if nn <= 5 else dg(g)
    M = splitmix_field(64 * 256, 330).reshape(64, 256, 2)
    out["prepmat"] = lib.prepare_matrix(M, splitmix_field(8, 331))
    # prove_fft on a vector of 512, prove_fft_matrix on 16 x 256
    m = splitmix_field(512, 340); rr = splitmix_field(10, 341)
    if is_ref:
        # previous_sum must equal q0(0)+q0(1) or the reference exits: take it from a first (unchecked-sum) run
        tmp = lib.prove_fft(m, rr, np.zeros(2, np.uint64))          # prove_fft only prints on mismatch
        res = lib.prove_fft(m, rr, lib.claim_of(tmp["poly"][0]))
    else:
        res = lib.prove_fft(m, rr)
    for kk, v in res.items():
        out["pfft_%s" % kk] = v
    Mx = splitmix_field(16 * 256, 350).reshape(16, 256, 2); rr2 = splitmix_field(9 + 4, 351)
    out["pfm_inputs_dg"] = dg(Mx)
    return out


def case_codeproofs_matrix(lib, claim=None):
    """prove_fft_matrix exits the process on a wrong previous_sum, so the reference run needs the claim
    computed by the (already pinned) restatement: gen_golden passes it in."""
    Mx = splitmix_field(16 * 256, 350).reshape(16, 256, 2); rr2 = splitmix_field(9 + 4, 351)
    if lib.__class__.__name__ == "Ref":
        res = lib.prove_fft_matrix(Mx, rr2, claim)
    else:
        res = lib.prove_fft_matrix(Mx, rr2)
    return {"pfm_%s" % kk: v for kk, v in res.items()}


def case_streamfold(lib):
    """compute{2,3,4}p_error_terms (src/sumcheck.cpp:374-432) and one batch_prod step (:1093-1136)"""
    n = 4096
    t = [splitmix_field(n, 400 + i) for i in range(8)]
    gate = (np.arange(n) * 7 % 5 < 2).astype(np.int32)               # 0/1 gate selectors
    out = dict(e2=lib.err2p(t[0], t[1], t[2], t[3]), e3=lib.err3p(t[0], gate, t[1], t[2], t[3], t[4]),
               e4=lib.err4p(t[0], t[1], t[2], gate, t[3], t[4], t[5], t[6]))
    batches, m = 3, 1024
    tb = [splitmix_field(batches * m, 420 + i).reshape(batches, m, 2) for i in range(6)]
    res = lib.batch_prod(tb[0], tb[1], tb[2], tb[3], tb[4], tb[5], np.array([1, 0], np.uint64), splitmix_field(batches, 430), splitmix_field(batches, 431),
                         splitmix_field(1, 432)[0], splitmix_field(batches, 433))
    out.update(bp_rand=res["rand"], bp_Kf=res["Kf"], bp_Kp=res["Kp"], bp_f1=dg(res["f1"]), bp_f2=dg(res["f2"]), bp_f3=dg(res["f3"]))
    return out


def case_lkpfold(lib):
    """compute{3,4}p_error_terms with has_lookups set (src/sumcheck.cpp:388-394, 413-427): the selector values
    prove_gate_consistency_lookups feeds them over its three rewrites (-1 .. 4), lookup_rand from a fixed seed"""
    n = 4096
    t = [splitmix_field(n, 440 + i) for i in range(8)]
    gate = ((np.arange(n) * 7 + 3) % 6 - 1).astype(np.int32)         # -1 .. 4
    lib.set_lookups(splitmix_field(2, 449))
    try:
        out = dict(e3=lib.err3p(t[0], gate, t[1], t[2], t[3], t[4]), e4=lib.err4p(t[0], t[1], t[2], gate, t[3], t[4], t[5], t[6]))
        for v in range(-1, 5):                                        # one selector value at a time: pins each branch of the map
            out["e3_%d" % (v + 1)] = lib.err3p(t[0][:64], np.full(64, v, np.int32), t[1][:64], t[2][:64], t[3][:64], t[4][:64])
            out["e4_%d" % (v + 1)] = lib.err4p(t[0][:64], t[1][:64], t[2][:64], np.full(64, v, np.int32), t[3][:64], t[4][:64], t[5][:64], t[6][:64])
    finally:
        lib.set_lookups(None)
    return out


def case_multree(lib):
    """batch_3product_sumcheck (src/sumcheck.cpp:275-372) and prove_multiplication_tree_new (:35-257)"""
    out = {}
    lens = [256, 64, 256, 4, 1024]
    tot = sum(lens)
    t = [splitmix_field(tot, 500 + i) for i in range(3)]
    res = lib.batch_3product_sumcheck(t[0], t[1], t[2], lens, splitmix_field(len(lens), 503))
    for k, v in res.items():
        out["b3_" + k] = v
    pr = np.array([17, 5], np.uint64)
    for (vectors, size, tag) in ((8, 256, "v8"), (1, 512, "v1"), (4, 64, "v4x")):
        x = splitmix_field(vectors * size, 510 + vectors).reshape(vectors, size, 2)
        px = splitmix_field(max(1, vectors.bit_length() - 1), 520) if tag == "v4x" else None
        if px is None and vectors > 1:
            lib.rng_reset()                                   # the reference draws r with generate_randomness
        res = lib.mul_tree(x, pr, px)
        for k, v in res.items():
            out["mt_%s_%s" % (tag, k)] = v
    return out


def case_innerpcs(lib):
    """shockwave_commit, change_form, whir_commit (src/Virgo.cpp:104-178): FFTs up to 2^17 here"""
    out = {}
    for (N, k) in ((1 << 10, 32), (1 << 16, 32), (1 << 13, 8)):
        p = splitmix_field(N, 800 + k)
        if N == 1 << 13:
            p[N // k:2 * N // k] = 0                    # an all-zero row takes the reference's no-FFT shortcut
        enc, lv = lib.shockwave_commit(p, k)
        out["sw_%d_%d_enc" % (N, k)] = dg(enc); out["sw_%d_%d_root" % (N, k)] = lv[-1].copy(); out["sw_%d_%d_lv" % (N, k)] = dg(lv)
    for logn in (1, 2, 5, 12):
        out["cf_%d" % logn] = dg(lib.change_form(splitmix_field(1 << logn, 810 + logn)))
    for N in (1 << 9, 1 << 14):
        com, lv = lib.whir_commit(splitmix_field(N, 820))
        out["wh_%d_com" % N] = dg(com); out["wh_%d_root" % N] = lv[-1].copy(); out["wh_%d_lv" % N] = dg(lv)
    return out


def case_gate(lib):
    """prove_gate_consistency_standard (src/sumcheck.cpp:434-501): the four tables' element 0 after the degree-4 gate sumcheck --
    a function of every round polynomial through the transcript"""
    from oracle.pyoracle import gate_standard_inputs
    out = {}
    for lg in (1, 3, 8, 12, 14):
        L, R, O, add = gate_standard_inputs(1 << lg, 40 + lg)
        out["gs_%d" % lg] = lib.gate_standard(L, R, O, add, splitmix_field(lg, 60 + lg))
    return out


CASES = dict(streamdrv=case_streamdrv, elastic_open=case_elastic_open, elastic_open2=case_elastic_open2, field=case_field, mimc=case_mimc, blake=case_blake, merkle=case_merkle, graph_encode=case_graph_encode,
             fft=case_fft, tensorcode=case_tensorcode, commit=case_commit, commit_rs=case_commit_rs, sumcheck=case_sumcheck, elastic=case_elastic, codeproofs=case_codeproofs, streamfold=case_streamfold, lkpfold=case_lkpfold, multree=case_multree, innerpcs=case_innerpcs, gate=case_gate)
