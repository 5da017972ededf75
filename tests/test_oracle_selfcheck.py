"""Self-consistency of the oracle restatements that cannot be pinned against oracle/_ref as whole functions:
the reference's own exit(-1) checks and the protocol identities they imply must hold on the CPU restatement."""
import ctypes

import numpy as np
import pytest

from oracle.pyoracle import splitmix_field


def _F(o, *xs):
    return [np.asarray(x, np.uint64).reshape(1, 2) for x in xs]


@pytest.mark.parametrize("logn", [1, 5, 12])
def test_gate_sumcheck_rounds_and_final_claim(oracle, logn):
    """src/sumcheck.cpp:875-929: every round's p(0)+p(1) equals the running sum, and the last running sum equals the gate
    expression evaluated at the six fully folded values."""
    n = 1 << logn
    tabs = [splitmix_field(n, 900 + i) for i in range(6)]
    a = splitmix_field(4, 910); r0 = splitmix_field(1, 911)[0]
    claim = oracle.gate_claim(tabs, a)
    res = oracle.gate_sumcheck(tabs, a, r0, claim)
    assert res["check"].tolist() == [1]
    fin = [res["fin"][i].reshape(1, 2) for i in range(6)]
    mul, add, fm = oracle.f_mul, oracle.f_add, lambda *xs: None
    add_, beta, L, R, O, mul_ = fin
    lr = add(mul(a[0:1], L), mul(a[1:2], R))
    want = add(add(mul(mul(add_, beta), lr), mul(a[2:3], mul(mul(mul_, beta), mul(L, R)))), mul(a[3:4], mul(beta, O)))
    assert np.array_equal(want.reshape(2), res["sum"])
    bad = claim.copy(); bad[1] ^= np.uint64(1)
    assert oracle.gate_sumcheck(tabs, a, r0, bad)["check"].tolist() == [0]


def test_open_standard_selfchecks(oracle):
    """open_standard prover side on test_PC(2^20,4,32)-shaped inputs: 'Error recursion 1/2', prove_fft_matrix's claimed sum,
    and for both shockwave_prove calls the WHIR round checks and 'Error in final verification step'."""
    libc = ctypes.CDLL(None)
    N, K = 1 << 20, 32
    trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(20)
    libc.srandom(31337)
    res = oracle.open_standard(poly, K, trs, x, 5900)
    assert res["checks"].tolist() == [1, 1, 1]
    for sp in ("sp_c", "sp_f"):
        assert res[sp]["wchecks"].tolist() == [1, 1]
        assert int(res[sp]["iters"][0]) >= 1


@pytest.mark.parametrize("logN,K", [(18, 32), (20, 32), (20, 16), (22, 32)])
def test_proof_size_matches_reference_stdout(oracle, logN, K):
    """End-to-end pin of the open restatement against the REAL reference: the proof size test_PC prints (fingerprints recorded from
    the reference binary, tests/golden/ps_fingerprints.json) is a function of every query index the open draws from libc -- in
    open_standard, in both shockwave_prove calls and in every WHIR round -- through the Merkle-path de-duplication.  The restatement,
    run on test_PC's own input sequence with the generator left alone, must reproduce it to the last bit."""
    import json, os
    from oracle.pyoracle import open_proof_size
    fp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ps_fingerprints.json")))["test_PC_ps_KB"]
    N = 1 << logN; trs = N // (K << 11)
    oracle.rng_reset(); poly = oracle.generate_randomness(N); oracle.expander_init_store(trs)
    x = oracle.generate_randomness(logN)                      # test_PC's order: poly, graphs, commit, x, open (src/Our_PC.cpp:758-822)
    res = oracle.open_standard(poly, K, trs, x, 5900)
    assert open_proof_size(res, N, K, trs) == fp["%d,%d" % (logN, K)]


@pytest.mark.parametrize("logN,logB", [(18, 14), (20, 16)])
def test_elastic_open_selfchecks(oracle, logN, logB):
    """Elastic_PC::open option 1 on test_Elastic_PC's own sequence (commit, x = generate_randomness(log N), open): prove_fft_matrix's two
    exit(-1) sum checks inside recursive_prover_RS (P2 against P0.vr[0], P5 against P3.vr[0]), the closing shockwave_prove's WHIR
    checks, and the structure of the replies (one entry per chunk; every chunk of the default stream is alike, so the columns agree)."""
    N, B = 1 << logN, 1 << logB
    oracle.rng_reset()
    lv = oracle.elastic_commit(N, B, 1)
    x = oracle.generate_randomness(logN)
    res = oracle.elastic_open(N, B, x, 700, lv)
    assert res["checks"].tolist() == [1, 1]
    assert res["sp_f"]["wchecks"].tolist() == [1, 1]
    assert res["reply"].shape == (700, N // B, 2)
    assert np.array_equal(res["reply"][:, 0], res["reply"][:, -1])
    assert 0 < int(res["ncols"][0]) <= 700
    # a Merkle path of the commitment re-hashes to its root (left|left rule: the sibling is carried, the parent hashes the left child twice)
    depth = (4 * B).bit_length() - 1
    assert res["paths"].shape == (700, depth, 32)


@pytest.mark.parametrize("distance,naive", [(5, True), (2, False), (1, True)])
def test_mul_tree_stream_shallow_selfchecks(oracle, distance, naive):
    """prove_multiplication_tree_stream_shallow restated (layers <= distance, the config-4 shape; the batched path; the naive path) on a
    stream whose reads all differ (stream kind 1: no counterpart in the reference, which repeats one chunk): every streaming sumcheck's
    K_partial must equal the claim handed down by the previous layer ("Error in sumcheck 0"), P1's claim the folded Kf ("... 1"), and
    sum b.vr P2's claim ("... 2")."""
    import ctypes
    B, vectors, size = 1 << 10, 8, 1 << 12
    oracle.stream_config(1, 4000)
    try:
        ctypes.CDLL(None).srandom(3)
        res = oracle.mul_tree_stream_shallow(vectors * size, B, vectors, size, np.array([32, 0], np.uint64), distance, splitmix_field(3, 9), naive=naive)
    finally:
        oracle.stream_config(0, 0)
    assert len(res["steps"]) == (min(distance, 4) if not naive else 4)
    for st in res["steps"]:
        assert st["checks"].tolist() == [1, 1, 1]


def test_gate_consistency_stream_selfchecks(oracle):
    """prove_gate_consistency's chunk loop, degree-4 sumcheck and Peval pass (src/sumcheck.cpp:796-975) on consistent synthetic gates:
    'Error in gate consistency 1/2/3' must all hold; a corrupted output column must trip check 1."""
    from oracle.pyoracle import gate_standard_inputs
    B, nch = 1 << 8, 8
    parts = [gate_standard_inputs(B, 100 + c) for c in range(nch)]
    L, R, O = [np.concatenate([p[i] for p in parts]) for i in range(3)]
    S = np.concatenate([p[3][:, 0] for p in parts]).astype(np.int32)
    res = oracle.gate_consistency_stream(L, R, O, S, B, splitmix_field(8, 3))
    assert res["checks"].tolist() == [1, 1, 1]
    assert np.array_equal(res["R"][0], np.array([1, 0], np.uint64))
    O2 = O.copy(); O2[3 * B + 5, 0] ^= np.uint64(1)
    assert oracle.gate_consistency_stream(L, R, O2, S, B, splitmix_field(8, 3))["checks"][0] == 0


def test_gate_consistency_lookups_stream_selfchecks(oracle):
    """prove_gate_consistency_lookups (src/sumcheck.cpp:503-795) on consistent synthetic gates with lookup rows: every reference self-check
    (gate consistency 1/2/3, the per-chunk Kf_M recomputation, the Kf_lkp recomputation) must hold; a corrupted addition gate trips check 1,
    a corrupted lookup row does not (its output cancels identically)."""
    from oracle.pyoracle import gate_lookup_inputs
    B, nch = 1 << 8, 8
    parts = [gate_lookup_inputs(B, 300 + c) for c in range(nch)]
    L, R, O, S = [np.concatenate([p[i] for p in parts]) for i in range(4)]
    lr = splitmix_field(2, 77); r = splitmix_field(8, 3)
    res = oracle.gate_consistency_lookups_stream(L, R, O, S, B, r, lr)
    assert res["checks"].tolist() == [1, 1, 1, 1, 1]
    assert np.array_equal(res["R"][0], np.array([1, 0], np.uint64))
    ia = 3 * B + int(np.nonzero(S[3 * B:4 * B] == 0)[0][0]); il = 3 * B + int(np.nonzero(S[3 * B:4 * B] == 2)[0][0])
    O2 = O.copy(); O2[ia, 0] ^= np.uint64(1)
    assert oracle.gate_consistency_lookups_stream(L, R, O2, S, B, r, lr)["checks"][0] == 0
    O3 = O.copy(); O3[il, 0] ^= np.uint64(1)
    assert oracle.gate_consistency_lookups_stream(L, R, O3, S, B, r, lr)["checks"].tolist() == [1, 1, 1, 1, 1]


@pytest.mark.parametrize("logn,K,trs,lin", [(16, 4, 16, 1), (18, 8, 64, 1), (16, 4, 16, 0), (20, 32, 16, 1)])
def test_commit_standard_threaded_restatement_is_identical(oracle, logn, K, trs, lin):
    """bench.py's all-cores CPU leg times orc_commit_standard_mt: every tree level and the tensor must equal the
    single-thread restatement's (which the golden fixtures and oracle/_ref pin) for any thread count, including ones that do not divide the ranges"""
    oracle.rng_reset(); poly = oracle.generate_randomness(1 << logn); oracle.expander_init_store(trs)
    lv, t = oracle.commit_standard(poly, K, trs, lin, want_tensor=True)
    for threads in (1, 3, 8):
        lv_mt, t_mt = oracle.commit_standard_mt(poly, K, trs, lin, threads, want_tensor=True)
        assert np.array_equal(lv, lv_mt) and np.array_equal(t, t_mt), threads


def test_open_transcript_fixture_is_self_consistent():
    """tests/golden/open_transcripts.json (the real reference's open_standard transcript up to its first SHA3 call, oracle/gen_open_transcript.py):
    the records kept in full hash to the digest recorded beside them, the run ended on the unresolved SHA3 symbol and nowhere else"""
    import hashlib, json, os
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "open_transcripts.json")))
    cases = [k for k in fix if k.startswith("test_pc_")]
    assert len(cases) >= 3
    for k in cases:
        d = fix[k]
        rec = np.array(d["records"], np.uint64).reshape(-1, 6)
        assert rec.shape[0] == d["count"] > 200 and hashlib.sha256(rec.tobytes()).hexdigest() == d["sha256"], k
        assert "undefined symbol: SHA3_256" in d["died_with"] and d["rc"] != 0, k
