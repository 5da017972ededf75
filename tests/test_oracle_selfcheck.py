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
