"""Direct cross-check of the C restatement against the real reference build (oracle/_ref), on
inputs beyond the committed fixtures.  Runs only where oracle/_ref exists (this container, or the
GPU box when the prebuilt library travelled); otherwise skipped -- test_oracle_golden.py is the
portable pin."""
import numpy as np
import pytest
import golden_cases
from oracle.pyoracle import splitmix_field


@pytest.mark.parametrize("name", ["field", "mimc", "blake", "merkle", "fft", "sumcheck", "elastic_open"])
def test_case_agrees(oracle, ref, name):
    a = golden_cases.CASES[name](oracle); b = golden_cases.CASES[name](ref)
    for k in a:
        assert np.array_equal(a[k], b[k]), (name, k)


@pytest.mark.parametrize("n", [14, 20, 50, 128, 333, 512, 2048])
def test_graph_and_encode_random_sizes(oracle, ref, n):
    # both libraries draw from the one process-wide libc generator: reset before EACH draw
    oracle.rng_reset(); l_o = oracle.expander_init_store(n)
    ref.rng_reset(); l_r = ref.expander_init_store(n)
    assert l_o == l_r
    ref.encode_reset_scratch()
    src = splitmix_field(n, n)
    d1, l1 = oracle.encode_monolithic(src); d2, l2 = ref.encode_monolithic(src)
    assert l1 == l2 and np.array_equal(d1, d2)


def test_commit_2e18_full_tensor(oracle, ref):
    poly, trs = golden_cases.test_pc_inputs(oracle, 1 << 18, 32)
    poly2, _ = golden_cases.test_pc_inputs(ref, 1 << 18, 32)
    assert np.array_equal(poly, poly2)
    lv1, t1 = oracle.commit_standard(poly, 32, trs, 1, want_tensor=True)
    lv2, t2 = ref.commit_standard(poly, 32, trs, 1, want_tensor=True)
    ref.release_commit()
    assert np.array_equal(lv1, lv2) and np.array_equal(t1, t2)


def test_elastic_open_passes_other_shape(oracle, ref):
    """aggregate / compute_aggregation_reply of Elastic_PC::open (option 1) at a shape outside the fixtures"""
    N, B = 1 << 17, 1 << 13
    beta = splitmix_field(N // B, 5)
    a1, r1 = oracle.elastic_aggregate(N, B, beta); a2, r2 = ref.elastic_aggregate(N, B, beta)
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)
    I = golden_cases.elastic_open_queries(B, 333, 7)
    assert np.array_equal(oracle.elastic_reply(N, B, I), ref.elastic_reply(N, B, I))
