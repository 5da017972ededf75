"""oracle/gen_sumcheck_2e24.py -- TEST INFRASTRUCTURE ONLY.

BASELINE config 2 ("sumcheck-only fold over 2^24 fieldElements, bit-exact vs sumcheck.cpp"): runs the REAL reference's
generate_2product_sumcheck_proof (src/sumcheck.cpp:2391-2460, oracle/_ref) once on C2's inputs -- v1 = splitmix_field(2^24, 1),
v2 = precompute_beta(splitmix_field(24, 9)), previous_r = F(33) -- and stores the complete transcript (24 round polynomials, challenges,
vr, final) in tests/golden/sumcheck2_2e24.npz.  tests/test_gpu_parity.py::test_sumcheck2_2e24_claim_consistency compares the device with it.
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle

if __name__ == "__main__":
    ref = pyoracle.Ref()
    n = 1 << 24
    v1 = pyoracle.splitmix_field(n, 1); v2 = ref.precompute_beta(pyoracle.splitmix_field(24, 9)); pr = np.array([33, 0], np.uint64)
    t = time.time(); res = ref.sumcheck2(v1, v2, pr)
    print("reference generate_2product_sumcheck_proof(2^24): %.1f s" % (time.time() - t))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sumcheck2_2e24.npz"), **res)
    print({k: v.shape for k, v in res.items()})
