"""C2 config: 2-product sumcheck over 2^24 field elements on one MI355X (tables resident in HBM).
Prints one JSON line with timing, per-kernel breakdown and the fold kernel's HBM roofline."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
splitmix_field = mod.splitmix_field
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
d1 = hb.fill_splitmix(n, 1)
d2 = hb.precompute_beta(splitmix_field(logn, 9), keep_on_device=True)
pr = np.array([33, 0], np.uint64)
for _ in range(2):
    hb.generate_2product_sumcheck_proof((d1, n), (d2, n), pr)
hb.profile(True); hb.profile_reset()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    res = hb.generate_2product_sumcheck_proof((d1, n), (d2, n), pr)
dt = (time.perf_counter() - t0) / reps
prof = hb.profile_report()
# first fused fold+poly launch: reads 2 tables of n, writes 2 of n/2 -> 48n bytes ; round 0 poly: 32n
ms_fold = prof.get("k_sc2_fold_poly", prof.get("k_sc2_double", (0.0, 0)))[0] / reps or 1e-9      # (tables >= 16384 go through k_sc2_double: no k_sc2_poly launch)
ms_poly = prof.get("k_sc2_poly", (0.0, 0))[0] / reps or 1e-9
total_bytes = 96 * n
out = {"config": "2-product sumcheck, n=2^%d" % logn, "seconds": dt, "f_mul_per_s": 6 * n / dt,
       "kernels_ms": {k: v[0] / reps for k, v in prof.items()},
       "hbm_GBs_fold_kernels": (64 * n) / ((ms_fold) * 1e-3) / 1e9, "hbm_GBs_poly0": 32 * n / (ms_poly * 1e-3) / 1e9,
       "algorithmic_bytes": total_bytes, "final_rand": [int(x) for x in res["fin"]]}
print(json.dumps(out))
hb.close()
