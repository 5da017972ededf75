"""Timings of the other BASELINE.json configs on one MI355X (the north-star line is bench.py):
  C2  2-product sumcheck over 2^24 elements           (scripts/bench_sumcheck.py has the per-kernel view)
  C3  Our_PC commit of 2^26 coefficients (K = 32)
  C5  Elastic_PC streaming commit of 2^30 coefficients, B = 2^20, opt 1 and 2, on ONE GPU (chunks generated on the device side once:
      the reference's default stream repeats the same chunk, src/witness_stream.cpp:2405-2411)
Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package

mod = load_package(); hb = mod.Hobbit(0)
splitmix_field = mod.splitmix_field
out = {}

def timed(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    hb.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    hb.sync(); return (time.perf_counter() - t0) / reps

# C2
n = 1 << 24
d1 = hb.fill_splitmix(n, 1); d2 = hb.precompute_beta(splitmix_field(24, 9), keep_on_device=True)
pr = np.array([33, 0], np.uint64)
out["C2_sumcheck2_2e24_ms"] = 1e3 * timed(lambda: hb.generate_2product_sumcheck_proof((d1, n), (d2, n), pr), reps=5)
del d1, d2

# C3
N, K = 1 << 26, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 2); hb.rng_reset(); hb.expander_init_store(trs)
def c3():
    c = hb.commit_standard((d, N), K, trs, 1); c.free()
out["C3_commit_2e26_ms"] = 1e3 * timed(c3, reps=5)
del d

# C5 (one GPU)
chunk = hb.to_device(hb.read_stream_PC(1 << 20))          # the host-side stream generator is not part of the path
for opt in (1, 2):
    hb.rng_reset()
    t = timed(lambda: hb.elastic_commit(1 << 30, 1 << 20, opt, chunk=chunk), reps=2, warm=1)
    out["C5_elastic_commit_2e30_B2e20_opt%d_s" % opt] = t
print(json.dumps(out))
hb.close()
