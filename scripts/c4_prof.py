"""Per-kernel view of config 4's streaming multiplication-tree prover (8 x 2^20, B = 2^18, distance 5) and gate provers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
B4 = 1 << 18; circ = 1 << 20
src4 = hb.chunk_source(0)
pr32 = np.array([32, 0], np.uint64); px3 = mod.splitmix_field(3, 9)
def run():
    return hb.mul_tree_stream_shallow(src4, 8 * circ, B4, 8, circ, pr32, 5, px3, naive=False)
run(); hb.sync()
t0 = time.perf_counter(); run(); hb.sync(); wall = 1e3 * (time.perf_counter() - t0)
hb.profile(True); hb.profile_reset(); run(); rep = hb.profile_report(); hb.profile(False)
tot = sum(t for t, n in rep.values()); nl = sum(n for t, n in rep.values())
print("mul_tree_stream: wall %.2f ms, kernels %.2f ms in %d launches" % (wall, tot, nl))
for k, (t, n) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:14]:
    print("   %-26s %7.3f ms %5d launches %6.1f us each" % (k, t, n, 1e3 * t / n))
hb.close()
