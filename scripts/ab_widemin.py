"""Same-box A/B of the sliced-ELL geometry threshold (steps with at least HOBBIT_ENC_WIDE_MIN outputs use 64-wide slices, one output per lane):
the graphs are re-finalized per value.  usage: ab_widemin.py 512,256,128 [logN] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
vals = sys.argv[1].split(","); logN = int(sys.argv[2]) if len(sys.argv) > 2 else 28; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << logN, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000)
roots = {}
for v in vals:
    os.environ["HOBBIT_ENC_WIDE_MIN"] = v
    hb.rng_reset(); hb.expander_init_store(trs)
    c = hb.commit_standard((d, N), K, trs, 1); roots[v] = c.root().tobytes().hex(); c.free()
    best = {}; tot = []
    for r in range(reps):
        hb.profile(True); hb.profile_reset()
        hb.timer_begin(); c = hb.commit_standard((d, N), K, trs, 1); ms = hb.timer_end_ms()
        rep = hb.profile_report(); hb.profile(False); c.free(); tot.append(ms)
        for k, (t, n) in rep.items():
            best[k] = min(best.get(k, 1e9), t)
    print("WIDE_MIN=%s: commit min %.2f ms   " % (v, min(tot)) + "  ".join("%s %.2f" % (k, t) for k, t in sorted(best.items(), key=lambda kv: -kv[1]) if t > 0.05))
print("roots identical:", len(set(roots.values())) == 1, list(roots.values())[0][:16])
hb.close()
