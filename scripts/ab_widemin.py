"""Same-box A/B of a switch that hobbit_graph_finalize reads (HOBBIT_ENC_WIDE_MIN: steps with at least that many outputs use 64-wide slices;
HOBBIT_ENC_BANK_SCHED: bank-conflict-aware slot order of the fat kernels' register-resident records): the graphs are re-finalized per value.
usage: ab_widemin.py VAR=a,b[,c] [logN] [reps]      (a bare list a,b,c means HOBBIT_ENC_WIDE_MIN)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
var, _, vl = sys.argv[1].rpartition("="); var = var or "HOBBIT_ENC_WIDE_MIN"
vals = vl.split(","); logN = int(sys.argv[2]) if len(sys.argv) > 2 else 28; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << logN, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000)
roots = {}
for v in vals:
    os.environ[var] = v
    hb.rng_reset(); hb.expander_init_store(trs)
    c = hb.commit_standard((d, N), K, trs, 1); roots[v] = c.root().tobytes().hex(); c.free()
    best = {}; tot = []
    for r in range(reps):
        hb.profile(True); hb.profile_reset()
        hb.timer_begin(); c = hb.commit_standard((d, N), K, trs, 1); ms = hb.timer_end_ms()
        rep = hb.profile_report(); hb.profile(False); c.free(); tot.append(ms)
        for k, (t, n) in rep.items():
            best[k] = min(best.get(k, 1e9), t)
    print(var + "=%s: commit min %.2f ms   " % (v, min(tot)) + "  ".join("%s %.2f" % (k, t) for k, t in sorted(best.items(), key=lambda kv: -kv[1]) if t > 0.05))
print("roots identical:", len(set(roots.values())) == 1, list(roots.values())[0][:16])
hb.close()
