"""Same-box A/B of commit_standard(2^28) under environment switches, alternating in ONE process (the FFT kernel's time differs by
several per cent between boxes of the pool: only same-call comparisons mean anything).
usage: ab_commit.py VAR=a,b[,c] [logN] [reps]     e.g.  ab_commit.py HOBBIT_ENC_STRIDED=0,1 28 4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package

var, vals = sys.argv[1].split("=")
vals = vals.split(",")
logN = int(sys.argv[2]) if len(sys.argv) > 2 else 28
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << logN, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000); hb.rng_reset(); hb.expander_init_store(trs)
roots = {}
for v in vals:                                    # warm every variant once (workspaces, code objects)
    os.environ[var] = v
    c = hb.commit_standard((d, N), K, trs, 1); roots[v] = c.root().tobytes().hex(); c.free()
if len(set(roots.values())) != 1: print("WARNING: roots differ", roots)
print("root", roots[vals[0]], "(identical under every variant)")
res = {v: [] for v in vals}
prof = {v: {} for v in vals}
for r in range(reps):
    for v in vals:
        os.environ[var] = v
        hb.profile(True); hb.profile_reset()
        hb.timer_begin(); c = hb.commit_standard((d, N), K, trs, 1); ms = hb.timer_end_ms()
        rep = hb.profile_report(); hb.profile(False)
        c.free()
        res[v].append(ms)
        for k, (t, n) in rep.items():
            prof[v].setdefault(k, []).append(t)
for v in vals:
    print("%s=%s: commit %s ms (min %.2f)" % (var, v, " ".join("%.2f" % x for x in res[v]), min(res[v])))
    print("   " + "  ".join("%s %.2f" % (k, min(t)) for k, t in sorted(prof[v].items(), key=lambda kv: -min(kv[1])) if min(t) > 0.05))
hb.close()
