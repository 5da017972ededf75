"""Is the commit slower right after a light phase?  Per-kernel times of commit_standard(2^28) when it follows (a) another commit directly,
(b) 12 ms of host sleep (GPU idle), (c) the open (latency-bound, ~20 % idle), alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << 28, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000); hb.rng_reset(); hb.expander_init_store(trs)
x = mod.splitmix_field(28, 5)
c = hb.commit_standard((d, N), K, trs, 1); hb.open_core((d, N), c, x, 5900, full=True); c.free()
def commit_prof():
    hb.profile(True); hb.profile_reset()
    c = hb.commit_standard((d, N), K, trs, 1)
    rep = hb.profile_report(); hb.profile(False)
    return c, {k: t for k, (t, n) in rep.items()}
res = {"after commit": [], "after 12 ms idle": [], "after open": []}
for it in range(4):
    c, _ = commit_prof(); c.free()
    c, r = commit_prof(); res["after commit"].append(r)
    hb.sync(); time.sleep(0.012)
    c2, r = commit_prof(); res["after 12 ms idle"].append(r); c2.free()
    hb.open_core((d, N), c, x, 5900, full=True)
    c3, r = commit_prof(); res["after open"].append(r); c3.free(); c.free()
for k, runs in res.items():
    names = ("k_fft4096", "k_transpose", "k_encode_A", "k_encode_B", "k_leaf_chain")
    print("%-18s " % k + "  ".join("%s %.2f" % (n.replace("k_", ""), min(r[n] for r in runs)) for n in names) + "   sum %.2f" % min(sum(r[n] for n in names) for r in runs))
hb.close()
