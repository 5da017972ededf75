"""Per-kernel profile of the Our_PC RS x RS (test_PC option 1) commit and open."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
for logN in [int(a) for a in sys.argv[1:]] or [26, 28]:
    N = 1 << logN
    d = hb.fill_splitmix(N, 4); x = mod.splitmix_field(logN, 8)
    c = hb.commit_standard((d, N), 32, 128, 0); c.free()
    hb.profile(True); hb.profile_reset()
    hb.timer_begin(); c = hb.commit_standard((d, N), 32, 128, 0); ms = hb.timer_end_ms()
    rep = hb.profile_report()
    print("2^%d commit %.2f ms" % (logN, ms))
    for k, (t, n) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]:
        print("   %-24s %8.3f ms  %5d launches" % (k, t, n))
    hb.open_standard_rs((d, N), c, x, 790)
    hb.profile_reset()
    hb.timer_begin(); r = hb.open_standard_rs((d, N), c, x, 790); ms = hb.timer_end_ms()
    rep = hb.profile_report(); hb.profile(False)
    print("2^%d open %.2f ms (kernels %.2f)" % (logN, ms, sum(t for t, n in rep.values())))
    for k, (t, n) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:10]:
        print("   %-24s %8.3f ms  %5d launches" % (k, t, n))
    c.free(); del d
hb.close()
