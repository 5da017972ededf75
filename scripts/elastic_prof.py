"""Per-kernel profile of the streaming Elastic commit (config 5 shape: B = 2^20) through the library's own event brackets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
logN = int(sys.argv[1]) if len(sys.argv) > 1 else 28
B = 1 << 20; N = 1 << logN
chunk = hb.to_device(hb.read_stream_PC(B))
for opt in (1, 2):
    hb.rng_reset()
    hb.elastic_commit(N, B, opt, chunk=chunk)
    hb.sync(); t0 = time.perf_counter()
    hb.elastic_commit(N, B, opt, chunk=chunk)
    hb.sync(); dt = time.perf_counter() - t0
    hb.profile(True); hb.profile_reset()
    hb.elastic_commit(N, B, opt, chunk=chunk)
    rep = hb.profile_report(); hb.profile(False)
    tot = sum(t for t, n in rep.values())
    print("opt %d: %.2f ms wall for %d chunks (%.1f us per chunk); kernels %.2f ms" % (opt, 1e3 * dt, N // B, 1e6 * dt / (N // B), tot))
    for k, (t, n) in sorted(rep.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %-24s %8.3f ms  %5d launches  %7.1f us each" % (k, t, n, 1e3 * t / n))
hb.close()
