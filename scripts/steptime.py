"""Host-side breakdown of one bench.py step (2^28, K = 32): commit_standard, open (full prover side), free, plus the open's stage times
(HOBBIT_TRACE=1 prints them from inside the library)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << 28, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000); hb.rng_reset(); hb.expander_init_store(trs)
x = np.stack([np.arange(1, 29, dtype=np.uint64), np.arange(7, 35, dtype=np.uint64)], axis=1)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    hb.sync()
    t0 = time.perf_counter(); c = hb.commit_standard((d, N), K, trs, 1); t1 = time.perf_counter()
    r = hb.open_core((d, N), c, x, 5900, full=True); t2 = time.perf_counter(); c.free(); t3 = time.perf_counter()
    print("commit %.2f ms  open %.2f ms  free %.2f ms  total %.2f" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)), flush=True)
hb.close()
