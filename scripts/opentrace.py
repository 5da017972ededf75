"""Stage-by-stage wall time of one open_standard at 2^28 (HOBBIT_TRACE=1 makes the library synchronise and print after every stage;
the stages are therefore serialised: their sum exceeds the untraced open).  usage: HOBBIT_TRACE=1 python scripts/opentrace.py"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
N, K = 1 << 28, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 1000); hb.rng_reset(); hb.expander_init_store(trs)
x = mod.splitmix_field(28, 5)
for it in range(4):
    c = hb.commit_standard((d, N), K, trs, 1); hb.sync()
    sys.stderr.write("---- open %d\n" % it); sys.stderr.flush()
    t0 = time.perf_counter(); r = hb.open_standard((d, N), c, x, 5900, want_paths=True); t1 = time.perf_counter(); c.free()
    sys.stderr.write("open_standard %.2f ms\n" % (1e3 * (t1 - t0)))
hb.close()
