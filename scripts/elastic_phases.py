"""Where the C4-shape Elastic commit / open spend their wall time: begin (allocations), pushes, finish, free."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
B = 1 << 18; N = 4 << 20; trs = B >> 11
chunk = hb.to_device(hb.read_stream_PC(B))
lv = hb.alloc(32 * 8 * B)
def once():
    t = [time.perf_counter()]
    e = ctypes.c_void_p()
    hb._chk(hb.lib.hobbit_elastic_begin(hb.ctx, B, trs, 0, 1, ctypes.byref(e))); hb.sync(); t.append(time.perf_counter())
    for _ in range(N // B): hb._chk(hb.lib.hobbit_elastic_push(hb.ctx, e, chunk.ptr))
    hb.sync(); t.append(time.perf_counter())
    hb._chk(hb.lib.hobbit_elastic_finish(hb.ctx, e, lv.ptr)); hb.sync(); t.append(time.perf_counter())
    hb.lib.hobbit_elastic_free(e); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
for i in range(5):
    print("begin %.2f  pushes %.2f  finish %.2f  free %.2f ms" % tuple(once()))
t0 = time.perf_counter(); hb.elastic_commit(N, B, 1, chunk=chunk); hb.sync(); print("elastic_commit wrapper %.2f ms" % (1e3 * (time.perf_counter() - t0)))
hb.close()
