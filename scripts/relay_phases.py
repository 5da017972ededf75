"""Where do the sporadic ~30 ms stalls of the sharded world-1 step sit?  Times sharded_commit_relay and the phases of sharded_open."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
par = mod.parallel
N, K = 1 << 28, 32; trs = N // (K << 11)
hb.rng_reset(); hb.expander_init_store(trs)
plan = par.ShardPlan(N, K, trs, 1, contiguous=True)
ops = par.HipOps(hb, torch.device("cuda", 0))
d = hb.alloc(16 * N)
for i in range(K):
    hb._chk(hb.lib.hobbit_fill_splitmix(hb.ctx, d.ptr + 16 * plan.M * i, plan.M, 2000 + i))
hb.sync()
x = mod.splitmix_field(28, 5)
marks = []
orig = {}
def wrap(obj, name):
    f = getattr(obj, name); orig[name] = f
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); marks.append((name, 1e3 * (time.perf_counter() - t0))); return r
    setattr(obj, name, g)
for n in ("encode_local", "chain_block", "tree_full", "aggregate_local", "sum_vectors", "open_from_aggregate", "gather_local", "tree_paths", "to_host", "empty_state"):
    wrap(ops, n)
import gc; gc.collect(); gc.disable()          # as bench.py does over its timed region
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    marks.clear()
    t0 = time.perf_counter()
    res = par.sharded_commit_relay(ops, None, plan, 0, (d.ptr, K))
    t1 = time.perf_counter()
    ops.set_local_chunks((d.ptr, K))
    o = par.sharded_open(ops, None, plan, 0, res, x, 5900)
    t2 = time.perf_counter()
    agg = {}
    for n, ms in marks: agg[n] = agg.get(n, 0) + ms
    print("step %2d commit %6.1f open %6.1f | " % (step, 1e3 * (t1 - t0), 1e3 * (t2 - t1)) + " ".join("%s %.1f" % (k, v) for k, v in agg.items()), flush=True)
hb.close()
