"""Timings of the other BASELINE.json configs on one MI355X (the north-star line is bench.py):
  C2  2-product sumcheck over 2^24 elements           (scripts/bench_sumcheck.py has the per-kernel view)
  C3  Our_PC commit of 2^26 coefficients (K = 32)
  C4  the MLP prover's math phases at MLP_test.sh's shape (B = 2^18, circuit 2^20: src/main.cpp:862-887) on synthetic resident streams:
      Elastic commit of the 4 * 2^20 witness (RS x RS), prove_multiplication_tree_stream_shallow(8 vectors x 2^20, distance 5),
      prove_gate_consistency over 2^20 gates, Elastic open.  The witness generator (Seval + witness_stream.cpp) is out of scope: what is
      timed is everything the prover computes once a chunk is in HBM (the reference spends 2.4-2.8 s of its 14 s there).
  C5  Elastic_PC streaming commit of 2^30 coefficients, B = 2^20, opt 1 and 2, on ONE GPU (chunks generated on the device side once:
      the reference's default stream repeats the same chunk, src/witness_stream.cpp:2405-2411), and the option-1 open (N = 2^26 too)
Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package

mod = load_package(); hb = mod.Hobbit(0)
splitmix_field = mod.splitmix_field
out = {}

def timed(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    hb.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    hb.sync(); return (time.perf_counter() - t0) / reps

# C2
n = 1 << 24
d1 = hb.fill_splitmix(n, 1); d2 = hb.precompute_beta(splitmix_field(24, 9), keep_on_device=True)
pr = np.array([33, 0], np.uint64)
out["C2_sumcheck2_2e24_ms"] = 1e3 * timed(lambda: hb.generate_2product_sumcheck_proof((d1, n), (d2, n), pr), reps=5)
del d1, d2

# C3
N, K = 1 << 26, 32; trs = N // (K << 11)
d = hb.fill_splitmix(N, 2); hb.rng_reset(); hb.expander_init_store(trs)
def c3():
    c = hb.commit_standard((d, N), K, trs, 1); c.free()
out["C3_commit_2e26_ms"] = 1e3 * timed(c3, reps=5)
del d

# Our_PC, linear_time == false (test_PC option 1: RS x RS, tensor_row_size = 128, 790 queries) at 2^26 and 2^28
for logN1 in (26, 28):
    N1 = 1 << logN1
    d = hb.fill_splitmix(N1, 4); x1 = splitmix_field(logN1, 8)
    def o1_commit():
        c = hb.commit_standard((d, N1), 32, 128, 0); c.free()
    out["O1_commit_rsrs_2e%d_ms" % logN1] = 1e3 * timed(o1_commit, reps=3)
    c1 = hb.commit_standard((d, N1), 32, 128, 0)
    def o1_open():
        r = hb.open_standard_rs((d, N1), c1, x1, 790)
        assert r["checks"].tolist() == [1, 1]
    out["O1_open_rsrs_2e%d_ms" % logN1] = 1e3 * timed(o1_open, reps=3)
    c1.free(); del d

# gate consistency with lookup gates (prove_gate_consistency_lookups) at config 4's trace shape
g = np.random.default_rng(2); circ_l = 1 << 20; Bl = 1 << 18; z = np.zeros(circ_l, np.uint64)
Sl = g.integers(0, 3, circ_l).astype(np.int32); Ll = g.integers(0, 1 << 30, circ_l).astype(np.uint64); Rl = g.integers(0, 1 << 30, circ_l).astype(np.uint64)
P61 = np.uint64((1 << 61) - 1)
Ol = np.stack([np.where(Sl == 0, Ll + Rl, np.where(Sl == 1, (Ll * Rl) % P61, g.integers(0, 1 << 60, circ_l).astype(np.uint64))), z], 1)
tsrc_l = hb.trace_source(np.stack([Ll, z], 1), np.stack([Rl, z], 1), Ol, Sl, Bl)
rl = splitmix_field(20, 3); lrand = splitmix_field(2, 77)
def gate_lk():
    r = hb.gate_consistency_lookups_stream(tsrc_l, circ_l // Bl, Bl, rl, lrand)
    assert r["checks"].tolist() == [1] * 5
out["gate_consistency_lookups_2e20_s"] = timed(gate_lk, reps=3)
del tsrc_l

# C4: math phases of the MLP prover (reference on one Xeon core: commit 5.08 s, mul-tree 5.53 s, gate 0.80 s, open 2.84 s)
import ctypes
B4 = 1 << 18; circ = 1 << 20
chunk_w = hb.to_device(hb.read_stream_PC(B4))
# (the tree stays in HBM, as in bench.py's Our_PC step; reading all of it back as the reference's host-side MT_hashes is timed separately)
out["C4_commit_witness_4x2e20_s"] = timed(lambda: hb.elastic_commit(4 * circ, B4, 1, chunk=chunk_w, levels="device"), reps=5)
out["C4_commit_witness_with_tree_readback_s"] = timed(lambda: hb.elastic_commit(4 * circ, B4, 1, chunk=chunk_w), reps=3)
lv_host, lv_dev = hb.elastic_commit(4 * circ, B4, 1, chunk=chunk_w, keep_levels=True)
src4 = hb.chunk_source(0)
pr32 = np.array([32, 0], np.uint64); px3 = splitmix_field(3, 9)
def c4_mul():
    r = hb.mul_tree_stream_shallow(src4, 8 * circ, B4, 8, circ, pr32, 5, px3, naive=False)
    assert all(st["checks"].tolist() == [1, 1, 1] for st in r["steps"]) and len(r["steps"]) == 4
out["C4_mul_tree_stream_8x2e20_s"] = timed(c4_mul, reps=3)
g = np.random.default_rng(1); z = np.zeros(circ, np.uint64)
sel = g.integers(0, 2, circ).astype(np.uint64); Lr = g.integers(0, 1 << 30, circ).astype(np.uint64); Rr = g.integers(0, 1 << 30, circ).astype(np.uint64)
P61 = np.uint64((1 << 61) - 1)
O = np.stack([np.where(sel == 1, Lr + Rr, (Lr * Rr) % P61), z], 1)
tsrc = hb.trace_source(np.stack([Lr, z], 1), np.stack([Rr, z], 1), O, sel.astype(np.int32), B4)
rg = splitmix_field(20, 3)
def c4_gate():
    r = hb.gate_consistency_stream(tsrc, circ // B4, B4, rg)
    assert r["checks"].tolist() == [1, 1, 1]
out["C4_gate_consistency_2e20_s"] = timed(c4_gate, reps=3)
chunk_r = hb.to_device(hb.read_stream(B4)); x4 = splitmix_field(22, 4)
def c4_open():
    r = hb.elastic_open(4 * circ, B4, x4, 700, commit_levels=lv_dev, chunk=chunk_r)
    assert r["checks"].tolist() == [1, 1]
out["C4_open_witness_s"] = timed(c4_open, reps=3)
out["C4_math_phases_total_s"] = sum(out[k] for k in ("C4_commit_witness_4x2e20_s", "C4_mul_tree_stream_8x2e20_s", "C4_gate_consistency_2e20_s", "C4_open_witness_s"))
out["C4_reference_one_core_s"] = {"commit": 5.08, "mul_tree": 5.53, "gate": 0.80, "open": 2.84, "note": "SURVEY.md 6; includes 2.4-2.8 s of stream generation"}
del chunk_w, lv_dev, chunk_r

# C5 (one GPU)
chunk = hb.to_device(hb.read_stream_PC(1 << 20))          # the host-side stream generator is not part of the path
for opt in (1, 2):
    hb.rng_reset()
    t = timed(lambda: hb.elastic_commit(1 << 30, 1 << 20, opt, chunk=chunk, levels="device"), reps=2, warm=1)
    out["C5_elastic_commit_2e30_B2e20_opt%d_s" % opt] = t
    if opt == 1:
        out["C5_elastic_commit_2e30_opt1_with_tree_readback_s"] = timed(lambda: hb.elastic_commit(1 << 30, 1 << 20, opt, chunk=chunk), reps=2, warm=0)
# C5 open (option 1): N = 2^26 and 2^30 with B = 2^20 (1024 chunks x 2 passes at 2^30)
chunk_r = hb.to_device(hb.read_stream(1 << 20))
for logN in (26, 30):
    N5 = 1 << logN
    lvh, lvd = hb.elastic_commit(N5, 1 << 20, 1, chunk=chunk, keep_levels=True)
    x5 = splitmix_field(logN, 6)
    def c5_open():
        r = hb.elastic_open(N5, 1 << 20, x5, 700, commit_levels=lvd, chunk=chunk_r)
        assert r["checks"].tolist() == [1, 1]
    out["C5_elastic_open_2e%d_B2e20_opt1_s" % logN] = timed(c5_open, reps=2, warm=1)
    del lvd
# C5 open (option 2, RS x expander: test_OurPC.sh's `./pigeon N 20 2`): 5900 queries, aux_commit, recursive_prover_Spielman_stream
for logN in (26, 30):
    N5 = 1 << logN
    hb.rng_reset()
    lvh, lvd = hb.elastic_commit(N5, 1 << 20, 2, chunk=chunk, keep_levels=True)      # (draws and uploads the graphs for trs = 64)
    x5 = splitmix_field(logN, 6)
    rs = hb.to_device(hb.read_stream(1 << 20))
    def c5_open2():
        r = hb.elastic_open2(N5, 1 << 20, x5, 5900, commit_levels=lvd, chunks=lambda i: rs)
        assert r["checks"].tolist() == [1]
    out["C5_elastic_open_2e%d_B2e20_opt2_s" % logN] = timed(c5_open2, reps=2, warm=1)
    del lvd
print(json.dumps(out))
hb.close()
