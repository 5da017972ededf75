"""TEST INFRASTRUCTURE.  Is the reference's Elastic_PC open, option 2 (RS x expander), deterministic as built?

update_reply_spielman (src/Elastic_PC.cpp:431-485) reads buff2[row >= tensor_row_size] after `buff2 = buff` shrank the vector: a read
past size() but inside the storage `vector<F> buff2(2*tensor_row_size)` retains.  This script runs the REAL reference (oracle/_ref)
through aggregate()'s linear_time branch and compute_aggregation_reply in two FRESH processes per shape (different heap histories: the
second one first allocates and frees a few hundred MB of junk) and compares sha256 digests of everything they return.

    python oracle/check_elastic_open2_determinism.py            # prints one JSON line per shape and a verdict
"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((1 << 20, 1 << 16), (1 << 22, 1 << 18), (1 << 24, 1 << 20))


def child(N, B, junk):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle import pyoracle
    from golden_cases import elastic_open2_inputs
    if junk:                                   # a different heap history: dirty, then free, a lot of small and large blocks
        import ctypes
        libc = ctypes.CDLL(None); libc.malloc.restype = ctypes.c_void_p; libc.free.argtypes = [ctypes.c_void_p]
        ps = []
        for i in range(20000):
            sz = 16 * (1 + (i * 7919) % 4096)
            p = libc.malloc(sz); ctypes.memset(p, 0xA5, sz); ps.append(p)
        for p in ps[::2] + ps[1::2]:
            libc.free(p)
    ref = pyoracle.Ref()
    ref.rng_reset(); ref.expander_init_store(B >> 14)
    x, I = elastic_open2_inputs(N, B)
    a = ref.elastic_aggregate2(N, B, ref.precompute_beta(x[:(N // B).bit_length() - 1]), I)
    rep = ref.elastic_reply2(N, B, I)
    h = lambda v: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
    print("RESULT " + json.dumps(dict(N=N, B=B, junk=junk, nr=int(a["aux"].shape[0]), aggr=h(a["aggr"]), cf_root=h(a["cf_root"]), cc_root=h(a["cc_root"]), aux=h(a["aux"]), reply=h(rep))))


if __name__ == "__main__":
    if len(sys.argv) == 4:
        child(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    ok = True
    for (N, B) in SHAPES:
        outs = []
        for junk in (0, 1):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), str(N), str(B), str(junk)], capture_output=True, text=True, check=True).stdout
            o = [l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]      # (the reference prints its own lines too)
            d = json.loads(o); d.pop("junk"); outs.append(d)
        same = outs[0] == outs[1]
        ok &= same
        print(json.dumps(dict(shape=[N, B], identical=same, run=outs[0] if same else outs)))
    print("VERDICT:", "deterministic as built" if ok else "NOT deterministic")
