"""oracle/gen_elastic_roots.py -- TEST INFRASTRUCTURE ONLY.

Runs the REAL reference's Elastic_PC commit (oracle/_ref, src/Elastic_PC.cpp:174-285) on the synthetic "test" stream at the
sizes the GPU tests check (C5: N = 2^30, B = 2^20, options 1 and 2; and N = 2^26) and records root + level digests.
One reference core, ~2 min at 2^26 and ~35 min at 2^30 per option.

Usage: python oracle/gen_elastic_roots.py <logN> <logB> <opt>   -> tests/golden/elastic_root_<logN>_<logB>_<opt>.npz
"""
import ctypes
import hashlib
import os
import shutil
import sys
import tempfile
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402


def main():
    logN, logB, opt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "libhobbit_ref.so")
    shutil.copy(pyoracle.REF_SO, so)              # a private copy: a rebuild of oracle/_ref cannot disturb a long run
    lib = pyoracle._dlopen_lazy(so)
    lib.ref_init(); lib.ref_rng_reset()           # fresh-process generator state: option 2 draws its graphs first (src/Elastic_PC.cpp:770)
    N, B = 1 << logN, 1 << logB
    lv = np.zeros((8 * B, 32), np.uint8)
    lib.ref_elastic_commit.restype = ctypes.c_size_t
    t0 = time.time()
    cnt = lib.ref_elastic_commit(ctypes.c_size_t(N), ctypes.c_size_t(B), ctypes.c_int(opt), lv.ctypes.data_as(ctypes.c_void_p))
    dt = time.time() - t0
    T = 4 * B
    assert cnt == 2 * T - 1
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()
    # leaf T-1 is undefined in the reference (a read past two heap arrays, see oracle/hobbit_oracle.c); it never feeds a parent
    out = dict(root=lv[cnt - 1].copy(), leaves_dg=sha(lv[:T - 1]), upper_dg=sha(lv[T:cnt]), ref_seconds=np.array([dt]))
    path = os.path.join(ROOT, "tests", "golden", "elastic_root_%d_%d_%d.npz" % (logN, logB, opt))
    np.savez_compressed(path, **out)
    print("Elastic 2^%d B=2^%d opt %d: reference commit %.1f s, root %s" % (logN, logB, opt, dt, lv[cnt - 1].tobytes().hex()), flush=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
