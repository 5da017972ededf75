"""oracle/gen_big_roots.py -- TEST INFRASTRUCTURE ONLY.

Runs the REAL reference's commit_standard (oracle/_ref, built from /root/reference) on the exact
test_PC(N,4,32) inputs at the C3 and north-star sizes and records what the GPU tests compare
against: root, sha256 of every Merkle level, sampled leaves, sampled tensor entries, five paths.
One reference core: ~2.5 min at 2^26, ~11 min and ~28 GB at 2^28.

Usage: python oracle/gen_big_roots.py 26 28     -> tests/golden/bigroot_2e26.npz, bigroot_2e28.npz
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

K = 32
NS = 64          # samples


def sample_plan(logn):
    """deterministic sample positions shared with the tests"""
    N = 1 << logn
    M = N // K
    trs = N // (K << 11)
    g = np.random.default_rng(1000 + logn)
    leaves = g.integers(0, M, NS)
    chunk = g.integers(0, K, NS); row = g.integers(0, 2 * trs, NS); col = g.integers(0, 4096, NS)
    row[:4] = [0, trs - 1, trs, 2 * trs - 1]
    q = [(0, 0), (5, 3), (4095, 2 * trs - 1), (100, trs), (2048, 7)]
    return M, trs, leaves, chunk, row, col, q


def main():
    # run from a private copy so that a rebuild of oracle/_ref cannot disturb a long run
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "libhobbit_ref.so")
    shutil.copy(pyoracle.REF_SO, so)
    lib = pyoracle._dlopen_lazy(so)
    lib.ref_init()
    for a in sys.argv[1:]:
        logn = int(a); N = 1 << logn
        M, trs, leaves, chunk, row, col, q = sample_plan(logn)
        lv = np.zeros((2 * M, 32), np.uint8)
        t0 = time.time()
        lib.ref_test_pc_commit.restype = ctypes.c_size_t
        cnt = lib.ref_test_pc_commit(ctypes.c_size_t(N), ctypes.c_int(K), lv.ctypes.data_as(ctypes.c_void_p))
        dt = time.time() - t0
        assert cnt == 2 * M - 1
        out = {"root": lv[cnt - 1].copy(), "leaves_s": lv[leaves].copy(), "ref_seconds": np.array([dt])}
        dgs, off, sz = [], 0, M
        while sz >= 1:
            dgs.append(np.frombuffer(hashlib.sha256(lv[off:off + sz].tobytes()).digest(), np.uint8).copy()); off += sz; sz //= 2
        out["level_dg"] = np.stack(dgs)
        ts = np.zeros((NS, 2), np.uint64)
        for i in range(NS):
            r = np.array([row[i]], np.uint32); c = np.array([col[i]], np.uint32)
            lib.ref_tensor_get(ctypes.c_int(int(chunk[i])), r.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(1),
                               ts[i:i + 1].ctypes.data_as(ctypes.c_void_p))
        out["tensor_s"] = ts
        depth = M.bit_length() - 1
        paths = np.zeros((len(q), depth, 32), np.uint8)
        for i, (c, r) in enumerate(q):
            d = lib.ref_open_tree_blake(ctypes.c_size_t(c), ctypes.c_size_t(r), ctypes.c_int(4096), paths[i].ctypes.data_as(ctypes.c_void_p))
            assert d == depth, (d, depth)
        out["paths"] = paths
        lib.ref_release_commit()
        path = os.path.join(ROOT, "tests", "golden", "bigroot_2e%d.npz" % logn)
        np.savez_compressed(path, **out)
        print("2^%d: reference commit_standard %.1f s, root %s -> %s" % (logn, dt, lv[cnt - 1].tobytes().hex(), path), flush=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
