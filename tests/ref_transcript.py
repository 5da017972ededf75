"""tests/ref_transcript.py -- TEST INFRASTRUCTURE (CPU; run here, where oracle/_ref exists).

Runs the REAL reference's own main() (oracle/_ref/libhobbit_ref.so: Seval, the witness streams, prove_circuit with its own commit,
prove_multiplication_tree_stream_shallow and prove_gate_consistency[_lookups]) with oracle/_ref/libref_recorder.so loaded in front of it:
a call-through recorder on mimc_hash and a stand-in for Elastic_PC::open, which ends in SHA3 and is the last prover call
(oracle/ref_recorder.cpp).  Prints one JSON line: the number of transcript hashes, sha256 of the whole (x, k, result) sequence and of every
block of 4096 records, the first and last record.

usage: ref_transcript.py 9 18 18 1 4 1024 256 256 16
"""
import ctypes, hashlib, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTLD_LAZY, RTLD_GLOBAL = 0x1, 0x100
BLOCK = 4096


def summarize(rec):
    """rec: (n, 6) uint64 -> dict"""
    raw = np.ascontiguousarray(rec, np.uint64)
    blocks = [hashlib.sha256(raw[i:i + BLOCK].tobytes()).hexdigest()[:16] for i in range(0, raw.shape[0], BLOCK)]
    return dict(count=int(raw.shape[0]), sha256=hashlib.sha256(raw.tobytes()).hexdigest(), blocks=blocks,
                first=[int(v) for v in raw[0]] if raw.shape[0] else [], last=[int(v) for v in raw[-1]] if raw.shape[0] else [])


def main():
    libc = ctypes.CDLL(None)
    dlopen = libc.dlopen; dlopen.restype = ctypes.c_void_p; dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
    dlsym = libc.dlsym; dlsym.restype = ctypes.c_void_p; dlsym.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    dlerror = libc.dlerror; dlerror.restype = ctypes.c_char_p

    def load(path):
        h = dlopen(path.encode(), RTLD_LAZY | RTLD_GLOBAL)
        if not h:
            sys.exit("dlopen %s: %s" % (path, dlerror().decode()))
        return h
    h_rec = load(os.path.join(ROOT, "oracle", "_ref", "libref_recorder.so"))          # first in the lookup order
    h_ref = load(os.path.join(ROOT, "oracle", "_ref", os.environ.get("HOBBIT_E2E_REFLIB", "libhobbit_ref.so")))
    for sym in (b"_Z9mimc_hashN5virgo12fieldElementES0_", b"_Z4open17stream_descriptorSt6vectorIN5virgo12fieldElementESaIS2_EERS0_IS0_I5_hashSaIS5_EESaIS7_EERdSB_"):
        assert dlsym(None, sym) == dlsym(h_rec, sym) and dlsym(h_rec, sym), "symbol %s does not resolve to the recorder" % sym.decode()
    nxt = dlsym(h_ref, b"_Z9mimc_hashN5virgo12fieldElementES0_")
    assert nxt and nxt != dlsym(h_rec, b"_Z9mimc_hashN5virgo12fieldElementES0_")
    ctypes.CFUNCTYPE(None, ctypes.c_void_p)(dlsym(h_rec, b"rec_set_next"))(nxt)
    F0 = ctypes.CFUNCTYPE(None); FS = ctypes.CFUNCTYPE(ctypes.c_size_t); FR = ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t)
    F0(dlsym(h_rec, b"rec_start"))()
    ref_main = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p))(dlsym(h_ref, b"main"))
    args = [b"pigeon"] + [a.encode() for a in sys.argv[1:]]
    argv = (ctypes.c_char_p * (len(args) + 1))(*args, None)
    sys.stdout.flush()
    rc = ref_main(len(args), argv)
    libc.fflush(None)
    F0(dlsym(h_rec, b"rec_stop"))()
    n = FS(dlsym(h_rec, b"rec_count"))()
    rec = np.zeros((n, 6), np.uint64)
    FR(dlsym(h_rec, b"rec_read"))(rec.ctypes.data, n)
    other = ctypes.CFUNCTYPE(ctypes.c_uint64)(dlsym(h_rec, b"rec_other_threads"))()
    out = summarize(rec); out["rc"] = rc; out["other_threads"] = int(other); out["args"] = sys.argv[1:]
    print("TRANSCRIPT " + json.dumps(out)); sys.stdout.flush()
    dump = os.environ.get("HOBBIT_TRANSCRIPT_DUMP")
    if dump:
        np.save(dump, rec)
    os._exit(rc)                                   # the reference leaves its Seval thread detached and blocked


if __name__ == "__main__":
    main()
