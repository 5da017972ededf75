"""tests/mlp_e2e.py -- TEST INFRASTRUCTURE (run as a subprocess by tests/test_mlp_end_to_end.py).

Config 4 end to end with the REAL witness generator: the reference's own `main()` -- Seval, the witness streams, `prove_circuit()` -- runs
out of oracle/_ref/libhobbit_ref.so (the reference compiled here from its own sources, nothing edited), and every prover function the
device path replaces (init_commitment, commit, prove_multiplication_tree_stream_shallow, prove_gate_consistency, open,
generate_randomness, mimc_hash, ...) resolves to the device-backed C++ mirror instead, which is loaded IN FRONT of it: plain ELF symbol
interposition, RTLD_GLOBAL | RTLD_LAZY (lazy because SHA3_256 stays unresolved in oracle/_ref; nothing on this path reaches it).  The
mirror is the -DHOBBIT_HOST_REFERENCE_BUILD flavour: no stream readers of its own, so its commit / open / sumchecks read the reference's
real "witness", "wiring_consistency_check_opt" and "transcript_stream" streams through the reference's read_stream / read_stream_PC /
read_trace.

usage: mlp_e2e.py 9 18 18 1 4 1024 256 256 16        (MLP_test.sh:1; the reference prints  "Pt : ..., Ps : ... KB, Vt : ..., streaming time: ...")
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hobbit-space-efficient-zksnark-with-optimal-prover-time_amd")
RTLD_LAZY, RTLD_GLOBAL = 0x1, 0x100


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
    if os.environ.get("HOBBIT_E2E_BACKTRACE"):
        import subprocess                           # debugging aid: a backtrace on SIGSEGV/SIGABRT (tests/aux/segv_bt.c), built on demand
        so = os.path.join(ROOT, "tests", "aux", "libsegv_bt.so")
        if not os.path.exists(so):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "aux", "segv_bt.c"), "-ldl"])
        bt = ctypes.CDLL(so, mode=RTLD_GLOBAL | RTLD_LAZY); bt.segv_bt_install()
    record = bool(os.environ.get("HOBBIT_E2E_TRANSCRIPT"))
    h_hip = load(os.path.join(PKG, "libhobbit_hip.so"))
    if record:
        # transcript run: Elastic_PC::open is stood in for exactly as in the reference-side recording (tests/ref_transcript.py, oracle/ref_recorder.cpp),
        # so that both transcripts end where the reference's run has to end (its open needs SHA3); the library records its own transcript hashes
        load(os.path.join(ROOT, "oracle", "_ref", "libref_openstub.so"))
    h_mir = load(os.path.join(PKG, "libhobbit_host_refmode.so"))          # first in the lookup order (after the open stand-in of a transcript run)
    h_ref = load(os.path.join(ROOT, "oracle", "_ref", os.environ.get("HOBBIT_E2E_REFLIB", "libhobbit_ref.so")))
    # the mirror's gate prover reads the trace through a hook: the reference's read_trace
    hook = dlsym(h_mir, b"hobbit_read_trace_hook")
    rt = dlsym(h_ref, b"_Z10read_traceR17stream_descriptorRSt6vectorIN5virgo12fieldElementESaIS3_EES6_S6_RS1_IiSaIiEE")
    assert hook and rt
    ctypes.c_void_p.from_address(hook).value = rt
    # who answers `commit`?  (must be the mirror)
    for sym in (b"_Z6commit17stream_descriptorR5_hashRSt6vectorIS2_IS0_SaIS0_EESaIS4_EE", b"_Z19generate_randomnessi"):
        a_def = dlsym(None, sym); a_mir = dlsym(h_mir, sym)
        assert a_def == a_mir and a_mir, "symbol %s does not resolve to the mirror" % sym.decode()
    ref_main = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p))(dlsym(h_ref, b"main"))
    args = [b"pigeon"] + [a.encode() for a in sys.argv[1:]]
    argv = (ctypes.c_char_p * (len(args) + 1))(*args, None)
    sys.stdout.flush()
    if record:
        ctypes.CFUNCTYPE(None, ctypes.c_int)(dlsym(h_hip, b"hobbit_transcript_record"))(1)
    rc = ref_main(len(args), argv)
    libc.fflush(None)
    if record:
        import json
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from ref_transcript import summarize
        ctypes.CFUNCTYPE(None, ctypes.c_int)(dlsym(h_hip, b"hobbit_transcript_record"))(0)
        n = ctypes.CFUNCTYPE(ctypes.c_size_t)(dlsym(h_hip, b"hobbit_transcript_count"))()
        rec = np.zeros((n, 6), np.uint64)
        ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t)(dlsym(h_hip, b"hobbit_transcript_read"))(rec.ctypes.data, n)
        out = summarize(rec); out["rc"] = rc; out["args"] = sys.argv[1:]
        print("TRANSCRIPT " + json.dumps(out)); sys.stdout.flush()
        if os.environ.get("HOBBIT_TRANSCRIPT_DUMP"):
            np.save(os.environ["HOBBIT_TRANSCRIPT_DUMP"], rec)
    os._exit(rc)                                   # the reference leaves its Seval thread detached and blocked


if __name__ == "__main__":
    main()
