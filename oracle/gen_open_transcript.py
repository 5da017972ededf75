"""oracle/gen_open_transcript.py -- TEST INFRASTRUCTURE ONLY (run here, where oracle/_ref exists).

The REAL reference's own test_PC flow -- commit_standard, x = generate_randomness, open_standard (src/Our_PC.cpp:757-826) -- run out of
oracle/_ref/libhobbit_ref.so with the call-through recorder on mimc_hash in front of it (oracle/ref_recorder.cpp), streaming every transcript hash to
a file.  open_standard reaches SHA3 (my_hhash, from the prebuilt lib/libXKCP.a that is never linked) inside its first shockwave_prove: the child
process dies there on the unresolved symbol -- nothing stands in for the library -- and what the reference's own recursive_prover_Spielman hashed
before that point (P1 .. P4, then shockwave_prove(C_c)'s sumcheck and prove_fft, then the rounds of its _whir_prove up to the first hash) is the
fixture: tests/golden/open_transcripts.json, compared hash for hash with the library's own record of the same run on the GPU
(tests/test_gpu_parity.py::test_open_standard_transcript_vs_reference).

usage: gen_open_transcript.py            (sizes in CASES)
       gen_open_transcript.py --child logN K      (internal)
"""
import ctypes, hashlib, json, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(20, 32), (22, 32), (21, 16), (24, 32)]          # (log2 N, K): test_PC(N, 4, K), trs = N / (K 2^11)
# the reference's own drivers, called as they are (ref_run_test_pc / ref_run_test_elastic in oracle/ref_shim.cpp): name -> (entry point, arguments)
DRIVERS = {
    "driver_test_pc4_2e20_K32": ("ref_run_test_pc", (1 << 20, 4, 32)),                 # must equal test_pc_2e20_K32 above (hand-sequenced): checked below
    "driver_test_pc1_2e20_K32": ("ref_run_test_pc", (1 << 20, 1, 32)),                 # RS x RS, tensor_row_size = 128: recursive_prover_RS
    "driver_test_pc1_2e22_K16": ("ref_run_test_pc", (1 << 22, 1, 16)),
    # Elastic_PC::open gives nothing this way: it checks the queried columns first (verify_claim_opt_blake -> SHA3, src/Elastic_PC.cpp:655-700) and dies
    # before its first transcript hash -- zero records for test_Elastic_PC(2^18 / 2^22, 1) and (2^20 / 2^22, 2) (ref_run_test_elastic; run 2026-10, kept out)
}
RTLD_LAZY, RTLD_GLOBAL = 0x1, 0x100


def child_driver(name):
    fn, args = DRIVERS[name]
    libc = ctypes.CDLL(None)
    dlopen = libc.dlopen; dlopen.restype = ctypes.c_void_p; dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
    dlsym = libc.dlsym; dlsym.restype = ctypes.c_void_p; dlsym.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    h_rec = dlopen(os.path.join(ROOT, "oracle", "_ref", "libref_recorder_mimc.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    h_ref = dlopen(os.path.join(ROOT, "oracle", "_ref", "libhobbit_ref.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    assert h_rec and h_ref
    sym = b"_Z9mimc_hashN5virgo12fieldElementES0_"
    assert dlsym(None, sym) == dlsym(h_rec, sym)
    ctypes.CFUNCTYPE(None, ctypes.c_void_p)(dlsym(h_rec, b"rec_set_next"))(dlsym(h_ref, sym))
    ctypes.CFUNCTYPE(None)(dlsym(h_ref, b"ref_init"))()
    ctypes.CFUNCTYPE(None)(dlsym(h_rec, b"rec_start"))()
    if fn == "ref_run_test_pc":
        ctypes.CFUNCTYPE(None, ctypes.c_size_t, ctypes.c_int, ctypes.c_int)(dlsym(h_ref, fn.encode()))(*args)
    else:
        ctypes.CFUNCTYPE(None, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int)(dlsym(h_ref, fn.encode()))(*args)
    print("UNEXPECTED: the driver returned"); sys.stdout.flush()


def child_main(args):
    """the reference's own main() (./pigeon <args>) under the mimc-only recorder"""
    libc = ctypes.CDLL(None)
    dlopen = libc.dlopen; dlopen.restype = ctypes.c_void_p; dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
    dlsym = libc.dlsym; dlsym.restype = ctypes.c_void_p; dlsym.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    h_rec = dlopen(os.path.join(ROOT, "oracle", "_ref", "libref_recorder_mimc.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    h_ref = dlopen(os.path.join(ROOT, "oracle", "_ref", "libhobbit_ref.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    assert h_rec and h_ref
    sym = b"_Z9mimc_hashN5virgo12fieldElementES0_"
    assert dlsym(None, sym) == dlsym(h_rec, sym)
    ctypes.CFUNCTYPE(None, ctypes.c_void_p)(dlsym(h_rec, b"rec_set_next"))(dlsym(h_ref, sym))
    ctypes.CFUNCTYPE(None)(dlsym(h_rec, b"rec_start"))()
    ref_main = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p))(dlsym(h_ref, b"main"))
    a = [b"pigeon"] + [x.encode() for x in args]
    argv = (ctypes.c_char_p * (len(a) + 1))(*a, None)
    ref_main(len(a), argv)
    print("UNEXPECTED: main returned"); sys.stdout.flush()


def child(logn, K):
    libc = ctypes.CDLL(None)
    dlopen = libc.dlopen; dlopen.restype = ctypes.c_void_p; dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
    dlsym = libc.dlsym; dlsym.restype = ctypes.c_void_p; dlsym.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    h_rec = dlopen(os.path.join(ROOT, "oracle", "_ref", "libref_recorder_mimc.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    h_ref = dlopen(os.path.join(ROOT, "oracle", "_ref", "libhobbit_ref.so").encode(), RTLD_LAZY | RTLD_GLOBAL)
    assert h_rec and h_ref
    sym = b"_Z9mimc_hashN5virgo12fieldElementES0_"
    assert dlsym(None, sym) == dlsym(h_rec, sym)
    ctypes.CFUNCTYPE(None, ctypes.c_void_p)(dlsym(h_rec, b"rec_set_next"))(dlsym(h_ref, sym))
    ctypes.CFUNCTYPE(None)(dlsym(h_ref, b"ref_init"))()
    ctypes.CFUNCTYPE(None)(dlsym(h_rec, b"rec_start"))()
    ctypes.CFUNCTYPE(None, ctypes.c_size_t, ctypes.c_int)(dlsym(h_ref, b"ref_test_pc_open"))(1 << logn, K)
    print("UNEXPECTED: open_standard returned"); sys.stdout.flush()


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_transcript import summarize
    out = {"source": "oracle/gen_open_transcript.py: the real reference's commit_standard + open_standard (oracle/_ref) under the mimc_hash recorder, up to the "
                     "first SHA3 call inside shockwave_prove(C_c); see oracle/ref_recorder.cpp"}
    for logn, K in CASES:
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "rec.bin")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(logn), str(K)], capture_output=True, text=True, cwd=td,
                               env=dict(os.environ, HOBBIT_REC_STREAM=f), timeout=3600)
            rec = np.fromfile(f, np.uint64).reshape(-1, 6) if os.path.exists(f) else np.zeros((0, 6), np.uint64)
        d = summarize(rec)
        d["died_with"] = (p.stderr.strip().splitlines() or [""])[-1][-160:]
        d["rc"] = p.returncode
        assert "UNEXPECTED" not in p.stdout and p.returncode != 0 and "SHA3" in d["died_with"], (p.returncode, p.stdout[-300:], p.stderr[-300:])
        # every record as well (a few hundred): the test compares record by record and says where a difference starts
        d["records"] = [[int(v) for v in r] for r in rec]
        out["test_pc_2e%d_K%d" % (logn, K)] = d
        print(logn, K, d["count"], d["sha256"][:16], "|", d["died_with"])
    for name in DRIVERS:
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "rec.bin")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--driver", name], capture_output=True, text=True, cwd=td,
                               env=dict(os.environ, HOBBIT_REC_STREAM=f), timeout=3600)
            rec = np.fromfile(f, np.uint64).reshape(-1, 6) if os.path.exists(f) else np.zeros((0, 6), np.uint64)
        d = summarize(rec)
        d["died_with"] = (p.stderr.strip().splitlines() or [""])[-1][-160:]
        d["rc"] = p.returncode
        assert "UNEXPECTED" not in p.stdout and p.returncode != 0 and "SHA3" in d["died_with"], (name, p.returncode, p.stdout[-300:], p.stderr[-300:])
        if name != "driver_test_pc4_2e20_K32":                       # (that one only has to equal the hand-sequenced case: digest kept, records not repeated)
            d["records"] = [[int(v) for v in r] for r in rec]
        d["driver"] = "%s%s" % DRIVERS[name]
        out[name] = d
        print(name, d["count"], d["sha256"][:16], "|", d["died_with"][-40:])
    assert out["driver_test_pc4_2e20_K32"]["sha256"] == out["test_pc_2e20_K32"]["sha256"], "the hand-sequenced test_PC flow and the reference's own driver differ"
    # `./pigeon 11 18 18 1`: the reference's own prove_circuit_standard (src/main.cpp:985-1087) on its own trace / memory streams -- both commitments,
    # prove_multiplication_tree_new, prove_gate_consistency_standard, then the first open_standard up to its first SHA3 call.  Records as a binary
    # fixture (tests/golden/standard_transcript.npz), summary here.
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "rec.bin")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--main", "11", "18", "18", "1"], capture_output=True, text=True, cwd=td,
                           env=dict(os.environ, HOBBIT_REC_STREAM=f), timeout=3600)
        rec = np.fromfile(f, np.uint64).reshape(-1, 6)
    d = summarize(rec)
    d["died_with"] = (p.stderr.strip().splitlines() or [""])[-1][-160:]; d["rc"] = p.returncode; d["cmd"] = "11 18 18 1"
    assert "UNEXPECTED" not in p.stdout and p.returncode != 0 and "SHA3" in d["died_with"], (p.returncode, p.stdout[-300:], p.stderr[-300:])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "standard_transcript.npz"), records=rec)
    out["main_standard_11_18_18_1"] = d
    print("main 11 18 18 1:", d["count"], d["sha256"][:16], "|", d["died_with"][-40:])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "open_transcripts.json"), "w"), indent=None, separators=(",", ":"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--driver":
        child_driver(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--main":
        child_main(sys.argv[2:])
    else:
        main()
