"""BASELINE config 4 end to end, with the reference's own witness generator.

The reference's `main()` (Seval oracle thread, witness / wiring / trace streams, prove_circuit) runs out of oracle/_ref/libhobbit_ref.so
-- the real reference compiled from its sources where they lie, test infrastructure -- and every prover function of the path
(init_commitment, commit, prove_multiplication_tree_stream_shallow, prove_gate_consistency[_lookups], open, generate_randomness,
mimc_hash ...) is answered by the device-backed mirror loaded in front of it (tests/mlp_e2e.py: ELF symbol interposition).  What is
checked:

  * the run completes: every consistency check the reference keeps in those functions and the mirror keeps too (sumcheck 0/1/2 of each
    streaming step, "Error in fft", the WHIR final check, the gate sumcheck's claim chain) exits non-zero on failure;
  * the proof size the run prints is the one the REAL reference prints for `./pigeon 9 18 18 1 4 1024 256 256 16`
    (`Ps : 559.000000 KB`): it fixes the libc rand() sequence over the whole run (the 700 open queries and their
    Merkle-path de-duplication), the reply length, every sumcheck's round count and the WHIR parameters of the RS open, which has no
    other stdout fingerprint;
  * for the lookup circuits the reference's own closing check `prods[0]*prod_f == prods[1]*prod_i` (src/main.cpp:937-942) prints "OK":
    it ties the products our streaming multiplication tree returns over the real "lookup_basic" stream to the reference's tables.

The libraries do not exist without a local build of oracle/_ref (`make -C oracle ref`, needs /root/reference): skipped then.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hobbit-space-efficient-zksnark-with-optimal-prover-time_amd")
NEEDED = [os.path.join(PKG, "libhobbit_hip.so"), os.path.join(PKG, "libhobbit_host_refmode.so"), os.path.join(ROOT, "oracle", "_ref", "libhobbit_ref.so")]

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not all(os.path.exists(p) for p in NEEDED), reason="oracle/_ref or the reference-build mirror is not built")]


def run(args, timeout=300):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mlp_e2e.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout + p.stderr


def final_line(out):
    m = re.search(r"Pt : ([0-9.]+), Ps : ([0-9.]+) KB, Vt : ([0-9.]+), streaming time: ([0-9.]+)", out)
    assert m, out[-2000:]
    return tuple(float(g) for g in m.groups())


def test_mlp_prover_end_to_end():
    rc, out = run([9, 18, 18, 1, 4, 1024, 256, 256, 16])
    assert rc == 0, out[-2000:]
    assert "Error" not in out and "error" not in out, out[-2000:]
    assert [l for l in out.splitlines() if l.startswith("OK ")] == ["OK 3", "OK 2", "OK 1", "OK 0"]
    pt, ps, vt, st = final_line(out)
    assert ps == 559.0, "proof size %r differs from the reference's stdout fingerprint 559.000000 KB" % ps
    print("MLP end to end: Pt %.3f s (streaming %.3f s), Ps %.4f KB" % (pt, st, ps))


@pytest.mark.parametrize("args", [(5, 18, 8, 1), (6, 18, 16, 1), (2, 18, 18, 1)], ids=["aes", "sql_range", "range_lookup"])
def test_lookup_circuits_end_to_end(args):
    """has_lookups branch of prove_circuit (src/main.cpp:887-945; test_aes.sh / sql_test.sh shapes at B = 2^18): two live commitments, two
    streaming multiplication trees (the range circuit's is 6 layers deep: the batched variant with commit_layers / generate_claims_opt /
    open_layers), the lookup gate prover, two opens, then the reference's own product check over its lookup tables"""
    rc, out = run(list(args))
    assert rc == 0, out[-2000:]
    assert ">> Error" not in out and "Error in" not in out, out[-2000:]
    assert out.rstrip().splitlines()[-1] == "OK", out[-2000:]
    final_line(out)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhobbit_ref_O0streams.so")), reason="oracle/_ref O0-streams flavour not built")
@pytest.mark.parametrize("args", [(1, 18, 18, 1), (7, 18, 18, 1)], ids=["arithmetic", "dummy"])
def test_arbitrary_circuit_end_to_end(args):
    """prove_arbitrary_circuit (src/main.cpp:812-857; test_arb.sh shape at B = 2^18): the MLP flow plus an open of the "circuit" stream,
    whose last chunks are all zero (compute_aggregation_reply skips them: shorter replies).  The reference's read_memory_circuit has no
    return statement, so this one loads the flavour of oracle/_ref whose stream readers are compiled as the reference's CMakeLists.txt
    compiles them, without optimisation (oracle/Makefile)."""
    env = dict(os.environ, HOBBIT_E2E_REFLIB="libhobbit_ref_O0streams.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mlp_e2e.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-2000:]
    assert "Error" not in out, out[-2000:]
    assert len([l for l in out.splitlines() if l.startswith("PC : ps")]) == 2
    final_line(out)


def test_prove_circuit_standard_end_to_end():
    """`./pigeon 11 18 18 1`: the reference's OWN prove_circuit_standard (src/main.cpp:985-1087; compiled out of the reference-build
    mirror) over the mirror's commit_standard (RS x RS circuit polynomial, then RS x expander witness: two live commitments),
    prove_multiplication_tree_new (mul_tree_proof returned by value into the reference's layout), prove_gate_consistency_standard and both
    open_standard modes on the real trace / memory streams; every consistency check of those functions exits non-zero on failure."""
    rc, out = run([11, 18, 18, 1])
    assert rc == 0, out[-2000:]
    assert "Error" not in out and "error" not in out, out[-2000:]
    assert out.count(">>OK") == 2, out[-2000:]
    m = re.search(r"Pt : ([0-9.]+), Vt : ([0-9.]+), Ps : ([0-9.]+)", out)
    assert m, out[-2000:]
    assert float(m.group(3)) > 0


def _transcript_cases():
    import json
    path = os.path.join(ROOT, "tests", "golden", "transcripts.json")
    if not os.path.exists(path):
        return []
    return [(k, v) for k, v in json.load(open(path)).items() if isinstance(v, dict)]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_openstub.so")), reason="oracle/_ref/libref_openstub.so is not built")
@pytest.mark.parametrize("name,want", _transcript_cases(), ids=[k for k, _ in _transcript_cases()])
def test_transcript_matches_reference(name, want):
    """The orchestration of the streaming provers against the REAL reference, message by message.  tests/golden/transcripts.json holds, per
    command, what a call-through recorder in front of the real reference's mimc_hash captured while the reference's OWN commit,
    prove_multiplication_tree_stream_shallow (twice for the lookup circuits) and prove_gate_consistency / prove_gate_consistency_lookups ran on
    the reference's own Seval streams (oracle/gen_transcripts.py, tests/ref_transcript.py, oracle/ref_recorder.cpp): every Fiat-Shamir hash --
    (input, key, result), i.e. every round polynomial coefficient, every claim that is hashed and every challenge, in order.  Here the same
    command runs with the device-backed mirror answering those functions and the library's own recorder on (hobbit_transcript_record): the
    number of hashes, the sha256 of the whole sequence, of every 4096-record block and the `Ps` the run prints must be identical.  Both runs
    end at the first Elastic_PC::open (it needs SHA3 in the reference; the open has its own tests and the `Ps : 559` fingerprint)."""
    import json
    env = dict(os.environ, HOBBIT_E2E_TRANSCRIPT="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mlp_e2e.py")] + want["cmd"].split(), capture_output=True, text=True, timeout=600, env=env)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("TRANSCRIPT ")]
    assert lines, out[-2000:]
    got = json.loads(lines[-1][11:])
    assert got["count"] == want["count"], (got["count"], want["count"])
    first_bad = next((i for i, (a, b) in enumerate(zip(got["blocks"], want["blocks"])) if a != b), None)
    assert first_bad is None, "transcripts diverge in block %d (records %d..)" % (first_bad, first_bad * 4096)
    assert got["sha256"] == want["sha256"] and got["first"] == want["first"] and got["last"] == want["last"]
    m = re.search(r"Ps : ([0-9.]+) KB", out)
    assert m and float(m.group(1)) == want["ps_truncated_run"], out[-500:]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_openstub.so")), reason="oracle/_ref is not built")
def test_standard_prover_transcript_matches_reference(tmp_path):
    """`./pigeon 11 18 18 1` (prove_circuit_standard, src/main.cpp:985-1087) against the REAL reference, hash for hash, as far as the reference can go:
    oracle/gen_open_transcript.py ran the reference's own main() under the call-through mimc_hash recorder until it died on its first SHA3 call (inside
    the first open_standard; the prebuilt library is not linked, nothing stands in for it) -- 1298 transcript hashes: both commitments' challenges,
    prove_multiplication_tree_new over the memory fingerprints, prove_gate_consistency_standard, then the opening's P1..P4 and its first
    shockwave_prove's sumchecks (tests/golden/standard_transcript.npz).  Here the same command runs with the device-backed mirror answering those
    functions (one thread) and the library's own recorder on: every reference record must appear, in order, in the library's transcript, which may
    hash at most one block in between that the reference only reaches later (P5 before shockwave_prove(C_c), as in the Our_PC opening tests)."""
    import json
    import numpy as np
    ref = np.load(os.path.join(ROOT, "tests", "golden", "standard_transcript.npz"), allow_pickle=False)["records"]
    summ = json.load(open(os.path.join(ROOT, "tests", "golden", "open_transcripts.json")))["main_standard_11_18_18_1"]
    assert ref.shape == (summ["count"], 6) and summ["count"] > 1000
    dump = str(tmp_path / "mine.npy")
    env = dict(os.environ, HOBBIT_E2E_TRANSCRIPT="1", HOBBIT_OPEN_THREADS="0", HOBBIT_TRANSCRIPT_DUMP=dump)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mlp_e2e.py")] + summ["cmd"].split(), capture_output=True, text=True, timeout=600, env=env)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and os.path.exists(dump), out[-2000:]
    mine = np.load(dump, allow_pickle=False)
    n = mine.shape[0]
    assert n > ref.shape[0]
    i = j = 0; gaps = []
    while i < len(ref) and j < n:
        if np.array_equal(ref[i], mine[j]):
            i += 1; j += 1
            continue
        hit = np.nonzero((mine[j + 1:] == ref[i]).all(axis=1))[0]
        assert len(hit), "reference record %d of %d is nowhere in the library's transcript (library record %d of %d)" % (i, len(ref), j, n)
        gaps.append((j, int(hit[0]) + 1)); j += int(hit[0]) + 1
    assert i == len(ref), "the library's transcript ends before the reference's recorded prefix (%d of %d matched)" % (i, len(ref))
    assert len(gaps) <= 1, "more than one block out of order: %s" % gaps
