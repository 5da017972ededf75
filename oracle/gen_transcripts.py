"""oracle/gen_transcripts.py -- TEST INFRASTRUCTURE ONLY.

Runs tests/ref_transcript.py (the REAL reference's main() out of oracle/_ref with a call-through recorder on mimc_hash, oracle/ref_recorder.cpp)
for every command in CONFIGS and writes tests/golden/transcripts.json: per command the number of transcript hashes the reference's own
commit / prove_multiplication_tree_stream_shallow / prove_gate_consistency[_lookups] computed on its own Seval streams up to the first
Elastic_PC::open, sha256 of the whole (input, k, result) sequence and of each block of 4096 records, the first and last record and the
`Ps` the truncated run prints.  tests/test_mlp_end_to_end.py::test_transcript_matches_reference replays each command through the
device-backed mirror on the GPU and compares.
"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "mlp": "9 18 18 1 4 1024 256 256 16",          # MLP_test.sh:1 (BASELINE config 4)
    "aes": "5 18 8 1",                             # test_aes.sh shape: prove_gate_consistency_lookups, two streaming multiplication trees
    "sql_range": "6 18 16 1",
    "range_lookup": "2 18 18 1",                   # its second tree is deeper than `distance`: the recording ends at open_layers' first open
    "arithmetic": "1 18 18 1",                     # prove_arbitrary_circuit
    "dummy": "7 18 18 1",
    "mlp_small_buffer": "9 16 18 1 4 1024 256 256 16",
}

if __name__ == "__main__":
    out = {"source": "oracle/gen_transcripts.py: the real reference (oracle/_ref) run by tests/ref_transcript.py; see oracle/ref_recorder.cpp"}
    names = sys.argv[1:] or list(CONFIGS)
    path = os.path.join(ROOT, "tests", "golden", "transcripts.json")
    if os.path.exists(path) and sys.argv[1:]:
        out = json.load(open(path))
    for name in names:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_transcript.py")] + CONFIGS[name].split(), capture_output=True, text=True, cwd="/tmp", timeout=1800)
        lines = [l for l in p.stdout.splitlines() if l.startswith("TRANSCRIPT ")]
        if p.returncode != 0 or not lines:
            print(name, "FAILED rc", p.returncode, p.stdout[-500:], p.stderr[-500:]); continue
        d = json.loads(lines[-1][11:])
        m = re.search(r"Ps : ([0-9.]+) KB", p.stdout)
        d["ps_truncated_run"] = float(m.group(1)) if m else None
        d["cmd"] = CONFIGS[name]; d.pop("args", None); d.pop("rc", None)
        out[name] = d
        print(name, d["count"], d["sha256"][:16], d["ps_truncated_run"], "other threads:", d["other_threads"])
    json.dump(out, open(path, "w"), indent=1)
