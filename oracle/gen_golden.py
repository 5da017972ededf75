"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.

Generates tests/golden/<case>.npz by running tests/golden_cases.py against the REAL reference
(oracle/_ref/libhobbit_ref.so, built from /root/reference by `make -C oracle ref`).  Run here
(where /root/reference exists); the fixtures are committed because the reference cannot travel
to the GPU box.  Usage:  python oracle/gen_golden.py [case ...]
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
import golden_cases  # noqa: E402


def main():
    if not pyoracle.ref_available():
        pyoracle.build_ref()
    ref = pyoracle.Ref()
    names = [a for a in sys.argv[1:] if a in golden_cases.CASES] or ([] if sys.argv[1:] else list(golden_cases.CASES))
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name in names:
        out = golden_cases.CASES[name](ref)
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print("%-14s %3d arrays  %7.1f KB" % (name, len(out), os.path.getsize(path) / 1024.0))
    if not sys.argv[1:] or "codeproofs_matrix" in sys.argv[1:]:
        # prove_fft_matrix exit()s on a wrong claimed sum: take the claim from the restatement
        orc = pyoracle.Oracle()
        claim = orc.claim_of(golden_cases.case_codeproofs_matrix(orc)["pfm_poly"][0])
        out = golden_cases.case_codeproofs_matrix(ref, claim)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "codeproofs_matrix.npz"), **out)
        print("codeproofs_matrix %d arrays" % len(out))


if __name__ == "__main__":
    main()
