#!/bin/bash
for a in 0 1 2 3 8 11; do
  HB_ABL=$a timeout -k 10 200 python bench.py --steps 2 --phase commit --no-cpu-baseline 2>/dev/null > gpurun_out/abl_$a.json
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$a.json")); k=d["kernels_ms_per_step"]
print("abl", $a, round(k["k_encode_A"],2), round(k["k_encode_B"],2))
PY
done
