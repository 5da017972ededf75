#!/bin/bash
# how often does a sharded world-1 step stall?  (bench.py --mode sharded, 40 steps, a few runs; prints median and the outlier steps)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sh_out; mkdir -p $O
for s in "$@"; do
  IFS=, read -ra kv <<< "$s"
  env "${kv[@]}" python3 $R/bench.py --mode sharded --steps 40 --no-cpu-baseline --no-dropin > $O/out.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python3 - "$s" $O/out.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-32s mean %.2f median %.2f outliers %s" % (sys.argv[1], j["ms_per_step"], j["step_ms_median"], [round(x, 1) for x in j["step_ms_outliers"]]))
PY
done
