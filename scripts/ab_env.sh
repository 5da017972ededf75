#!/bin/bash
# like ab_env.sh, but each setting is a comma-separated list of VAR=value pairs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_env; mkdir -p $O
for s in "$@"; do
  IFS=, read -ra kv <<< "$s"
  env "${kv[@]}" python3 $R/bench.py --steps 20 --no-cpu-baseline --no-dropin > $O/out.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python3 - "$s" $O/out.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = j.get("kernels_ms_per_step", {})
print("%-44s ms/step %.2f median %.2f root %s | leaf %.2f fft %.2f" % (sys.argv[1], j["ms_per_step"], j["step_ms_median"], str(j.get("root"))[:16], k.get("k_leaf_chain", 0), k.get("k_fft4096", 0)))
PY
done
