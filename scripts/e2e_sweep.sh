#!/bin/bash
# runs tests/mlp_e2e.py over a list of reference CLI lines (one GPU process at a time; stops after a timeout/kill)
out=gpurun_out/e2e; mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "== [$i] pigeon $line"
  timeout -k 10 ${E2E_TIMEOUT:-180} python3 tests/mlp_e2e.py $line > $out/run_$i.log 2>&1
  rc=$?
  echo "rc=$rc"; tail -n 6 $out/run_$i.log
  if [ $rc -ge 124 ] && [ $rc -ne 255 ]; then echo "stopping after rc=$rc"; break; fi
done
