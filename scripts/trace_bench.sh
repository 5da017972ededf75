#!/bin/bash
# rocprofv3 kernel trace of three default bench steps (environment switches pass through); the CSV stays in gpurun_out/trace_bench/ for
# scripts/trace_step.py (one step as a timeline) and scripts/open_timeline.py
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_bench; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HOBBIT_BENCH_NOPROF=1
rocprofv3 --kernel-trace -d $O/kt -o tb --output-format csv -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-dropin > $O/bench.json 2> $O/bench.err
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
cp $T $O/kernel_trace.csv; rm -rf $O/kt
python3 $R/scripts/trace_step.py $O/kernel_trace.csv 4 > $O/step.txt
python3 $R/scripts/open_timeline.py $O/kernel_trace.csv | tail -12
