#!/bin/bash
# kernel trace of a few steps with the tree on the tail stream (experiment): timeline analysis by scripts/trace_overlap.py
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_tail; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HOBBIT_COMMIT_TAIL=side HOBBIT_LEAF_GRID=${LEAF_GRID:-1024} HOBBIT_BENCH_NOPROF=1
rocprofv3 --kernel-trace -d $O/kt -o tt --output-format csv -- python3 $R/bench.py --steps 3 --no-cpu-baseline --no-dropin > $O/bench.json 2> $O/bench.err
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/trace_overlap.py $T > $O/timeline.txt
cp $T $O/kernel_trace.csv; rm -rf $O/kt
tail -5 $O/timeline.txt
