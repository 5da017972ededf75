#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_open; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o op --output-format csv -- python3 $R/scripts/opentrace.py > $O/run.log 2>&1
python3 $R/scripts/open_gaps.py $(find $O -name "*kernel_trace.csv" | head -1)
rm -f $(find $O -name "*kernel_trace.csv")
