#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: kernel trace + stats of the default bench command, then FETCH_SIZE and WRITE_SIZE in
# passes of their own (counters never together with other trace domains), commit phase only (the kernels the roofline is about).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o r03 --output-format csv -- python3 $R/bench.py --steps 5 --no-cpu-baseline --no-dropin > $O/bench_stats.json 2> $O/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o r03f --output-format csv -- python3 $R/bench.py --phase commit --steps 2 --no-cpu-baseline --no-dropin > $O/bench_fetch.json 2> $O/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o r03w --output-format csv -- python3 $R/bench.py --phase commit --steps 2 --no-cpu-baseline --no-dropin > $O/bench_write.json 2> $O/bench_write.err
find $O -name "*.csv" | head -20
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 $R/scripts/hbm_traffic.py $F $W $O/r03_hbm_traffic_commit_2e28.json python3 bench.py --phase commit --steps 2 --no-cpu-baseline
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $O/r03_kernel_stats_commit_open_2e28.csv
# the raw counter files are large: keep only the summaries
rm -rf $O/fetch $O/write
find $O/stats -name "*kernel_trace.csv" -delete
