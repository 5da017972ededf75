#!/bin/bash
# SQ counter passes (one commit each, no other trace domain) for the commit kernels: where do the waves' cycles go?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HOBBIT_COMMIT_PIPE=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o sq --output-format csv -- python3 $R/bench.py --phase commit --steps 1 --warmup 1 --no-cpu-baseline --no-dropin > $O/$tag.json 2> $O/$tag.err
done
python3 $R/scripts/sq_counters.py $(find $O -name "*counter_collection.csv") > $O/summary.txt
cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
