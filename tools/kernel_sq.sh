#!/bin/bash
# tools/kernel_sq.sh <tag> <workload> <kernel-name-substring> — SQ counters (instruction mix, issue / wait cycles) of one kernel
# of a bench workload, two rocprofv3 --pmc passes (GPU box, repo root); summaries in gpurun_out/<tag>_<workload>_sq{1,2}.txt
set -u
TAG=$1; W=$2; K=$3
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  OUT=/tmp/prof_${TAG}_${W}_sq$i
  rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $OUT -o run -- python $ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT.log 2>&1)
  DB=$(find $OUT -name '*_results.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $set -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extra"; python tools/rocprof_summary.py $DB | grep "$K"; } > gpurun_out/${TAG}_${W}_sq$i.txt
  cat gpurun_out/${TAG}_${W}_sq$i.txt
done
