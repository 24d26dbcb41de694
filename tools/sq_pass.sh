#!/bin/bash
# tools/sq_pass.sh <workload> <tag> [bench args...] — SQ (wave scheduler) counters of one bench workload, two rocprofv3
# --pmc passes (run on the GPU box from the repo root); summaries in gpurun_out/<tag>_sq{1,2}.txt
set -u
W=$1; TAG=$2; shift 2
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extra $*"
mkdir -p gpurun_out
i=0
SET1=${SQ_SET1:-"SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY"}
SET2=${SQ_SET2:-"SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU"}
for set in "$SET1" "$SET2"; do
  i=$((i+1))
  OUT=/tmp/prof_${TAG}_sq$i
  rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $OUT -o run -- $CMD > $OUT.log 2>&1)
  DB=$(find $OUT -name '*_results.db' | head -1)
  if [ -n "$DB" ]; then
    { echo "# $CMD"; echo "# rocprofv3 --pmc $set"; python tools/rocprof_summary.py $DB | grep -v "rocclr\|at::native"; } > gpurun_out/${TAG}_sq$i.txt
  else
    { echo "no results db"; tail -20 $OUT.log; } > gpurun_out/${TAG}_sq$i.txt
  fi
done
