#!/bin/bash
# tools/dyn_sq.sh <tag> — SQ counters of dyn_kernel on tools/dyn_probe.py's graph (two rocprofv3 --pmc passes; GPU box, repo root)
set -u
TAG=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  OUT=/tmp/prof_${TAG}_dsq$i
  rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $OUT -o run -- python $ROOT/tools/dyn_probe.py > $OUT.log 2>&1)
  DB=$(find $OUT -name '*_results.db' | head -1)
  { echo "# rocprofv3 --pmc $set -- python tools/dyn_probe.py"; python tools/rocprof_summary.py $DB | grep "dyn_kernel"; } > gpurun_out/${TAG}_dyn_sq$i.txt
  cat gpurun_out/${TAG}_dyn_sq$i.txt
done
