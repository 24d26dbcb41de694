#!/bin/bash
# tools/kernel_sq.sh <tag> <kernel-name-substring> <bench workload> — SQ counters of one kernel of a bench workload
# (two rocprofv3 --pmc passes; GPU box, repo root).  The generalisation of tools/dyn_sq.sh.
set -u
TAG=$1
KERN=$2
WL=$3
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  OUT=/tmp/prof_${TAG}_ksq$i
  rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $OUT -o run -- python $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-live-pmc --sustain 0 > $OUT.log 2>&1)
  DB=$(find $OUT -name '*_results.db' | head -1)
  { echo "# rocprofv3 --pmc $set -- python bench.py --workload $WL --steps 2 --warmup 1"; python tools/rocprof_summary.py $DB | grep "$KERN"; } > gpurun_out/${TAG}_${WL}_sq$i.txt
  cat gpurun_out/${TAG}_${WL}_sq$i.txt
done
