#!/bin/bash
# tools/pmc_pass.sh <workload> <tag> [bench args...] — the three rocprofv3 passes the roofline numbers come from, for one
# bench workload (run on the GPU box from the repo root, e.g. through gpurun):
#   1. --kernel-trace --stats            per-kernel durations
#   2. --kernel-trace --pmc FETCH_SIZE   HBM read bytes  (own pass: TCC counter slots, MI355X_MICROARCH.md)
#   3. --kernel-trace --pmc WRITE_SIZE   HBM write bytes
# Summaries (tools/rocprof_summary.py) land in gpurun_out/<tag>_{stats,fetch,write}.txt; copy what is judged into profiles/.
set -u
W=$1; TAG=$2; shift 2
export TMPDIR=/tmp
ROOT=$(pwd)
# the counter passes count bytes per launch: plain allocations, every kernel exactly five times (placement does not change traffic)
CMD_PMC="python $ROOT/bench.py --workload $W --steps 3 --warmup 1 --sustain 0 --no-cpu-baseline --no-extra --arena-gb 0 --no-preroll $*"
# the stats pass is the DRIVER'S protocol (--steps 20 --warmup 5) out of the graded arena, without the `cold` record: every launch of
# the workload's kernels is of the kind the bench line's kernel_ms averages (the pre-roll's launches included: same kernel, same memory)
CMD_STATS="python $ROOT/bench.py --workload $W --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-extra --no-live-pmc --no-cold $*"
mkdir -p gpurun_out
for pass in stats fetch write; do
  OUT=/tmp/prof_${TAG}_${pass}
  rm -rf $OUT
  case $pass in
    stats) FLAGS="--kernel-trace --stats" ;;
    fetch) FLAGS="--kernel-trace --pmc FETCH_SIZE" ;;
    write) FLAGS="--kernel-trace --pmc WRITE_SIZE" ;;
  esac
  if [ $pass = stats ]; then CMD=$CMD_STATS; else CMD=$CMD_PMC; fi
  (cd /tmp && rocprofv3 $FLAGS -d $OUT -o run -- $CMD > $OUT.log 2>&1)
  # (the stats pass: the bench line the PROFILED process printed — its roofline.kernel_ms is what the summary's average is held against)
  if [ $pass = stats ]; then grep -h '^{"metric"' $OUT.log | tail -1 > $ROOT/gpurun_out/${TAG}_stats_bench.json; fi
  DB=$(find $OUT -name '*_results.db' | head -1)
  if [ -n "$DB" ]; then
    { echo "# $CMD"; echo "# rocprofv3 $FLAGS"; python tools/rocprof_summary.py $DB; } > gpurun_out/${TAG}_${pass}.txt
  else
    { echo "no results db for $pass"; tail -20 $OUT.log; } > gpurun_out/${TAG}_${pass}.txt
  fi
done
