# measurement aid: per-pass times of the lane-per-stream kernels under WAA_LANES_DEBUG (run from the repo root)
export TMPDIR=/tmp
ROOT=$(pwd)
for dbg in ${LANES_DEBUG_LIST:-0 4 0 4}; do
  echo "== debug=$dbg"
  OUT=/tmp/lp_$dbg; rm -rf $OUT
  (cd /tmp && WAA_LANES_DEBUG=$dbg timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $ROOT/bench.py --workload c1a --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT.log 2>&1)
  DB=$(find $OUT -name '*_results.db' | head -1)
  python tools/rocprof_summary.py $DB | grep -i "lanes" | head -3
done
