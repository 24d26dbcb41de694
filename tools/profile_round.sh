#!/bin/bash
# tools/profile_round.sh <tag> — the round's measurement batch (GPU box, from the repo root): rocprofv3 stats + FETCH/WRITE
# passes of the roofline workloads, one bench line per workload, the default bench line, the GPU test log with durations.
set -u
TAG=$1
mkdir -p gpurun_out
for w in c2 t1 c3 c4 c5 c1a os2 hrtf echo; do
  bash tools/pmc_pass.sh $w ${TAG}_$w > /dev/null 2>&1
done
for w in c2k c4 echo fb fbq fm osc iir2 iir8 os2 os4 hrtf; do
  python bench.py --workload $w --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench_$w.json 2>/dev/null
done
python bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_gputests_durations.log 2>&1
tail -3 gpurun_out/${TAG}_gputests_durations.log
ls gpurun_out | grep $TAG | wc -l
