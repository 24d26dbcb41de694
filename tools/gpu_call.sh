#!/bin/bash
# tools/gpu_call.sh <tag> <section> [<section> ...] — one gpurun call's worth of GPU work (run from the repo root on the GPU
# box); every section has its own time limit and writes under gpurun_out/<tag>_*.  Sections:
#   tests      pytest -m gpu with durations
#   abfft      same-process A/B of the N = 16384 convolver transforms (three register passes vs the round-2 radix-4-in-LDS form)
#   bench      the default bench line (+ the detail file)
#   pmc:<w>    rocprofv3 stats / FETCH_SIZE / WRITE_SIZE passes of bench workload <w>
#   line:<w>   one bench line of workload <w>
#   fuzz:<n>   tools/fuzz_campaign.py over n seeds per generator
#   ranks:<n>  `python bench.py --gpus n` with no launcher (it spawns the ranks; WAA_BENCH_SHARE_GPU: they share this box's GPU)
#   dyn        tools/dyn_probe.py: dyn_kernel's quantum pipeline against the one-wavefront form
#   t:<files>  pytest -m gpu -x on the '+'-separated test files (ta:<files>: without -x, every failure listed)
#   convnoise  tools/conv_noise_probe.py: the cost of the convolver's noise floor in a dynamic plan, A/B
#   determinism  tools/determinism_campaign.py, one process on the device (the same bits twice?)
#   fuzzvariants the campaign's two variants on fresh seeds: FUZZ_MIXED_COUNTS=1, WAA_POISON_ALLOC=1
#   box        copy floor of this box (tools/stream_probe) + rocm-smi clocks: C2 ran 1.35 ... 1.60 ms depending on the box
#   plantrace:<w>  WAA_PLAN_TRACE of workload <w> (measurement build): where build_plan's host time goes
set -u
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for S in "$@"; do
  echo "=== $S"
  case $S in
    tests)  timeout 900 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${TAG}_gputests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/${TAG}_gputests.log ;;
    testsall) timeout 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_gputests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/${TAG}_gputests.log ;;
    abfft)  AB_REPS=2 AB_ITERS=5 timeout 300 python tools/ab_env.py WAA_CONV_FFT_R4 t1 c3 > gpurun_out/${TAG}_ab_fft.txt 2>&1; cat gpurun_out/${TAG}_ab_fft.txt | tail -12 ;;
    ab:*)   V=${S#ab:}; AB_REPS=2 AB_ITERS=5 timeout 300 python tools/ab_env.py ${V%%@*} ${V#*@} > gpurun_out/${TAG}_ab_${V%%[@=]*}.txt 2>&1; tail -12 gpurun_out/${TAG}_ab_${V%%[@=]*}.txt ;;
    bench)  timeout 1200 python bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "rc=$?"; wc -c gpurun_out/${TAG}_bench_default.json; cat gpurun_out/${TAG}_bench_default.json; tail -3 gpurun_out/${TAG}_bench_default.err ;;
    pmc:*)  W=${S#pmc:}; timeout 600 bash tools/pmc_pass.sh $W ${TAG}_$W > /dev/null 2>&1; head -12 gpurun_out/${TAG}_${W}_stats.txt ;;
    line:*) W=${S#line:}; timeout 300 python bench.py --workload $W --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err; cat gpurun_out/${TAG}_bench_$W.json | cut -c1-1500 ;;
    fuzz:*) N=${S#fuzz:}; timeout 1500 python tools/fuzz_campaign.py --first 100000 --count $N --jobs 8 --out gpurun_out/${TAG}_fuzz.json 2> gpurun_out/${TAG}_fuzz.err | cut -c1-3000; tail -3 gpurun_out/${TAG}_fuzz.err ;;
    determinism) # the same process, the same bits twice?  (tools/determinism_campaign.py, one process on the device, ~4 min)
            timeout 600 python tools/determinism_campaign.py --first 800000 --count 25000 --repeats 6 --rebuild-every 8 --jobs 1 --hot-repeats 300 --seconds 240 --out gpurun_out/${TAG}_determinism.json 2> gpurun_out/${TAG}_determinism.err | cut -c1-1500 ;;
    fuzzvariants) # the two campaign variants on fresh seeds: mixed channel counts, poisoned fresh device memory
            FUZZ_MIXED_COUNTS=1 timeout 600 python tools/fuzz_campaign.py --first 300000 --count 4000 --jobs 8 --out gpurun_out/${TAG}_fuzz_mixed.json 2> /dev/null | cut -c1-1500
            WAA_POISON_ALLOC=1 timeout 600 python tools/fuzz_campaign.py --first 400000 --count 4000 --jobs 8 --out gpurun_out/${TAG}_fuzz_poisoned.json 2> /dev/null | cut -c1-1500 ;;
    box)    # which kind of box is this?  the copy floor of the C2 access pattern (tools/stream_probe, built in-tree) + clocks
            { echo "# $(date -u) $(hostname)"; rocm-smi --showclocks --showpower --showmemuse 2>/dev/null | grep -v "^=\|^$" | head -30;
              timeout 120 tools/stream_probe 2>&1 | head -20; } > gpurun_out/${TAG}_box.txt 2>&1; grep -E "linear 256x65536|stream tile 2048 |sclk|mclk|fclk" gpurun_out/${TAG}_box.txt | head -8 ;;
    plantrace:*) W=${S#plantrace:}; WAA_USE_MEASURE_LIB=1 WAA_PLAN_TRACE=1 timeout 300 python bench.py --workload $W --steps 2 --warmup 1 --sustain 0 --no-cpu-baseline --no-extra 2> gpurun_out/${TAG}_plantrace_$W.txt | cut -c1-400; grep "\[plan\]" gpurun_out/${TAG}_plantrace_$W.txt | head -20 ;;
    ranks:*) N=${S#ranks:}; # the N-rank line rehearsed on this 1-GPU box: plain `bench.py --gpus N` starts the ranks itself (they share device 0 over gloo)
            WAA_BENCH_SHARE_GPU=1 WAA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $N --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_${N}rank_shared.json 2> gpurun_out/${TAG}_bench_${N}rank_shared.err; echo "rc=$?"; cut -c1-700 gpurun_out/${TAG}_bench_${N}rank_shared.json; tail -2 gpurun_out/${TAG}_bench_${N}rank_shared.err ;;
    dyn)    # the quantum-serial probe graph: pipelined stages (default) against the one-wavefront form, twice each
            for rep in 1 2; do for v in 0 1; do
              if [ $v = 1 ]; then export WAA_DYN_NO_PIPE=1; else unset WAA_DYN_NO_PIPE; fi
              echo "## WAA_DYN_NO_PIPE=${WAA_DYN_NO_PIPE:-unset}"; timeout 300 python tools/dyn_probe.py 2>&1 | grep -E "dynamic-count group|dyn_kernel|first render"
            done; done > gpurun_out/${TAG}_dyn_probe.txt 2>&1; unset WAA_DYN_NO_PIPE; cat gpurun_out/${TAG}_dyn_probe.txt ;;
    convnoise) # what the convolver's noise floor costs in a dynamic plan: with it, and with round 3's clearing only
            for v in 0 1 0 1; do if [ $v = 1 ]; then export WAA_NO_CONV_NOISE_FLOOR=1; else unset WAA_NO_CONV_NOISE_FLOOR; fi
              echo "## WAA_NO_CONV_NOISE_FLOOR=${WAA_NO_CONV_NOISE_FLOOR:-unset}"; timeout 300 python tools/conv_noise_probe.py 2>&1 | grep -vE "hostname|Warning"
            done > gpurun_out/${TAG}_conv_noise_probe.txt 2>&1; unset WAA_NO_CONV_NOISE_FLOOR; cat gpurun_out/${TAG}_conv_noise_probe.txt ;;
    t:*)    F=${S#t:}; timeout 900 python -m pytest ${F//+/ } -m gpu -q -x > gpurun_out/${TAG}_tests_sel.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/${TAG}_tests_sel.log ;;
    ta:*)   F=${S#ta:}; timeout 900 python -m pytest ${F//+/ } -m gpu -q > gpurun_out/${TAG}_tests_all.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_tests_all.log | cut -c1-300 | tail -30 ;;
    *) echo "unknown section $S" ;;
  esac
done
