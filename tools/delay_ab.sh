set -u
python -m pytest tests/test_delay.py tests/test_cycles.py tests/test_fuzz_graphs.py tests/test_param_modulation.py tests/test_edge_fold.py -m gpu -x -q 2>&1 | tail -4
for v in fold nofold; do
  if [ $v = nofold ]; then export WAA_NO_DELAY_FOLD=1; fi
python bench.py --workload echo --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('echo $v ms', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
