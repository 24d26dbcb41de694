set -u
python -m pytest tests/test_oversample.py -m gpu -x -q 2>&1 | tail -2
for v in w8 w4 d e f; do
  unset WAA_QGEMM_DEBUG WAA_QGEMM_W4
  case $v in w4) export WAA_QGEMM_W4=1;; d|e|f) export WAA_QGEMM_DEBUG=$v;; esac
python bench.py --workload os2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('os2 $v ms', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
