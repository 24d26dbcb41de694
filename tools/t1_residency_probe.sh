#!/bin/bash
# tools/t1_residency_probe.sh — would cache-resident spectra make T1's three convolver kernels faster?  (GPU box)
# Round-5 review, item 3: T1 moves 5 x its compulsory bytes because the 2x-expanded spectra X and Y (15.5 MB per pair of contexts)
# round-trip HBM between the forward transform, the product and the inverse transform.  Sub-batches whose X / Y scratch stays in the
# 256 MB Infinity Cache would avoid that — IF the kernels are bound by that traffic.  Cheapest test: render T1 with batches so small
# that X + Y fit the cache as they are (8 contexts: 124 MB, 16: 248 MB) and compare the kernels' time PER CONTEXT with the
# full batch's (1024 contexts: 15.9 GB of spectra).  Prints one line per batch size.
mkdir -p gpurun_out
for N in 8 16 32 64 256 1024; do
  timeout 300 python bench.py --workload t1 --instances $N --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-extra --no-live-pmc --arena-gb 32 2>/dev/null |
    python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; n=d['config']['contexts_per_gpu']
print('t1 %5d contexts: step %.3f ms = %.2f us/context |'%(n, d['ms_per_step'], d['ms_per_step']*1e3/n), ' | '.join('%s %.3f ms = %.2f us/ctx'%(k.replace('conv_','').replace('_kernel',''), v, v*1e3/n) for k,v in sorted(r.get('kernel_ms_all', {}).items())))
"
done
