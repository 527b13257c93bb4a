#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
show() { grep '^{' | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1: value %.2f M  ms/step %.4f  kernel ms/step %.4f  triad %.0f GB/s' % (j['value']/1e6, j['ms_per_step'], r['kernel_avg_ms'], r['peak_measured_triad'] or 0))"; }
for it in 10 100 400; do for i in 1 2; do LL_BENCH_TRIAD_ITERS=$it python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "triad iters $it"; done; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | show "with cpu baseline (the exact line of the driver)"; done
