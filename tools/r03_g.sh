#!/bin/bash
# the driver's own invocation (20 timed steps after 5) against longer runs, with and without the HBM triad calibration up front
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
show() { grep '^{' | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1: value %.2f M  ms/step %.4f  kernel ms/step %.4f  launches %d' % (j['value']/1e6, j['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches_timed']))"; }
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style            "; done
for i in 1 2 3; do LL_BENCH_TRIAD_FIRST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style, triad first"; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --steps-per-launch 1 --no-cpu-baseline 2>/dev/null | show "driver-style, spl 1      "; done
python bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | show "2048 steps, spl 32       "
python bench.py --gpus 1 --steps 2048 --warmup 256 --steps-per-launch 128 --no-cpu-baseline 2>/dev/null | show "2048 steps, spl 128      "
python bench.py --gpus 1 --steps 200 --warmup 200 --no-cpu-baseline 2>/dev/null | show "200 steps after 200      "
python bench.py --gpus 1 --steps 20 --warmup 200 --no-cpu-baseline 2>/dev/null | show "20 steps after 200       "
