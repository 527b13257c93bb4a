#!/bin/bash
# What a resident RCCL kernel costs the step kernel, measured with RCCL itself on ONE GPU: bench.py's N > 1 control flow with a
# one-rank communicator, every gather repeated R more times so that RCCL's kernel is resident for as long as an 8-rank gather would be, with
# the one-wave-per-SIMD builds (512 registers per wave: no other wave fits on the SIMD) and with the 256-register builds (LL_SHARE_SIMDS=1).
cd "$(dirname "$0")/.." || exit 1
OUT_ERR=${OUT_ERR:-/dev/null}
B="python bench.py --gpus 1 --steps 1024 --warmup 128 --no-cpu-baseline"
for R in 0 8 24; do
  for SH in 0 1; do
    for MODE in async none; do
      L=$(LL_BENCH_FORCE_GATHER=1 LL_BENCH_GATHER_REPEAT=$R LL_SHARE_SIMDS=$SH $B --gather-mode $MODE 2>$OUT_ERR | grep '^{' | tail -1)
      echo "$L" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); g=j['config']['gather']
print('extra gathers $R  share_simds $SH  mode %-5s  ms/step %.4f  kernel ms/step %.4f  value %.2f M  stream stall %.1f ms' % (g['mode'], j['ms_per_step'], j['roofline']['kernel_avg_ms'], j['value']/1e6, g['stream_stall_ms_total']))"
    done
  done
done
