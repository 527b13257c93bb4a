#!/bin/bash
# Round 6, one gpurun call: the whole GPU suite; the kernel-time price of the spec switches round 6 added (leg edges, the XROWS twins after their idle turns were skipped);
# leg edges priced on the five policies; the bars policy's budgeted experiments.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${1:-r06i}; mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.log
{
for spec in "" "leg_edges=0" "self_friction=0.25"; do
  echo "== PMC spec '$spec'"; LL_SWEEP_SPEC=$spec timeout 300 python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1" 2>&1 | grep kernel
  LL_SWEEP_SPEC=$spec timeout 600 python tools/sweep_epmc.py "4096:1:32,4096:1:1,4096:2:32,4096:3:32" 2>&1 | grep kernel
done
for spec in "" "leg_edges=0" "pair_friction=0.25" "max_pair=4" "self_friction=0.25,pair_friction=0.25,max_pair=4"; do
  echo "== SEPMC spec '$spec'"; LL_SWEEP_SPEC=$spec timeout 300 python tools/sweep_sepmc.py "2048:0:32,2048:0:1,2048:1:32" 2>&1 | grep kernel
done
} > $OUT/spec_price.txt 2>&1
tail -40 $OUT/spec_price.txt
tools/r06_bars_experiments.sh ${1:-r06i}/bars
timeout 900 python tools/spec_table.py --engine --variants 'spec:;no leg edges (rounds 1 - 5):leg_edges=0'  > $OUT/leg_edges_table.md 2> $OUT/leg_edges_table.err; tail -4 $OUT/leg_edges_table.md
