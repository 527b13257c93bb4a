#!/bin/bash
# Round 6, closing gpurun call: the whole GPU suite, everything profiles/r06_* is built from (tools/profile.sh), and the price of the XROWS options as they ship.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
tools/profile.sh r06
{
for spec in "" "leg_edges=1" "self_friction=0.25"; do
  echo "== spec '$spec'"
  LL_SWEEP_SPEC=$spec timeout 300 python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1" 2>&1 | grep kernel
  LL_SWEEP_SPEC=$spec timeout 600 python tools/sweep_epmc.py "4096:1:32,4096:1:1,4096:2:32,4096:3:32" 2>&1 | grep kernel
done
for spec in "" "leg_edges=1" "pair_friction=0.25" "max_pair=4" "self_friction=0.25,pair_friction=0.25,max_pair=4"; do
  echo "== SEPMC spec '$spec'"; LL_SWEEP_SPEC=$spec timeout 300 python tools/sweep_sepmc.py "2048:0:32,2048:0:1,2048:1:32" 2>&1 | grep kernel
done
} > $OUT/xrows_price.txt 2>&1
tail -30 $OUT/xrows_price.txt
