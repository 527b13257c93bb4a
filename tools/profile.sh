#!/bin/bash
# Everything profiles/ is built from, in one gpurun call:
#   gpurun --timeout 900 -- 'tools/profile.sh r01'      then here:   python tools/profile_summarize.py r01
# 1. bench.py as the driver runs it (with the CPU baseline)       -> gpurun_out/<tag>/bench.log
# 2. rocprofv3 --kernel-trace --stats of the same command          -> .../stats
# 3. PMC passes, each on its own (no trace domains mixed in): SQ issue counters, FETCH_SIZE, WRITE_SIZE
# 4. batch-size sweep and the per-wave timeline of the ablation build (if tools/_build/libllenv_abl.so travelled)
TAG=${1:-r01}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.log 2>$OUT/bench.err          # exactly what the driver runs at N = 1
B="python bench.py --gpus 1 --steps 320 --warmup 32 --no-cpu-baseline --no-single-step-leg"     # (multiples of the 32 control steps a launch runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_under_rocprof.log 2>&1
S="python bench.py --gpus 1 --steps 96 --warmup 32 --no-cpu-baseline --no-single-step-leg"     # (three launches of 32 control steps: bench.py never times fewer than three)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- $S > /dev/null 2>&1
# LL_PROFILE_LEAN=1: what a call with few GPU-minutes left affords -- shorter sweeps, no CPU-baseline legs for EPMC / SEPMC, one policy rollout
if [ -n "$LL_PROFILE_LEAN" ]; then
  python tools/sweep.py "1024:4:10:10:32,4096:4:10:10:32,16384:4:10:10:32,65536:4:10:10:32,4096:4:10:10:1,65536:4:10:10:1" > $OUT/sweep.txt 2>&1
  (echo "== friction_mode=0 (the pyramid builds)"; LL_SWEEP_SPEC=friction_mode=0 python tools/sweep.py "4096:4:10:10:32,65536:4:10:10:1") >> $OUT/sweep.txt 2>&1
else
python tools/sweep.py "1024:4:10:10:32,2048:4:10:10:32,4096:4:10:10:32,8192:4:10:10:32,16384:4:10:10:32,32768:4:10:10:32,65536:4:10:10:32,4096:4:10:10:1,4096:4:10:10:8,4096:4:10:10:128,65536:4:10:10:1" > $OUT/sweep.txt 2>&1
fi
if [ -f tools/_build/libllenv_abl.so ]; then
  LL_DEBUG_FLAGS=16 LL_LIB=tools/_build/libllenv_abl.so python tools/timeline.py 4096 > $OUT/timeline.txt 2>&1
  tools/ablate.sh run "0 1 2 3 256 512" "4096:4:10:10,4096:4:1:10" > $OUT/ablation.txt 2>&1
fi
# 5. the neighbours of the path: EPMC / SEPMC (bench line, kernel stats, the same PMC passes, sweeps) and the closed actor loop with the trained policy
for W in epmc sepmc; do
  python bench.py --workload $W ${LL_PROFILE_LEAN:+--no-cpu-baseline} > $OUT/${W}_bench.log 2>$OUT/${W}_bench.err      # the default run, CPU baseline leg included
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- python bench.py --workload $W --gpus 1 --steps 192 --warmup 32 --no-cpu-baseline > /dev/null 2>&1
  SW="python bench.py --workload $W --gpus 1 --steps 64 --warmup 32 --no-cpu-baseline"
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/${W}_pmc_sq -- $SW > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${W}_pmc_fetch -- $SW > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${W}_pmc_write -- $SW > /dev/null 2>&1
done
if [ -n "$LL_PROFILE_LEAN" ]; then
  python tools/sweep_epmc.py "4096:0:32,4096:1:32,4096:2:32,4096:3:32,4096:1:1,65536:1:1" > $OUT/epmc_sweep.txt 2>&1
  python tools/sweep_sepmc.py "2048:0:32,2048:1:32,2048:0:1,32768:0:1" > $OUT/sepmc_sweep.txt 2>&1
  python tools/rollout_policy.py 4096 300 hip > $OUT/policy_rollout.txt 2>&1
else
python tools/sweep_epmc.py "4096:0:32,4096:1:32,4096:2:32,4096:3:32,4096:1:1,16384:1:1,65536:1:1" > $OUT/epmc_sweep.txt 2>&1
python tools/sweep_sepmc.py "2048:0:32,2048:1:32,2048:0:1,8192:0:1,32768:0:1,32768:1:1" > $OUT/sepmc_sweep.txt 2>&1
(python tools/rollout_policy.py 4096 300 hip; python tools/rollout_policy.py 65536 200 hip; python tools/rollout_policy.py 4096 300 torch) > $OUT/policy_rollout.txt 2>&1
fi
find $OUT -name "*.csv" -size +20M -delete
tail -c 600 $OUT/bench.log
