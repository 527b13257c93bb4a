#!/bin/bash
# One gpurun call of a build -> measure iteration:   gpurun --timeout 900 -- 'tools/gpu_round.sh <tag> [quick]'
#   1. pytest -m gpu                                   -> gpurun_out/<tag>/gpu_tests.log
#   2. bench.py as the driver runs it (CPU baseline on in the full mode)  -> bench.log
#   3. batch-size sweep                                -> sweep.txt
#   4. rocprofv3 --kernel-trace --stats of bench.py, then (separate passes) SQ issue counters, FETCH_SIZE, WRITE_SIZE
#   5. timeline + ablations of the ablation build, if tools/_build/libllenv_abl.so travelled
TAG=${1:-r02}
MODE=${2:-full}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?" >> $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
if [ "$MODE" = full ]; then
  timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 200 > $OUT/bench.log 2>$OUT/bench.err
else
  timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench.log 2>$OUT/bench.err
fi
tail -c 1500 $OUT/bench.log
timeout 300 python tools/sweep.py "1024:4,4096:4,8192:4,16384:4,65536:4,4096:4:1:10" > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
B="python bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_under_rocprof.log 2>&1
S="python bench.py --gpus 1 --steps 30 --warmup 3 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -- $S > /dev/null 2>&1
if [ "$MODE" = full ]; then
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $S > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- $S > /dev/null 2>&1
fi
if [ -f tools/_build/libllenv_abl.so ]; then
  LL_DEBUG_FLAGS=16 LL_LIB=tools/_build/libllenv_abl.so timeout 200 python tools/timeline.py 4096 > $OUT/timeline.txt 2>&1
  if [ "$MODE" = full ]; then
    timeout 300 tools/ablate.sh run "0 1 2 3 256 512" "4096:4:10:10,4096:4:1:10" > $OUT/ablation.txt 2>&1
  fi
fi
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*_kernel_trace.csv" | head -3
