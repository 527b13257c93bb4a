#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sweep_epmc.py "4096:1:1,4096:1:32,4096:0:1,4096:0:32,4096:3:1,4096:3:32,65536:1:1,65536:1:16" > $OUT/epmc_sweep.txt 2>&1
python tools/sweep_sepmc.py "2048:0:1,2048:0:32,2048:1:1,2048:1:32,32768:0:1,32768:0:16" > $OUT/sepmc_sweep.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -k "multi_step or overlap" > $OUT/pytest.log 2>&1
python bench.py --workload epmc --no-cpu-baseline > $OUT/epmc_bench.log 2>&1
python bench.py --workload sepmc --no-cpu-baseline > $OUT/sepmc_bench.log 2>&1
cat $OUT/epmc_sweep.txt $OUT/sepmc_sweep.txt; tail -3 $OUT/pytest.log
