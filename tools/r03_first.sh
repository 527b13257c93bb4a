#!/bin/bash
# round 3, first GPU call: the full -m gpu suite at the new HEAD, the multi-step launch sweep, the default bench line
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sweep.py "4096:4:10:10:1,4096:4:10:10:8,4096:4:10:10:32,4096:4:10:10:128,4096:4:10:10:1,4096:4:10:10:32,65536:4:10:10:1,65536:4:10:10:32,16384:4:10:10:1,16384:4:10:10:32" > $OUT/sweep_spl.txt 2>&1
python bench.py > $OUT/bench.log 2>$OUT/bench.err
python bench.py --steps-per-launch 1 --no-cpu-baseline > $OUT/bench_spl1.log 2>>$OUT/bench.err
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log; cat $OUT/sweep_spl.txt; tail -c 1500 $OUT/bench.log
