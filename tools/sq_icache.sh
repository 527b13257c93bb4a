#!/bin/bash
# Instruction-cache counters of the PMC step kernel.  gpurun -- 'tools/sq_icache.sh <tag>'
TAG=${1:-ic}; cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
S="python bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES -d $OUT/p1 -- $S > /dev/null 2>&1
find $OUT -name "*.csv" -size +20M -delete
python tools/sq_raw.py $OUT
