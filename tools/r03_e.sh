#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
OUT_ERR=$OUT/simd_sharing.err tools/simd_sharing.sh > $OUT/simd_sharing.txt 2>&1
python tools/deviation_table.py --engine --only "sliding direction" --engine-episodes 4096 > $OUT/dev_pmc_engine_dirs.md 2>$OUT/dev.err
cat $OUT/simd_sharing.txt; tail -5 $OUT/simd_sharing.err; tail -3 $OUT/dev_pmc_engine_dirs.md
