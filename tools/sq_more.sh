#!/bin/bash
# More SQ counters for the PMC step kernel (two passes).  gpurun -- 'tools/sq_more.sh <tag>'
TAG=${1:-sq}; cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
S="python bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_BRANCH -d $OUT/p1 -- $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $OUT/p2 -- $S > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/p3 -- $S > /dev/null 2>&1
find $OUT -name "*.csv" -size +20M -delete
python - <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob('gpurun_out/%s/p*' % sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sq/p*')):
    pass
PY
python tools/sq_raw.py $OUT
