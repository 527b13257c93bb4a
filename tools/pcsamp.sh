#!/bin/bash
# PC sampling of the step kernel (rocprofv3 beta): where the wave's cycles go, per instruction.  gpurun -- 'tools/pcsamp.sh <tag> [method] [lib]'
TAG=${1:-pcs}; METHOD=${2:-stochastic}; LIB=${3:-}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
UNIT=cycles; IV=65536
if [ "$METHOD" = host_trap ]; then UNIT=time; IV=1; fi
LL_LIB=$LIB timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $IV \
   --kernel-trace --output-format csv -d $OUT/pcs -- python tools/sweep.py "4096:4" > $OUT/log.txt 2>&1
echo rc $? >> $OUT/log.txt
tail -5 $OUT/log.txt
find $OUT -type f | head; du -sh $OUT
find $OUT -name "*.csv" -size +40M -delete
