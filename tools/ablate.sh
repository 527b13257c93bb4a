#!/bin/bash
# Timing ablations of pmc_step_kernel (numbers quoted in DESIGN.md 5).  Builds a SEPARATE library with -DPMC_ABLATION --
# the shipped libllenv.so has no such switches -- and runs tools/sweep.py against it.
#   tools/ablate.sh build                      (here: hipcc cross-compiles)
#   gpurun -- 'tools/ablate.sh run "0 1 2 3 7 11" "4096:4:10:1,4096:4:10:10"'
set -e
cd "$(dirname "$0")/.."
OUT=tools/_build/libllenv_abl.so
if [ "$1" = build ]; then
  mkdir -p tools/_build
  FLAGS=$(python -c "import __graft_entry__ as g; print(' '.join(g.HIP_FLAGS))")
  /opt/rocm/bin/hipcc $FLAGS -DPMC_ABLATION -o $OUT lifelike_agility_and_play_amd/csrc/llenv.hip
else
  for f in $2; do echo "LL_DEBUG_FLAGS=$f"; LL_DEBUG_FLAGS=$f LL_LIB=$OUT python tools/sweep.py "$3"; done
fi
