#!/bin/bash
# A/B timing of step-kernel variants ON ONE BOX (box-to-box clocks differ by a few percent, so variants are only comparable inside one gpurun call).
#   tools/ab.sh build NAME "-DFLAG=1 ..."     (here; repeat per variant)  -> tools/_build/ab_NAME.so
#   gpurun -- 'tools/ab.sh run "A B" "4096:4,65536:4" 3'                   alternates the variants, 3 rounds
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/_build
  FLAGS=$(python -c "import __graft_entry__ as g; print(' '.join(g.HIP_FLAGS))")
  /opt/rocm/bin/hipcc $FLAGS $3 -o tools/_build/ab_$2.so lifelike_agility_and_play_amd/csrc/llenv.hip
else
  for r in $(seq 1 ${4:-3}); do
    for v in $2; do echo "== $v (round $r)"; LL_LIB=tools/_build/ab_$v.so python tools/sweep.py "$3"; done
  done
fi
