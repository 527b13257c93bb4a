#!/bin/bash
# A/B of library variants over the three envs on ONE box:  tools/ab3.sh "P0 P1 P2" rounds
cd "$(dirname "$0")/.."
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    echo "== $v (round $r)"
    LL_LIB=tools/_build/ab_$v.so python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,65536:4:10:10:32"
    LL_LIB=tools/_build/ab_$v.so python tools/sweep_epmc.py "4096:1:32,4096:1:1,65536:1:1"
    LL_LIB=tools/_build/ab_$v.so python tools/sweep_sepmc.py "2048:0:32,2048:0:1,32768:0:1"
  done
done
