#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "chase_tag or sliding or torque or traverse" > $OUT/pytest.log 2>&1
python tools/deviation_envs.py sepmc --no-oracle --engine-arenas 512 --engine-steps 1000 > $OUT/dev_sepmc_engine.md 2>$OUT/dev.err
python tools/rollout_sepmc_policy.py 512 1000 > $OUT/sepmc_policy_rollout.txt 2>>$OUT/dev.err
tail -3 $OUT/pytest.log; cat $OUT/sepmc_policy_rollout.txt; tail -2 $OUT/dev_sepmc_engine.md
