#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | tail -80 > $OUT/pytest_gpu.log
(for g in ifou iufo ifuo fiou; do python tools/rollout_epmc_policy.py hurdle 1024 500 - $g 1.0; done; python tools/rollout_epmc_policy.py hurdle 1024 500 - ifou 0.0; python tools/rollout_epmc_policy.py cube 1024 700 - ifou 1.0; python tools/rollout_epmc_policy.py cube 1024 700 - iufo 1.0) > $OUT/epmc_policy_rollout.txt 2>&1
python tools/deviation_envs.py epmc --no-oracle --engine-envs 4096 > $OUT/dev_epmc_engine.md 2>$OUT/dev.err
python tools/deviation_envs.py sepmc --no-oracle --engine-arenas 2048 --engine-steps 1000 > $OUT/dev_sepmc_engine.md 2>>$OUT/dev.err
python tools/deviation_table.py --engine > $OUT/dev_pmc_engine.md 2>>$OUT/dev.err
tail -4 $OUT/pytest_gpu.log; cat $OUT/epmc_policy_rollout.txt; tail -3 $OUT/dev_epmc_engine.md; tail -2 $OUT/dev_sepmc_engine.md
