#!/bin/bash
# The bars ("hole") policy's three budgeted experiments (round-5 review, "Next round" #7), one gpurun call: the reference's third EPMC checkpoint
# (test_scripts/environmental_level/test_environmental_level_env.py:89-90) at 1024 episodes per leg, protocol of tools/rollout_epmc_policy.py.
#   (i)   knee-limit rows in the crouch: the spec (a joint 0.04 rad past its limit is stopped, not pushed back) / limit_erp_deep=0.2 / the speculative rows of rounds 1 - 4
#   (ii)  the landing of a bound: contact ERP 0.08 (spec) against 0.2 (per-leg ERPs do not exist: all contacts)
#   (iii) the passive wheel cylinders at the knees: with (spec) / shrunk to a millimetre (LL_MODEL_NO_WHEELS=1)
# plus what round 6 added to the spec table: leg edges off (rounds 1 - 5), self friction 0.25.  Decided in advance: a leg that moves "reached" by more than 3 s.e. (42 of 1024) is a lead; otherwise the paragraph is written.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${1:-r06_bars}; mkdir -p $OUT
run() { name=$1; shift; (env "$@" OMP_NUM_THREADS=2 python tools/rollout_epmc_policy.py hole 1024 600 > $OUT/$name.txt 2>&1; echo "== $name ($*)"; tail -3 $OUT/$name.txt) & }
run spec LL_X=0
run limit_erp_deep_0.2 LL_SPEC=limit_erp_deep=0.2
run limit_speculative LL_SPEC=limit_speculative=1
run contact_erp_0.2 LL_SPEC=erp=0.2
run no_wheels LL_MODEL_NO_WHEELS=1
run no_leg_edges LL_SPEC=leg_edges=0
run self_friction_0.25 LL_SPEC=self_friction=0.25
wait
