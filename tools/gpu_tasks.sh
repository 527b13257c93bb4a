#!/bin/bash
# The GPU-side jobs of a round as named targets (one gpurun call each):   gpurun --timeout 1500 -- 'tools/gpu_tasks.sh <target> [tag]'
# Outputs go to gpurun_out/<tag>/; what is to be judged is copied into profiles/ afterwards (tools/profile_summarize.py for the profile target).
#   inertia   the -m gpu suite, then the five trained policies under both model blobs on the engine (tools/inertia_table.py)
#   tests     the -m gpu suite alone
#   valu      tools/valu_issue_bench.hip: wave64 VALU issue rate per SIMD at 1 .. 8 resident waves
#   profile   tools/profile.sh (bench lines, rocprofv3 stats + PMC passes, sweeps, timeline)
#   cone      the cone-coupled friction builds: GPU parity, kernel time, the trained tracking policy under pyramid and cone
#   final     the -m gpu suite at HEAD, then the three bench lines and the large-batch sweep points
#   driver    the driver's own invocation against longer runs, with and without the HBM triad first
TARGET=${1:-tests}
TAG=${2:-r05_$TARGET}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
show() { grep '^{' | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1: value %.2f M  ms/step %.4f  kernel ms/step %.4f  launches %d' % (j['value']/1e6, j['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches_timed']))"; }
gpu_tests() { timeout 1500 python -m pytest tests -m gpu -q -rA "$@" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^ERROR|pytest rc" $OUT/pytest_gpu.log | tail -8; }
case $TARGET in
  tests) [ $# -ge 1 ] && shift; [ $# -ge 1 ] && shift; gpu_tests "$@" ;;     # tools/gpu_tasks.sh tests TAG -k expr
  valu) tools/_build/valu_issue_bench 2000 | tee $OUT/valu_issue.txt ;;
  inertia)
    tools/_build/valu_issue_bench 2000 > $OUT/valu_issue.txt 2>&1; cat $OUT/valu_issue.txt
    gpu_tests
    python tools/inertia_table.py --engine > $OUT/inertia_engine.md 2> $OUT/inertia_engine.err
    cat $OUT/inertia_engine.md; tail -3 $OUT/inertia_engine.err
    python bench.py --no-cpu-baseline > $OUT/bench.log 2>$OUT/bench.err; tail -c 800 $OUT/bench.log ;;
  profile) tools/profile.sh $TAG ;;
  round2)        # second call of round 4: the suite at the new HEAD, what the register-budget builds disagree on, the p2p hand-off A/B
    gpu_tests -x
    python tools/diag_budgets.py > $OUT/diag_budgets.txt 2>&1; cat $OUT/diag_budgets.txt
    cp gpurun_out/two_rank/*.txt gpurun_out/two_rank/*.json $OUT/ 2>/dev/null; cat $OUT/p2p_no_cu.txt | head -8 ;;
  driver)
    for i in 1 2 3; do LL_BENCH_TRIAD_FIRST=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style, no triad first"; done
    for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style (triad first)  "; done
    python bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | show "2048 steps                  " ;;
  round3)        # the whole suite (no -x), then what the ray families of the EPMC / SEPMC observation cost (ablation build, if it travelled)
    gpu_tests
    cp gpurun_out/two_rank/*.txt gpurun_out/two_rank/*.json $OUT/ 2>/dev/null; head -8 $OUT/p2p_no_cu.txt
    if [ -f tools/_build/libllenv_abl.so ]; then
      for f in 0 32 64 128; do echo "LL_DEBUG_FLAGS=$f (32: no rays, 64: height grid only, 128: height grid + fan)"; LL_DEBUG_FLAGS=$f LL_LIB=tools/_build/libllenv_abl.so python tools/sweep_epmc.py "4096:1:32,4096:1:1,4096:0:32"; done > $OUT/epmc_ray_ablation.txt 2>&1
      cat $OUT/epmc_ray_ablation.txt
    fi ;;
  rays)          # A/B of the ray-phase variants on ONE box (tools/_build/ab_<v>.so built here with tools/ab.sh build), then the EPMC / SEPMC GPU tests
    for r in 1 2; do for v in ${AB_VARIANTS:-before rc1 rc3 rc7}; do
      echo "== $v (round $r)"
      LL_LIB=tools/_build/ab_$v.so python tools/sweep_epmc.py "4096:1:32,4096:3:32,4096:2:32,65536:1:1"
      LL_LIB=tools/_build/ab_$v.so python tools/sweep_sepmc.py "2048:0:32,2048:1:32,32768:0:1"
    done; done > $OUT/ray_ab.txt 2>&1
    cat $OUT/ray_ab.txt
    gpu_tests -k "epmc or sepmc" ;;
  cone)          # the cone-coupled friction builds (LLM_SPEC_FRICTION_MODE = 2): GPU parity, kernel time next to the pyramid's, and the trained tracking policy under both
    gpu_tests -k "test_gpu_parity"
    for r in 1 2; do for sp in "" "friction_mode=2"; do echo "== spec '$sp' (round $r)"; LL_SWEEP_SPEC=$sp python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,65536:4:10:10:1,65536:4:10:10:8"; done; done > $OUT/cone_sweep.txt 2>&1
    cat $OUT/cone_sweep.txt
    for v in "spec (as shipped)" "friction cone-coupled"; do python tools/deviation_table.py --engine --only "$v" 2>&1 | tail -1; done > $OUT/cone_policy.txt; cat $OUT/cone_policy.txt ;;
  r05a)          # round 5, first call: the engine twins of the joint-limit rule and the penetration-recovery switches -- GPU parity, kernel time, and the five policies per variant (one process per variant)
    gpu_tests -k "bullet_limit or two_erp"
    for r in 1 2; do for sp in "" "limit_speculative=0" "limit_speculative=0,erp=0.08,limit_erp=0.2,max_depen_speed=1e30"; do echo "== spec '$sp' (round $r)"; LL_SWEEP_SPEC=$sp python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,65536:4:10:10:1"; LL_SWEEP_SPEC=$sp python tools/sweep_epmc.py "4096:1:32,65536:1:1"; LL_SWEEP_SPEC=$sp python tools/sweep_sepmc.py "2048:0:32,32768:0:1"; done; done > $OUT/limit_sweep.txt 2>&1
    cat $OUT/limit_sweep.txt
    i=0
    while IFS= read -r v; do
      OMP_NUM_THREADS=2 python tools/spec_table.py --engine --variants "$v" > $OUT/engine_$i.md 2> $OUT/engine_$i.err &
      i=$((i+1))
    done < tools/r05_variants.txt
    wait
    cat $OUT/engine_*.md | grep -v "^| simulator\|^|---" ; tail -n 2 $OUT/engine_*.err | tail -20 ;;
  r05b)          # round 5, second call: the per-step table versions of the multi-step launch (bit-identity, cost), the 8-rank and p2p tests, driver-style lines, limit_erp_deep priced
    timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rA -k "multi_step" > $OUT/pytest_multi.log 2>&1; echo "multi-step rc $?"; tail -3 $OUT/pytest_multi.log
    for r in 1 2; do python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:16,4096:4:10:10:8,4096:4:10:10:128,4096:4:10:10:1,8192:4:10:10:32,65536:4:10:10:8"; done > $OUT/table_versions_sweep.txt 2>&1
    cat $OUT/table_versions_sweep.txt
    for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style"; done
    python bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline > $OUT/bench_2048.json 2>/dev/null; show "2048 steps" < $OUT/bench_2048.json
    gpu_tests -k "bench_ or p2p or gather or rccl or env_api"
    cp gpurun_out/two_rank/* $OUT/ 2>/dev/null
    i=0
    for v in "bullet limits + contact ERP 0.08, no cap:limit_speculative=0,erp=0.08,limit_erp=0.2,max_depen_speed=1e30" "the same + deep limit rows without push-back:limit_speculative=0,erp=0.08,limit_erp=0.2,max_depen_speed=1e30,limit_erp_deep=0"; do
      OMP_NUM_THREADS=4 python tools/spec_table.py --engine --variants "$v" > $OUT/engine_$i.md 2> $OUT/engine_$i.err &
      i=$((i+1))
    done
    wait
    cat $OUT/engine_*.md | grep -v "^| simulator\|^|---" ;;
  r05c)          # round 5, third call: the whole -m gpu suite at the round-5 spec (no -x: every cap that needs re-observing shows), the chunk-7 reproduction, bench lines
    gpu_tests
    timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/chunk7.txt 2>&1; cat $OUT/chunk7.txt | cut -c1-700
    for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "driver-style"; done
    python bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline > $OUT/bench_2048.json 2>/dev/null; show "2048 steps" < $OUT/bench_2048.json
    cp gpurun_out/two_rank/* $OUT/ 2>/dev/null ;;
  r05d)          # round 5, fourth call: the tests that changed since r05c, then the chunk-7 reproduction with its per-field report
    gpu_tests -k "round4_spec or trunk_on_edges or mocap_discontinuity or every_observation or p2p_pull_occupies or terrain_physics or larger_batch or pyramid"
    timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/chunk7.txt 2>&1; cat $OUT/chunk7.txt | cut -c1-1800
    cp gpurun_out/two_rank/p2p_no_cu.txt $OUT/ 2>/dev/null ;;
  chunk7)        # the seven-ray SEPMC build against the host build of the same source, with where the differences sit
    timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/chunk7.txt 2>&1; grep -o "^[a-z (]*chunk[^{]*\|'detail': {.*" $OUT/chunk7.txt | cut -c1-1500 ;;
  mfma)          # A/B on one box: the Gram blocks on the matrix cores (the in-tree library) against the DPP form (tools/_build/ab_nomfma.so, -DLL_MFMA_GRAM=0); SEPMC larger-batch knobs; then parity
    for r in 1 2 3; do for v in "" tools/_build/ab_nomfma.so; do echo "== ${v:-in-tree (MFMA Gram)} (round $r)"
      LL_LIB=$v python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,8192:4:10:10:32,65536:4:10:10:1"; LL_LIB=$v python tools/sweep_epmc.py "4096:1:32,65536:1:1"; LL_LIB=$v python tools/sweep_sepmc.py "2048:0:32,32768:0:1"; done; done > $OUT/mfma_ab.txt 2>&1
    cat $OUT/mfma_ab.txt
    for r in 1 2; do for v in conelds noreload; do echo "== $v (round $r)"; LL_LIB=tools/_build/ab_$v.so python tools/sweep_sepmc.py "32768:0:1,2048:0:32"; done; done > $OUT/sepmc2_ab.txt 2>&1
    cat $OUT/sepmc2_ab.txt
    gpu_tests -k "test_gpu_parity or test_gpu_epmc or test_gpu_sepmc" ;;
  r05g)          # the whole suite at HEAD (MFMA Gram off, cone scalars in LDS for the larger-batch SEPMC build), the MFMA experiment (fi = 1) through the parity tests, large-batch sweeps
    gpu_tests
    LL_TEST_LIB=tools/_build/ab_mfma.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sepmc.py tests/test_gpu_epmc.py -m gpu -q -k "single_control_step or multi_step or partial_wave or auto_reset or pair_physics or robot_robot or terrain_physics" > $OUT/pytest_mfma.log 2>&1; echo "MFMA-Gram library through the parity tests: rc $?"; tail -3 $OUT/pytest_mfma.log
    python tools/sweep.py "4096:4:10:10:32,65536:4:10:10:1" > $OUT/sweeps.txt 2>&1; python tools/sweep_epmc.py "4096:1:32,65536:1:1" >> $OUT/sweeps.txt 2>&1; python tools/sweep_sepmc.py "2048:0:32,32768:0:1" >> $OUT/sweeps.txt 2>&1; cat $OUT/sweeps.txt ;;
  r05h)          # the bars policy's fall histogram; driver-style lines with and without the triad in front; the EPMC / SEPMC bench lines with their CPU legs
    python tools/hole_fall_histogram.py 1024 600 > $OUT/hole_fall_histogram.txt 2>/dev/null; cat $OUT/hole_fall_histogram.txt
    for i in 1 2 3; do LL_BENCH_TRIAD_FIRST=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee -a $OUT/driver_style_raw.txt | show "driver-style, no triad first"; done > $OUT/driver_style.txt
    for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee -a $OUT/driver_style_raw.txt | show "driver-style (triad first)  "; done >> $OUT/driver_style.txt
    python bench.py --gpus 1 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | show "2048 steps                  " >> $OUT/driver_style.txt; cat $OUT/driver_style.txt
    python bench.py --workload epmc > $OUT/epmc_bench.log 2>/dev/null; python bench.py --workload sepmc > $OUT/sepmc_bench.log 2>/dev/null; tail -c 300 $OUT/sepmc_bench.log ;;
  pipe)          # A/B on one box: cone turns with the second select in the next turn's wait state (in-tree) against the round-4 turn (tools/_build/ab_nopipe.so); then parity
    for r in 1 2 3; do for v in "" tools/_build/ab_nopipe.so; do echo "== ${v:-in-tree (pipelined cone turns)} (round $r)"
      LL_LIB=$v python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,65536:4:10:10:1"; LL_LIB=$v python tools/sweep_epmc.py "4096:1:32,65536:1:1"; LL_LIB=$v python tools/sweep_sepmc.py "2048:0:32,32768:0:1"; done; done > $OUT/cone_pipe_ab.txt 2>&1
    cat $OUT/cone_pipe_ab.txt
    gpu_tests -k "test_gpu_parity or pair_physics or terrain_physics or multi_step" ;;
  ab)            # A/B on one box: the in-tree library against tools/_build/ab_base.so (a previous build); then the parity tests of the changed paths
    for r in 1 2 3; do for v in "" tools/_build/ab_base.so; do echo "== ${v:-in-tree} (round $r)"
      LL_LIB=$v python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,4096:4:10:10:8,65536:4:10:10:1"; LL_LIB=$v python tools/sweep_epmc.py "4096:1:32"; LL_LIB=$v python tools/sweep_sepmc.py "2048:0:32"; done; done > $OUT/ab.txt 2>&1
    cat $OUT/ab.txt
    gpu_tests -k "test_gpu_parity or pair_physics or terrain_physics or multi_step or env_api" ;;
  r05i)          # which builds of the chase-tag kernels go wrong (seven rays per chunk and its variants, GPU against GPU); where the p2p hand-off's launch gaps come from; soak; the suite at HEAD
    timeout 900 python tools/diag_sepmc_builds.py tools/_build/diag/libllenv_c7.so tools/_build/diag/libllenv_c5.so tools/_build/diag/libllenv_c7O2.so tools/_build/diag/libllenv_c7prealloc.so tools/_build/diag/libllenv_c7bperm.so tools/_build/diag/libllenv_c7fence.so > $OUT/sepmc_builds.txt 2>&1; cut -c1-420 $OUT/sepmc_builds.txt
    timeout 600 python tools/diag_p2p_gaps.py > $OUT/p2p_gaps.txt 2>&1; cat $OUT/p2p_gaps.txt
    timeout 600 python tools/soak.py 60000 8000 8000 > $OUT/soak.txt 2>&1; cat $OUT/soak.txt
    gpu_tests ;;
  r05j)          # the chase-tag builds again, observations included, and against the host build; the SEPMC GPU tests with the closed corners; soak
    timeout 600 python tools/diag_sepmc_builds.py tools/_build/diag/libllenv_c7.so tools/_build/diag/libllenv_c5.so tools/_build/diag/libllenv_c7pyr.so > $OUT/sepmc_builds.txt 2>&1; cut -c1-600 $OUT/sepmc_builds.txt
    timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/chunk7.txt 2>&1; cut -c1-900 $OUT/chunk7.txt
    gpu_tests -k "sepmc"
    timeout 600 python tools/soak.py 60000 8000 8000 > $OUT/soak.txt 2>&1; cat $OUT/soak.txt ;;
  r05k)          # was the seven-ray failure of profiles/r05_sepmc_chunk7_diag.txt the build's or a stale library's?  Libraries rebuilt from the source of that commit (efa7b7a): seven against three rays, GPU against GPU and against that commit's host build
    D=tools/_build/diag
    LL_DIAG_REF=$D/libllenv_old_c3.so timeout 600 python tools/diag_sepmc_builds.py $D/libllenv_old_c7.so > $OUT/old_builds.txt 2>&1; cut -c1-600 $OUT/old_builds.txt
    LL_DIAG_EMUL=$D/libllenv_old_emul.so LL_DIAG_LIB3=$D/libllenv_old_c3.so LL_DIAG_LIB7=$D/libllenv_old_c7.so timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/old_chunk7.txt 2>&1; cut -c1-500 $OUT/old_chunk7.txt
    LL_DIAG_LIB3=$D/libllenv_old_c3.so LL_DIAG_LIB7=$D/libllenv_old_c7.so timeout 600 python tools/diag_sepmc_chunk7.py > $OUT/old_libs_new_emul.txt 2>&1; cut -c1-500 $OUT/old_libs_new_emul.txt ;;
  r05l)          # round 4's own diagnosis re-run on libraries rebuilt from round 4's source (commit 9b75351, its python tree under tools/_build/r04b_tree); today's builds at 201 arenas (a partial last wave) and over 40 steps
    (cd tools/_build/r04b_tree && timeout 600 python tools/diag_sepmc_rays.py s7) > $OUT/r04_tree_rays.txt 2>&1; cut -c1-400 $OUT/r04_tree_rays.txt
    D=tools/_build/diag
    LL_DIAG_N=201 LL_DIAG_STEPS=40 timeout 600 python tools/diag_sepmc_builds.py $D/libllenv_c7.so $D/libllenv_c5.so 2>&1 | grep -v "after step [0-9]*: robots off    0\|'flag_info': 0, 'flag_info_cheat': 0, 'with_flag': 0, 'control_spd': 0}; bitwise-equal observation rows 402 of 402" > $OUT/builds_201.txt; cut -c1-500 $OUT/builds_201.txt
    LL_DIAG_REF=$D/libllenv_r04_c3.so LL_DIAG_N=201 LL_DIAG_STEPS=40 timeout 600 python tools/diag_sepmc_builds.py $D/libllenv_r04_c7.so 2>&1 | grep -v "after step [0-9]*: robots off    0\|'flag_info': 0, 'flag_info_cheat': 0, 'with_flag': 0, 'control_spd': 0}; bitwise-equal observation rows 402 of 402" > $OUT/builds_r04final_201.txt; cut -c1-500 $OUT/builds_r04final_201.txt ;;
  r05m)          # bisect over the source revisions between round 4's seven-ray failure and today (libraries under tools/_build/bis, built here)
    timeout 900 python tools/diag_sepmc_bisect.py 9b75351 7477004 4596d44 e6ab84b a9cbc3e 184caf2 033b865 aab0946 884b8d2 19cfcab 01da58d 34ca010 > $OUT/bisect.txt 2>&1; cut -c1-700 $OUT/bisect.txt ;;
  r05n)          # A/B on one box: seven / five / three rays per chunk in the one-wave-per-SIMD chase-tag kernels (tools/_build/diag, built from this source)
    for r in 1 2 3; do for v in "" tools/_build/diag/libllenv_c5.so tools/_build/diag/libllenv_c7.so; do echo "== ${v:-in-tree (three rays)} (round $r)"; LL_LIB=$v python tools/sweep_sepmc.py "2048:0:32,2048:1:32,2048:0:1"; done; done > $OUT/ray_chunk_ab.txt 2>&1
    cat $OUT/ray_chunk_ab.txt | cut -c1-200 ;;
  r05o)          # the host-build nets under the PMC and EPMC step kernels (report first, then the tests as written)
    python - > $OUT/nets_report.txt 2>&1 <<'PY'
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import subprocess; subprocess.check_call(['make', '-C', 'tests/emul', '-s'])
import parity_common as pc, epmc_parity_common as ec
from lifelike_agility_and_play_amd import mocap, urdf_model
E = 'tests/emul/_build/libllenv_emul.so'
t0 = time.time(); print('PMC', pc.check_engine_against_host_build(urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), E, report_only=True), '%.0f s' % (time.time() - t0), flush=True)
for el in (1, 3):
    t0 = time.time(); print('EPMC element', el, ec.check_engine_against_host_build(E, element=el, report_only=True), '%.0f s' % (time.time() - t0), flush=True)
PY
    cut -c1-1500 $OUT/nets_report.txt
    gpu_tests -k "every_observation" ;;
  ab2)           # A/B on one box, sweeps only: the in-tree library against tools/_build/ab_base.so
    for r in 1 2 3; do for v in "" tools/_build/ab_base.so; do echo "== ${v:-in-tree} (round $r)"
      LL_LIB=$v python tools/sweep.py "4096:4:10:10:32,4096:4:10:10:1,65536:4:10:10:1"; LL_LIB=$v python tools/sweep_epmc.py "4096:1:32"; LL_LIB=$v python tools/sweep_sepmc.py "2048:0:32"; done; done > $OUT/ab.txt 2>&1
    grep "==\|steps/s" $OUT/ab.txt | cut -c1-150 ;;
  final)         # the round's closing call: the whole -m gpu suite at HEAD, then the three bench lines against the committed counters
    gpu_tests
    python bench.py > $OUT/bench.log 2>$OUT/bench.err; tail -c 400 $OUT/bench.log
    for W in epmc sepmc; do python bench.py --workload $W --no-cpu-baseline > $OUT/${W}_bench.log 2>$OUT/${W}_bench.err; tail -c 300 $OUT/${W}_bench.log; done
    python tools/sweep.py "4096:4:10:10:32,16384:4:10:10:32,65536:4:10:10:32,65536:4:10:10:1" > $OUT/sweep_tail.txt 2>&1; cat $OUT/sweep_tail.txt
    python tools/sweep_epmc.py "65536:1:1" >> $OUT/sweep_tail.txt 2>&1; python tools/sweep_sepmc.py "32768:0:1" >> $OUT/sweep_tail.txt 2>&1; tail -2 $OUT/sweep_tail.txt ;;
  *) echo "unknown target $TARGET"; exit 2 ;;
esac
