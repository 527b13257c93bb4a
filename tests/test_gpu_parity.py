"""GPU parity tests: the real HIP library (lifelike_agility_and_play_amd/csrc/libllenv.so) through the C ABI,
against the reference goldens and the float64 oracle.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

import parity_common as pc
from lifelike_agility_and_play_amd import capi

pytestmark = pytest.mark.gpu


def E_lib_is_hip(lib):
    import ctypes
    return isinstance(lib, ctypes.CDLL) and lib._name.endswith('csrc/libllenv.so') and 'emul' not in lib._name


def test_library_is_the_hip_build(model_blob, mocap_table):
    """The default library is the HIP build, not the host emulation the CPU tests compile: it lives in csrc/, carries a gfx950 code
    object and no emulation entry point, is mapped into this process next to the HIP runtime, and launches on a real stream."""
    import os
    lib = capi.load_library()
    assert lib.ll_abi_version() == 2
    assert lib._name == capi.DEFAULT_LIB and os.path.basename(os.path.dirname(lib._name)) == 'csrc'
    assert not hasattr(lib, 'emu_substep')                                  # tests/emul/emul.cpp only
    blob = open(lib._name, 'rb').read()
    assert b'gfx950' in blob and b'pmc_step_kernel' in blob                 # the offload bundle
    maps = open('/proc/self/maps').read()
    assert 'csrc/libllenv.so' in maps and 'libamdhip64' in maps         # (the host build may be mapped too: the nets of this suite load it as the CHECKER, by explicit path)
    assert E_lib_is_hip(lib)
    E = pc.make_engine(model_blob, mocap_table, 4, None)
    assert E.device_ptrs().stream                                           # a hipStream_t; the emulation reports NULL
    E.close()


def test_reset_against_reference_goldens(golden, model_blob, mocap_table):
    pc.check_reset_against_goldens(golden, model_blob, mocap_table, None)


def test_single_control_step_parity(golden, orc, model_blob, mocap_table):
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=48, n_steps=12)
    print('config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))


def test_sliding_direction_friction_variant(golden, orc, model_blob, mocap_table):
    """LLM_SPEC_FRICTION_DIRS = 1 (first friction direction along the contact point's sliding velocity, Bullet's published default rule: the one
    audit switch that exists in the kernel too) against the oracle with the same switch.  The rule is discontinuous where a point stops
    sliding (|v_lat|^2 = 1.2e-7: btPlaneSpace1 below, the velocity above) and ill-conditioned just above, so this VARIANT -- not the shipped
    spec -- is held to the standing bars for 98 % of the samples and to 2e-2 for all."""
    st = pc.run_lockstep(golden, orc, model_blob, mocap_table, None, 32, 12, 7, resync=True, spec=dict(friction_dirs=1))
    c, v = np.asarray(st['config']), np.asarray(st['vel'])
    print('friction_dirs=1: config err p50 / p98 / max', np.percentile(c, [50, 98, 100]), 'vel (rel)', np.percentile(v, [50, 98, 100]))
    assert np.percentile(c, 98) < 1e-4 and np.percentile(v, 98) < 1e-3 and c.max() < 2e-2, (np.percentile(c, [98, 100]), np.percentile(v, [98, 100]))


def test_pyramid_friction_variant(golden, orc, model_blob, mocap_table):
    """LLM_SPEC_FRICTION_MODE = 0 (all t1 rows, then all t2 rows, box bounds: the spec of rounds 1 - 3, still a build of every step kernel)
    against the oracle with the same switch: held to the bars of the shipped spec, in the occupancy-1 build and (5000 envs) the occupancy-2 build."""
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=48, n_steps=12, spec=dict(friction_mode=0))
    print('friction_mode=0: config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=24, n_steps=8, total_envs=5000, spec=dict(friction_mode=0))
    print('friction_mode=0, 5000 envs: config err max', np.max(st['config']), 'vel (rel)', np.max(st['vel']))


def test_round4_spec_variant(golden, orc, model_blob, mocap_table):
    """The spec of rounds 1 - 4 as an A/B leg of the ENGINE (LLM_SPEC_LIMIT_SPECULATIVE = 1 with its gate, ERP 0.2 on every row, the 0.5 m/s cap; default since
    round 5: btMultiBodyJointLimitConstraint's rule, contact ERP 0.08, no cap) against the oracle under the same switches, at the standing bars, in the
    occupancy-1 build and (5000 envs) the occupancy-2 build; then the penetration-recovery variants on robots set INTO the ground."""
    spec = dict(limit_speculative=1, erp=0.2, limit_erp=0.2, limit_erp_deep=-1, max_depen_speed=0.5)
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=48, n_steps=12, spec=spec)
    print('round-4 spec: config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=24, n_steps=8, total_envs=5000, spec=spec)
    print('round-4 spec, 5000 envs: config err max', np.max(st['config']), 'vel (rel)', np.max(st['vel']))
    for sp in (dict(), spec, dict(erp=0.2, erp_deep=0.08), dict(erp=0.2)):
        print(sp or 'spec', 'base rise per control step at 5 / 30 / 45 / 80 mm:', pc.check_deep_penetration_against_oracle(orc, model_blob, mocap_table, None, spec=sp))
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=32, n_steps=8, spec=dict(limit_speculative=1))
    print('speculative limit rows alone: config err max', np.max(st['config']), 'vel (rel)', np.max(st['vel']))


def test_free_running_episode_statistics(golden, orc, model_blob, mocap_table):
    print(pc.check_rollout_statistics(golden, orc, model_blob, mocap_table, None, n_envs=512))


def test_policy_driven_parity(golden, orc, model_blob, mocap_table):
    st = pc.check_policy_driven_parity(golden, orc, model_blob, mocap_table, None, n_envs=48)
    print('policy-driven: config err 50/99/max', np.percentile(st['config'], [50, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 99, 100]))


def test_partial_wave_and_odd_batch_sizes(golden, orc, model_blob, mocap_table):
    for n in (1, 5, 17):
        st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=n, n_steps=3, seed=n)
        assert len(st['config']) > 0


def test_full_size_invariants(golden, model_blob, mocap_table):
    """BASELINE config 2 size (4096 envs, all clips, random policy, auto-reset): size-independent properties."""
    n = 4096
    E = pc.make_engine(model_blob, mocap_table, n, None, auto_reset=1, seed=3)
    E.reset()
    info0 = E.episode_info()
    assert info0['clip'].min() >= 0 and info0['clip'].max() < E.n_clips and len(np.unique(info0['clip'])) > 40
    done_total, rew = 0, []
    for t in range(60):
        E.fill_random_actions(pc.SIGMA)
        E.step()
        r, d, why = E.reward_done()
        s, o = E.state(), E.obs()
        assert np.isfinite(s).all() and np.isfinite(o).all()
        np.testing.assert_allclose(np.linalg.norm(s[:, 3:7], axis=1), 1.0, atol=1e-5)     # unit quaternions
        assert (r >= 0).all() and (r <= 1.0 + 1e-6).all()                                   # convex combination of exp(-x)
        assert ((why != 0) == d).all()
        done_total += int(d.sum())
        rew.append(r.mean())
        info = E.episode_info()
        assert (info['steps'][d] == 0).all()                                               # re-seeded inside the kernel
        assert (info['steps'][~d] >= 1).all()
    c = E.counters()
    assert c['env_steps'] == 60 * n and c['episodes'] == done_total and c['nonfinite'] == 0
    assert done_total > 0
    p, avg_r, avg_l = E.sampling_table()
    assert abs(p.sum() - 1.0) < 1e-9 and (p > 0).all()
    E.close()


def test_both_register_budgets_compute_the_same(golden, orc, model_blob, mocap_table):
    """pmc_step_kernel<1> (one wavefront per SIMD, batches up to 4096 envs: the build every oracle parity test exercises) and
    pmc_step_kernel<2> (256 registers, larger batches) are two compilations of one source.  They are not bit-identical (the compiler fuses
    multiply-adds differently in the two), so the larger-batch build is held to the smaller one with the bars the smaller one is held to the
    oracle with: every env, every step, 1e-4 configuration / 1e-3 relative velocity, re-synchronised after each control step; rewards
    and termination reasons equal.  A sample outside those bars is examined with the oracle (parity_common.oracle_self_deviation): it
    must be a step that is ill-conditioned in the float64 oracle itself and stay within 4 x the oracle's own deviation; counted, capped."""
    B1 = pc.make_oracle_batch(orc, model_blob, mocap_table, n_envs=1)
    ill = []
    n_small, n_big = 96, 4096 + 160
    rng = np.random.default_rng(21)
    clip = rng.integers(0, 62, n_big).astype(np.int32)
    t0 = rng.uniform(0.0, 1.0, n_big)
    A = pc.make_engine(model_blob, mocap_table, n_small, None, auto_reset=0, seed=1)
    B = pc.make_engine(model_blob, mocap_table, n_big, None, auto_reset=0, seed=1)
    A.reset(clip=clip[:n_small], t0=t0[:n_small]); B.reset(clip=clip, t0=t0)
    assert np.array_equal(A.state(), B.state()[:n_small]) and np.array_equal(A.obs(), B.obs()[:n_small])
    worst_c = worst_v = 0.0
    n_reason = 0
    for t in range(40):
        act = (rng.normal(size=(n_big, 12)) * 0.3).astype(np.float32)
        pre = A.state().astype(np.float64)
        A.step_host(act[:n_small]); B.step_host(act)
        sa, sb_all = A.state().astype(np.float64), B.state()
        sb = sb_all[:n_small].astype(np.float64)
        err = np.abs(np.stack([pc.quat_align(sb[i], sa[i]) for i in range(n_small)]) - sa)
        ce = np.maximum(err[:, 0:7].max(1), err[:, 13:25].max(1))
        ve = np.maximum(err[:, 7:13].max(1), err[:, 25:37].max(1)) / (1.0 + np.abs(sa[:, 25:37]).max(1))
        ra, rb = A.reward_done(), B.reward_done()
        bad = (ce >= 1e-4) | (ve >= 1e-3)
        live = ~(ra[1] | rb[1][:n_small])
        for i in np.nonzero(bad & live)[0]:
            cc, cv, _ = pc.oracle_self_deviation(B1, pre[i], act[i])
            ill.append((float(ce[i]), float(ve[i]), cc, cv))
        ce[bad], ve[bad] = 0.0, 0.0
        worst_c, worst_v = max(worst_c, ce.max()), max(worst_v, ve.max())
        np.testing.assert_allclose(ra[0][~bad], rb[0][:n_small][~bad], atol=2e-5)
        n_reason += int((ra[2] != rb[2][:n_small]).sum())
        sb_all[:n_small] = A.state()
        B.set_state(sb_all)
    print('register budgets: worst config %.2e, velocity %.2e outside of %d samples examined with the oracle: %s' % (worst_c, worst_v, len(ill), ill))
    assert worst_c < 1e-4 and worst_v < 1e-3, (worst_c, worst_v)
    assert len(ill) <= 2, ill                                                   # of 96 x 40 samples (about 1 in 3000 under random actions; observed on MI355X: 1)
    for (ce, ve, cc, cv) in ill:
        assert ce <= max(1e-4, pc.ILL_FACTOR * cc) and ve <= max(1e-3, pc.ILL_FACTOR * cv), ('between the builds', ce, ve, 'oracle self-deviation', cc, cv)
    assert n_reason <= 1 + len(ill), n_reason                                   # (a threshold test may fall either way once)
    A.close(); B.close()


def test_larger_batch_build_against_the_oracle(golden, orc, model_blob, mocap_table):
    """pmc_step_kernel<2> (batches above 4096 envs: every sweep figure above 21 M env-steps/s) held to the float64 oracle DIRECTLY, with the
    bars of the occupancy-1 build: 4096 + 256 envs, the oracle in lock-step with 48 of them spread over the first, middle and last
    wavefronts of the grid, every sample."""
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, None, n_envs=48, n_steps=12, total_envs=4096 + 256)
    print('occupancy-2 build vs oracle: config err 50/99/max', np.percentile(st['config'], [50, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 99, 100]))


def test_step_random_is_fill_then_step(model_blob, mocap_table):
    """ll_step_random (actions drawn inside the step kernel) == ll_fill_random_actions + ll_step, bit for bit, including the
    recorded actions and the sampling table the step leaves behind; two batch sizes, so both kernel variants run."""
    for n in (70, 4200):
        A = pc.make_engine(model_blob, mocap_table, n, None, auto_reset=1, seed=11)
        B = pc.make_engine(model_blob, mocap_table, n, None, auto_reset=1, seed=11)
        A.reset(); B.reset()
        for t in range(25):
            A.fill_random_actions(pc.SIGMA); A.step()
            B.step_random(pc.SIGMA)
        assert np.array_equal(A.obs(), B.obs()) and np.array_equal(A.state(), B.state())
        ra, rb = A.reward_done(), B.reward_done()
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[2], rb[2])
        ta, tb = A.sampling_table(), B.sampling_table()
        assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
        assert A.counters() == B.counters() and A.counters()['episodes'] > 0
        A.close(); B.close()


def test_multi_step_launch(model_blob, mocap_table):
    """ll_step_random_n: k control steps in one launch == k launches, bit for bit (both kernel variants: 70 / 4096 and 4200 envs)."""
    from lifelike_agility_and_play_amd import gather
    import torch
    if not torch.cuda.is_available():
        pytest.skip('torch.cuda is not available on this box (needed to read the unroll buffers)')

    def read_ring(addr, shape):
        return gather.device_tensor(addr, shape).cpu().numpy()
    pc.check_multi_step_launch(model_blob, mocap_table, None, read_ring, sizes=(70, 4096, 4200), k=7, n_launches=5)        # (4096: the benchmark's own kernel at its own size)
    pc.check_multi_step_launch(model_blob, mocap_table, None, read_ring, sizes=(70, 4200), k=7, n_launches=3, spec=dict(friction_mode=0))   # the pyramid builds


def test_deterministic_mode_runs_multi_step_calls_as_single_launches(model_blob, mocap_table, monkeypatch):
    """LL_DETERMINISTIC=1 (read when an engine is created): ll_step_random_n(k) is k single launches -- for a device that is shared with other kernels, where a multi-step
    launch may have to re-seed from the newest table version there is (counted by ll_get_table_sync) and its result hangs on timing"""
    import math
    from lifelike_agility_and_play_amd import capi
    RW = {'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05}
    PT = ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('LL_DETERMINISTIC', mode)
        cfg = capi.make_config(256, control_freq=50.0, sim_freq=500.0, kd=0.5, reward_weights=RW, prop_type=PT, prioritized_sample_factor=3.0, auto_reset=1, seed=3)
        E = capi.Engine(cfg, model_blob, mocap_table)
        E.reset(); E.enable_kernel_timing(True)
        E.step_random_n(math.exp(-2), 8); E.sync()
        _, launches, steps = E.kernel_time_stats()
        out[mode] = (launches, steps, E.state().copy(), E.table_sync())
        E.close()
    assert out['0'][:2] == (1, 8) and out['1'][:2] == (8, 8), (out['0'][:2], out['1'][:2])
    np.testing.assert_array_equal(out['0'][2], out['1'][2])


def test_contact_rich_parity(golden, orc, model_blob, mocap_table):
    out = pc.check_contact_rich_parity(golden, orc, model_blob, mocap_table, None)
    print('contact-rich: config err', np.percentile(out['config'], [50, 100]), 'vel', np.percentile(out['vel'], [50, 100]))


def test_trained_reference_policy_tracks_in_our_simulator():
    out = pc.check_trained_policy_tracks(None, n_envs=256, n_steps=300)
    print('trained PMC policy on GPU: mean reward/step %.3f, tracked %.0f%%' % (out['mean_reward'], 100 * out['tracked']))


def test_reset_onto_a_mocap_discontinuity(orc, model_blob, mocap_table):
    print('worst configuration error against the oracle: %.2e' % pc.check_reset_onto_a_mocap_discontinuity(orc, model_blob, mocap_table, None))


def test_obstacle_variant(golden, orc, model_blob, mocap_table):
    n = pc.check_obstacle_variant(golden, orc, model_blob, mocap_table, None)
    print('obstacle variant: %d episodes ended on the box' % n)
    n = pc.check_obstacle_variant(golden, orc, model_blob, mocap_table, None, total_envs=4096 + 128)        # pmc_step_kernel<2, true, ., .>: the larger-batch build
    print('obstacle variant, 4224 envs: %d episodes ended on the box' % n)


def test_scripted_episodes_against_reference_goldens(golden, model_blob, mocap_table):
    pc.check_scripted_episodes_against_goldens(golden, model_blob, mocap_table, None)


def test_auto_reset_equals_manual_reset(model_blob, mocap_table):
    assert pc.check_auto_reset_equals_manual_reset(model_blob, mocap_table, None, n_envs=70) >= 5


def test_self_collision_parity(golden, orc, model_blob, mocap_table):
    out = pc.check_self_collision_parity(golden, orc, model_blob, mocap_table, None, n_envs=32)
    assert out['stopped'] >= 16


def test_self_collision_with_friction_parity(golden, orc, model_blob, mocap_table):
    """LLM_SPEC_SELF_FRICTION = 0.25 on the GPU (round 6: engine twin of the oracle's switch), one-wave-per-SIMD build; see tests/test_kernel_logic_emul.py"""
    out = pc.check_self_collision_parity(golden, orc, model_blob, mocap_table, None, n_envs=32, spec=dict(self_friction=0.25))
    print('self friction 0.25 on the GPU: worst config %.2e vel %.2e, moved %.3f' % (out['config'].max(), out['vel'].max(), out['moved']))
    assert out['stopped'] >= 16 and out['moved'] > 1e-3
    import parity_common as pc2
    E = pc2.make_engine(model_blob, mocap_table, 8)                       # the switch needs the cone builds and flat ground: anything else is refused, not ignored
    E.set_spec(friction_mode=0, self_friction=0.25)
    E.reset()
    with pytest.raises(Exception):
        E.step_random(pc2.SIGMA); E.sync()
    E.close()


def test_nonfinite_guard(model_blob, mocap_table):
    pc.check_nonfinite_guard(model_blob, mocap_table, None)


def test_reset_argument_handling(model_blob, mocap_table):
    pc.check_reset_argument_handling(model_blob, mocap_table, None)


def test_every_observation_entry_against_the_host_build_of_the_kernel_source(model_blob, mocap_table):
    """BASELINE config 2's size and the larger-batch build: parity_common.check_engine_against_host_build."""
    import os
    import subprocess
    emul_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
    subprocess.check_call(['make', '-C', emul_dir, '-s', '-j2'])
    lib = os.path.join(emul_dir, '_build', 'libllenv_emul.so')
    print('4096 envs (one wave per SIMD):', pc.check_engine_against_host_build(model_blob, mocap_table, lib))
    print('8192 envs (the 256-register build):', pc.check_engine_against_host_build(model_blob, mocap_table, lib, n_envs=8192, steps=8, seed=23))
