"""Kernel LOGIC tests on the CPU: the kernel body (csrc/pmc_step.hpp) and the engine host code are compiled for
the host with a 4-wide stand-in for the GPU quad (tests/emul) and driven through the same C ABI as the product.
These prove the algorithm (float32, quad formulation) against the float64 oracle and the reference goldens here,
where no GPU exists; tests/test_gpu_parity.py repeats the same checks on the real HIP library."""
import os
import subprocess

import numpy as np
import pytest

import parity_common as pc

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='session')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s', '-j2'])
    return EMUL_LIB


def test_reset_against_reference_goldens(golden, model_blob, mocap_table, emul_lib):
    pc.check_reset_against_goldens(golden, model_blob, mocap_table, emul_lib)


def test_single_control_step_parity(golden, orc, model_blob, mocap_table, emul_lib):
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=24, n_steps=10)
    print('config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))


def test_sliding_direction_friction_variant(golden, orc, model_blob, mocap_table, emul_lib):
    """LLM_SPEC_FRICTION_DIRS = 1 (first friction direction along the contact point's sliding velocity, Bullet's published default rule: the one
    audit switch that exists in the kernel too) against the oracle with the same switch.  The rule is discontinuous where a point stops
    sliding (|v_lat|^2 = 1.2e-7: btPlaneSpace1 below, the velocity above) and ill-conditioned just above, so this VARIANT -- not the shipped
    spec -- is held to the standing bars for 98 % of the samples and to 2e-2 for all."""
    st = pc.run_lockstep(golden, orc, model_blob, mocap_table, emul_lib, 24, 10, 7, resync=True, spec=dict(friction_dirs=1))
    c, v = np.asarray(st['config']), np.asarray(st['vel'])
    print('friction_dirs=1: config err p50 / p98 / max', np.percentile(c, [50, 98, 100]), 'vel (rel)', np.percentile(v, [50, 98, 100]))
    assert np.percentile(c, 98) < 1e-4 and np.percentile(v, 98) < 1e-3 and c.max() < 2e-2, (np.percentile(c, [98, 100]), np.percentile(v, [98, 100]))


def test_free_running_episode_statistics(golden, orc, model_blob, mocap_table, emul_lib):
    print(pc.check_rollout_statistics(golden, orc, model_blob, mocap_table, emul_lib, n_envs=64))


def test_policy_driven_parity(golden, orc, model_blob, mocap_table, emul_lib):
    st = pc.check_policy_driven_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=16)
    print('policy-driven: config err 50/99/max', np.percentile(st['config'], [50, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 99, 100]))


def test_contact_rich_parity(golden, orc, model_blob, mocap_table, emul_lib):
    out = pc.check_contact_rich_parity(golden, orc, model_blob, mocap_table, emul_lib)
    print('contact-rich: config err', np.percentile(out['config'], [50, 100]), 'vel', np.percentile(out['vel'], [50, 100]))


def test_trained_reference_policy_tracks_in_our_simulator(emul_lib):
    out = pc.check_trained_policy_tracks(emul_lib)
    print('trained PMC policy: mean reward/step %.3f, tracked %.0f%%' % (out['mean_reward'], 100 * out['tracked']))


def test_trajectory_ring(model_blob, mocap_table, emul_lib):
    import ctypes
    def read_ring(addr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)).reshape(shape).copy()

    def write_dev(addr, arr):
        np.ctypeslib.as_array((ctypes.c_float * arr.size).from_address(addr))[:] = arr.ravel()
    pc.check_trajectory_ring(model_blob, mocap_table, emul_lib, read_ring, write_dev)


def test_multi_step_launch(model_blob, mocap_table, emul_lib):
    import ctypes

    def read_ring(addr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)).reshape(shape).copy()
    pc.check_multi_step_launch(model_blob, mocap_table, emul_lib, read_ring)
    pc.check_multi_step_launch(model_blob, mocap_table, emul_lib, read_ring, n_launches=2, spec=dict(friction_mode=0))    # the pyramid builds


def test_obstacle_variant(golden, orc, model_blob, mocap_table, emul_lib):
    n = pc.check_obstacle_variant(golden, orc, model_blob, mocap_table, emul_lib)
    print('obstacle variant: %d episodes ended on the box' % n)


def test_scripted_episodes_against_reference_goldens(golden, model_blob, mocap_table, emul_lib):
    pc.check_scripted_episodes_against_goldens(golden, model_blob, mocap_table, emul_lib)


def test_host_build_net_runs(model_blob, mocap_table, emul_lib):
    # (the checker of tests/test_gpu_parity.py::test_every_observation_entry_..., here the host build against itself: every difference is exactly zero)
    out = pc.check_engine_against_host_build(model_blob, mocap_table, emul_lib, n_envs=192, steps=10, gpu_lib=emul_lib)
    assert out['worst_obs'] == 0.0 and out['left_out'] == 0 and out['reseeded'] >= 3, out


def test_auto_reset_equals_manual_reset(model_blob, mocap_table, emul_lib):
    assert pc.check_auto_reset_equals_manual_reset(model_blob, mocap_table, emul_lib) >= 5


def test_self_collision_parity(golden, orc, model_blob, mocap_table, emul_lib):
    out = pc.check_self_collision_parity(golden, orc, model_blob, mocap_table, emul_lib)
    assert out['stopped'] >= 8


def test_self_collision_with_friction_parity(golden, orc, model_blob, mocap_table, emul_lib):
    """LLM_SPEC_SELF_FRICTION = 0.25 (Bullet's 0.5 x 0.5 for two robot links; round 6: the engine twin of what had been an oracle-only switch): the leg-leg contact's two
    tangential rows in the kernel source against the oracle's, standing bars; the friction must have had something to act on"""
    out = pc.check_self_collision_parity(golden, orc, model_blob, mocap_table, emul_lib, spec=dict(self_friction=0.25))
    print('self friction 0.25: worst config %.2e vel %.2e; the switch moved the oracle\'s joint rates by up to %.3f rad/s' % (out['config'].max(), out['vel'].max(), out['moved']))
    assert out['stopped'] >= 8 and out['moved'] > 1e-3


def test_nonfinite_guard(model_blob, mocap_table, emul_lib):
    pc.check_nonfinite_guard(model_blob, mocap_table, emul_lib)


def test_reset_argument_handling(model_blob, mocap_table, emul_lib):
    pc.check_reset_argument_handling(model_blob, mocap_table, emul_lib)


def test_multi_step_launch_must_fit_the_unroll_ring(model_blob, mocap_table, emul_lib):
    """ll_step_random_n with unrolls recorded: a launch longer than the ring (unroll_length x n_buffers rows per env) would overwrite rows of its
    own and is refused with LL_EINVAL before anything runs (round-3 advice); running from one block into the next is the caller's business."""
    from lifelike_agility_and_play_amd import capi
    E = pc.make_engine(model_blob, mocap_table, 8, emul_lib, auto_reset=1, seed=3)
    E.reset()
    E.step_random_n(0.1, 20)                      # no unrolls: any length
    E.enable_unrolls(8, 2)
    E.step_random_n(0.1, 5)
    assert E.unroll_position() == (0, 5)
    with pytest.raises(capi.LLError) as ei:
        E.step_random_n(0.1, 17)
    assert ei.value.code == capi.LL_EINVAL and 'unroll ring' in str(ei.value)
    assert E.unroll_position() == (0, 5)          # nothing ran
    E.step_random_n(0.1, 16)                      # exactly the ring
    assert E.unroll_position() == (2, 5)
    E.close()


def test_pyramid_friction_variant(golden, orc, model_blob, mocap_table, emul_lib):
    """LLM_SPEC_FRICTION_MODE = 0 (all t1 rows, then all t2 rows, box bounds: the spec of rounds 1 - 3, still a build of every step kernel)
    against the oracle with the same switch: held to the bars of the shipped spec (the cone, which every other test of this file runs)."""
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=24, n_steps=10, spec=dict(friction_mode=0))
    print('friction_mode=0: config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))


ROUND4_SPEC = dict(limit_speculative=1, erp=0.2, limit_erp=0.2, limit_erp_deep=-1, max_depen_speed=0.5)   # rounds 1 - 4: speculative limit rows + gate, ERP 0.2, push-out capped
SPECULATIVE_LIMITS = dict(limit_speculative=1)             # ... the limit rule alone
TWO_ERPS = dict(erp=0.2, erp_deep=0.08)                    # btContactSolverInfo's m_erp / m_erp2 around m_splitImpulsePenetrationThreshold as the RIGID-body solver picks them


def test_round4_spec_variant(golden, orc, model_blob, mocap_table, emul_lib):
    """The spec of rounds 1 - 4 as an A/B leg of engine and oracle (LLM_SPEC_LIMIT_SPECULATIVE = 1 with its gate, ERP 0.2 on every row, the 0.5 m/s cap): since
    round 5 the default is btMultiBodyJointLimitConstraint's rule -- a row only once the limit is passed -- with contact ERP 0.08 and no cap
    (profiles/r05_limit_rows.md).  Both held to the oracle under the same switches at the standing bars; and the two limit rules ARE different simulators."""
    for spec in (ROUND4_SPEC, SPECULATIVE_LIMITS):
        st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=24, n_steps=10, spec=spec)
        print(spec, 'config err 50/90/99/max', np.percentile(st['config'], [50, 90, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 90, 99, 100]))
    A = pc.make_engine(model_blob, mocap_table, 16, emul_lib, seed=5, auto_reset=0); B = pc.make_engine(model_blob, mocap_table, 16, emul_lib, seed=5, auto_reset=0)
    A.set_spec(**SPECULATIVE_LIMITS)
    assert B.get_spec('limit_speculative') == 0.0 and A.get_spec('limit_speculative') == 1.0 and B.get_spec('erp') == np.float32(0.08) and B.get_spec('limit_erp') == np.float32(0.2)
    A.reset(); B.reset()
    over = 0
    lo, hi = model_blob[241:253], model_blob[253:265]                     # LLM_OFF_Q_LO / _HI
    for t in range(20):
        A.step_random(0.5); B.step_random(0.5)
        q = B.state()[:, 13:25]
        over += int(((q < lo - 1e-4) | (q > hi + 1e-4)).sum())
    assert over > 0                                                         # Bullet's rule lets a joint overshoot (and walks it back within 0.04 rad); the speculative rule stops it AT the limit
    assert np.abs(A.state()[:, 13:25] - B.state()[:, 13:25]).max() > 1e-3
    qa = A.state()[:, 13:25]
    assert ((qa > lo - 2e-3) & (qa < hi + 2e-3)).all()
    A.close(); B.close()


def test_two_erp_variant(golden, orc, model_blob, mocap_table, emul_lib):
    """LLM_SPEC_ERP_DEEP in engine and oracle alike; a robot dropped INTO the ground meets the deep branch.  (Not the spec: btMultiBodyConstraintSolver takes
    m_erp2 at every depth; the two-ERP rule is the rigid-body solver's, priced in profiles/r05_limit_rows.md.)"""
    st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=16, n_steps=6, spec=TWO_ERPS)
    print('erp_deep=0.08, no cap: config err 50/99/max', np.percentile(st['config'], [50, 99, 100]), 'vel (rel)', np.percentile(st['vel'], [50, 99, 100]))
    pc.check_deep_penetration_against_oracle(orc, model_blob, mocap_table, emul_lib, spec=TWO_ERPS)
    pc.check_deep_penetration_against_oracle(orc, model_blob, mocap_table, emul_lib, spec=dict())
    pc.check_deep_penetration_against_oracle(orc, model_blob, mocap_table, emul_lib, spec=ROUND4_SPEC)


def test_friction_mode_switch_is_validated(model_blob, mocap_table, emul_lib):
    """ll_set_spec_param / ll_epmc_set_spec_param / ll_sepmc_set_spec_param (LLM_SPEC_FRICTION_MODE): every engine has the cone-coupled solve (2,
    the default: LLM_FRICTION_MODE) and the pyramid (0); the oracle-only modes are refused loudly, nothing is silently ignored."""
    from lifelike_agility_and_play_amd import capi
    import epmc_parity_common as ec
    import sepmc_parity_common as SC
    engines = [pc.make_engine(model_blob, mocap_table, 4, emul_lib), pc.make_engine(model_blob, mocap_table, 4, emul_lib, set_obstacle=1),
               ec.make_engine(ec.env_config(1), 4, emul_lib), SC.make_engine(SC.env_config((0, 0, 0)), 2, emul_lib)]
    for E in engines:
        assert E.get_spec('friction_mode') == 2.0 == capi.LLM_FRICTION_MODE
        for bad in (1, 3, -1, 0.5):
            with pytest.raises(capi.LLError):
                E.set_spec(friction_mode=bad)
        E.set_spec(friction_mode=0); assert E.get_spec('friction_mode') == 0.0
        E.set_spec(friction_mode=2); assert E.get_spec('friction_mode') == 2.0
        E.close()


def test_cone_scalars_through_the_row_scratch_equal_registers(model_blob, mocap_table, emul_lib):
    """lanes.hpp WithConeInLds -- what the larger-batch GPU builds run: the cone round's 32 cross scalars wait in the row's LDS scratch and are read
    block by block -- against the register version, bit for bit (the host build runs the variant under LL_EMUL_PARK; flat ground and set_obstacle)."""
    import os
    for kw in (dict(), dict(set_obstacle=1)):
        A = pc.make_engine(model_blob, mocap_table, 8, emul_lib, seed=3, auto_reset=1, **kw)
        B = pc.make_engine(model_blob, mocap_table, 8, emul_lib, seed=3, auto_reset=1, **kw)
        A.reset(); B.reset()
        try:
            for t in range(25):
                os.environ.pop('LL_EMUL_PARK', None)
                A.step_random(pc.SIGMA)
                os.environ['LL_EMUL_PARK'] = '1'
                B.step_random(pc.SIGMA)
                np.testing.assert_array_equal(A.state(), B.state())
                np.testing.assert_array_equal(A.obs(), B.obs())
        finally:
            os.environ.pop('LL_EMUL_PARK', None)
        assert np.abs(A.state()[:, 25:37]).max() > 0.1
        A.close(); B.close()


def test_gram_blocks_as_a_background_job_equal_the_shuffle_statement(orc, model_blob, mocap_table, emul_lib):
    """lanes.hpp WithGramPipe -- what the one-wave-per-SIMD PMC cone kernels run since round 6: the five Gram blocks of a contact's rows are formed one k at a time
    (on the GPU: one MFMA at a time, between pieces of the next row's arithmetic) and collected later.  Same scalars up to the order of two additions (the base part is
    summed from zero and the joint part added last): the host build of the piped statement (LL_EMUL_GRAM_PIPE) stays within float32 rounding of the plain one over
    25 free-running steps with contacts, and holds the standing bars against the oracle."""
    import os
    for kw in (dict(), dict(set_obstacle=1)):
        A = pc.make_engine(model_blob, mocap_table, 8, emul_lib, seed=3, auto_reset=0, **kw)
        B = pc.make_engine(model_blob, mocap_table, 8, emul_lib, seed=3, auto_reset=0, **kw)
        A.reset(); B.reset()
        worst = 0.0
        try:
            for t in range(25):
                B.set_state(A.state())                               # compare ONE step at a time from identical states (a contact's discontinuities amplify rounding over many)
                os.environ.pop('LL_EMUL_GRAM_PIPE', None)
                A.step_random(pc.SIGMA)
                os.environ['LL_EMUL_GRAM_PIPE'] = '1'
                B.step_random(pc.SIGMA)
                sa, sb = A.state(), B.state()
                vel_scale = 1.0 + np.abs(sa[:, 25:37]).max(axis=1, keepdims=True)
                worst = max(worst, np.abs(sa[:, :25] - sb[:, :25]).max(), (np.abs(sa[:, 25:37] - sb[:, 25:37]) / vel_scale).max())
        finally:
            os.environ.pop('LL_EMUL_GRAM_PIPE', None)
        print('piped against plain, worst over 25 single steps:', worst)
        assert worst < 2e-4, worst
        assert np.abs(A.state()[:, 25:37]).max() > 0.1
        A.close(); B.close()
    os.environ['LL_EMUL_GRAM_PIPE'] = '1'
    try:
        golden = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pmc_golden.npz'))
        st = pc.check_single_step_parity(golden, orc, model_blob, mocap_table, emul_lib, n_envs=16, n_steps=6)
        print('piped host build against the oracle: config err max %.2e' % st['config'].max())
    finally:
        os.environ.pop('LL_EMUL_GRAM_PIPE', None)


def test_reset_onto_a_mocap_discontinuity(orc, model_blob, mocap_table, emul_lib):
    print('worst configuration error against the oracle: %.2e' % pc.check_reset_onto_a_mocap_discontinuity(orc, model_blob, mocap_table, emul_lib))
