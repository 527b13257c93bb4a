"""SEPMC kernel logic on the CPU: csrc/sepmc_step.hpp compiled for the host (tests/emul; the two robots of an arena run as two
threads that meet where the GPU rows exchange registers) and driven through the C ABI of include/llenv_sepmc.h."""
import os
import subprocess

import numpy as np
import pytest

import sepmc_parity_common as SC
from oracle import sepmc_oracle as SO

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s', '-j2'])
    return EMUL_LIB


def test_reset_cases_against_reference_goldens(emul_lib):
    SC.check_engine_reset_cases(emul_lib)


def test_scripted_episodes_against_reference_goldens(emul_lib, model_blob):
    SC.check_engine_episodes(emul_lib, SO.BlobModel(model_blob))


def test_free_running_invariants(emul_lib):
    out = SC.check_free_running(emul_lib, n_arenas=4, steps=80)
    print(out)


def test_flag_handover_by_physical_contact(emul_lib):
    SC.check_flag_handover_physical(emul_lib)


def test_robot_robot_contact(emul_lib):
    print(SC.check_robot_robot_contact(emul_lib))


def test_arena_corners_are_closed(emul_lib):
    print(SC.check_arena_corners_are_closed(emul_lib))


def test_trained_reference_policy_plays_chase_tag(emul_lib):
    print(SC.check_trained_policy_plays_chase_tag(emul_lib))


def test_per_robot_torque_limit(emul_lib):
    SC.check_per_robot_torque_limit(emul_lib)


def test_multi_step_launch(emul_lib):
    SC.check_multi_step_launch(emul_lib)


def test_pair_physics_against_oracle(emul_lib):
    print(SC.check_pair_physics_against_oracle(emul_lib))


def test_pyramid_friction_variant(emul_lib):
    """LLM_SPEC_FRICTION_MODE = 0 (ll_sepmc_set_spec_param: the pyramid of rounds 1 - 3) against the two-robot oracle under the same switch"""
    import epmc_parity_common as ec
    with ec.spec_variant(friction_mode=0):
        print(SC.check_pair_physics_against_oracle(emul_lib))
        SC.check_multi_step_launch(emul_lib)


def test_round4_spec_variant(emul_lib):
    """The spec of rounds 1 - 4 (speculative limit rows with their gate, ERP 0.2 on every row, push-out capped at 0.5 m/s) as an A/B leg with two robots, engine
    against the oracle under the same switches (default since round 5: Bullet's limit rule, contact ERP 0.08, no cap); then the rigid-body solver's two-ERP rule."""
    import epmc_parity_common as ec
    with ec.spec_variant(limit_speculative=1, erp=0.2, limit_erp=0.2, limit_erp_deep=-1, max_depen_speed=0.5):
        print(SC.check_pair_physics_against_oracle(emul_lib))
        SC.check_multi_step_launch(emul_lib)
    with ec.spec_variant(erp=0.2, erp_deep=0.08):
        # (robots of this case set START up to 5 cm inside each other; without a cap their push-out hangs on the closest-point normal of two crossing
        # capsule axes, and the second ERP is a step in the bias at -0.04: up to two of the 48 robots are ill-conditioned in the oracle itself)
        print(SC.check_pair_physics_against_oracle(emul_lib, cap_ill=2))


def test_reset_of_a_subset_of_arenas(emul_lib):
    """ll_sepmc_reset(arena_ids): only the listed arenas are re-seeded (new arena, roles, flag, poses, zeroed history); the
    others keep their state, episode scalars and observations bit for bit."""
    E = SC.make_engine(SC.env_config((1, 1, 0)), 5, emul_lib, auto_reset=0, seed=17)
    E.reset()
    for t in range(3):
        E.step_host(np.full((5, 2, 12), 0.05, np.float32))
    s0, o0, b0, ep0 = E.state().copy(), E.obs().copy(), E.boxes()[0].copy(), {k: v.copy() for k, v in E.episode().items()}
    E.reset(arena_ids=[1, 3])
    s1, o1, b1, ep1 = E.state(), E.obs(), E.boxes()[0], E.episode()
    keep, redo = [0, 2, 4], [1, 3]
    np.testing.assert_array_equal(s1[keep], s0[keep]); np.testing.assert_array_equal(o1[keep], o0[keep]); np.testing.assert_array_equal(b1[keep], b0[keep])
    for k in ('flag_x', 'counter', 'friction', 'with_flag0'):
        np.testing.assert_array_equal(ep1[k][keep], ep0[k][keep])
    assert np.all(ep1['counter'][redo] == 0) and np.all(ep1['flag_x'][redo] != ep0['flag_x'][redo]) and np.all(s1[redo][:, :, 2] == 0.5)
    assert np.all(o1[redo][:, :, 99:135] == 0.0)                                   # action history of a fresh episode
    assert not np.array_equal(b1[redo], b0[redo])                                  # new cubes
    E.close()


def test_free_running_against_the_oracle_env(emul_lib):
    print(SC.check_free_running_against_oracle_env(emul_lib))


def test_free_running_with_another_prop_type(emul_lib):
    """A shorter prop (CTG:90-104: any list of the five keys, in the given order) moves every later observation block."""
    print(SC.check_free_running_against_oracle_env(emul_lib, n_steps=3, prop_type=['e_g', 'joint_pos'], element_sets=((1, 0, 1),)))


def test_free_running_with_observation_noise(emul_lib):
    print(SC.check_free_running_against_oracle_env(emul_lib, n_steps=3, element_sets=((0, 1, 0),), noisy=True))


def test_parked_variant_equals_plain(emul_lib):
    SC.check_parked_variant_equals_plain(emul_lib)


def test_pair_physics_with_bullets_pair_rows(emul_lib):
    """Round 6: the engine twins of LLM_SPEC_PAIR_FRICTION (0.25 = 0.5 x 0.5), LLM_SPEC_MAX_PAIR (4: a manifold's four points) and LLM_SPEC_SELF_FRICTION (0.25) -- the XROWS build of the
    kernel source -- against the oracle under the same switches: two robots in contact, one control step, standing bars"""
    print(SC.check_pair_physics_against_oracle(emul_lib, spec=dict(pair_friction=0.25)))
    print(SC.check_pair_physics_against_oracle(emul_lib, spec=dict(max_pair=4), cap_ill=2))
    print(SC.check_pair_physics_against_oracle(emul_lib, spec=dict(pair_friction=0.25, max_pair=4, self_friction=0.25), cap_ill=2))


def test_rays_by_a_kernel_of_their_own_equal_the_fused_rays(emul_lib):
    SC.check_split_rays_equal_fused(emul_lib)



def test_game_statistics_against_the_oracle_env(emul_lib):
    """the mechanism of the GPU test of the same name at a size the CPU build affords (the distribution bars are asserted on the GPU, 256 games)"""
    SC.check_game_statistics(emul_lib, n_arenas=4, frac_tol=0.6, len_tol=1.5, ks_p=0.0, n_se=3.0)
