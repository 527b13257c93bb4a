"""EPMC kernel source (csrc/epmc_step.hpp) compiled for the host (tests/emul) against the reference goldens and the oracle.
tests/test_gpu_epmc.py repeats the same checks on the HIP library."""
import os
import subprocess

import pytest

import epmc_parity_common as ec

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s', '-j2'])
    return EMUL_LIB


def test_terrain_and_reset_against_reference_goldens(emul_lib):
    ec.check_terrain_and_reset_against_goldens(emul_lib)


def test_scripted_episodes_against_reference_goldens(emul_lib):
    ec.check_scripted_episodes_against_goldens(emul_lib)


def test_ray_casting_against_oracle(emul_lib):
    assert ec.check_ray_casting_against_oracle(emul_lib) > 100


def test_host_build_net_runs(emul_lib):
    out = ec.check_engine_against_host_build(emul_lib, n_envs=96, steps=6, gpu_lib=emul_lib)
    assert out['worst_prop'] == 0.0 and out['worst_tail'] == 0.0 and out['reseeded'] >= 96, out


def test_free_running_invariants(emul_lib):
    assert ec.check_free_running_invariants(emul_lib, n_envs=16, n_steps=70) > 0


def test_trained_reference_policies_traverse_our_terrain(emul_lib):
    print(ec.check_trained_policies_traverse(emul_lib))


def test_multi_step_launch(emul_lib):
    ec.check_multi_step_launch(emul_lib)


def test_terrain_physics_against_oracle(emul_lib):
    out = ec.check_terrain_physics_against_oracle(emul_lib, cap_ill=4)        # (3 of the 32 cases are ill-conditioned in the oracle under the cone; 1 under the pyramid)
    assert out['n_terrain'] >= 10


def test_pyramid_friction_variant(emul_lib):
    """LLM_SPEC_FRICTION_MODE = 0 (ll_epmc_set_spec_param: the pyramid of rounds 1 - 3) against the oracle under the same switch"""
    with ec.spec_variant(friction_mode=0):
        ec.check_terrain_physics_against_oracle(emul_lib)
        ec.check_multi_step_launch(emul_lib)


def test_round4_spec_variant(emul_lib):
    """The spec of rounds 1 - 4 (speculative limit rows with their gate, ERP 0.2 on every row, push-out capped at 0.5 m/s) as an A/B leg on terrain, engine
    against the oracle under the same switches; since round 5 the default is Bullet's limit rule, contact ERP 0.08, no cap (profiles/r05_limit_rows.md).
    Then the rigid-body solver's two-ERP rule (LLM_SPEC_ERP_DEEP), the other priced variant."""
    with ec.spec_variant(limit_speculative=1, erp=0.2, limit_erp=0.2, limit_erp_deep=-1, max_depen_speed=0.5):
        ec.check_terrain_physics_against_oracle(emul_lib, cap_ill=4)
        ec.check_multi_step_launch(emul_lib)
    with ec.spec_variant(erp=0.2, erp_deep=0.08):
        ec.check_terrain_physics_against_oracle(emul_lib, cap_ill=4)


def test_trunk_on_edges_against_oracle(emul_lib):
    out = ec.check_trunk_on_edges_against_oracle(emul_lib)
    print(out['n_edge_felt'], max(out['config']), max(out['vel']))


def test_legs_on_edges_against_oracle(emul_lib):
    out = ec.check_legs_on_edges_against_oracle(emul_lib)
    print('legs on hurdle edges: %d of 16 cases feel the leg edges; worst config %.2e, velocity %.2e' % (out['n_edge_felt'], max(out['config']), max(out['vel'])))


def test_free_running_against_the_oracle_env(emul_lib):
    print(ec.check_free_running_against_oracle_env(emul_lib))


def test_free_running_with_noise_and_without_edge_cylinders(emul_lib):
    noise = {'pos_x_bias': [-0.1, 0.1], 'pos_y_bias': [-0.1, 0.1], 'yaw_bias': [-0.2, 0.2], 'pos_z_bias': [-0.02, 0.02]}
    print(ec.check_free_running_against_oracle_env(emul_lib, n_steps=3, elements=(2, 1), aux=None, obs_rand=noise))


def test_parked_variant_equals_plain(emul_lib):
    ec.check_parked_variant_equals_plain(emul_lib)



def test_rays_by_a_kernel_of_their_own_equal_the_fused_rays(emul_lib):
    ec.check_split_rays_equal_fused(emul_lib, n=6, n_steps=24)
    ec.check_split_rays_equal_fused(emul_lib, n=4, n_steps=12, elements=(1,), noise=True)


def test_game_statistics_against_the_oracle_env(emul_lib):
    """the mechanism of the GPU test of the same name (recorded uniforms, both sides to the end of every episode) at a size the CPU build affords;
    the distribution bars proper are asserted on the GPU with 128 episodes per policy"""
    ec.check_game_statistics(emul_lib, n_per_policy=6, policies=('hurdle',), frac_tol=0.35, len_tol=0.5, ks_p=0.01, n_se=3.0)
