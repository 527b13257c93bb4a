"""EPMC on the MI355X: the HIP library (through the C ABI of include/llenv_epmc.h) against the reference goldens and the oracle."""
import pytest

import epmc_parity_common as ec

pytestmark = pytest.mark.gpu


def test_terrain_and_reset_against_reference_goldens():
    ec.check_terrain_and_reset_against_goldens(None)


def test_scripted_episodes_against_reference_goldens():
    ec.check_scripted_episodes_against_goldens(None)


def test_ray_casting_against_oracle():
    assert ec.check_ray_casting_against_oracle(None) > 100


def test_free_running_invariants():
    assert ec.check_free_running_invariants(None, n_envs=300, n_steps=80) > 0
    assert ec.check_free_running_invariants(None, n_envs=4200, n_steps=12, element=3) >= 0


def test_env_api_contract_gpu():
    from test_epmc_env_api import check_single_env_contract
    check_single_env_contract(None)


def test_terrain_physics_against_oracle():
    out = ec.check_terrain_physics_against_oracle(None, n_envs=48)
    assert out['n_terrain'] >= 30 and out['n_felt'] >= 20


def test_trunk_on_edges_against_oracle():
    out = ec.check_trunk_on_edges_against_oracle(None, n_envs=48)
    assert out['n_edge_felt'] >= 24


def test_free_running_against_the_oracle_env_gpu():
    print(ec.check_free_running_against_oracle_env(None))
