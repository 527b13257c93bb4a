import os
import sys

import numpy as np
import pytest

if os.environ.get('LL_TEST_LIB'):
    # run the suite against ANOTHER build of the HIP library (an A/B leg under tools/_build/, e.g. -DLL_MFMA_GRAM=1): test infrastructure only
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lifelike_agility_and_play_amd import capi as _capi
    _capi.DEFAULT_LIB = os.path.abspath(os.environ['LL_TEST_LIB'])

try:                # torch bundles its own copy of the HIP runtime under the same SONAME as the system one libllenv.so links: whichever is
    import torch    # noqa: F401  loaded first serves the process.  Load torch's first, the order bench.py has (and every full-suite run had)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
POLICY_WEIGHTS = os.path.join(ROOT, 'lifelike_agility_and_play_amd', 'assets', 'pmc_policy.npz')   # the reference's trained PMC policy (tools/extract_policy.py)

# test_primitive_level_env.py:18-38 / example_pmc_train.sh:67-79
PMC_REWARD_WEIGHTS = {'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05}
PMC_PROP_TYPE = ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(GOLDEN_DIR, 'pmc_golden.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def model_blob():
    from lifelike_agility_and_play_amd import urdf_model
    return urdf_model.default_model_blob()


@pytest.fixture(scope='session')
def mocap_table():
    """The shipped float64 clip table (the same numbers the reference parses from JSON)."""
    from lifelike_agility_and_play_amd import mocap
    return mocap.load_mocap('', 1.0 / 50.0)


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def make_oracle_batch(orc, model_blob, table, n_envs=1, **kw):
    kw.setdefault('reward_weights', PMC_REWARD_WEIGHTS)
    kw.setdefault('prop_type', PMC_PROP_TYPE)
    cfg = orc.make_config(n_envs=n_envs, **kw)
    return orc.OracleBatch(cfg, model_blob, table)
