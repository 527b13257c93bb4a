"""The reference's EPMC plug-in surface (create_playground_game / create_playground_env, create_pybullet_envs.py:67-101, :150-154)
over the engine.  CPU tests run the host build of the kernel source; test_gpu_epmc.py repeats the contract check on the GPU."""
import os
import re
import subprocess
from collections import OrderedDict

import numpy as np
import pytest

import lifelike_agility_and_play_amd as lla
from conftest import ROOT
from lifelike_agility_and_play_amd import epmc_capi
from test_epmc_oracle_golden import env_config

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s'])
    return EMUL_LIB


def check_single_env_contract(lib_path):
    cfg = env_config(1)
    cfg['max_steps'] = 5
    env = lla.create_playground_game(lib_path=lib_path, **cfg)
    sp = env.observation_space.spaces[0].spaces                               # PGE:128-139
    assert list(sp.keys()) == ['prop', 'prop_a', 'percep_2d', 'percep_1d', 'percep_front', 'target']
    assert [sp[k].shape for k in sp] == [(99,), (36,), (25, 13), (128,), (25, 13), (3,)]
    act = env.action_space.spaces[0].spaces                                    # PGE:141-145
    assert act['A_Z'].n == 256 and act['A_LLC'].shape == (12,)
    out = env.reset(inter_kwargs={'x': 1})
    assert isinstance(out, tuple) and len(out) == 1 and isinstance(out[0], OrderedDict)
    o = out[0]
    assert [o[k].shape for k in o] == [(99,), (36,), (25, 13), (128,), (25, 13), (3,)]
    assert (o['prop_a'] == 0).all() and np.allclose(o['prop'][:33], o['prop'][66:])           # PGE:277-284 pre-filled history
    done, n = False, 0
    while not done:
        a = {'A_Z': 3, 'A_LLC': np.zeros(12)} if n % 2 else np.zeros(12)                      # PGE:322 accepts both
        (o,), (r,), done, info = env.step([a])
        assert isinstance(r, float) and isinstance(done, bool)
        n += 1
        assert (info == {}) == (not done)
    assert n <= 5 and sorted(info) == sorted(['ave_spd', 'max_spd', 'reward_vel', 'reward_rotation', 'reward_dist', 'reward_avg_spd'])   # PGE:356-362
    env.close()
    e2 = lla.create_playground_env(lib_path=lib_path, **env_config(0))
    assert list(e2.observation_space.spaces.keys())[0] == 'prop' and e2.action_space.spaces['A_LLC'].shape == (12,)   # CPE:150-154
    e2.close()
    b = lla.create_playground_game(lib_path=lib_path, num_envs=6, seed=4, **env_config(3))
    ob = b.reset()
    assert ob.shape == (6, 916)
    ob, r, d, info = b.step(np.zeros((6, 12), np.float32))
    assert ob.shape == (6, 916) and r.shape == (6,) and d.dtype == bool and info['episode_info'].shape == (6, 6)
    parts = b.split(ob)
    assert parts['percep_2d'].shape == (6, 25, 13) and parts['target'].shape == (6, 3)
    b.close()


def test_single_env_contract(emul_lib):
    check_single_env_contract(emul_lib)


def test_factory_errors(emul_lib):
    with pytest.raises(AssertionError):
        lla.create_playground_game(lib_path=emul_lib, **dict(env_config(1), arena_id='Nope'))   # CPE:69-71
    c = env_config(1); del c['arena_id']
    with pytest.raises(KeyError):
        lla.create_playground_game(lib_path=emul_lib, **c)                                       # CPE:68
    with pytest.raises(TypeError):
        lla.create_playground_game(lib_path=emul_lib, **dict(env_config(1), prop_type='joint_pos'))   # PGE:122
    bad = env_config(1); bad['env_randomize_config'] = dict(bad['env_randomize_config'], element_id=7)
    with pytest.raises(epmc_capi.capi.LLError):
        lla.create_playground_game(lib_path=emul_lib, **bad)                                     # BSE:249-250 'Unknown element id.'


def test_push_counts_use_python_floor_division():
    """PR:45-53 evaluates `x // time_step` on floats: 0.2 // 0.002 is 100, 1.0 // 0.002 is 499 (not 500), -0.5 // 0.002 is -250 (0.5 // 0.002 is 249)."""
    cfg = epmc_capi.make_epmc_config(1, env_config(1))
    assert (cfg.push_count0, cfg.push_interval_step, cfg.push_duration_step) == (-250, 499, 100)


def test_header_binding_and_library_agree():
    text = open(os.path.join(ROOT, 'include', 'llenv_epmc.h')).read()
    declared = sorted(set(re.findall(r'\b(ll_epmc_[a-z0-9_]+)\s*\(', text)))
    assert declared == epmc_capi.EXPORTED_SYMBOLS
    import __graft_entry__ as g
    g.build_hip()
    lib = epmc_capi.load_library()                # resolves every ll_epmc_* name in the HIP build or raises
    for name in declared:
        assert hasattr(lib, name), name
