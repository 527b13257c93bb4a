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
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s', '-j2'])
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


def check_numpy_stream_reproduces_reference(lib_path):
    """`np.random.seed(s)` + create_playground_game(...).reset() builds the terrain, target, friction and command period the
    REFERENCE built from the same seed (the 24 cases of golden set T), and consumes exactly as many draws."""
    import numpy as np
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'epmc_golden.npz'))
    for k in range(len(g['t_element'])):
        element, seed = int(g['t_element'][k]), int(g['t_seed'][k])
        np.random.seed(1000 * element + seed)                                # what gen_epmc_golden.py seeded the reference with
        aux = None if np.isnan(g['t_aux'][k]) else float(g['t_aux'][k])
        env = lla.create_playground_game(lib_path=lib_path, **env_config(element, aux=aux))
        env.reset()
        rows, cnt = env._engine.statics()
        n = int(g['t_n_statics'][k])
        assert cnt[0] == n
        np.testing.assert_allclose(rows[0, :n], g['t_statics'][k][:n], rtol=1e-5, atol=2e-5)
        ep = env._engine.episode()
        np.testing.assert_allclose([ep['target_x'][0], ep['target_y'][0], ep['target_z'][0]], g['t_target'][k], atol=2e-5)
        assert abs(ep['friction'][0] - g['t_friction'][k]) < 1e-5 and int(ep['cmd_vary_freq'][0]) == int(g['t_cmd_freq'][k])
        np.testing.assert_allclose([ep['push_fx'][0], ep['push_fy'][0], ep['push_fz'][0]], g['t_push_force'][k], rtol=1e-5, atol=1e-4)
        follow = np.random.uniform()                                           # the stream must now stand where the reference left it
        np.random.seed(1000 * element + seed)
        for kind, a, b, v in g['t_draws'][k][:g['t_n_draws'][k]]:
            pass
        np.random.uniform(0.4, 3.0)                                            # the constructor's draw (not in the recorded log)
        for kind, a, b, v in g['t_draws'][k][:g['t_n_draws'][k]]:
            got = np.random.uniform(a, b) if kind == 0 else (np.random.randint(int(a), int(b)) if kind == 1 else np.random.rand())
            assert abs(got - v) < 1e-12
        assert abs(np.random.uniform() - follow) < 1e-15
        env.close()


def check_numpy_stream_through_steps(lib_path):
    """The same for the draws made while stepping (joystick targets, commanded speed, push forces): after reset() and as many
    step() calls as the reference's golden episodes ran, NumPy's global stream stands exactly where the reference left it."""
    import numpy as np
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'epmc_golden.npz'))
    for e in range(len(g['e_element'])):
        noise_on = not np.isnan(g['e_noise'][e][0])
        obs_rand = {'pos_x_bias': [-0.1, 0.1], 'pos_y_bias': [-0.1, 0.1], 'yaw_bias': [-0.2, 0.2], 'pos_z_bias': [-0.02, 0.02]} if noise_on else None
        cmd = {0: (7, 8), 1: (25, 200)}.get(e, (9999, 10000))
        np.random.seed(5000 + e)
        env = lla.create_playground_game(lib_path=lib_path, **env_config(int(g['e_element'][e]), obs_rand=obs_rand, cmd_range=cmd))
        env.reset()
        for t in range(int(g['e_n'][e])):
            env.step([np.zeros(12)])
        ep = env._engine.episode()
        np.testing.assert_allclose([ep['target_x'][0], ep['target_y'][0]], g['e_target'][e][int(g['e_n'][e]) - 1][:2], rtol=1e-4, atol=1e-2) \
            if int(g['e_element'][e]) != 0 else None                          # (a joystick target is relative to where the robot stands)
        assert abs(ep['target_spd'][0] - g['e_target_spd'][e][int(g['e_n'][e]) - 1]) < 1e-5
        follow = np.random.uniform()
        np.random.seed(5000 + e)
        np.random.uniform(0.4, 3.0)
        for kind, a, b, v in g['e_draws'][e][:g['e_n_draws'][e]]:
            got = np.random.uniform(a, b) if kind == 0 else (np.random.randint(int(a), int(b)) if kind == 1 else np.random.rand())
            assert abs(got - v) < 1e-12
        assert abs(np.random.uniform() - follow) < 1e-15, e
        env.close()


def test_numpy_stream_through_steps(emul_lib):
    check_numpy_stream_through_steps(emul_lib)


def test_numpy_stream_reproduces_reference(emul_lib):
    check_numpy_stream_reproduces_reference(emul_lib)
