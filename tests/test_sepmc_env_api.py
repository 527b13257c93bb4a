"""The reference's SEPMC plug-in surface (create_chase_tag_game / create_chase_tag_env, create_pybullet_envs.py:104-161) over the
CPU build of the kernel source: spaces, return shapes, and -- for the 15 golden reset cases -- that `np.random.seed(s)` rebuilds the
reference's arena, roles, flag, friction, push and start poses and leaves NumPy's global stream where the reference leaves it."""
import os
import subprocess

import numpy as np
import pytest

import lifelike_agility_and_play_amd as lla
from lifelike_agility_and_play_amd import chase_tag
import sepmc_parity_common as SC

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, '_build', 'libllenv_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    subprocess.check_call(['make', '-C', EMUL_DIR, '-s', '-j2'])
    return EMUL_LIB


def test_contract_and_seeded_resets(emul_lib):
    g = SC.load_golden()
    for k in range(len(g['t_seed'])):
        es, seed_i = k // 3, k % 3
        seed = 7000 + 10 * es + seed_i                       # gen_sepmc_golden.py
        noisy = bool(g['t_noise_on'][k])
        cfg = SC.env_config(g['t_elements'][k], noisy)
        np.random.seed(seed)
        env = chase_tag.create_chase_tag_game(lib_path=emul_lib, **cfg)
        if k == 0:
            assert len(env.observation_space.spaces) == 2 and env.observation_space.spaces[0].spaces['percept_2d'].shape == (25, 13)
            assert env.action_space.spaces[1].spaces['A_LLC'].shape == (12,) and env.action_space.spaces[0].spaces['A_Z'].n == 256
        # the reference shares ONE start-orientation dict between all envs of the process (CTG:224-229); a fresh engine starts from the pristine one
        if not np.allclose(g['t_prev_orn'][k], g['init_states_info'][3:7]):
            env._engine.reset(draws=env._draws.reset()[None], prev_orn=g['t_prev_orn'][k][None])
            obs = env._obs()
        else:
            obs = env.reset()
        # stream position: replay the constructor draw + the logged reset draws and compare the next number
        mine = np.random.rand()
        np.random.seed(seed)
        np.random.uniform(0.4, 3.0)
        for kind, a, b, v in g['t_draws'][k][:g['t_n_draws'][k]]:
            x = np.random.uniform(a, b) if int(kind) == 0 else (np.random.randint(int(a), int(b)) if int(kind) == 1 else np.random.rand())
            assert abs(x - v) < 1e-12
        assert mine == np.random.rand()
        assert isinstance(obs, list) and len(obs) == 2 and list(obs[0].keys())[:3] == ['prop', 'prop_a', 'percept_2d']
        E = env._engine
        rows, n = E.boxes()
        np.testing.assert_allclose(rows[0][:n[0]], g['t_boxes'][k][:g['t_n_boxes'][k]][:, 1:7], atol=1e-6)
        ep = E.episode()
        np.testing.assert_allclose([ep['flag_x'][0], ep['flag_y'][0], ep['flag_z'][0]], g['t_flag'][k], atol=1e-6)
        assert bool(ep['with_flag0'][0] > 0.5) == bool(g['t_with_flag'][k]) and abs(ep['friction'][0] - g['t_friction'][k]) < 1e-6
        np.testing.assert_allclose(E.state()[0], g['t_state'][k], atol=2e-6)
        for r in range(2):
            want_o = g['t_obs'][k][r]
            np.testing.assert_allclose(obs[r]['prop'], want_o[:99], atol=SC.OBS_TOL)
            np.testing.assert_allclose(obs[r]['percept_vec'], want_o[913:918], atol=SC.OBS_TOL)
            np.testing.assert_allclose(obs[r]['oppo_info_cheat'][1:], want_o[913 + 21:913 + 35], atol=SC.OBS_TOL)     # ([0] is the visibility, from real rays here)
            np.testing.assert_allclose(obs[r]['flag_info'], want_o[913 + 35:913 + 42], atol=SC.OBS_TOL)
            np.testing.assert_allclose(obs[r]['with_flag'], want_o[913 + 49:913 + 51], atol=0)
            np.testing.assert_allclose(obs[r]['control_spd'], want_o[913 + 51:], atol=1e-6)
        # a few steps: lists of two, a bool, the info keys; no draws are made before the push schedule starts unless the flag moves
        before = np.random.get_state()
        for t in range(3):
            o, rew, done, info = env.step([{'A_LLC': np.zeros(12)}, {'A_LLC': np.zeros(12)}])
            assert len(o) == 2 and len(rew) == 2 and isinstance(done, bool) and sorted(info) == ['avg_spd0', 'avg_spd1', 'max_spd0', 'max_spd1']
            if E.episode()['switch'][0] > 0.5:
                before = None
        if before is not None:
            after = np.random.get_state()
            assert (before[1] == after[1]).all() and before[2] == after[2]
        env.close()


def test_untupled_and_batched(emul_lib):
    cfg = SC.env_config((1, 0, 1))
    env = chase_tag.create_chase_tag_env(lib_path=emul_lib, **cfg)                       # CPE:157-161
    assert env.observation_space.spaces['oppo_info'].shape == (15,) and env.action_space.spaces['A_HLC'].shape == (1,)
    env.close()
    b = lla.create_chase_tag_game(lib_path=emul_lib, num_envs=3, seed=4, **cfg)
    obs = b.reset()
    assert obs.shape == (3, 2, 965)
    obs, rew, done, extra = b.step(np.zeros((3, 2, 12)))
    assert rew.shape == (3, 2) and done.shape == (3,) and extra['speeds'].shape == (3, 4)
    d = b.split(obs)
    assert d['percept_front'].shape == (3, 2, 25, 13) and d['with_flag'].shape == (3, 2, 2)
    b.close()
    with pytest.raises(TypeError):
        chase_tag.create_chase_tag_game(lib_path=emul_lib, **dict(cfg, prop_type='joint_pos'))
    with pytest.raises(AssertionError):
        chase_tag.create_chase_tag_game(lib_path=emul_lib, **dict(cfg, arena_id='Playground'))
