"""The EPMC oracle (oracle/epmc_oracle.py, NumPy float64) against tests/golden/epmc_golden.npz, which the reference's own
PlayGroundEnv produced here through a fake BulletClient (tests/golden/gen_epmc_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import epmc_oracle as eo  # noqa: E402

TOL = 1e-9


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'epmc_golden.npz'))


def env_config(element_id, aux=0.02, obs_rand=None, cmd_range=(25, 200)):      # the dict gen_epmc_golden.py passed to the reference
    return {
        'arena_id': 'Playground', 'render': False, 'control_freq': 50.0,
        'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
        'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': obs_rand or {},
        'env_randomize_config': {
            'element_id': element_id, 'height_range': [0.0, 0.0], 'friction_range': [0.4, 3.0],
            'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2,
                                     'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
            'cmd_vary_freq_range': list(cmd_range), 'target_spd_range': [0.5, 3.0], 'auxiliary_radius': aux,
            'hole_config': {'min_gap_height': 0.25, 'max_gap_height': 0.25},
        },
    }


def scripted_rays(call, n):          # the fake client's answers (same arithmetic as gen_epmc_golden.scripted_rays)
    i = np.arange(n)
    return ((i * 7 + call * 13) % 10) < 7, (((i * 37 + call * 101) % 1009) + 0.5) / 1009.0


class ScriptedRays(object):
    def __init__(self, call0=0):
        self.call = call0

    def __call__(self, f, t):
        hits, fracs = [], []
        for n in (eo.N_HEIGHT, eo.N_HORIZ, eo.N_FRONT):
            h, fr = scripted_rays(self.call, n)
            hits.append(h); fracs.append(fr)
            self.call += 1
        return np.concatenate(hits), np.concatenate(fracs)


def percep_checks(obs):
    out = []
    for a, b in ((135, 460), (460, 588), (588, 913)):
        v = obs[a:b]
        i = np.arange(len(v))
        out += [v.sum(), (v * (i + 1)).sum() / len(v), (v * np.where(i % 2 == 0, 1.0, -1.0)).sum()]
    return np.array(out)


def make_env(g, cfg, prev_orn):
    init = g['init_states_info'].copy()
    init[3:7] = prev_orn
    return eo.EpmcOracleEnv(cfg, init)


def test_terrain_generation_and_reset(g):
    """BSE reset() for element ids 0..3 (24 seeded cases, with and without auxiliary edge cylinders): same bodies in the same
    order, same target, friction, command period, start pose, push force and first observation, from the same draws."""
    for k in range(len(g['t_element'])):
        aux = None if np.isnan(g['t_aux'][k]) else float(g['t_aux'][k])
        env = make_env(g, env_config(int(g['t_element'][k]), aux=aux), g['t_prev_orn'][k])
        draws = eo.LoggedDraws(g['t_draws'][k][:g['t_n_draws'][k]])
        obs = env.reset(draws, ScriptedRays())
        assert draws.exhausted()
        n = g['t_n_statics'][k]
        assert len(env.statics) == n
        np.testing.assert_allclose(env.statics, g['t_statics'][k][:n], rtol=0, atol=1e-12)
        np.testing.assert_allclose(env.target_pos, g['t_target'][k], atol=1e-12)
        assert abs(env.foot_friction - g['t_friction'][k]) < 1e-12 and env.cmd_vary_freq == g['t_cmd_freq'][k]
        np.testing.assert_allclose(env.state, g['t_init_state'][k], atol=1e-12)
        np.testing.assert_allclose(env.push.force, g['t_push_force'][k], atol=1e-12)
        np.testing.assert_allclose(obs, g['t_reset_obs'][k], rtol=TOL, atol=TOL)


def test_scripted_episodes(g):
    """8 episodes through reset()/step() with scripted physics and ray answers: ray end points, observations, the joystick
    and average-speed rewards, termination by reach / fall, target updates, push schedule, end-of-episode info."""
    n_done = 0
    for e in range(len(g['e_element'])):
        noise_on = not np.isnan(g['e_noise'][e][0])
        obs_rand = {'pos_x_bias': [-0.1, 0.1], 'pos_y_bias': [-0.1, 0.1], 'yaw_bias': [-0.2, 0.2], 'pos_z_bias': [-0.02, 0.02]} if noise_on else None
        cmd = {0: (7, 8), 1: (25, 200)}.get(e, (9999, 10000))
        env = make_env(g, env_config(int(g['e_element'][e]), obs_rand=obs_rand, cmd_range=cmd), g['e_prev_orn'][e])
        draws = eo.LoggedDraws(g['e_draws'][e][:g['e_n_draws'][e]])
        rays = ScriptedRays(int(g['e_ray_call0'][e]))
        obs0 = env.reset(draws, rays)
        np.testing.assert_allclose(obs0, g['e_reset_obs'][e], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(env.state, g['e_init_state'][e], atol=1e-12)
        np.testing.assert_allclose(np.stack(env.last_rays), np.stack([g['e_ray_from'][e][0], g['e_ray_to'][e][0]]), rtol=0, atol=1e-10)
        if noise_on:
            np.testing.assert_allclose([env.noise[k] for k in ('pos_x_bias', 'pos_y_bias', 'yaw_bias', 'pos_z_bias')], g['e_noise'][e], atol=1e-12)
        n = int(g['e_n'][e])
        for t in range(n):
            state = g['e_state'][e][t]
            obs, r, d, info = env.step(g['e_action'][e][t], draws, lambda k, tgt, f: state if k == 9 else None, rays)
            if t < g['e_obs_full'].shape[1]:
                np.testing.assert_allclose(obs, g['e_obs_full'][e][t], rtol=TOL, atol=TOL)
                np.testing.assert_allclose(np.stack(env.last_rays), np.stack([g['e_ray_from'][e][t + 1], g['e_ray_to'][e][t + 1]]), rtol=0, atol=1e-10)
            np.testing.assert_allclose(np.concatenate([obs[:135], obs[913:]]), g['e_obs_core'][e][t], rtol=TOL, atol=TOL)
            np.testing.assert_allclose(percep_checks(obs), g['e_obs_checks'][e][t], rtol=1e-9, atol=1e-7)
            assert abs(r - g['e_reward'][e][t]) < 1e-12 and d == bool(g['e_done'][e][t]), (e, t)
            np.testing.assert_allclose(env.target_pos, g['e_target'][e][t], atol=1e-9)
            assert abs(env.target_spd - g['e_target_spd'][e][t]) < 1e-12
            for k in range(10):                                              # PR:56-86: which substeps are pushed, and how hard
                on = env.applied[k] is not None
                assert on == bool(g['e_force_on'][e][t][k]), (e, t, k)
                if on:
                    np.testing.assert_allclose(env.applied[k], g['e_force'][e][t][k], atol=1e-12)
            if d:
                n_done += 1
                np.testing.assert_allclose([info[k] for k in ('ave_spd', 'max_spd', 'reward_vel', 'reward_rotation', 'reward_dist', 'reward_avg_spd')],
                                           g['e_info'][e], rtol=1e-10, atol=1e-12)
        assert draws.exhausted(), e
    assert n_done == 3


def test_ray_casting_spec():
    """cast_rays (this build's rayTestBatch spec): plane, box faces, misses, origin inside a box, parallel rays."""
    statics = np.array([[0, 2.0, 0.0, 0.5, 0.5, 1.0, 0.5, 0], [1, 1.0, 0, 0.3, 0.02, 2.0, 0, 0], [0, 8.0, 0, 0, 0, 0, 0, 0]], dtype=np.float64)
    f = np.array([[0, 0, 0.3], [0, 0, 0.3], [2.0, 0.2, 10.0], [5.0, 0, 10.0], [2.0, 0.0, 0.5], [0, 3.0, 0.3], [0, 0, 0.3]], dtype=np.float64)
    t = np.array([[20, 0, 0.3], [-20, 0, 0.3], [2.0, 0.2, -10.0], [5.0, 0, -10.0], [9.0, 0.0, 0.5], [20, 3.0, 0.3], [4.0, 0, -0.5]], dtype=np.float64)
    hit, frac = eo.cast_rays(f, t, statics)
    assert list(hit) == [True, False, True, True, False, False, True]
    np.testing.assert_allclose(frac[[0, 2, 3, 6]], [1.5 / 20, 9.0 / 20, 0.5, 1.5 / 4.0], atol=1e-12)   # box x face, box top, plane, box before plane
