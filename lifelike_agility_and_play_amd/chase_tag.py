"""Host-side mirror of the reference's SEPMC factory (create_pybullet_envs.py:104-140, :157-161) over the HIP engine.

``create_chase_tag_game(**env_config)`` takes the reference's env_config (same keys, same defaults, same exceptions) and returns an
object with ChaseTagGameEnv's contract (CTG:105-135, :263, :378):
    observation_space = Tuple([Dict{prop, prop_a, percept_2d (25,13), percept_1d (128,), percept_front (25,13), percept_vec (5,),
                                    oppo_info (15,), oppo_info_cheat (15,), flag_info (7,), flag_info_cheat (7,), with_flag (2,),
                                    control_spd (1,)}] * 2)
    action_space      = Tuple([Dict{A_HLC: Box(1), A_Z: Discrete(256), A_LLC: Box(12)}] * 2)
    reset(**kw) -> [obs0, obs1]          step([a0, a1]) -> ([obs0, obs1], [r0, r1], done, info)        a = {'A_LLC': 12 floats}
Extra keys switch to the batched engine: ``num_envs`` (arenas), ``device``, ``seed``, ``auto_reset``, ``lib_path``.

Randomness: as playground.py -- the 1-arena game makes the reference's np.random calls in the reference's order (the
constructor's friction draw CTG:61, reset CTG:263-299, the two-robot push schedule PR:78-86, a flag move CTG:231-236) and hands
the values to the engine; the draws of a flag move that did not happen are given back (the generator state is restored).
"""
import warnings
from collections import OrderedDict

import numpy as np

from . import epmc_capi, sepmc_capi, urdf_model
from .spaces import Box, Dict, Discrete, Tuple

ENGINE_KEYS = ('num_envs', 'device', 'seed', 'auto_reset', 'lib_path', 'urdf_path')
OBS_PARTS = (('percept_2d', 325, (25, 13)), ('percept_1d', 128, (128,)), ('percept_front', 325, (25, 13)), ('percept_vec', 5, (5,)), ('oppo_info', 15, (15,)),
             ('oppo_info_cheat', 15, (15,)), ('flag_info', 7, (7,)), ('flag_info_cheat', 7, (7,)), ('with_flag', 2, (2,)), ('control_spd', 1, (1,)))


def _spaces(prop_type):
    if not isinstance(prop_type, list):
        raise TypeError("Expected 'prop_type' to be a list.")                  # CTG:104
    full = {'joint_pos': 12, 'joint_vel': 12, 'root_lin_vel_loc': 3, 'root_ang_vel_loc': 3, 'e_g': 3}
    prop = sum(full[e] for e in prop_type) * 3                                 # CTG:90-105, stack_frame_num 3
    obs = Dict(OrderedDict([('prop', Box(0, 0, shape=(prop,))), ('prop_a', Box(0, 0, shape=(36,)))] + [(k, Box(0, 0, shape=shape)) for k, _, shape in OBS_PARTS]))
    act = Dict(OrderedDict([('A_HLC', Box(0, 0, shape=(1,))), ('A_Z', Discrete(256)), ('A_LLC', Box(0, 0, shape=(12,)))]))      # CTG:128-135
    return obs, act, prop


def _split(row, prop):
    out = OrderedDict([('prop', row[..., :prop]), ('prop_a', row[..., prop:prop + 36])])
    a = prop + 36
    for k, n, shape in OBS_PARTS:
        out[k] = row[..., a:a + n].reshape(row.shape[:-1] + shape)
        a += n
    return out


def _build_engine(env_config, num_arenas, auto_reset):
    arena_id = env_config['arena_id']                                          # KeyError like CPE:105
    assert arena_id in ['CTG']                                                 # CPE:106-108
    if env_config.get('render', False):
        warnings.warn('render is ignored: the batched engine has no GUI')
    cfg = sepmc_capi.make_sepmc_config(num_arenas, env_config, auto_reset=auto_reset, seed=env_config.get('seed', 0), device=env_config.get('device', 0))
    urdf_path = env_config.get('urdf_path', None)
    blob = urdf_model.UrdfModel(urdf_path).blob() if urdf_path else urdf_model.default_model_blob()
    return sepmc_capi.SepmcEngine(cfg, blob, lib_path=env_config.get('lib_path', None))


def _llc(action):
    a = action['A_LLC'] if isinstance(action, dict) and 'A_LLC' in action else action      # CTG:379
    return np.asarray(a, dtype=np.float32)


class _ReferenceDraws(object):
    """The np.random calls of ChaseTagGameEnv / BulletStaticsV4 / PushRandomizer, in their order, returned as the uniforms in
    [0, 1) that make the engine reproduce the drawn values (ll_sepmc_reset h_draws, ll_sepmc_set_step_draws)."""

    def __init__(self, env_config):
        self.rc = env_config.get('env_randomize_config', {})
        self.el = env_config.get('element_config', {}) or {}
        self.obs_rand = env_config.get('obs_randomization') or {}
        self.push = self.rc.get('disturb_force_config')
        self.n_sub = int((1.0 / env_config.get('control_freq', 25.0)) / epmc_capi.TIME_STEP)
        self.u = []
        np.random.uniform(*self.rc['friction_range'])                          # CTG:61: the constructor's own friction draw
        self.max_tau = env_config.get('max_tau', 18.0)
        vals = [float(np.random.uniform(*self.max_tau)) if isinstance(self.max_tau, list) else self.max_tau for _ in range(2)]   # LR:244, one per robot
        self.max_tau_value, self.max_tau_robot1 = vals[0], vals[1]             # each robot is clipped with its own draw

    def _uniform(self, a, b):
        v = np.random.uniform(a, b)
        self.u.append((v - a) / (b - a) if b > a else 0.0)
        return v

    def _randint(self, a, b):
        v = np.random.randint(a, b)
        self.u.append((v - a + 0.5) / (b - a))
        return v

    def _force(self):                                                          # PR:88-98
        self._uniform(0, 2 * np.pi)
        self._uniform(*self.push['horizontal_force'])
        self._uniform(*self.push['vertical_force'])

    def reset(self):
        self.u = []
        self._uniform(0.5, 3.0)                                                # CTG:264
        if self.el.get('rand_cube'):                                           # BS4:904-945
            n = self._randint(5, 6)
            for _ in range(n):
                self._uniform(0.05, 0.25); self._uniform(-2.0, 2.0); self._uniform(-2.0, 2.0); self._uniform(0.5, 1.0); self._uniform(0.5, 1.0)
        if self.el.get('hurdle'):
            self._uniform(0.05, 0.15)                                          # BS4:954
        if self.el.get('hole'):
            self._uniform(0.25, 0.3)                                           # BS4:1003
        self._randint(0, 2)                                                    # CTG:268
        self._uniform(*self.rc['friction_range'])                              # CTG:279
        if self.push is not None:                                              # CTG:284-285
            self._force()
            self.count = -self.push.get('start_time', 0.) // epmc_capi.TIME_STEP
            self.interval = self.push.get('interval_time', 5.) // epmc_capi.TIME_STEP
            self.duration = self.push.get('duration_time', 0.5) // epmc_capi.TIME_STEP
        if isinstance(self.max_tau, list):                                     # CTG:285-287: one draw per robot into an attribute the torque
            np.random.uniform(*self.max_tau); np.random.uniform(*self.max_tau)  # clip never reads; the global stream moves on
        drawn = {}
        for k in self.obs_rand:                                                # CTG:207-210, in the dict's own order ...
            n0 = len(self.u)
            self._uniform(*self.obs_rand[k])
            drawn[k] = self.u.pop(n0)
        self.u += [drawn[k] for k in epmc_capi.NOISE_KEYS if k in drawn]       # ... handed over in the engine's fixed key order
        for _ in range(4):
            self._uniform(-2.0, 2.0)                                           # CTG:213-214
        for _ in range(2):
            self.u.append(float(np.random.rand()))                             # CTG:219
        self._uniform(-2.0, 2.0); self._uniform(-2.0, 2.0)                     # CTG:233
        out = np.full(sepmc_capi.LLS_MAX_DRAWS, 0.5, np.float32)
        out[:len(self.u)] = self.u
        return out

    def step(self):
        """The draws a step certainly makes (pushes), then -- provisionally -- the two of a flag move; returns (uniforms, state to
        restore if the flag did not move)."""
        self.u = []
        if self.push is not None:
            for _ in range(self.n_sub):                                        # PR:56-86
                self.count += 1
                if self.count > 0:
                    if self.count % self.interval == 0:
                        self._force()
                        self.count = 0
                    if self.count < self.duration:
                        self._force(); self._force()                           # PR:84-85: redrawn after each of the two robots
        saved = np.random.get_state()
        self._uniform(-2.0, 2.0); self._uniform(-2.0, 2.0)                     # CTG:233 (only if the flag changes hands)
        return np.array(self.u, dtype=np.float32), saved


class ChaseTagGame(object):
    """ChaseTagGameEnv: one arena, two robots, reference semantics (no auto-reset)."""

    def __init__(self, env_config):
        self._draws = _ReferenceDraws(env_config)                                 # the constructor's draws, in the reference's order
        self._engine = _build_engine(dict(env_config, max_tau=self._draws.max_tau_value, max_tau_robot1=self._draws.max_tau_robot1), 1, auto_reset=0)
        obs, act, self._prop = _spaces(env_config['prop_type'])
        self.n_max = 2
        self.observation_space, self.action_space = Tuple([obs] * 2), Tuple([act] * 2)    # CTG:125, :128-135

    def _obs(self):
        o = self._engine.obs()[0].astype(np.float64)
        return [_split(o[0], self._prop), _split(o[1], self._prop)]

    def reset(self, **kwargs):
        self._engine.reset(draws=self._draws.reset()[None])
        return self._obs()

    def step(self, rl_actions):
        u, saved = self._draws.step()
        self._engine.set_step_draws(u[None])
        self._engine.step_host(np.array([_llc(rl_actions[0]), _llc(rl_actions[1])]).reshape(1, 2, 12))
        r, d, _ = self._engine.reward_done()
        if self._engine.episode()['switch'][0] < 0.5:
            np.random.set_state(saved)                                          # the flag stayed: its two draws were never made
        v = self._engine.info()[0]
        info = {'avg_spd0': float(v[0]), 'avg_spd1': float(v[1]), 'max_spd0': float(v[2]), 'max_spd1': float(v[3])}     # CTG:404-409
        return self._obs(), [float(r[0][0]), float(r[0][1])], bool(d[0]), info

    def close(self):
        self._engine.close()


class BatchedChaseTagEnv(object):
    """num_envs arenas in lockstep on one GPU: arrays [arena][robot][...] in and out; finished arenas are re-seeded inside the
    step kernel when ``auto_reset`` (default)."""

    def __init__(self, env_config):
        self.num_envs = int(env_config['num_envs'])
        self.auto_reset = bool(env_config.get('auto_reset', True))
        self.engine = _build_engine(env_config, self.num_envs, int(self.auto_reset))
        obs, act, self.prop_size = _spaces(env_config['prop_type'])
        self.single_observation_space, self.single_action_space = obs, act
        self.observation_space, self.action_space = Tuple([obs] * 2), Tuple([act] * 2)
        self.obs_dim = self.engine.obs_dim

    def reset(self, arena_ids=None):
        self.engine.reset(arena_ids)
        return self.engine.obs()

    def step(self, actions):
        self.engine.step_host(_llc(actions))
        r, d, why = self.engine.reward_done()
        return self.engine.obs(), r, d, {'done_reason': why, 'speeds': self.engine.info()}

    def split(self, obs):
        return _split(obs, self.prop_size)

    def close(self):
        self.engine.close()


class _Untupled(object):
    def __init__(self, game):
        self._game = game
        self.observation_space = game.observation_space.spaces[0]              # CPE:157-161
        self.action_space = game.action_space.spaces[0]

    def __getattr__(self, name):
        return getattr(self._game, name)


def create_chase_tag_game(**env_config):
    unknown = [k for k in env_config if k not in ('arena_id', 'render', 'control_freq', 'kp', 'kd', 'max_tau', 'prop_type', 'max_steps', 'obs_randomization',
                                                  'env_randomize_config', 'element_config') + ENGINE_KEYS]
    if unknown:
        warnings.warn('create_chase_tag_game: unused keys %s' % unknown)
    if int(env_config.get('num_envs', 1)) > 1:
        return BatchedChaseTagEnv(env_config)
    return ChaseTagGame(env_config)


def create_chase_tag_env(**env_config):
    return _Untupled(create_chase_tag_game(**env_config))
