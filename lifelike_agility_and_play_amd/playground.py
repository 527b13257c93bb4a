"""Host-side mirror of the reference's EPMC factory (create_pybullet_envs.py:67-101, :150-154) over the HIP engine.

``create_playground_game(**env_config)`` takes the reference's env_config (same keys, same defaults, same exceptions) and
returns an object with PlayGroundEnv's contract behind SingleAgentWrapper (CPE:6-18):
    observation_space = Tuple([Dict{prop, prop_a, percep_2d (25,13), percep_1d (128,), percep_front (25,13), target (3,)}])
    action_space      = Tuple([Dict{A_Z: Discrete(256), A_LLC: Box(12)}])
    reset(**kw) -> (OrderedDict,)        step([a]) -> ((obs,), (reward,), done, info)       a = {'A_LLC': 12 floats} or 12 floats
Extra keys switch to the batched engine: ``num_envs``, ``device``, ``seed``, ``auto_reset``, ``lib_path``.

Randomness: the reference draws terrain, friction, pushes and joystick commands from NumPy's global MT19937 stream.  The 1-env
game does the same -- `_ReferenceDraws` makes the reference's np.random calls in the reference's order (including the friction
drawn by the constructor, PGE:92) and hands the values to the engine -- so `np.random.seed(s)` reproduces the reference's
terrain, targets and pushes.  The batched env draws from the engine's Philox stream (keyed on seed, env, episode) in the same order.
"""
import warnings
from collections import OrderedDict

import numpy as np

from . import epmc_capi, urdf_model
from .spaces import Box, Dict, Discrete, Tuple

ENGINE_KEYS = ('num_envs', 'device', 'seed', 'auto_reset', 'lib_path', 'urdf_path')
INFO_KEYS = ('ave_spd', 'max_spd', 'reward_vel', 'reward_rotation', 'reward_dist', 'reward_avg_spd')   # PGE:356-362


def _spaces(prop_type):
    if not isinstance(prop_type, list):
        raise TypeError("Expected 'prop_type' to be a list.")                  # PGE:122
    full = {'joint_pos': 12, 'joint_vel': 12, 'root_lin_vel_loc': 3, 'root_ang_vel_loc': 3, 'e_g': 3}
    prop = sum(full[e] for e in prop_type) * 3                                 # PGE:108-123, stack_frame_num 3
    obs = Dict(OrderedDict([('prop', Box(0, 0, shape=(prop,))), ('prop_a', Box(0, 0, shape=(36,))), ('percep_2d', Box(0, 0, shape=(25, 13))),
                            ('percep_1d', Box(0, 0, shape=(128,))), ('percep_front', Box(0, 0, shape=(25, 13))), ('target', Box(0, 0, shape=(3,)))]))
    act = Dict(OrderedDict([('A_Z', Discrete(256)), ('A_LLC', Box(0, 0, shape=(12,)))]))      # PGE:141-145
    return obs, act, prop


def _split(row, prop):
    a = prop + 36
    return OrderedDict([('prop', row[..., :prop]), ('prop_a', row[..., prop:a]),
                        ('percep_2d', row[..., a:a + 325].reshape(row.shape[:-1] + (25, 13))), ('percep_1d', row[..., a + 325:a + 453]),
                        ('percep_front', row[..., a + 453:a + 778].reshape(row.shape[:-1] + (25, 13))), ('target', row[..., a + 778:a + 781])])


def _build_engine(env_config, num_envs, auto_reset):
    arena_id = env_config['arena_id']                                          # KeyError like CPE:68
    assert arena_id in ['Playground']                                          # CPE:69-71
    if env_config.get('render', False):
        warnings.warn('render is ignored: the batched engine has no GUI')
    cfg = epmc_capi.make_epmc_config(num_envs, env_config, auto_reset=auto_reset, seed=env_config.get('seed', 0), device=env_config.get('device', 0))
    urdf_path = env_config.get('urdf_path', None)
    blob = urdf_model.UrdfModel(urdf_path).blob() if urdf_path else urdf_model.default_model_blob()
    return epmc_capi.EpmcEngine(cfg, blob, lib_path=env_config.get('lib_path', None))


def _llc(action):
    a = action['A_LLC'] if isinstance(action, dict) and 'A_LLC' in action else action      # PGE:322
    return np.asarray(a, dtype=np.float32)


class _ReferenceDraws(object):
    """The np.random calls of PlayGroundEnv / BulletStatics / PushRandomizer, in their order, returned as the uniforms in [0, 1)
    that make the engine reproduce the drawn values (ll_epmc_reset h_draws, ll_epmc_set_step_draws)."""

    def __init__(self, env_config):
        self.rc = env_config['env_randomize_config']
        self.element = int(self.rc['element_id'])
        self.obs_rand = env_config.get('obs_randomization') or {}
        self.push = self.rc.get('disturb_force_config')
        self.n_sub = int((1.0 / env_config.get('control_freq', 50.0)) / epmc_capi.TIME_STEP)
        self.u = []
        np.random.uniform(*self.rc['friction_range'])                          # PGE:92: the constructor's own friction draw
        self.max_tau = env_config.get('max_tau', 16.0)
        self.max_tau_value = float(np.random.uniform(*self.max_tau)) if isinstance(self.max_tau, list) else self.max_tau   # LR:244, right after it

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
        self._uniform(*self.rc['friction_range'])                              # PGE:209
        if self.push is not None:                                              # PGE:213-214
            self._force()
            self.count = -self.push.get('start_time', 0.) // epmc_capi.TIME_STEP
            self.interval = self.push.get('interval_time', 5.) // epmc_capi.TIME_STEP
        if self.element != 0:                                                  # BSE:27-28, :166-170
            self._uniform(0.02, 0.5)
            self._uniform(1.0, 20.0)
            if self.element in (1, 2):                                         # BSE:200-222
                hc = self.rc['hole_config'] if self.element == 2 else {}
                n = self._randint(1, 10)
                for half in range(2):
                    for _ in range(n):
                        if self.element == 1:
                            self._uniform(0.05, 0.15); self._uniform(1.0, 3.0)   # height, distance (BSE:325, :345)
                        else:
                            self._uniform(1.0, 3.0); self._uniform(hc.get('min_gap_height', 0.25), hc.get('max_gap_height', 0.3))   # BSE:400-401
                    if half == 0:
                        self._uniform(-1.0, 1.0)
            else:                                                              # BSE:187-198
                n = self._randint(1, 5)
                for half in range(2):
                    for _ in range(n):
                        self._uniform(0.0, 1.0)
                    if half == 0:
                        self._uniform(-3.0, 3.0)
        cr = self.rc.get('cmd_vary_freq_range', [25, 200])
        self.cmd_freq = self._randint(*cr)                                     # PGE:223
        if isinstance(self.max_tau, list):                                     # PGE:236: lands in an attribute the torque clip never reads,
            np.random.uniform(*self.max_tau)                                   # but the global stream moves on
        drawn = {}
        for k in self.obs_rand:                                                # PGE:176-179, in the dict's own order ...
            n0 = len(self.u)
            self._uniform(*self.obs_rand[k])
            drawn[k] = self.u.pop(n0)
        self.u += [drawn[k] for k in epmc_capi.NOISE_KEYS if k in drawn]       # ... handed over in the engine's fixed key order
        self.u.append(float(np.random.rand()))                                 # PGE:183
        self.counter = 0
        out = np.full(epmc_capi.LLE_MAX_DRAWS, 0.5, np.float32)
        out[:len(self.u)] = self.u
        return out

    def step(self):
        self.u = []
        if self.element == 0 and self.counter % self.cmd_freq == 0:
            self._uniform(0, 2 * np.pi)                                        # PGE:303
        if self.counter % self.cmd_freq == 0:
            self._uniform(*self.rc['target_spd_range'])                        # PGE:313
        if self.push is not None:
            for _ in range(self.n_sub):                                        # PR:56-71
                self.count += 1
                if self.count > 0 and self.count % self.interval == 0:
                    self._force()
                    self.count = 0
        self.counter += 1
        return np.array(self.u, dtype=np.float32)


class PlaygroundGame(object):
    """PlayGroundEnv behind SingleAgentWrapper, one robot, reference semantics (no auto-reset)."""

    def __init__(self, env_config):
        self._draws = _ReferenceDraws(env_config)                                 # the constructor's draws, in the reference's order
        self._engine = _build_engine(dict(env_config, max_tau=self._draws.max_tau_value), 1, auto_reset=0)
        obs, act, self._prop = _spaces(env_config['prop_type'])
        self.observation_space, self.action_space = Tuple([obs]), Tuple([act])    # CPE:9-10
        self.env = self

    def _obs(self):
        return _split(self._engine.obs()[0].astype(np.float64), self._prop)

    def reset(self, **kwargs):                                                  # CPE:12-14
        self._engine.reset(draws=self._draws.reset()[None])
        return (self._obs(),)

    def step(self, action):                                                     # CPE:16-18 uses action[0]
        self._engine.set_step_draws(self._draws.step()[None])
        self._engine.step_host(_llc(action[0]).reshape(1, 12))
        r, d, _ = self._engine.reward_done()
        info = {}
        if d[0]:
            info = dict(zip(INFO_KEYS, [float(x) for x in self._engine.info()[0]]))
        return (self._obs(),), (float(r[0]),), bool(d[0]), info

    def close(self):
        self._engine.close()


class BatchedPlaygroundEnv(object):
    """num_envs playgrounds in lockstep on one GPU: arrays in, arrays out; finished envs are re-seeded (new terrain, friction,
    start yaw) inside the step kernel when ``auto_reset`` (default)."""

    def __init__(self, env_config):
        self.num_envs = int(env_config['num_envs'])
        self.auto_reset = bool(env_config.get('auto_reset', True))
        self.engine = _build_engine(env_config, self.num_envs, int(self.auto_reset))
        obs, act, self.prop_size = _spaces(env_config['prop_type'])
        self.single_observation_space, self.single_action_space = obs, act
        self.observation_space, self.action_space = Tuple([obs]), Tuple([act])
        self.obs_dim = self.engine.obs_dim

    def reset(self, env_ids=None):
        self.engine.reset(env_ids)
        return self.engine.obs()

    def step(self, actions):
        self.engine.step_host(_llc(actions))
        r, d, why = self.engine.reward_done()
        return self.engine.obs(), r, d, {'done_reason': why, 'episode_info': self.engine.info()}

    def split(self, obs):
        return _split(obs, self.prop_size)

    def close(self):
        self.engine.close()


class _Untupled(object):
    def __init__(self, game):
        self._game = game
        self.observation_space = game.observation_space.spaces[0]              # CPE:150-154
        self.action_space = game.action_space.spaces[0]

    def __getattr__(self, name):
        return getattr(self._game, name)


def create_playground_game(**env_config):
    unknown = [k for k in env_config if k not in ('arena_id', 'render', 'control_freq', 'kp', 'kd', 'max_tau', 'prop_type', 'max_steps',
                                                  'obs_randomization', 'env_randomize_config') + ENGINE_KEYS]
    if unknown:
        warnings.warn('create_playground_game: unused keys %s' % unknown)
    if int(env_config.get('num_envs', 1)) > 1:
        return BatchedPlaygroundEnv(env_config)
    return PlaygroundGame(env_config)


def create_playground_env(**env_config):
    return _Untupled(create_playground_game(**env_config))
