"""The reference's env plug-in surface over the HIP batch stepper.

Mirrors ``lifelike.sim_envs.pybullet_envs.create_pybullet_envs`` (CPE) for the PMC tracking task:

    create_tracking_game(**env_config)   CPE:21-64    -> multi-agent tuple API (1-tuples), what ``--outer_env`` names
    create_tracking_env(**env_config)    CPE:143-147  -> same object with the two spaces un-tupled

``env_config`` takes the reference's keys with the reference's defaults (CPE:28-59).  Extra, optional keys select the
batched engine: ``num_envs`` (default 1), ``device``, ``seed``, ``auto_reset``, ``keep_terminal_obs``, ``lib_path``.

* ``num_envs == 1``  -> :class:`TrackingGame`: drop-in for an unmodified TLeague actor -- ``reset(**kw)`` returns
  ``(OrderedDict(prop, prop_a, future),)``, ``step([a])`` returns ``((obs,), (reward,), done, {})``; episodes never
  auto-reset (PLE never does); clip and start time are drawn from NumPy's global RNG with the reference's two calls
  (ML:60, ML:51), so ``np.random.seed(s)`` reproduces the reference's (clip, t0).
* ``num_envs > 1``   -> :class:`BatchedTrackingEnv`: array API for rollout workers.

There is no CPU fallback: the engine needs libllenv.so (HIP, gfx950) and a GPU.
"""
import os
import warnings
from collections import OrderedDict

import numpy as np

from . import capi, mocap, urdf_model
from .spaces import Box, Dict, Tuple

ENGINE_KEYS = ('num_envs', 'device', 'seed', 'auto_reset', 'keep_terminal_obs', 'lib_path', 'urdf_path')
KNOWN_KEYS = ('arena_id', 'render', 'control_freq', 'sim_freq', 'kp', 'kd', 'max_tau', 'data_path', 'prop_type',
              'prioritized_sample_factor', 'set_obstacle', 'obstacle_height', 'reward_weights', 'foot_lateral_friction',
              'video_path', 'enable_gui') + ENGINE_KEYS


def _build_engine(env_config, num_envs, auto_reset):
    """CPE:27-59 create_single_env: same keys, same defaults."""
    enable_render = env_config.get('render', False)
    control_freq = env_config.get('control_freq', 25.0)
    sim_freq = env_config.get('sim_freq', 500.0)
    kp = env_config.get('kp', 50.0)
    kd = env_config.get('kd', 1.0)
    max_tau = env_config.get('max_tau', 18.0)
    data_path = env_config.get('data_path', '')
    prop_type = env_config.get('prop_type', '')
    prioritized_sample_factor = env_config.get('prioritized_sample_factor', 0.0)
    set_obstacle = env_config.get('set_obstacle', False)
    obstacle_height = env_config.get('obstacle_height', 0.0)                 # quirk Q7: factory default 0.0
    reward_weights = env_config.get('reward_weights', None)
    video_path = env_config.get('video_path', None)
    if video_path is not None:
        assert isinstance(video_path, str) and video_path.endswith('.mp4')  # PLE:54
    if enable_render or video_path is not None:
        warnings.warn('render / video_path are ignored: the batched engine has no GUI (PLE:58-60 is PyBullet-only)')
    if not isinstance(prop_type, list):
        raise TypeError("Expected 'prop_type' to be a list.")                # PLE:113
    if isinstance(max_tau, tuple):                                          # the reference only understands a list (LR:244): a tuple
        raise ValueError('operands could not be broadcast together: max_tau must be a scalar or a [lo, hi] list (LR:244-245)')   # reaches np.ones_like(...) * tuple
    if isinstance(max_tau, list):                                           # LR:244 draws once at construction; the per-episode redraw
        max_tau = float(np.random.uniform(*max_tau))                        # of PLE:153 never reaches the torque clip (quirk Q1)
    policy_step = 1.0 / control_freq
    table = mocap.load_mocap(data_path, policy_step)
    urdf_path = env_config.get('urdf_path', None)
    blob = urdf_model.UrdfModel(urdf_path).blob() if urdf_path else urdf_model.default_model_blob()
    cfg = capi.make_config(num_envs, control_freq=control_freq, sim_freq=sim_freq, kp=kp, kd=kd, max_tau=max_tau,
                           foot_lateral_friction=env_config.get('foot_lateral_friction', 0.5),
                           reward_weights=reward_weights, prop_type=prop_type,
                           prioritized_sample_factor=prioritized_sample_factor, set_obstacle=set_obstacle,
                           obstacle_height=obstacle_height, auto_reset=auto_reset,
                           keep_terminal_obs=bool(env_config.get('keep_terminal_obs', False)) and bool(auto_reset),
                           seed=env_config.get('seed', 0), device=env_config.get('device', 0))
    eng = capi.Engine(cfg, blob, table, lib_path=env_config.get('lib_path', None))
    return eng, table, list(prop_type)


def _spaces(prop_type):
    prop_size = sum(capi.PROP_SIZES[e] for e in prop_type) * 3             # PLE:109-114 stack_frame_num = 3
    return Dict(OrderedDict([('prop', Box(0, 0, shape=(prop_size,))), ('prop_a', Box(0, 0, shape=(36,))),
                             ('future', Box(0, 0, shape=(72,)))])), Box(0, 0, shape=(12,))     # PLE:117-124


def _split_obs(row, prop_size):
    return OrderedDict([('prop', row[..., :prop_size]), ('prop_a', row[..., prop_size:prop_size + 36]),
                        ('future', row[..., prop_size + 36:])])             # PLE:292-296


class TrackingGame(object):
    """PrimitiveLevelEnv wrapped in SingleAgentWrapper (CPE:6-18), one robot, reference semantics."""

    def __init__(self, env_config):
        self._max_tau_cfg = env_config.get('max_tau', 18.0)                 # PLE:47 keeps the configured value, list or scalar
        self._engine, self._table, self._prop_type = _build_engine(env_config, 1, auto_reset=0)
        obs_space, act_space = _spaces(self._prop_type)
        self.observation_space = Tuple([obs_space])                         # CPE:9
        self.action_space = Tuple([act_space])                              # CPE:10
        self._prop_size = obs_space.spaces['prop'].shape[0]
        self.env = self                                                     # gym.Wrapper exposes .env
        self.sampled_data_idx = None                                        # PLE:135
        self.time = 0                                                       # PLE:139
        self.reward_sum = 0.0                                               # PLE:138
        self.avg_episode_len = np.zeros(self._table.n_clips)               # PLE:134

    @property
    def _prioritized_sample_probability(self):
        return self._engine.sampling_table()[0]

    def _obs(self):
        row = self._engine.obs()[0]
        return _split_obs(row.astype(np.float64), self._prop_size)          # the reference returns float64 arrays

    def reset(self, **kwargs):                                              # CPE:12-14 (kwargs accepted and dropped)
        if isinstance(self._max_tau_cfg, list):                             # PLE:153: the redraw lands in an attribute apply_action never
            np.random.uniform(*self._max_tau_cfg)                           # reads (quirk Q1), but the global stream moves on -- so does ours
        prob = self._engine.sampling_table()[0]
        n = self._table.n_clips
        clip = int(np.random.choice(range(n), p=prob))                      # ML:60
        duration = self._table.frame_step * (int(self._table.clip_len[clip]) - self._table.margin - 1)   # ML:50
        t0 = np.random.uniform(0, 1) * duration                             # ML:51
        self._engine.reset(clip=[clip], t0=[t0])
        self.sampled_data_idx, self.time, self.reward_sum = clip, t0, 0.0
        return (self._obs(),)

    def step(self, action):                                                 # CPE:16-18 uses action[0]
        a = np.asarray(action[0], dtype=np.float32).reshape(1, 12)
        self._engine.step_host(a)
        r, d, _ = self._engine.reward_done()
        info = self._engine.episode_info()
        self.time = float(info['time'][0])
        self.reward_sum = float(info['reward_sum'][0])
        if d[0]:
            self.avg_episode_len = self._engine.sampling_table()[2]         # PLE:237
        return (self._obs(),), (float(r[0]),), bool(d[0]), {}               # PLE:245: info is {}

    def close(self):                                                        # PLE:428-431
        self._engine.close()


class _UntupledSpaces(object):
    """create_tracking_env: same env, spaces un-tupled (CPE:143-147)."""

    def __init__(self, game):
        self._game = game
        self.observation_space = game.observation_space.spaces[0]
        self.action_space = game.action_space.spaces[0]

    def __getattr__(self, name):
        return getattr(self._game, name)


class BatchedTrackingEnv(object):
    """num_envs robots in lockstep on one GPU.  Arrays in, arrays out; finished envs are re-seeded inside the step
    kernel when ``auto_reset`` (default) -- ``obs`` then holds the first observation of the new episode; with
    ``keep_terminal_obs=True`` the last one of the finished episode is kept in ``terminal_obs()`` as well."""

    def __init__(self, env_config):
        self.num_envs = int(env_config['num_envs'])
        self.auto_reset = bool(env_config.get('auto_reset', True))
        self.engine, self.table, self.prop_type = _build_engine(env_config, self.num_envs, int(self.auto_reset))
        obs_space, act_space = _spaces(self.prop_type)
        self.single_observation_space, self.single_action_space = obs_space, act_space
        self.observation_space = Tuple([obs_space])
        self.action_space = Tuple([act_space])
        self.prop_size = obs_space.spaces['prop'].shape[0]
        self.obs_dim = self.engine.obs_dim

    def reset(self, env_ids=None, clip=None, t0=None):
        self.engine.reset(env_ids, clip, t0)
        return self.engine.obs()

    def step(self, actions):
        self.engine.step_host(actions)
        r, d, why = self.engine.reward_done()
        return self.engine.obs(), r, d, {'done_reason': why}

    def step_device(self, d_actions_ptr=None):
        """Zero-copy path: actions already in HBM (device address) or in the engine's own action buffer."""
        self.engine.step(d_actions_ptr)

    def split(self, obs):
        return _split_obs(obs, self.prop_size)

    def terminal_obs(self):
        return self.engine.terminal_obs()

    def close(self):
        self.engine.close()


def create_tracking_game(**env_config):
    arena_id = env_config['arena_id']
    assert arena_id in ['LeggedRobotTracking', ]                            # CPE:22-25
    unknown = sorted(set(env_config) - set(KNOWN_KEYS))
    if unknown:
        warnings.warn('ignoring unknown env_config keys: %s' % unknown)
    if int(env_config.get('num_envs', 1)) > 1:
        return BatchedTrackingEnv(env_config)
    return TrackingGame(env_config)


def create_tracking_env(**env_config):
    return _UntupledSpaces(create_tracking_game(**env_config))
