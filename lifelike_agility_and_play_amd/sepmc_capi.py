"""ctypes binding of include/llenv_sepmc.h (the SEPMC / ChaseTagGameEnv engine inside libllenv.so)."""
import ctypes as C
import os
import math

import numpy as np

from . import capi
from .epmc_capi import NOISE_KEYS, TIME_STEP, default_init_state

LLS_N_RAYS, LLS_N_VIS, LLS_MAX_BOXES, LLS_MAX_DRAWS, LLS_MAX_CONTACTS = 778, 21, 12, 64, 8
BODY_PLANE, BODY_STATIC, BODY_FLAG, BODY_ROBOT0, BODY_ROBOT1 = 0, 1, 2, 3, 4
DONE_FALL, DONE_TIME, DONE_CATCH, DONE_NONFINITE = 1, 2, 8, 16


class LLSepmcConfig(C.Structure):   # struct ll_sepmc_config
    _fields_ = [('abi_version', C.c_int32), ('n_arenas', C.c_int32), ('device', C.c_int32), ('auto_reset', C.c_int32),
                ('control_freq', C.c_double), ('kp', C.c_double), ('kd', C.c_double), ('max_tau', C.c_double),
                ('max_steps', C.c_int32), ('prop_order', C.c_int32 * 5), ('rand_cube', C.c_int32), ('hurdle', C.c_int32), ('hole', C.c_int32),
                ('solver_iterations', C.c_int32), ('friction_range', C.c_double * 2), ('push_enabled', C.c_int32), ('push_count0', C.c_int32),
                ('push_interval_step', C.c_int32), ('push_duration_step', C.c_int32), ('horizontal_force', C.c_double * 2),
                ('vertical_force', C.c_double * 2), ('push_strength_ratio', C.c_double), ('visible_angle', C.c_double), ('control_spd', C.c_double),
                ('noise_enabled', C.c_int32 * 4), ('noise_range', (C.c_double * 2) * 4), ('seed', C.c_uint64), ('max_tau_robot1', C.c_double)]


def make_sepmc_config(n_arenas, env_config, auto_reset=0, seed=0, device=0, solver_iterations=10):
    """From the dict `create_chase_tag_game(**env_config)` takes (create_pybullet_envs.py:104-140), same keys and defaults."""
    prop_type = env_config['prop_type'] if 'prop_type' in env_config else None
    if not isinstance(prop_type, list):
        raise TypeError("Expected 'prop_type' to be a list.")                       # CTG:104
    rc = env_config.get('env_randomize_config', {})
    el = env_config.get('element_config', {}) or {}
    cfg = LLSepmcConfig()
    cfg.abi_version, cfg.n_arenas, cfg.device, cfg.auto_reset = capi.LL_ABI_VERSION, int(n_arenas), int(device), int(auto_reset)
    cfg.control_freq = float(env_config.get('control_freq', 25.0))
    cfg.kp, cfg.kd = float(env_config.get('kp', 50.0)), float(env_config.get('kd', 1.0))
    max_tau = env_config.get('max_tau', 18.0)
    cfg.max_tau_robot1 = float(env_config.get('max_tau_robot1', 0.0))               # (the 1-arena shim passes the second robot's own draw, LR:244)
    if isinstance(max_tau, (list, tuple)):                                           # one draw per LeggedRobot at construction (LR:244, CTG:62-72)
        max_tau, cfg.max_tau_robot1 = float(np.random.uniform(*max_tau)), float(np.random.uniform(*max_tau))
    cfg.max_tau = float(max_tau)
    cfg.max_steps = int(env_config.get('max_steps', 1000))
    for i in range(5):
        cfg.prop_order[i] = capi.PROP_IDS[prop_type[i]] if i < len(prop_type) else -1   # KeyError mirrors CTG:102
    cfg.rand_cube, cfg.hurdle, cfg.hole = int(bool(el.get('rand_cube'))), int(bool(el.get('hurdle'))), int(bool(el.get('hole')))
    cfg.solver_iterations = int(solver_iterations)
    cfg.friction_range[0], cfg.friction_range[1] = [float(x) for x in rc['friction_range']]    # CTG:61 (KeyError like the reference)
    if 'disturb_force_config' in rc:                                                 # CTG:154-157, PR:24-54
        pc = rc['disturb_force_config']
        start, interval, duration = pc.get('start_time', 0.), pc.get('interval_time', 5.), pc.get('duration_time', 0.5)
        assert duration <= interval                                                  # PR:34
        cfg.push_enabled = 1
        cfg.push_count0 = int(-start // TIME_STEP)                                   # Python's float floor division, as PR:45-53 evaluates them
        cfg.push_interval_step = int(interval // TIME_STEP)
        cfg.push_duration_step = int(duration // TIME_STEP)
        hf, vf = pc.get('horizontal_force', 20), pc.get('vertical_force', 5)
        assert isinstance(hf, list) and isinstance(vf, list)                         # PR:90-91
        cfg.horizontal_force[0], cfg.horizontal_force[1] = float(hf[0]), float(hf[1])
        cfg.vertical_force[0], cfg.vertical_force[1] = float(vf[0]), float(vf[1])
        cfg.push_strength_ratio = float(pc.get('push_strength_ratio', 1.0))
    cfg.visible_angle = math.pi                                                      # CTG:31; the factory never passes another
    cfg.control_spd = float(rc['control_spd']) if 'control_spd' in rc else -1.0      # CTG:361
    obs_rand = env_config.get('obs_randomization') or {}
    for i, k in enumerate(NOISE_KEYS):
        if k in obs_rand:
            cfg.noise_enabled[i] = 1
            cfg.noise_range[i][0], cfg.noise_range[i][1] = float(obs_rand[k][0]), float(obs_rand[k][1])
    if ('pos_x_bias' in obs_rand) and ('pos_y_bias' not in obs_rand):
        raise KeyError('pos_y_bias')                                                 # CTG:505-507 reads both under the x key
    cfg.seed = int(seed)
    return cfg


_V = C.c_void_p
_SIGS = {
    'll_sepmc_create': (C.c_int, [C.POINTER(LLSepmcConfig), _V, C.c_int, _V, C.POINTER(_V)]),
    'll_sepmc_destroy': (C.c_int, [_V]),
    'll_sepmc_reset': (C.c_int, [_V, _V, C.c_int, _V, _V]),
    'll_sepmc_step': (C.c_int, [_V, _V]),
    'll_sepmc_set_actions': (C.c_int, [_V, _V]),
    'll_sepmc_fill_random_actions': (C.c_int, [_V, C.c_float]),
    'll_sepmc_step_scripted': (C.c_int, [_V, _V, _V, _V, _V, _V, _V, _V, C.c_int]),
    'll_sepmc_set_step_draws': (C.c_int, [_V, _V, C.c_int]),
    'll_sepmc_script_reset': (C.c_int, [_V, _V, _V, _V]),
    'll_sepmc_sync': (C.c_int, [_V]),
    'll_sepmc_set_spec_param': (C.c_int, [_V, C.c_int, C.c_double]),
    'll_sepmc_get_spec_param': (C.c_int, [_V, C.c_int, C.POINTER(C.c_double)]),
    'll_sepmc_obs_dim': (C.c_int, [_V]),
    'll_sepmc_get_obs': (C.c_int, [_V, _V]),
    'll_sepmc_get_reward_done': (C.c_int, [_V, _V, _V, _V]),
    'll_sepmc_get_state': (C.c_int, [_V, _V]),
    'll_sepmc_set_state': (C.c_int, [_V, _V]),
    'll_sepmc_get_episode': (C.c_int, [_V, _V]),
    'll_sepmc_get_info': (C.c_int, [_V, _V]),
    'll_sepmc_get_boxes': (C.c_int, [_V, _V, _V]),
    'll_sepmc_get_rays': (C.c_int, [_V, _V, _V, _V, _V]),
    'll_sepmc_get_vis': (C.c_int, [_V, _V]),
    'll_sepmc_get_push_trace': (C.c_int, [_V, _V, C.POINTER(C.c_int32)]),
    'll_sepmc_get_counters': (C.c_int, [_V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'll_sepmc_device_ptrs': (C.c_int, [_V, C.POINTER(capi.LLDevicePtrs)]),
    'll_sepmc_kernel_time_ms': (C.c_int, [_V, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'll_sepmc_kernel_time_stats': (C.c_int, [_V, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'll_sepmc_step_random_n': (C.c_int, [_V, C.c_float, C.c_int]),
    'll_sepmc_enable_kernel_timing': (C.c_int, [_V, C.c_int]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)
_bound = {}


def load_library(path=None):
    lib = capi.load_library(path)
    key = id(lib)
    if key not in _bound:
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)               # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _bound[key] = True
    return lib


_ptr = capi._ptr
EP_FIELDS = ('flag_x', 'flag_y', 'flag_z', 'with_flag0', 'friction', 'fix_spd', 'counter', 'push_fx', 'push_fy', 'push_fz', 'pos_x_bias', 'pos_y_bias', 'yaw_bias',
             'pos_z_bias', 'last_two_rob_pos_diff_len', 'last_esc_flag_pos_diff_len', 'switch', 'visible0', 'visible1', 'who0', 'who_taker')


class SepmcEngine(object):
    """One batch of chase-tag arenas (two robots each) on one GPU (ll_sepmc_engine).  Per-robot arrays are [arena][robot][...]."""

    def __init__(self, cfg, model_blob, init_state=None, lib_path=None):
        self.lib = load_library(lib_path)
        self.n_arenas = int(cfg.n_arenas)
        self._pid = os.getpid()
        self.h = C.c_void_p()
        blob = np.ascontiguousarray(model_blob, dtype=np.float64)
        init = np.ascontiguousarray(default_init_state() if init_state is None else init_state, dtype=np.float64)
        assert init.shape == (37,)
        self._chk(self.lib.ll_sepmc_create(C.byref(cfg), _ptr(blob), int(blob.size), _ptr(init), C.byref(self.h)))
        self.obs_dim = int(self.lib.ll_sepmc_obs_dim(self.h))
        self.n_sub = int((1.0 / cfg.control_freq) / TIME_STEP)                   # CTG:57

    def _chk(self, rc):
        if rc != 0:
            raise capi.LLError(rc, self.lib.ll_last_error().decode())

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            if getattr(self, '_pid', None) == os.getpid():      # (a fork()ed child inherits the object, not the HIP context: it must not destroy it)
                self.lib.ll_sepmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:     # noqa: BLE001
            pass

    def reset(self, arena_ids=None, draws=None, prev_orn=None):
        ids = None if arena_ids is None else np.ascontiguousarray(arena_ids, dtype=np.int32)
        n = self.n_arenas if ids is None else len(ids)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.float32).reshape(n, LLS_MAX_DRAWS)
        po = None if prev_orn is None else np.ascontiguousarray(prev_orn, dtype=np.float32).reshape(n, 4)
        self._chk(self.lib.ll_sepmc_reset(self.h, _ptr(ids), n, _ptr(d), _ptr(po)))

    def step(self, d_actions_ptr=None):
        self._chk(self.lib.ll_sepmc_step(self.h, d_actions_ptr))

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_arenas, 2, 12)
        self._chk(self.lib.ll_sepmc_set_actions(self.h, _ptr(a)))
        self.step()

    def _script(self, ray_hit, ray_frac, vis_blocked):
        A = self.n_arenas
        return (np.ascontiguousarray(ray_hit, dtype=np.uint8).reshape(A, 2, LLS_N_RAYS), np.ascontiguousarray(ray_frac, dtype=np.float32).reshape(A, 2, LLS_N_RAYS),
                np.ascontiguousarray(vis_blocked, dtype=np.uint8).reshape(A, LLS_N_VIS))

    def step_scripted(self, actions, state, ray_hit, ray_frac, vis_blocked, contacts, draws=None):
        A = self.n_arenas
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(A, 2, 12)
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(A, 2, 37)
        h, f, v = self._script(ray_hit, ray_frac, vis_blocked)
        c = np.ascontiguousarray(contacts, dtype=np.int32).reshape(A, LLS_MAX_CONTACTS, 4)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.float32).reshape(A, -1)
        self._chk(self.lib.ll_sepmc_step_scripted(self.h, _ptr(a), _ptr(s), _ptr(h), _ptr(f), _ptr(v), _ptr(c), _ptr(d), 0 if d is None else d.shape[1]))

    def set_step_draws(self, draws):
        d = np.ascontiguousarray(draws, dtype=np.float32).reshape(self.n_arenas, -1)
        self._chk(self.lib.ll_sepmc_set_step_draws(self.h, _ptr(d) if d.shape[1] else None, d.shape[1]))

    def script_reset(self, ray_hit, ray_frac, vis_blocked):
        h, f, v = self._script(ray_hit, ray_frac, vis_blocked)
        self._chk(self.lib.ll_sepmc_script_reset(self.h, _ptr(h), _ptr(f), _ptr(v)))

    def fill_random_actions(self, sigma):
        self._chk(self.lib.ll_sepmc_fill_random_actions(self.h, float(sigma)))

    def sync(self):
        self._chk(self.lib.ll_sepmc_sync(self.h))

    def set_spec(self, **kw):
        """ll_sepmc_set_spec_param: the physics-spec switches of include/llenv_model.h (the robot and its solver are the PMC engine's), e.g.
        set_spec(friction_mode=2)."""
        for k, v in kw.items():
            self._chk(self.lib.ll_sepmc_set_spec_param(self.h, capi.SPEC_IDS[k], float(v)))

    def get_spec(self, key):
        v = C.c_double()
        self._chk(self.lib.ll_sepmc_get_spec_param(self.h, capi.SPEC_IDS[key], C.byref(v)))
        return v.value

    def obs(self):
        o = np.empty((self.n_arenas, 2, self.obs_dim), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_obs(self.h, _ptr(o)))
        return o

    def reward_done(self):
        r = np.empty((self.n_arenas, 2), dtype=np.float32); d = np.empty(self.n_arenas, dtype=np.uint8); w = np.empty(self.n_arenas, dtype=np.uint8)
        self._chk(self.lib.ll_sepmc_get_reward_done(self.h, _ptr(r), _ptr(d), _ptr(w)))
        return r, d.astype(bool), w

    def state(self):
        s = np.empty((self.n_arenas, 2, 37), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_state(self.h, _ptr(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32).reshape(self.n_arenas, 2, 37)
        self._chk(self.lib.ll_sepmc_set_state(self.h, _ptr(s)))

    def episode(self):
        e = np.empty((self.n_arenas, 21), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_episode(self.h, _ptr(e)))
        return {k: e[:, i] for i, k in enumerate(EP_FIELDS)}

    def info(self):
        v = np.empty((self.n_arenas, 4), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_info(self.h, _ptr(v)))
        return v

    def boxes(self):
        rows = np.empty((self.n_arenas, LLS_MAX_BOXES, 6), dtype=np.float32); n = np.empty(self.n_arenas, dtype=np.int32)
        self._chk(self.lib.ll_sepmc_get_boxes(self.h, _ptr(rows), _ptr(n)))
        return rows, n

    def rays(self):
        f = np.empty((self.n_arenas, 2, LLS_N_RAYS, 3), dtype=np.float32); t = np.empty_like(f)
        h = np.empty((self.n_arenas, 2, LLS_N_RAYS), dtype=np.uint8); fr = np.empty((self.n_arenas, 2, LLS_N_RAYS), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_rays(self.h, _ptr(f), _ptr(t), _ptr(h), _ptr(fr)))
        return f, t, h.astype(bool), fr

    def vis(self):
        v = np.empty((self.n_arenas, LLS_N_VIS, 8), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_vis(self.h, _ptr(v)))
        return v

    def push_trace(self):
        n = C.c_int32(0)
        rows = np.empty((self.n_arenas, 2, self.n_sub, 4), dtype=np.float32)
        self._chk(self.lib.ll_sepmc_get_push_trace(self.h, _ptr(rows), C.byref(n)))
        assert n.value == self.n_sub
        return rows

    def counters(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.ll_sepmc_get_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(arena_steps=a.value, episodes=b.value, nonfinite=c.value)

    def device_ptrs(self):
        """Device addresses of obs / reward / done / actions and the engine's stream (gather.engine_tensors, gather.use_engine_stream)."""
        p = capi.LLDevicePtrs()
        self._chk(self.lib.ll_sepmc_device_ptrs(self.h, C.byref(p)))
        return p

    def enable_kernel_timing(self, on=True):
        self._chk(self.lib.ll_sepmc_enable_kernel_timing(self.h, 1 if on else 0))

    def step_random_n(self, sigma, n_steps):
        """n_steps x {fill_random_actions(sigma); step()} in one launch (ll_sepmc_step_random_n)"""
        self._chk(self.lib.ll_sepmc_step_random_n(self.h, float(sigma), int(n_steps)))

    def kernel_time_stats(self):
        ms, n, st = C.c_double(), C.c_int(), C.c_int64()
        self._chk(self.lib.ll_sepmc_kernel_time_stats(self.h, C.byref(ms), C.byref(n), C.byref(st)))
        return ms.value, n.value, st.value

    def kernel_time_ms(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.lib.ll_sepmc_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
