"""ctypes binding of include/llenv_epmc.h (the EPMC / PlayGroundEnv engine inside libllenv.so)."""
import ctypes as C
import os

import numpy as np

from . import capi

LLE_N_RAYS, LLE_MAX_STATICS, LLE_MAX_DRAWS = 778, 104, 64
NOISE_KEYS = ('pos_x_bias', 'pos_y_bias', 'yaw_bias', 'pos_z_bias')
TIME_STEP = 1.0 / 500.0            # PGE:82


class LLEpmcConfig(C.Structure):   # struct ll_epmc_config
    _fields_ = [('abi_version', C.c_int32), ('n_envs', C.c_int32), ('device', C.c_int32), ('auto_reset', C.c_int32),
                ('control_freq', C.c_double), ('kp', C.c_double), ('kd', C.c_double), ('max_tau', C.c_double),
                ('max_steps', C.c_int32), ('prop_order', C.c_int32 * 5), ('element_id', C.c_int32), ('solver_iterations', C.c_int32),
                ('friction_range', C.c_double * 2), ('push_enabled', C.c_int32), ('push_count0', C.c_int32),
                ('push_interval_step', C.c_int32), ('push_duration_step', C.c_int32), ('horizontal_force', C.c_double * 2),
                ('vertical_force', C.c_double * 2), ('push_strength_ratio', C.c_double), ('cmd_vary_freq_range', C.c_int32 * 2),
                ('target_spd_range', C.c_double * 2), ('auxiliary_radius', C.c_double), ('hole_gap_height', C.c_double * 2),
                ('noise_enabled', C.c_int32 * 4), ('noise_range', (C.c_double * 2) * 4), ('seed', C.c_uint64)]


def default_init_state():
    """LeggedRobot.get_init_states_info() (LR:116-117) as a 37-vector; data shipped in assets/."""
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'epmc_init_state.npy')).astype(np.float64)


def make_epmc_config(n_envs, env_config, auto_reset=0, seed=0, device=0, solver_iterations=10):
    """From the dict `create_playground_game(**env_config)` takes (create_pybullet_envs.py:67-101), same keys and defaults."""
    prop_type = env_config['prop_type'] if 'prop_type' in env_config else None
    if not isinstance(prop_type, list):
        raise TypeError("Expected 'prop_type' to be a list.")                       # PGE:122
    rc = env_config['env_randomize_config'] if 'env_randomize_config' in env_config else None
    cfg = LLEpmcConfig()
    cfg.abi_version, cfg.n_envs, cfg.device, cfg.auto_reset = capi.LL_ABI_VERSION, int(n_envs), int(device), int(auto_reset)
    cfg.control_freq = float(env_config.get('control_freq', 50.0))
    cfg.kp, cfg.kd = float(env_config.get('kp', 50.0)), float(env_config.get('kd', 1.0))
    max_tau = env_config.get('max_tau', 16.0)
    if isinstance(max_tau, (list, tuple)):                                           # PGE:235-236 redraws per episode; drawn once here
        max_tau = float(np.random.uniform(*max_tau))
    cfg.max_tau = float(max_tau)
    cfg.max_steps = int(env_config.get('max_steps', 1000))
    for i in range(5):
        cfg.prop_order[i] = capi.PROP_IDS[prop_type[i]] if i < len(prop_type) else -1   # KeyError mirrors PGE:120
    cfg.element_id = int(rc['element_id'])
    cfg.solver_iterations = int(solver_iterations)
    cfg.friction_range[0], cfg.friction_range[1] = [float(x) for x in rc['friction_range']]
    if 'disturb_force_config' in rc:                                                 # PGE:158-161, PR:24-54
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
    cr = rc.get('cmd_vary_freq_range', [25, 200])                                    # PGE:169
    cfg.cmd_vary_freq_range[0], cfg.cmd_vary_freq_range[1] = int(cr[0]), int(cr[1])
    cfg.target_spd_range[0], cfg.target_spd_range[1] = [float(x) for x in rc['target_spd_range']]
    aux = rc['auxiliary_radius']                                                     # PGE:80 (KeyError like the reference if absent)
    cfg.auxiliary_radius = -1.0 if aux is None else float(aux)
    hc = rc['hole_config'] if cfg.element_id == 2 else {}                            # PGE:206-207
    cfg.hole_gap_height[0], cfg.hole_gap_height[1] = float(hc.get('min_gap_height', 0.25)), float(hc.get('max_gap_height', 0.3))
    obs_rand = env_config.get('obs_randomization') or {}
    for i, k in enumerate(NOISE_KEYS):
        if k in obs_rand:
            cfg.noise_enabled[i] = 1
            cfg.noise_range[i][0], cfg.noise_range[i][1] = float(obs_rand[k][0]), float(obs_rand[k][1])
    if ('pos_x_bias' in obs_rand) != ('pos_y_bias' in obs_rand):
        raise KeyError('pos_y_bias')                                                 # PGE:388-391 reads both under the x key
    cfg.seed = int(seed)
    return cfg


_SIGS = {
    'll_epmc_create': (C.c_int, [C.POINTER(LLEpmcConfig), C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    'll_epmc_destroy': (C.c_int, [C.c_void_p]),
    'll_epmc_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'll_epmc_step': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_set_actions': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_step_scripted': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'll_epmc_script_reset_rays': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_epmc_set_step_draws': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'll_epmc_sync': (C.c_int, [C.c_void_p]),
    'll_epmc_set_spec_param': (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    'll_epmc_get_spec_param': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    'll_epmc_obs_dim': (C.c_int, [C.c_void_p]),
    'll_epmc_get_obs': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_get_reward_done': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_epmc_get_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_set_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_get_episode': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_get_info': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_epmc_get_statics': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_epmc_get_rays': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_epmc_get_push_trace': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    'll_epmc_get_counters': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'll_epmc_device_ptrs': (C.c_int, [C.c_void_p, C.POINTER(capi.LLDevicePtrs)]),
    'll_epmc_kernel_time_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'll_epmc_kernel_time_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'll_epmc_step_random_n': (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    'll_epmc_enable_kernel_timing': (C.c_int, [C.c_void_p, C.c_int]),
    'll_epmc_fill_random_actions': (C.c_int, [C.c_void_p, C.c_float]),
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
EP_FIELDS = ('target_x', 'target_y', 'target_z', 'target_spd', 'friction', 'cmd_vary_freq', 'counter', 'push_fx', 'push_fy', 'push_fz',
             'pos_x_bias', 'pos_y_bias', 'yaw_bias', 'pos_z_bias', 'last_pos_diff_len', 'init_pos_diff_len', 'total_spd', 'max_spd', 'episode')


class EpmcEngine(object):
    """One batch of PlayGround environments on one GPU (ll_epmc_engine)."""

    def __init__(self, cfg, model_blob, init_state=None, lib_path=None):
        self.lib = load_library(lib_path)
        self.n_envs = int(cfg.n_envs)
        self._pid = os.getpid()
        self.h = C.c_void_p()
        blob = np.ascontiguousarray(model_blob, dtype=np.float64)
        init = np.ascontiguousarray(default_init_state() if init_state is None else init_state, dtype=np.float64)
        assert init.shape == (37,)
        self._chk(self.lib.ll_epmc_create(C.byref(cfg), _ptr(blob), int(blob.size), _ptr(init), C.byref(self.h)))
        self.obs_dim = int(self.lib.ll_epmc_obs_dim(self.h))
        self.n_sub = int((1.0 / cfg.control_freq) / TIME_STEP)                   # PGE:86

    def _chk(self, rc):
        if rc != 0:
            raise capi.LLError(rc, self.lib.ll_last_error().decode())

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            if getattr(self, '_pid', None) == os.getpid():      # (a fork()ed child inherits the object, not the HIP context: it must not destroy it)
                self.lib.ll_epmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:     # noqa: BLE001
            pass

    def reset(self, env_ids=None, draws=None, prev_orn=None):
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        n = self.n_envs if ids is None else len(ids)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.float32).reshape(n, LLE_MAX_DRAWS)
        po = None if prev_orn is None else np.ascontiguousarray(prev_orn, dtype=np.float32).reshape(n, 4)
        self._chk(self.lib.ll_epmc_reset(self.h, _ptr(ids), n, _ptr(d), _ptr(po)))

    def step(self, d_actions_ptr=None):
        self._chk(self.lib.ll_epmc_step(self.h, d_actions_ptr))

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, 12)
        self._chk(self.lib.ll_epmc_set_actions(self.h, _ptr(a)))
        self.step()

    def step_scripted(self, actions, state, ray_hit, ray_frac, draws=None):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, 12)
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(self.n_envs, 37)
        h = np.ascontiguousarray(ray_hit, dtype=np.uint8).reshape(self.n_envs, LLE_N_RAYS)
        f = np.ascontiguousarray(ray_frac, dtype=np.float32).reshape(self.n_envs, LLE_N_RAYS)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.float32).reshape(self.n_envs, -1)
        self._chk(self.lib.ll_epmc_step_scripted(self.h, _ptr(a), _ptr(s), _ptr(h), _ptr(f), _ptr(d), 0 if d is None else d.shape[1]))

    def set_step_draws(self, draws):
        """Uniforms for the draws of the next step only, [n_envs][k] (k may be 0: the step must then not draw)."""
        d = np.ascontiguousarray(draws, dtype=np.float32).reshape(self.n_envs, -1)
        self._chk(self.lib.ll_epmc_set_step_draws(self.h, _ptr(d) if d.shape[1] else None, d.shape[1]))

    def script_reset_rays(self, ray_hit, ray_frac):
        h = np.ascontiguousarray(ray_hit, dtype=np.uint8).reshape(self.n_envs, LLE_N_RAYS)
        f = np.ascontiguousarray(ray_frac, dtype=np.float32).reshape(self.n_envs, LLE_N_RAYS)
        self._chk(self.lib.ll_epmc_script_reset_rays(self.h, _ptr(h), _ptr(f)))

    def fill_random_actions(self, sigma):
        self._chk(self.lib.ll_epmc_fill_random_actions(self.h, float(sigma)))

    def sync(self):
        self._chk(self.lib.ll_epmc_sync(self.h))

    def set_spec(self, **kw):
        """ll_epmc_set_spec_param: the physics-spec switches of include/llenv_model.h (the robot and its solver are the PMC engine's), e.g.
        set_spec(friction_mode=2)."""
        for k, v in kw.items():
            self._chk(self.lib.ll_epmc_set_spec_param(self.h, capi.SPEC_IDS[k], float(v)))

    def get_spec(self, key):
        v = C.c_double()
        self._chk(self.lib.ll_epmc_get_spec_param(self.h, capi.SPEC_IDS[key], C.byref(v)))
        return v.value

    def obs(self):
        o = np.empty((self.n_envs, self.obs_dim), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_obs(self.h, _ptr(o)))
        return o

    def reward_done(self):
        r = np.empty(self.n_envs, dtype=np.float32); d = np.empty(self.n_envs, dtype=np.uint8); w = np.empty(self.n_envs, dtype=np.uint8)
        self._chk(self.lib.ll_epmc_get_reward_done(self.h, _ptr(r), _ptr(d), _ptr(w)))
        return r, d.astype(bool), w

    def state(self):
        s = np.empty((self.n_envs, 37), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_state(self.h, _ptr(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32).reshape(self.n_envs, 37)
        self._chk(self.lib.ll_epmc_set_state(self.h, _ptr(s)))

    def episode(self):
        e = np.empty((self.n_envs, 19), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_episode(self.h, _ptr(e)))
        return {k: e[:, i] for i, k in enumerate(EP_FIELDS)}

    def info(self):
        v = np.empty((self.n_envs, 6), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_info(self.h, _ptr(v)))
        return v

    def statics(self):
        rows = np.empty((self.n_envs, LLE_MAX_STATICS, 8), dtype=np.float32); n = np.empty(self.n_envs, dtype=np.int32)
        self._chk(self.lib.ll_epmc_get_statics(self.h, _ptr(rows), _ptr(n)))
        return rows, n

    def rays(self):
        f = np.empty((self.n_envs, LLE_N_RAYS, 3), dtype=np.float32); t = np.empty_like(f)
        h = np.empty((self.n_envs, LLE_N_RAYS), dtype=np.uint8); fr = np.empty((self.n_envs, LLE_N_RAYS), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_rays(self.h, _ptr(f), _ptr(t), _ptr(h), _ptr(fr)))
        return f, t, h.astype(bool), fr

    def push_trace(self):
        n = C.c_int32(0)
        rows = np.empty((self.n_envs, self.n_sub, 4), dtype=np.float32)
        self._chk(self.lib.ll_epmc_get_push_trace(self.h, _ptr(rows), C.byref(n)))
        assert n.value == self.n_sub
        return rows

    def counters(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.ll_epmc_get_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(env_steps=a.value, episodes=b.value, nonfinite=c.value)

    def device_ptrs(self):
        """Device addresses of obs / reward / done / actions and the engine's stream (gather.engine_tensors, gather.use_engine_stream)."""
        p = capi.LLDevicePtrs()
        self._chk(self.lib.ll_epmc_device_ptrs(self.h, C.byref(p)))
        return p

    def enable_kernel_timing(self, on=True):
        self._chk(self.lib.ll_epmc_enable_kernel_timing(self.h, 1 if on else 0))

    def step_random_n(self, sigma, n_steps):
        """n_steps x {fill_random_actions(sigma); step()} in one launch (ll_epmc_step_random_n)"""
        self._chk(self.lib.ll_epmc_step_random_n(self.h, float(sigma), int(n_steps)))

    def kernel_time_stats(self):
        ms, n, st = C.c_double(), C.c_int(), C.c_int64()
        self._chk(self.lib.ll_epmc_kernel_time_stats(self.h, C.byref(ms), C.byref(n), C.byref(st)))
        return ms.value, n.value, st.value

    def kernel_time_ms(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.lib.ll_epmc_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
