"""ctypes binding of the C ABI in include/llenv.h (libllenv.so, built from csrc/llenv.hip for gfx950).

This is the thin layer the north star asks for: Python keeps the reference's env API (envs.py) and calls
the HIP batch stepper through plain pointers and sizes.  There is no CPU fallback: if the shared library is
missing, or no GPU is visible, constructing an Engine raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, 'csrc', 'libllenv.so')

PROP_IDS = {'joint_pos': 0, 'joint_vel': 1, 'root_lin_vel_loc': 2, 'root_ang_vel_loc': 3, 'e_g': 4}   # PLE:102-108
PROP_SIZES = {'joint_pos': 12, 'joint_vel': 12, 'root_lin_vel_loc': 3, 'root_ang_vel_loc': 3, 'e_g': 3}
RW_KEYS = ['joint_pos', 'joint_vel', 'end_effector', 'root_pose', 'root_vel']                          # PLE:352-357
PLE_DEFAULT_REWARD_WEIGHTS = {'joint_pos': 0.6, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.15,
                              'root_vel': 0.1}                                                       # PLE:359-363

DONE_FALL, DONE_CLIP_END, DONE_DIVERGED, DONE_COLLISION, DONE_NONFINITE = 1, 2, 4, 8, 16
SPEC_IDS = dict(limit_gate=0, max_depen_speed=1, link_damping=2, max_contacts_per_leg=3, self_collision=4, self_margin=5, max_self=6,
                erp=7, contact_margin=8, self_friction=9, warm_start=10, trunk_edges=11, select_eps=12,
                friction_mode=13, row_order=14, max_coord_vel=15, limit_erp=16, pair_friction=17, max_pair=18, friction_dirs=19, limit_speculative=20, gyro=21, friction_keep=22, erp_deep=23, erp_deep_below=24, limit_erp_deep=25, leg_edges=26)                                       # include/llenv_model.h LLM_SPEC_*
LL_SELECT_EPS = 1e-5                                                                                         # include/llenv_model.h LLM_SELECT_EPS
LLM_FRICTION_MODE = 2                                                                                        # include/llenv_model.h LLM_FRICTION_MODE: cone-coupled friction (0: the pyramid)
LL_DONE_FALL, LL_DONE_CLIP_END, LL_DONE_DIVERGED, LL_DONE_COLLISION, LL_DONE_NONFINITE = 1, 2, 4, 8, 16      # include/llenv.h:65-69
LL_OK, LL_EINVAL, LL_ENOMEM, LL_EHIP, LL_ESTATE, LL_ENODEV = 0, -1, -2, -3, -4, -5                           # include/llenv.h:57-62


class LLConfig(C.Structure):
    """struct ll_config (include/llenv.h)."""
    _fields_ = [('abi_version', C.c_int32), ('n_envs', C.c_int32), ('device', C.c_int32), ('auto_reset', C.c_int32),
                ('control_freq', C.c_double), ('sim_freq', C.c_double), ('kp', C.c_double), ('kd', C.c_double),
                ('max_tau', C.c_double), ('foot_lateral_friction', C.c_double), ('reward_weights', C.c_double * 5),
                ('prop_order', C.c_int32 * 5), ('set_obstacle', C.c_int32), ('obstacle_height', C.c_double),
                ('prioritized_sample_factor', C.c_double), ('solver_iterations', C.c_int32), ('keep_terminal_obs', C.c_int32),
                ('seed', C.c_uint64)]


class LLDevicePtrs(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('done', C.c_void_p), ('done_reason', C.c_void_p),
                ('actions', C.c_void_p), ('terminal_obs', C.c_void_p), ('obs_dim', C.c_int32), ('n_envs', C.c_int32),
                ('stream', C.c_void_p)]


class LLError(RuntimeError):
    def __init__(self, code, msg):
        super(LLError, self).__init__('llenv error %d: %s' % (code, msg))
        self.code = code


def make_config(n_envs, control_freq=25.0, sim_freq=500.0, kp=50.0, kd=1.0, max_tau=18.0, foot_lateral_friction=0.5,
                reward_weights=None, prop_type=None, prioritized_sample_factor=0.0, set_obstacle=False,
                obstacle_height=0.0, auto_reset=0, seed=0, device=0, solver_iterations=10, keep_terminal_obs=False):
    """Defaults are the factory's (create_pybullet_envs.py:28-59)."""
    if not isinstance(prop_type, (list, tuple)):
        raise TypeError("Expected 'prop_type' to be a list.")                     # PLE:113
    cfg = LLConfig()
    cfg.abi_version = LL_ABI_VERSION
    cfg.n_envs, cfg.device, cfg.auto_reset = int(n_envs), int(device), int(auto_reset)
    cfg.control_freq, cfg.sim_freq, cfg.kp, cfg.kd = float(control_freq), float(sim_freq), float(kp), float(kd)
    cfg.max_tau, cfg.foot_lateral_friction = float(max_tau), float(foot_lateral_friction)
    rw = reward_weights if reward_weights is not None else PLE_DEFAULT_REWARD_WEIGHTS
    for i, k in enumerate(RW_KEYS):
        cfg.reward_weights[i] = float(rw[k])
    if len(prop_type) > 5 or len(set(prop_type)) != len(prop_type):
        raise ValueError('prop_type entries must be distinct keys of %s' % sorted(PROP_IDS))
    for i in range(5):
        cfg.prop_order[i] = PROP_IDS[prop_type[i]] if i < len(prop_type) else -1   # KeyError mirrors PLE:111
    cfg.set_obstacle, cfg.obstacle_height = int(bool(set_obstacle)), float(obstacle_height)
    cfg.prioritized_sample_factor = float(prioritized_sample_factor)
    cfg.solver_iterations = int(solver_iterations)
    cfg.keep_terminal_obs = int(bool(keep_terminal_obs))
    cfg.seed = int(seed)
    return cfg


LL_ABI_VERSION = 2          # include/llenv.h; checked against the library in load_library

_SIGS = {
    'll_last_error': (C.c_char_p, []),
    'll_abi_version': (C.c_int, []),
    'll_model_blob_len': (C.c_int, []),
    'll_create': (C.c_int, [C.POINTER(LLConfig), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    'll_destroy': (C.c_int, [C.c_void_p]),
    'll_load_mocap': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]),
    'll_load_mocap_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]),
    'll_load_obstacles': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'll_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'll_step': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_step_scripted': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_probe_pd_torque': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'll_set_spec_param': (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    'll_get_spec_param': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    'll_fill_random_actions': (C.c_int, [C.c_void_p, C.c_float]),
    'll_step_random': (C.c_int, [C.c_void_p, C.c_float]),
    'll_step_random_n': (C.c_int, [C.c_void_p, C.c_float, C.c_int]),
    'll_sync': (C.c_int, [C.c_void_p]),
    'll_set_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_enable_unrolls': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    'll_pg_ptrs': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'll_pg_mark_current': (C.c_int, [C.c_void_p]),
    'll_unroll_position': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'll_finish_unroll': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    'll_device_ptrs': (C.c_int, [C.c_void_p, C.POINTER(LLDevicePtrs)]),
    'll_get_obs': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_terminal_obs': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_reward_done': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_set_actions': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_set_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_ref_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_episode_info': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_get_sampling_table': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_set_sampling_table': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_feet': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_get_counters': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'll_get_table_sync': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_get_episode_histogram': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_kernel_time_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'll_enable_kernel_timing': (C.c_int, [C.c_void_p, C.c_int]),
    'll_kernel_time_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)

_libs = {}


def load_library(path=None):
    path = os.path.abspath(path or DEFAULT_LIB)
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)           # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.ll_abi_version() != LL_ABI_VERSION:
            raise ImportError('%s speaks ABI version %d, this binding %d: rebuild the library (__graft_entry__.build())' % (path, lib.ll_abi_version(), LL_ABI_VERSION))
        _libs[path] = lib
    return _libs[path]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine(object):
    """One batch of environments on one GPU (ll_engine)."""

    def __init__(self, cfg, model_blob, mocap_table, lib_path=None):
        self.lib = load_library(lib_path)
        self.cfg = cfg
        self._pid = os.getpid()      # a handle is only ever destroyed by the process that created it (a fork()ed child inherits the Python object, not the HIP context)
        self.h = C.c_void_p()
        blob = np.ascontiguousarray(model_blob, dtype=np.float64)
        self._chk(self.lib.ll_create(C.byref(cfg), _ptr(blob), int(blob.size), C.byref(self.h)))
        frames = np.ascontiguousarray(mocap_table.frames)
        clip_len = np.ascontiguousarray(mocap_table.clip_len, dtype=np.int32)
        if frames.dtype == np.float64:
            self._chk(self.lib.ll_load_mocap_f64(self.h, _ptr(frames), _ptr(clip_len), len(clip_len), float(mocap_table.frame_step)))
        else:
            frames = frames.astype(np.float32)
            self._chk(self.lib.ll_load_mocap(self.h, _ptr(frames), _ptr(clip_len), len(clip_len), float(mocap_table.frame_step)))
        self.n_envs = cfg.n_envs
        self.n_clips = len(clip_len)
        if cfg.set_obstacle:                                   # PLE:159-160 / utils/obstacle.py
            cnt, tab = mocap_table.obstacles()
            tab = np.ascontiguousarray(tab, dtype=np.float64)
            self._chk(self.lib.ll_load_obstacles(self.h, _ptr(cnt), _ptr(tab) if len(tab) else None, len(cnt)))
        p = self.device_ptrs()
        self.obs_dim = p.obs_dim

    def _chk(self, rc):
        if rc != 0:
            raise LLError(rc, self.lib.ll_last_error().decode())

    def close(self):
        if self.h:
            if getattr(self, '_pid', None) == os.getpid():
                self.lib.ll_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- control ------------------------------------------------------------------------------------
    def reset(self, env_ids=None, clip=None, t0=None):
        ids = np.ascontiguousarray(env_ids, dtype=np.int32) if env_ids is not None else None
        n = len(ids) if ids is not None else self.n_envs
        cl = np.ascontiguousarray(clip, dtype=np.int32) if clip is not None else None
        tt = np.ascontiguousarray(t0, dtype=np.float64) if t0 is not None else None
        for a in (cl, tt):
            if a is not None and len(a) != n:
                raise ValueError('clip / t0 must have one entry per reset env')
        self._chk(self.lib.ll_reset(self.h, _ptr(ids), n, _ptr(cl), _ptr(tt)))

    def step(self, d_actions_ptr=None):
        """d_actions_ptr: integer device address of float32 [n_envs][12], or None for the engine's action buffer."""
        self._chk(self.lib.ll_step(self.h, C.c_void_p(d_actions_ptr) if d_actions_ptr else None))

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, 12)
        self._chk(self.lib.ll_set_actions(self.h, _ptr(a)))
        self.step(None)

    def step_scripted(self, actions, state, feet=None):
        """Parity hook (ll_step_scripted): host actions [n][12], physics result state [n][37], optional feet [n][2][4][3]."""
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n_envs, 12)
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(self.n_envs, 37)
        f = np.ascontiguousarray(feet, dtype=np.float32).reshape(self.n_envs, 24) if feet is not None else None
        self._chk(self.lib.ll_set_actions(self.h, _ptr(a)))
        self._chk(self.lib.ll_step_scripted(self.h, None, _ptr(s), _ptr(f)))

    def probe_pd_torque(self, q, qd, x, mode=0):
        """Parity probe (ll_probe_pd_torque): the PD torques of LR:137-141 for rows of joint_pos, joint_vel and a target (mode 0) or action (mode 1)."""
        rows = np.ascontiguousarray(np.concatenate([np.asarray(q), np.asarray(qd), np.asarray(x)], axis=1), dtype=np.float32)
        assert rows.shape[1] == 36
        tau = np.empty((len(rows), 12), dtype=np.float32)
        self._chk(self.lib.ll_probe_pd_torque(self.h, _ptr(rows), len(rows), int(mode), _ptr(tau)))
        return tau

    def set_spec(self, **kw):
        """Deviation study (ll_set_spec_param): e.g. set_spec(limit_gate=1e30, max_contacts_per_leg=2)."""
        for k, v in kw.items():
            self._chk(self.lib.ll_set_spec_param(self.h, SPEC_IDS[k], float(v)))

    def get_spec(self, key):
        v = C.c_double()
        self._chk(self.lib.ll_get_spec_param(self.h, SPEC_IDS[key], C.byref(v)))
        return v.value

    def fill_random_actions(self, sigma):
        self._chk(self.lib.ll_fill_random_actions(self.h, float(sigma)))

    def step_random(self, sigma):
        """fill_random_actions(sigma) + step() as one kernel launch."""
        self._chk(self.lib.ll_step_random(self.h, float(sigma)))

    def step_random_n(self, sigma, n_steps):
        """n_steps iterations of the random-policy loop in ONE launch (ll_step_random_n)."""
        self._chk(self.lib.ll_step_random_n(self.h, float(sigma), int(n_steps)))

    def sync(self):
        self._chk(self.lib.ll_sync(self.h))

    def set_stream(self, stream_handle):
        if stream_handle is not None and int(stream_handle) == 0:
            # torch's default stream reports handle 0, which ll_set_stream reads as "back to the private stream": work queued by
            # torch and by the engine would then be unordered.  Share an explicit stream instead (gather.bind_torch_stream).
            raise ValueError('the legacy default stream (handle 0) cannot be shared; pass a torch.cuda.Stream handle, or None for the private stream')
        self._chk(self.lib.ll_set_stream(self.h, C.c_void_p(int(stream_handle)) if stream_handle is not None else None))

    def enable_unrolls(self, unroll_length, n_buffers=2):
        """-> (device address, row_floats) of the [n_buffers][n_envs][unroll_length][row_floats] unroll buffers every step writes into"""
        buf, w = C.c_void_p(), C.c_int()
        self._chk(self.lib.ll_enable_unrolls(self.h, int(unroll_length), int(n_buffers), C.byref(buf), C.byref(w)))
        return buf.value, w.value

    def pg_ptrs(self):
        """-> device addresses of the [n_envs] neglogp and value buffers a policy fills next to the action buffer"""
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.ll_pg_ptrs(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pg_mark_current(self):
        """the pg buffers now hold the policy's outputs for the current observation (ll_pg_mark_current)"""
        self._chk(self.lib.ll_pg_mark_current(self.h))

    def unroll_position(self):
        """-> (index of the unroll the next step writes into, its time step there)"""
        k, t = C.c_int64(), C.c_int()
        self._chk(self.lib.ll_unroll_position(self.h, C.byref(k), C.byref(t)))
        return k.value, t.value

    def finish_unroll(self, buffer, gamma, lam, d_bootstrap_value=None):
        self._chk(self.lib.ll_finish_unroll(self.h, int(buffer), float(gamma), float(lam), C.c_void_p(int(d_bootstrap_value)) if d_bootstrap_value else None))

    def device_ptrs(self):
        p = LLDevicePtrs()
        self._chk(self.lib.ll_device_ptrs(self.h, C.byref(p)))
        return p

    # ---- host copies ------------------------------------------------------------------------------------
    def obs(self):
        o = np.empty((self.n_envs, self.obs_dim), dtype=np.float32)
        self._chk(self.lib.ll_get_obs(self.h, _ptr(o)))
        return o

    def terminal_obs(self):
        o = np.empty((self.n_envs, self.obs_dim), dtype=np.float32)
        self._chk(self.lib.ll_get_terminal_obs(self.h, _ptr(o)))
        return o

    def reward_done(self):
        r = np.empty(self.n_envs, dtype=np.float32)
        d = np.empty(self.n_envs, dtype=np.uint8)
        why = np.empty(self.n_envs, dtype=np.uint8)
        self._chk(self.lib.ll_get_reward_done(self.h, _ptr(r), _ptr(d), _ptr(why)))
        return r, d.astype(bool), why

    def state(self):
        s = np.empty((self.n_envs, 37), dtype=np.float32)
        self._chk(self.lib.ll_get_state(self.h, _ptr(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32).reshape(self.n_envs, 37)
        self._chk(self.lib.ll_set_state(self.h, _ptr(s)))

    def ref_state(self):
        s = np.empty((self.n_envs, 37), dtype=np.float32)
        self._chk(self.lib.ll_get_ref_state(self.h, _ptr(s)))
        return s

    def feet(self):
        a = np.empty((self.n_envs, 4, 3), dtype=np.float32)
        b = np.empty((self.n_envs, 4, 3), dtype=np.float32)
        self._chk(self.lib.ll_get_feet(self.h, _ptr(a), _ptr(b)))
        return a, b

    def episode_info(self):
        clip = np.empty(self.n_envs, dtype=np.int32); steps = np.empty(self.n_envs, dtype=np.int32)
        t = np.empty(self.n_envs, dtype=np.float64); rs = np.empty(self.n_envs, dtype=np.float32)
        self._chk(self.lib.ll_get_episode_info(self.h, _ptr(clip), _ptr(t), _ptr(steps), _ptr(rs)))
        return dict(clip=clip, time=t, steps=steps, reward_sum=rs)

    def sampling_table(self):
        p, a, l = (np.empty(self.n_clips) for _ in range(3))
        self._chk(self.lib.ll_get_sampling_table(self.h, _ptr(p), _ptr(a), _ptr(l)))
        return p, a, l

    def set_sampling_table(self, avg_reward_sum):
        a = np.ascontiguousarray(avg_reward_sum, dtype=np.float64)
        assert a.shape == (self.n_clips,)
        self._chk(self.lib.ll_set_sampling_table(self.h, _ptr(a)))

    def table_sync(self):
        """episodes that re-seeded inside a multi-step launch from an older table version than the exact one (ll_get_table_sync; 0 unless the chip was shared)"""
        a = C.c_uint64()
        self._chk(self.lib.ll_get_table_sync(self.h, C.byref(a)))
        return a.value

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.ll_get_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(env_steps=a.value, episodes=b.value, nonfinite=c.value)

    def episode_histogram(self):
        """finished episodes by length: counts[b] = episodes of 2^b .. 2^(b+1) - 1 control steps (b = 15: and longer)"""
        c = np.zeros(16, dtype=np.uint64)
        self._chk(self.lib.ll_get_episode_histogram(self.h, _ptr(c)))
        return c

    def enable_kernel_timing(self, on=True):
        self._chk(self.lib.ll_enable_kernel_timing(self.h, int(on)))

    def kernel_time_ms(self):
        ms, n = C.c_double(), C.c_int()
        self._chk(self.lib.ll_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_stats(self):
        """-> (average ms per launch, launches, control steps those launches ran)"""
        ms, n, st = C.c_double(), C.c_int(), C.c_int64()
        self._chk(self.lib.ll_kernel_time_stats(self.h, C.byref(ms), C.byref(n), C.byref(st)))
        return ms.value, n.value, st.value
