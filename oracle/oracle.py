"""ctypes binding of oracle/pmc_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libpmc_oracle.so')

PROP_IDS = {'joint_pos': 0, 'joint_vel': 1, 'root_lin_vel_loc': 2, 'root_ang_vel_loc': 3, 'e_g': 4}
RW_KEYS = ['joint_pos', 'joint_vel', 'end_effector', 'root_pose', 'root_vel']


class LLConfig(C.Structure):
    """Mirror of ll_config (include/llenv.h)."""
    _fields_ = [('abi_version', C.c_int32), ('n_envs', C.c_int32), ('device', C.c_int32), ('auto_reset', C.c_int32),
                ('control_freq', C.c_double), ('sim_freq', C.c_double), ('kp', C.c_double), ('kd', C.c_double),
                ('max_tau', C.c_double), ('foot_lateral_friction', C.c_double), ('reward_weights', C.c_double * 5),
                ('prop_order', C.c_int32 * 5), ('set_obstacle', C.c_int32), ('obstacle_height', C.c_double),
                ('prioritized_sample_factor', C.c_double), ('solver_iterations', C.c_int32), ('keep_terminal_obs', C.c_int32),
                ('seed', C.c_uint64)]


def make_config(n_envs=1, control_freq=50.0, sim_freq=500.0, kp=50.0, kd=0.5, max_tau=18.0, foot_lateral_friction=0.5,
                reward_weights=None, prop_type=None, prioritized_sample_factor=3.0, auto_reset=0, seed=0, device=0,
                set_obstacle=False, obstacle_height=0.0, solver_iterations=10):
    cfg = LLConfig()
    cfg.abi_version = 2          # include/llenv.h LL_ABI_VERSION
    cfg.n_envs, cfg.device, cfg.auto_reset = n_envs, device, auto_reset
    cfg.control_freq, cfg.sim_freq, cfg.kp, cfg.kd, cfg.max_tau = control_freq, sim_freq, kp, kd, max_tau
    cfg.foot_lateral_friction = foot_lateral_friction
    rw = reward_weights or {'joint_pos': 0.6, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.15, 'root_vel': 0.1}
    for i, k in enumerate(RW_KEYS):
        cfg.reward_weights[i] = rw[k]
    pt = prop_type if prop_type is not None else ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']
    for i in range(5):
        cfg.prop_order[i] = PROP_IDS[pt[i]] if i < len(pt) else -1
    cfg.set_obstacle, cfg.obstacle_height = int(set_obstacle), obstacle_height
    cfg.prioritized_sample_factor = prioritized_sample_factor
    cfg.solver_iterations = solver_iterations
    cfg.seed = seed
    return cfg


def _cpu_key():
    """What -march=native means on this host: the library is rebuilt when the snapshot lands on a different CPU (the GPU box)."""
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    import hashlib
                    return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return 'unknown'


def build(force=False):
    """make (-O3 -march=native, OpenMP); serialised by a file lock so that concurrent test processes do not race on the .so."""
    import fcntl
    os.makedirs(os.path.join(HERE, '_build'), exist_ok=True)
    stamp = os.path.join(HERE, '_build', '.cpu')
    with open(os.path.join(HERE, '_build', '.lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        key = _cpu_key()
        old = open(stamp).read() if os.path.exists(stamp) else ''
        subprocess.check_call(['make', '-C', HERE, '-s'] + (['-B'] if (force or old != key) else []))
        if old != key:
            with open(stamp, 'w') as f:
                f.write(key)
    return LIB


_lib = None
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


def _p(a):
    return a.ctypes.data_as(dp)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_reward.restype = C.c_double
        _lib.orc_energy.restype = C.c_double
        _lib.orc_motion_duration.restype = C.c_double
    return _lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- spec overrides (include/llenv_model.h LLM_SPEC_*; deviation study) ------------------------
SPEC_IDS = dict(limit_gate=0, max_depen_speed=1, link_damping=2, max_contacts_per_leg=3, self_collision=4, self_margin=5, max_self=6,
                erp=7, contact_margin=8, self_friction=9, warm_start=10, trunk_edges=11, select_eps=12,
                friction_mode=13, row_order=14, max_coord_vel=15, limit_erp=16, pair_friction=17, max_pair=18, friction_dirs=19, limit_speculative=20, gyro=21, friction_keep=22, erp_deep=23, erp_deep_below=24, limit_erp_deep=25, leg_edges=26)


def set_spec(**kw):
    for k, v in kw.items():
        if lib().orc_set_spec_param(C.c_int(SPEC_IDS[k]), C.c_double(float(v))) != 0:
            raise ValueError('bad spec override %s=%r' % (k, v))


def reset_spec():
    lib().orc_reset_spec()


# ---- stateless pieces -------------------------------------------------------------------------
def pd_torque(kp, kd, max_tau, q, qd, tgt_joint_pos):
    """LR:119-148 (orc_pd_torque): the torques apply_action hands to PyBullet, for one robot."""
    out = np.zeros(12)
    lib().orc_pd_torque(C.c_double(kp), C.c_double(kd), C.c_double(max_tau), _p(f64(q)), _p(f64(qd)), _p(f64(tgt_joint_pos)), _p(out))
    return out


def mocap_locate(t, frame_step):
    fid, frac = C.c_int32(), C.c_double()
    lib().orc_mocap_locate(C.c_double(t), C.c_double(frame_step), C.byref(fid), C.byref(frac))
    return fid.value, frac.value


def mocap_interp(fc, fn, frac, frame_step):
    out = np.zeros(37)
    lib().orc_mocap_interp(_p(f64(fc)), _p(f64(fn)), C.c_double(frac), C.c_double(frame_step), _p(out))
    return out


def mocap_future(frames_at_fid, frac, frame_step):
    out = np.zeros((4, 37))
    fr = f64(frames_at_fid)
    lib().orc_mocap_future(_p(fr), C.c_double(frac), C.c_double(frame_step), _p(out))
    return out


def prop(state, prop_order=(0, 1, 3, 2, 4)):
    po = (C.c_int32 * 5)(*(list(prop_order) + [-1] * (5 - len(prop_order))))
    out = np.zeros(33)
    n = lib().orc_prop(_p(f64(state)), po, _p(out))
    return out[:n]


def calc_future(base_pos, base_orn, fut76):
    out = np.zeros(72)
    lib().orc_calc_future(_p(f64(base_pos)), _p(f64(base_orn)), _p(f64(fut76)), _p(out))
    return out


def reward(dyn, kin, feet_dyn, feet_kin, weights):
    return lib().orc_reward(_p(f64(dyn)), _p(f64(kin)), _p(f64(feet_dyn).ravel()), _p(f64(feet_kin).ravel()), _p(f64(weights)))


def check_fall(quat):
    return bool(lib().orc_check_fall(_p(f64(quat))))


def check_diverged(dyn, kin):
    return bool(lib().orc_check_diverged(_p(f64(dyn)), _p(f64(kin))))


# ---- batch env ----------------------------------------------------------------------------------
class OracleBatch(object):
    def __init__(self, cfg, model_blob, mocap_table):
        self.cfg = cfg
        blob = f64(model_blob)
        self.h = C.c_void_p(lib().orc_create(C.byref(cfg), _p(blob), C.c_int(blob.size)))
        assert self.h.value, 'orc_create failed'
        fr = f64(mocap_table.frames)
        cl = np.ascontiguousarray(mocap_table.clip_len, dtype=np.int32)
        lib().orc_load_mocap(self.h, _p(fr), cl.ctypes.data_as(ip), C.c_int(len(cl)), C.c_double(mocap_table.frame_step))
        self.n_envs = cfg.n_envs
        self.n_clips = len(cl)
        if cfg.set_obstacle:
            cnt, tab = mocap_table.obstacles()
            tab = f64(tab)
            lib().orc_load_obstacles(self.h, np.ascontiguousarray(cnt, dtype=np.int32).ctypes.data_as(ip), _p(tab), C.c_int(len(cnt)))
        self.obs_dim = lib().orc_obs_dim(self.h)

    def __del__(self):
        if getattr(self, 'h', None) and self.h.value:
            lib().orc_destroy(self.h)
            self.h = None

    def reset_env(self, env, clip, t0):
        obs = np.zeros(self.obs_dim)
        rc = lib().orc_reset_env(self.h, C.c_int(env), C.c_int(int(clip)), C.c_double(t0), _p(obs))
        assert rc == 0, rc
        return obs

    def step_env(self, env, action, scripted_dyn=None, feet_dyn=None, feet_kin=None):
        obs = np.zeros(self.obs_dim)
        r, d = C.c_double(), C.c_int()
        sd = _p(f64(scripted_dyn)) if scripted_dyn is not None else None
        fd = _p(f64(feet_dyn).ravel()) if feet_dyn is not None else None
        fk = _p(f64(feet_kin).ravel()) if feet_kin is not None else None
        lib().orc_step_env(self.h, C.c_int(env), _p(f64(action)), sd, fd, fk, _p(obs), C.byref(r), C.byref(d))
        return obs, r.value, bool(d.value)

    def step_all(self, actions):
        a = f64(actions)
        obs = np.zeros((self.n_envs, self.obs_dim)); rew = np.zeros(self.n_envs); done = np.zeros(self.n_envs, dtype=np.int32)
        lib().orc_step_all(self.h, _p(a), _p(obs), _p(rew), done.ctypes.data_as(ip))
        return obs, rew, done.astype(bool)

    def step_all_mt(self, actions, n_threads):
        """step_all over the host's cores (bench.py's cpu_baseline leg only; table-update order is not deterministic)."""
        a = f64(actions)
        obs = np.zeros((self.n_envs, self.obs_dim)); rew = np.zeros(self.n_envs); done = np.zeros(self.n_envs, dtype=np.int32)
        lib().orc_step_all_mt(self.h, _p(a), _p(obs), _p(rew), done.ctypes.data_as(ip), C.c_int(int(n_threads)))
        return obs, rew, done.astype(bool)

    def get_state(self, env):
        s = np.zeros(37); lib().orc_get_state(self.h, C.c_int(env), _p(s)); return s

    def set_state(self, env, s):
        lib().orc_set_state(self.h, C.c_int(env), _p(f64(s)))

    def get_ref_state(self, env):
        s = np.zeros(37); lib().orc_get_ref_state(self.h, C.c_int(env), _p(s)); return s

    def get_feet(self, env):
        a, b = np.zeros((4, 3)), np.zeros((4, 3)); lib().orc_get_feet(self.h, C.c_int(env), _p(a), _p(b)); return a, b

    def episode_info(self, env):
        clip, steps, reason, fid = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        t, rs, frac = C.c_double(), C.c_double(), C.c_double()
        lib().orc_get_episode_info(self.h, C.c_int(env), C.byref(clip), C.byref(t), C.byref(steps), C.byref(rs), C.byref(reason), C.byref(fid), C.byref(frac))
        return dict(clip=clip.value, time=t.value, steps=steps.value, reward_sum=rs.value, done_reason=reason.value, frame_id=fid.value, frac=frac.value)

    def sampling_table(self):
        p, a, l = np.zeros(self.n_clips), np.zeros(self.n_clips), np.zeros(self.n_clips)
        lib().orc_get_sampling_table(self.h, _p(p), _p(a), _p(l)); return p, a, l

    def set_sampling_table(self, prob, avg_r):
        lib().orc_set_sampling_table(self.h, _p(f64(prob)), _p(f64(avg_r)))

    def meta(self):
        m, fr = C.c_int32(), C.c_int32(); ms = np.zeros(self.n_clips)
        lib().orc_get_margin(self.h, C.byref(m), C.byref(fr), _p(ms)); return m.value, fr.value, ms

    def motion_duration(self, clip):
        return lib().orc_motion_duration(self.h, C.c_int(clip))

    def fk_feet(self, state):
        out = np.zeros((4, 3)); lib().orc_fk_feet(self.h, _p(f64(state)), _p(out)); return out

    def forward_dynamics(self, state, tau):
        acc = np.zeros(18); rc = lib().orc_forward_dynamics(self.h, _p(f64(state)), _p(f64(tau)), _p(acc)); assert rc == 0; return acc

    def substep(self, state, tau):
        s = f64(state).copy(); nc = C.c_int32(); lam = np.zeros(12 + 3 * 24); acc = np.zeros(18)
        rc = lib().orc_substep(self.h, _p(s), _p(f64(tau)), C.byref(nc), _p(lam), _p(acc)); assert rc == 0
        return s, nc.value, lam, acc

    def substep_terrain(self, state, tau, mu_foot, shapes, box_mu_scale, push=None):
        """One substep with EPMC terrain (records x0 x1 y0 y1 z0 z1 rod r) and the push on the FR hip link (epmc parity tests)."""
        s = f64(state).copy(); nc = C.c_int32(); lam = np.zeros(12 + 3 * 24)
        sh = f64(np.asarray(shapes).reshape(-1, 8)) if len(shapes) else np.zeros((0, 8))
        pu = None if push is None else f64(push)
        rc = lib().orc_substep_terrain(self.h, _p(s), _p(f64(tau)), C.c_double(mu_foot), C.c_int(len(sh)), _p(sh) if len(sh) else None, C.c_double(box_mu_scale),
                                       _p(pu) if pu is not None else None, C.byref(nc), _p(lam))
        assert rc == 0
        return s, nc.value, lam

    @staticmethod
    def selection_margin(reset=True):
        """How close the find_contacts calls of this thread since the last reset came to the deepest-K rule's discontinuity (a candidate's depth
        crossing deepest + LLM_SELECT_EPS), in metres: parity tests set cases aside in which float32 rounding can flip the pick."""
        f = lib().orc_selection_margin
        f.restype = C.c_double
        return float(f(C.c_int(1 if reset else 0)))

    def list_contacts(self, state, mu_foot, shapes, box_mu_scale):
        """The contacts kept in this configuration: rows [leg, candidate index (9 sub + jj), depth, P(3), n(3), mu, body] (tests)."""
        out = np.zeros((16, 11))
        sh = f64(np.asarray(shapes).reshape(-1, 8)) if len(shapes) else np.zeros((0, 8))
        n = lib().orc_list_contacts(self.h, _p(f64(state)), C.c_double(mu_foot), C.c_int(len(sh)), _p(sh) if len(sh) else None, C.c_double(box_mu_scale), _p(out))
        return out[:n]

    def substep_pair(self, state0, state1, tau0, tau1, mu_foot, shapes0, shapes1, box_mu_scale, push0=None, push1=None):
        """One substep of the two robots of a SEPMC arena (sepmc parity tests).  Returns (state0, state1, shared rows [n][8])."""
        s0, s1 = f64(state0).copy(), f64(state1).copy()
        sh = [f64(np.asarray(x).reshape(-1, 8)) if len(x) else np.zeros((0, 8)) for x in (shapes0, shapes1)]
        pu = [None if x is None else f64(x) for x in (push0, push1)]
        rows = np.zeros((2, 8))
        n = lib().orc_substep_pair(self.h, _p(s0), _p(s1), _p(f64(tau0)), _p(f64(tau1)), C.c_double(mu_foot), C.c_int(len(sh[0])), _p(sh[0]) if len(sh[0]) else None,
                                   C.c_int(len(sh[1])), _p(sh[1]) if len(sh[1]) else None, C.c_double(box_mu_scale), _p(pu[0]) if pu[0] is not None else None,
                                   _p(pu[1]) if pu[1] is not None else None, _p(rows))
        assert n >= 0
        return s0, s1, rows[:n]

    def touch(self, state0, state1, shapes0, flag0, shapes1, flag1):
        """SEPMC contact classes of both robots: [[static, flag, robot] of robot 0, ... of robot 1]; flag* = index of the flag in shapes* (-1 none)."""
        sh = [f64(np.asarray(x).reshape(-1, 8)) if len(x) else np.zeros((0, 8)) for x in (shapes0, shapes1)]
        out = np.zeros(6, dtype=np.int32)
        lib().orc_touch(self.h, _p(f64(state0)), _p(f64(state1)), C.c_int(len(sh[0])), _p(sh[0]) if len(sh[0]) else None, C.c_int(flag0), C.c_int(len(sh[1])),
                        _p(sh[1]) if len(sh[1]) else None, C.c_int(flag1), _p(out))
        return out.reshape(2, 3)

    def momentum(self, state):
        out = np.zeros(6); lib().orc_momentum(self.h, _p(f64(state)), _p(out)); return out

    def energy(self, state):
        return lib().orc_energy(self.h, _p(f64(state)))


def set_link_damping(k):
    lib().orc_set_link_damping(C.c_double(k))


def set_self_collision(on):
    lib().orc_set_self_collision(C.c_int(1 if on else 0))
