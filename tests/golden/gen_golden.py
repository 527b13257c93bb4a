"""Generate golden vectors by IMPORTING the reference Python (build container only).

The reference needs gym / pybullet / tleague, none of which exist here; tiny stub
modules stand in for them.  Only the reference's own code
computes the expected values; the stubs merely hold state.  The fake BulletClient
cannot step physics: a scripted list of dynamic-robot states is injected instead
(one per control step), and foot positions returned by ``getLinkStates`` are
scripted inputs too.  Outputs -> tests/golden/pmc_golden.npz (inputs + expected).

    python tests/golden/gen_golden.py
"""
import os
import sys
import time
import types
from collections import OrderedDict

import numpy as np

REF_SRC = '/root/reference/src'
MOCAP_DIR = '/root/reference/data/mocap_data'
HERE = os.path.dirname(os.path.abspath(__file__))

JOINT_NAMES = ['joint_FR1', 'joint_FR2', 'joint_FR3', 'joint_FR4', 'joint_FRW', 'joint_FL1', 'joint_FL2',
               'joint_FL3', 'joint_FL4', 'joint_FLW', 'joint_HR1', 'joint_HR2', 'joint_HR3', 'joint_HR4',
               'joint_HRW', 'joint_HL1', 'joint_HL2', 'joint_HL3', 'joint_HL4', 'joint_HLW',
               'joint_front_handle', 'joint_hind_handle']          # URDF order (max.urdf)


# ----------------------------------------------------------------------------- stubs
def install_stubs():
    tl = types.ModuleType('tleague'); tlu = types.ModuleType('tleague.utils'); tll = types.ModuleType('tleague.utils.logger')
    tll.log = lambda *a, **k: None
    tl.utils = tlu; tlu.logger = tll
    sys.modules.update({'tleague': tl, 'tleague.utils': tlu, 'tleague.utils.logger': tll})

    gym = types.ModuleType('gym'); spaces = types.ModuleType('gym.spaces')

    class Env(object):
        pass

    class Wrapper(object):
        def __init__(self, env):
            self.env = env

        def step(self, a):
            return self.env.step(a)

    class Box(object):
        def __init__(self, lo, hi, shape=None):
            self.shape = shape

    class Dict(object):
        def __init__(self, d):
            self.spaces = d

    class Tuple(object):
        def __init__(self, l):
            self.spaces = l
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    spaces.Box, spaces.Dict, spaces.Tuple = Box, Dict, Tuple
    sys.modules.update({'gym': gym, 'gym.spaces': spaces})

    pb = types.ModuleType('pybullet'); pb.GUI, pb.DIRECT = 1, 2
    pbu = types.ModuleType('pybullet_utils'); bc = types.ModuleType('pybullet_utils.bullet_client')
    bc.BulletClient = FakeBulletClient
    pbu.bullet_client = bc
    sys.modules.update({'pybullet': pb, 'pybullet_utils': pbu, 'pybullet_utils.bullet_client': bc})
    time.sleep = lambda s: None                                    # defeat the real-time throttle (PLE:241-244)


class FakeBulletClient(object):
    """State-holding stand-in: stores what reset* writes, returns it from get*."""

    def __init__(self, connection_mode=None):
        self.bodies = []
        self.script = []            # scripted dyn states (37,) consumed one per 10 stepSimulation calls
        self.n_sim = 0
        self.feet = {0: np.zeros((4, 3)), 1: np.zeros((4, 3))}     # scripted getLinkStates positions

    def __getattr__(self, name):
        if name.isupper():
            return 0
        return lambda *a, **k: None

    def loadURDF(self, *a, **k):
        self.bodies.append(dict(p=[0, 0, 0], q=[0, 0, 0, 1], v=[0, 0, 0], w=[0, 0, 0], j=np.zeros((22, 2))))
        return len(self.bodies) - 1

    def getNumJoints(self, i):
        return 22

    def getJointInfo(self, i, j):
        return (j, JOINT_NAMES[j].encode())

    def resetBasePositionAndOrientation(self, i, p, q):
        self.bodies[i]['p'], self.bodies[i]['q'] = list(p), list(q)

    def resetBaseVelocity(self, i, v, w):
        self.bodies[i]['v'], self.bodies[i]['w'] = list(v), list(w)

    def resetJointState(self, i, j, p, v=0.0):
        self.bodies[i]['j'][j] = (p, v)

    def getBasePositionAndOrientation(self, i):
        return tuple(self.bodies[i]['p']), tuple(self.bodies[i]['q'])

    def getBaseVelocity(self, i):
        return tuple(self.bodies[i]['v']), tuple(self.bodies[i]['w'])

    def getJointStates(self, i, idx):
        return [(self.bodies[i]['j'][j][0], self.bodies[i]['j'][j][1], None, 0.0) for j in idx]

    def getLinkStates(self, i, idx, **k):
        return [(tuple(self.feet[i][n]), None, None, None, None, None, (0.0, 0.0, 0.0), None) for n in range(len(idx))]

    def getContactPoints(self, **k):
        return []

    def isConnected(self):
        return 0

    def stepSimulation(self):
        self.n_sim += 1
        if self.n_sim % 10 == 0 and self.script:
            s = self.script.pop(0)
            set_dyn(self, s)


LEG_IDX = [0, 1, 2, 5, 6, 7, 10, 11, 12, 15, 16, 17]


def set_dyn(client, s, body=0):
    b = client.bodies[body]
    b['p'], b['q'], b['v'], b['w'] = list(s[0:3]), list(s[3:7]), list(s[7:10]), list(s[10:13])
    for n, j in enumerate(LEG_IDX):
        b['j'][j] = (s[13 + n], s[25 + n])


def state_vec(d):
    return np.concatenate([d['base_pos'], d['base_orn'], d['base_lin_vel'], d['base_ang_vel'], d['joint_pos'], d['joint_vel']]).astype(np.float64)


def fut_vec(f):
    return np.concatenate([np.concatenate([x['base_pos'], x['base_orn'], x['joint_pos']]) for x in f]).astype(np.float64)


def obs_vec(o):
    return np.concatenate([o['prop'], o['prop_a'], o['future']]).astype(np.float64)


def rand_state(rng, big=False):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if not big:                      # near upright, like a running robot
        from scipy.spatial.transform import Rotation as R
        q = R.from_euler('xyz', rng.uniform(-0.5, 0.5, 3) * [1, 1, 6]).as_quat()
    return np.concatenate([rng.uniform(-2, 2, 2), rng.uniform(0.1, 0.6, 1), q, rng.normal(size=3), rng.normal(size=3) * 2,
                           rng.uniform(-1.5, 1.5, 12), rng.normal(size=12) * 3])


PMC_CONFIG = dict(                                                   # test_primitive_level_env.py:18-38
    arena_id='LeggedRobotTracking', render=False, data_path=MOCAP_DIR, control_freq=50.0,
    prop_type=['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
    prioritized_sample_factor=3.0, set_obstacle=False, obstacle_height=0.2, kp=50.0, kd=0.5, max_tau=18,
    reward_weights={'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05})


def main():
    install_stubs()
    sys.path.insert(0, REF_SRC)
    from lifelike.sim_envs.pybullet_envs.create_pybullet_envs import create_tracking_game
    from lifelike.sim_envs.pybullet_envs.primitive_level_env.motion_lib import MotionLib
    from lifelike.sim_envs.pybullet_envs.primitive_level_env.primitive_level_env import PrimitiveLevelEnv
    from lifelike.sim_envs.pybullet_envs.legged_robot.legged_robot import LeggedRobot
    from lifelike.utils.obstacle import obstacles_in_frame, get_obstacle_pose
    from lifelike.utils import constants as C

    out = OrderedDict()
    rng = np.random.default_rng(20240807)
    names = sorted(f for f in os.listdir(MOCAP_DIR) if f.endswith('txt'))
    out['clip_names'] = np.array(names)
    sub_names = ['dog_quad_walkrun_001_ret.txt', 'dog_jump_002_ret.txt', 'dog_play_001_ret_mir.txt']

    # ---- K1: MotionLib meta (SURVEY §4 KAT K1) --------------------------------------
    ml_all = MotionLib(MOCAP_DIR, 0.02)
    out['k1_frame_step'] = np.float64(ml_all.frame_step)
    out['k1_margin'] = np.int64(ml_all.margin)
    out['k1_data_len'] = np.array(ml_all.data_len)
    out['k1_max_steps'] = np.array(ml_all.max_steps)

    # ---- G1: mocap interpolation at (clip, t0, n_substeps) ----------------------------
    g1_clip, g1_t0, g1_n, g1_state, g1_fut, g1_end, g1_fid, g1_frac = [], [], [], [], [], [], [], []
    for sn in sub_names:
        c = names.index(sn)
        ml_all.frames = ml_all.data[c]; ml_all.num_frames = len(ml_all.frames)
        dur = ml_all.frame_step * (ml_all.num_frames - ml_all.margin - 1)
        for k in range(50):
            t0 = rng.uniform(0, 1) * dur
            n = int(rng.integers(0, 400)) if k % 5 else 0
            if k == 7:
                t0, n = dur - 0.05, 30           # runs just past the end-of-clip test
            if k == 8:
                t0, n = 0.0, 0
            if k == 9:
                t0, n = 5 * ml_all.frame_step, 0  # exactly on a frame boundary
            n = min(n, int((dur - t0) / 0.002) + 8)   # the env terminates at is_ended(); never run far past it
            t = t0
            ml_all.step(t0)                        # reset() leaves frame_id/frac at t0 (ML:52-53)
            for _ in range(n):
                ml_all.step(t)
                t += 0.002
            g1_clip.append(c); g1_t0.append(t0); g1_n.append(n)
            g1_state.append(state_vec(ml_all.get_states_info()))
            g1_fut.append(fut_vec(ml_all.get_states_info_future()))
            g1_end.append(ml_all.is_ended()); g1_fid.append(ml_all.frame_id); g1_frac.append(ml_all.frame_fraction)
    out.update(g1_clip=np.array(g1_clip), g1_t0=np.array(g1_t0), g1_n=np.array(g1_n), g1_state=np.array(g1_state),
               g1_future=np.array(g1_fut), g1_ended=np.array(g1_end), g1_frame_id=np.array(g1_fid), g1_frac=np.array(g1_frac))

    # ---- full env through the fake client -----------------------------------------------
    env = create_tracking_game(**PMC_CONFIG)
    ple = env.env
    client = ple._bullet_client
    out['obs_space_shapes'] = np.array([ple.observation_space.spaces[k].shape[0] for k in ['prop', 'prop_a', 'future']])

    # ---- G2: reset obs for 64 seeds, restricted to golden-subset clips for f64 pinning --
    g2 = dict(seed=[], clip=[], t0=[], obs=[], kin=[])
    seed = 0
    while len(g2['seed']) < 64:
        np.random.seed(seed)
        ple._prioritized_sample_probability[:] = 1.0 / len(names)
        o = env.reset()[0]
        if True:
            g2['seed'].append(seed); g2['clip'].append(ple.sampled_data_idx); g2['t0'].append(ple.time)
            g2['obs'].append(obs_vec(o)); g2['kin'].append(state_vec(ple._legged_robot_kin.get_states_info()))
        seed += 1
    out.update({'g2_' + k: np.array(v) for k, v in g2.items()})

    # ---- K4: reset on the walkrun clip alone, seed 123 ------------------------
    cfg1 = dict(PMC_CONFIG); cfg1['data_path'] = os.path.join(MOCAP_DIR, 'dog_quad_walkrun_001_ret.txt')
    env1 = create_tracking_game(**cfg1)
    np.random.seed(123)
    o = env1.reset()[0]
    out['k4_t0'] = np.float64(env1.env.time); out['k4_obs'] = obs_vec(o)

    # ---- G3: prop + future at random dyn states -------------------------------------------
    g3_state, g3_prop, g3_fut_in, g3_fut = [], [], [], []
    for k in range(256):
        s = rand_state(rng, big=(k % 4 == 0))
        d = dict(base_pos=s[0:3], base_orn=s[3:7], base_lin_vel=s[7:10], base_ang_vel=s[10:13], joint_pos=list(s[13:25]), joint_vel=list(s[25:37]))
        prop = ple._cfg_prop(ple._prepare_full_prop(d), PMC_CONFIG['prop_type'])
        fut_states = []
        for h in range(4):
            f = rand_state(rng, big=(k % 8 == 0))
            if k % 16 == 1:
                f[3:7] = s[3:7]                      # zero relative rotation (angle ~ 0 branch)
            fut_states.append(dict(base_pos=list(f[0:3]), base_orn=list(f[3:7]), joint_pos=list(f[13:25])))
        fut = PrimitiveLevelEnv.calculate_future(list(s[0:3]), list(s[3:7]), fut_states)
        g3_state.append(s); g3_prop.append(prop); g3_fut.append(fut)
        g3_fut_in.append(np.concatenate([np.concatenate([x['base_pos'], x['base_orn'], x['joint_pos']]) for x in fut_states]))
    out.update(g3_state=np.array(g3_state), g3_prop=np.array(g3_prop), g3_future_in=np.array(g3_fut_in), g3_future=np.array(g3_fut))

    # ---- G4: reward + three termination flags at random (dyn, kin, feet) ---------------------
    g4 = dict(dyn=[], kin=[], feet_dyn=[], feet_kin=[], reward=[], fall=[], diverged=[])
    import io, contextlib
    for k in range(256):
        kin = rand_state(rng)
        scale = [0.02, 0.1, 0.3, 1.0][k % 4]
        dyn = kin + rng.normal(size=37) * scale
        dyn[3:7] = rand_state(rng, big=(k % 16 == 5))[3:7] if k % 2 else kin[3:7]
        if k % 2:
            from scipy.spatial.transform import Rotation as R
            dyn[3:7] = (R.from_rotvec(rng.normal(size=3) * scale) * R.from_quat(kin[3:7])).as_quat()
        if k == 3:
            dyn[0:3] = kin[0:3] + np.array([1.0, 0, 0]) * (1.0 + 1e-9)   # squared position error just above 1
        if k == 7:
            dyn[0:3] = kin[0:3] + np.array([1.0, 0, 0]) * (1.0 - 1e-9)
        if k == 11:
            from scipy.spatial.transform import Rotation as R
            dyn[3:7] = (R.from_rotvec([0, 0, np.pi - 1e-6]) * R.from_quat(kin[3:7])).as_quat()
        if k == 13:
            dyn[3:7] = -kin[3:7]; dyn[0:3] = kin[0:3]                     # antipodal quaternion, angle 0
        fd = rng.uniform(-1, 1, (4, 3)); fk = fd + rng.normal(size=(4, 3)) * 0.05 * scale
        set_dyn(client, dyn, 0); set_dyn(client, kin, 1)
        client.feet[0], client.feet[1] = fd, fk
        with contextlib.redirect_stdout(io.StringIO()):
            r = ple._compute_reward()
            st = ple._legged_robot.get_states_info()
            fall = LeggedRobot.check_terminate(st)
            div = ple._check_dyn_kin_difference(st)
        for kk, v in zip(g4.keys(), [dyn, kin, fd, fk, r, fall, div]):
            g4[kk].append(v)
    out.update({'g4_' + k: np.array(v) for k, v in g4.items()})
    # K3
    set_dyn(client, state_vec(C.STATES_INFO_12), 0); set_dyn(client, state_vec(C.STATES_INFO_12_RUN_0), 1)
    client.feet[0], client.feet[1] = np.zeros((4, 3)), np.full((4, 3), 0.01)
    out['k3_reward'] = np.float64(ple._compute_reward())
    out['k3_dyn'] = state_vec(C.STATES_INFO_12); out['k3_kin'] = state_vec(C.STATES_INFO_12_RUN_0)

    # ---- G5/G6: scripted episodes through env.step (history stacking, done, sampling table) ----
    ep = dict(seed=[], clip=[], t0=[], actions=[], dyn=[], feet_dyn=[], feet_kin=[], obs=[], reward=[], done=[], n=[],
              prob_after=[], avg_len_after=[], reset_obs=[])
    T = 12
    ple._prioritized_sample_probability[:] = 1.0 / len(names)
    ple._avg_reward_sum[:] = 0.0
    for e in range(12):
        seed = 1000 + e
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            o0 = obs_vec(env.reset()[0])
        kin0 = state_vec(ple._legged_robot_kin.get_states_info())
        acts = rng.normal(size=(T, 12)) * 0.1353
        dyn_seq, obs_seq, rew_seq, done_seq, fd_seq, fk_seq = [], [], [], [], [], []
        n_done = T
        for t in range(T):
            # scripted dyn state: drifts away from the reference; episode e%3==0 falls over at step 8
            s = kin0 + rng.normal(size=37) * 0.02 * (t + 1)
            s[3:7] = kin0[3:7] / np.linalg.norm(kin0[3:7])
            if e % 3 == 0 and t >= 8:
                from scipy.spatial.transform import Rotation as R
                s[3:7] = (R.from_quat(kin0[3:7]) * R.from_euler('x', 1.2)).as_quat()
            if e % 3 == 1 and t >= 10:
                s[0:3] = kin0[0:3] + [3.0, 0, 0]
            fd = rng.uniform(-1, 1, (4, 3)); fk = fd + rng.normal(size=(4, 3)) * 0.01
            client.script = [s]; client.feet[0], client.feet[1] = fd, fk
            with contextlib.redirect_stdout(io.StringIO()):
                (o,), (r,), d, info = env.step([acts[t]])
            dyn_seq.append(s); obs_seq.append(obs_vec(o)); rew_seq.append(r); done_seq.append(d); fd_seq.append(fd); fk_seq.append(fk)
            if d:
                n_done = t + 1
                break
        pad = lambda a, shape: np.concatenate([np.array(a), np.zeros((T - len(a),) + shape)], 0)
        ep['seed'].append(seed); ep['clip'].append(ple.sampled_data_idx); ep['t0'].append(ple.time - n_done * 10 * 0.002)
        ep['actions'].append(acts); ep['dyn'].append(pad(dyn_seq, (37,))); ep['obs'].append(pad(obs_seq, (207,)))
        ep['feet_dyn'].append(pad(fd_seq, (4, 3))); ep['feet_kin'].append(pad(fk_seq, (4, 3)))
        ep['reward'].append(pad(rew_seq, ())); ep['done'].append(pad(done_seq, ())); ep['n'].append(n_done)
        ep['prob_after'].append(ple._prioritized_sample_probability.copy()); ep['avg_len_after'].append(ple.avg_episode_len.copy())
        ep['reset_obs'].append(o0)
    out.update({'g5_' + k: np.array(v) for k, v in ep.items()})
    # exact t0 of each scripted episode (time accumulates, so recompute from a fresh seeded reset)
    t0s = []
    for e in range(12):
        np.random.seed(1000 + e)
        ple._prioritized_sample_probability[:] = 1.0 / len(names) if e == 0 else ep['prob_after'][e - 1]
        env.reset(); t0s.append(ple.time)
    out['g5_t0'] = np.array(t0s)

    # ---- G7: obstacle extraction for every clip (PMC obstacle variant) -------------
    ob_clip, ob_n, ob_pos, ob_orn, ob_time, ob_pose = [], [], [], [], [], []
    for c, nm in enumerate(names):
        ob = ml_all.obstacles_info[c]
        if ob is None:
            continue
        for i in range(len(ob['time'])):
            p, q = get_obstacle_pose(ob['pos'][i], ob['orn_otho'][i])
            ob_clip.append(c); ob_pos.append(ob['pos'][i]); ob_orn.append(ob['orn_otho'][i]); ob_time.append(ob['time'][i])
            ob_pose.append(np.concatenate([p, q]))
    out.update(g7_clip=np.array(ob_clip), g7_pos=np.array(ob_pos), g7_orn=np.array(ob_orn), g7_time=np.array(ob_time), g7_pose=np.array(ob_pose))

    np.savez_compressed(os.path.join(HERE, 'pmc_golden.npz'), **out)
    print('wrote', os.path.join(HERE, 'pmc_golden.npz'), {k: getattr(v, 'shape', None) for k, v in out.items()})
    print('K3 reward', out['k3_reward'], 'K4 t0', out['k4_t0'], 'sum|prop|', np.abs(out['k4_obs'][:99]).sum(), 'sum|future|', np.abs(out['k4_obs'][135:]).sum())


if __name__ == '__main__':
    main()
