"""Golden vectors for what the reference ASKS of PyBullet on the PMC path (build container only; imports the reference).

pmc_golden.npz (gen_golden.py) pins everything the reference computes itself.  What `stepSimulation()` computes cannot be
pinned here (the pybullet wheel is absent), but every INPUT the reference hands to the physics engine can -- this script
records them through a call-logging fake BulletClient:

  G8   PD torques: the `forces=` of every setJointMotorControlArray(TORQUE_CONTROL) call of LeggedRobot.apply_action (LR:119-148)
       for 320 (q, qd, target) triples, incl. the +-3 rad target clip, the +-max_tau torque clip and list-valued max_tau (LR:244);
       plus the ten torque sets of whole env.step() calls (PLE:199-206: target = joint_pos + action, re-applied per substep)
  G9   the configuration calls of world construction, in order, with their arguments (loadURDF flags, collision filter
       groups, changeDynamics, the zero-force POSITION_CONTROL call, gravity, solver iterations, time step, sub steps):
       LR:207-264, :266-308, PLE:56-82 -- and of obstacle creation (PLE:173-193) with set_obstacle=True
  G10  the global NumPy stream around list-valued max_tau: construction draws once (LR:244), every reset() draws once more
       BEFORE the clip / start-time draws (PLE:153) -- (clip, t0) and the next uniform after each reset, for 6 seeds

    python tests/golden/gen_pmc_config_golden.py        -> tests/golden/pmc_config_golden.npz
"""
import contextlib
import io
import json
import os
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G   # noqa: E402  (the stubs and the state-holding fake client)


class Sym(int):
    """An ALL-CAPS pybullet constant, kept symbolic: sums / ors of constants stay readable in the log."""
    def __new__(cls, names):
        o = int.__new__(cls, 0)
        o.names = tuple(names)
        return o

    def __add__(self, other):
        return Sym(self.names + (other.names if isinstance(other, Sym) else (repr(other),)))
    __or__ = __radd__ = __ror__ = __add__

    def __repr__(self):
        return '+'.join(self.names)


def plain(x):
    """JSON-able rendering of a call argument."""
    if isinstance(x, Sym):
        return repr(x)
    if isinstance(x, (np.floating, float)):
        return float(x)
    if isinstance(x, (np.integer, int)):
        return int(x)
    if isinstance(x, (list, tuple, np.ndarray)):
        return [plain(v) for v in x]
    if isinstance(x, str):
        return os.path.basename(x) if x.endswith('.urdf') else x
    return repr(x)


class RecordingClient(G.FakeBulletClient):
    """The state-holding fake of gen_golden.py that also logs every call it is asked to make."""
    LOGGED = ('loadURDF', 'setCollisionFilterGroupMask', 'changeDynamics', 'setJointMotorControlArray', 'setGravity', 'setPhysicsEngineParameter',
              'setTimeStep', 'createCollisionShape', 'createMultiBody', 'removeBody', 'configureDebugVisualizer', 'applyExternalForce')

    def __init__(self, connection_mode=None):
        G.FakeBulletClient.__init__(self, connection_mode)
        self.calls = []
        self.n_bodies_extra = 0

    def __getattr__(self, name):
        if name.isupper():
            return Sym((name,))
        if name in self.LOGGED:
            def rec(*a, **k):
                self.calls.append([name, plain(a), {kk: plain(v) for kk, v in k.items()}])
                if name == 'createCollisionShape':
                    return 77
                if name == 'createMultiBody':
                    self.bodies.append(dict(p=[0, 0, 0], q=[0, 0, 0, 1], v=[0, 0, 0], w=[0, 0, 0], j=np.zeros((22, 2))))
                    return len(self.bodies) - 1
                return None
            return rec
        return lambda *a, **k: None

    def loadURDF(self, *a, **k):
        self.calls.append(['loadURDF', plain(a), {kk: plain(v) for kk, v in k.items()}])
        return G.FakeBulletClient.loadURDF(self, *a, **k)

    def resetBasePositionAndOrientation(self, i, p, q):
        if i >= 3:                                              # bodies 0, 1, 2 are robot, ghost, plane: anything later is the obstacle
            self.calls.append(['resetBasePositionAndOrientation', [int(i), plain(p), plain(q)], {}])
        G.FakeBulletClient.resetBasePositionAndOrientation(self, i, p, q)

    def torque_calls(self):
        return [c for c in self.calls if c[0] == 'setJointMotorControlArray' and c[2].get('controlMode') == 'TORQUE_CONTROL']


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def main():
    G.install_stubs()
    sys.modules['pybullet_utils.bullet_client'].BulletClient = RecordingClient
    sys.path.insert(0, G.REF_SRC)
    from lifelike.sim_envs.pybullet_envs.create_pybullet_envs import create_tracking_game

    out = OrderedDict()
    rng = np.random.default_rng(20260927)

    # ---- G9: configuration calls -------------------------------------------------------------------------------------
    env = create_tracking_game(**G.PMC_CONFIG)
    ple = env.env
    client = ple._bullet_client
    out['g9_calls_json'] = np.array(json.dumps(client.calls))
    cfg_ob = dict(G.PMC_CONFIG); cfg_ob.update(set_obstacle=True, obstacle_height=0.2)
    env_ob = create_tracking_game(**cfg_ob)
    names = sorted(f for f in os.listdir(G.MOCAP_DIR) if f.endswith('txt'))
    jump = names.index('dog_jump_002_ret.txt')
    n0 = len(env_ob.env._bullet_client.calls)
    np.random.seed(3)
    p = np.zeros(len(names)); p[jump] = 1.0
    env_ob.env._motion_generator.prioritized_sample_probability = p      # a clip that has obstacles (G7)
    with quiet():
        env_ob.reset()
        first = [c for c in env_ob.env._bullet_client.calls[n0:]]
        n1 = len(env_ob.env._bullet_client.calls)
        env_ob.reset()                                                     # second episode: the old box is removed first (PLE:174-175)
        second = [c for c in env_ob.env._bullet_client.calls[n1:]]
    out['g9_obstacle_calls_json'] = np.array(json.dumps({'first_reset': first, 'second_reset': second, 'clip': names[jump]}))

    # ---- G8a: apply_action at (q, qd, target) triples ------------------------------------------------------------------
    robot = ple._legged_robot
    q_l, qd_l, tgt_l, tau_l = [], [], [], []
    for k in range(256):
        q = rng.uniform(-2.5, 2.5, 12)
        qd = rng.normal(size=12) * [3.0, 10.0, 30.0][k % 3]
        tgt = q + rng.normal(size=12) * [0.05, 0.2, 0.6, 2.0][k % 4]
        if k % 16 == 3:
            tgt[:6] = [3.0, -3.0, 3.0 + 1e-9, -3.0 - 1e-9, 7.5, -7.5]          # the +-3 rad clip and its edge (LR:126-127)
        if k % 16 == 5:                                                         # tau exactly at / just inside / just outside +-18 (kd * qd = 0)
            qd[:6] = 0.0
            tgt[:6] = q[:6] + np.array([18.0, -18.0, 18.0 + 1e-9, -18.0 - 1e-9, 18.0 - 1e-9, -18.0 + 1e-9]) / 50.0
            tgt[:6] = np.clip(tgt[:6], -3, 3)
        s = np.zeros(37); s[6] = 1.0; s[13:25] = q; s[25:37] = qd
        G.set_dyn(client, s, 0)
        n0 = len(client.calls)
        robot.apply_action(np.array(tgt))
        calls = client.calls[n0:]
        assert len(calls) == 1 and calls[0][2]['controlMode'] == 'TORQUE_CONTROL' and calls[0][2]['jointIndices'] == G.LEG_IDX
        q_l.append(q); qd_l.append(qd); tgt_l.append(tgt); tau_l.append(calls[0][2]['forces'])
    out.update(g8_q=np.array(q_l), g8_qd=np.array(qd_l), g8_target=np.array(tgt_l), g8_tau=np.array(tau_l, dtype=np.float64),
               g8_kp=np.float64(G.PMC_CONFIG['kp']), g8_kd=np.float64(G.PMC_CONFIG['kd']), g8_max_tau=np.float64(G.PMC_CONFIG['max_tau']))

    # ---- G8b: list-valued max_tau: one draw at construction (LR:244) sets the clip for the life of the env ----------------
    lst = dict(seed=[], max_tau=[], q=[], qd=[], target=[], tau=[])
    for seed in (11, 12, 13, 14):
        np.random.seed(seed)
        cfg_l = dict(G.PMC_CONFIG); cfg_l['max_tau'] = [6.0, 14.0]
        env_l = create_tracking_game(**cfg_l)
        r_l, c_l = env_l.env._legged_robot, env_l.env._bullet_client
        with quiet():
            env_l.reset(); env_l.reset()                                   # PLE:153 redraws into `max_taus` -- the clip must not move
        for k in range(16):
            q = rng.uniform(-2.0, 2.0, 12); qd = rng.normal(size=12) * 5.0; tgt = q + rng.normal(size=12) * 0.5
            s = np.zeros(37); s[6] = 1.0; s[13:25] = q; s[25:37] = qd
            G.set_dyn(c_l, s, 0)
            n0 = len(c_l.calls)
            r_l.apply_action(np.array(tgt))
            lst['seed'].append(seed); lst['max_tau'].append(float(r_l._max_taus[0])); lst['q'].append(q); lst['qd'].append(qd)
            lst['target'].append(tgt); lst['tau'].append(c_l.calls[n0][2]['forces'])
    out.update({'g8l_' + k: np.array(v) for k, v in lst.items()})

    # ---- G8c: whole env.step(): ten torque sets per control step, target = joint_pos(at step start) + action (PLE:199-206) ----
    st = dict(clip=[], t0=[], action=[], q=[], qd=[], tau=[])
    for e in range(8):
        np.random.seed(500 + e)
        with quiet():
            env.reset()
        kin0 = G.state_vec(ple._legged_robot_kin.get_states_info())
        s = kin0 + rng.normal(size=37) * 0.05
        s[3:7] = kin0[3:7]
        s[25:37] = rng.normal(size=12) * 4.0
        G.set_dyn(client, s, 0)
        a = rng.normal(size=12) * [0.1353, 0.6, 2.5][e % 3]
        client.script = []                                                 # the fake never moves the robot inside the step
        n0 = len(client.calls)
        with quiet():
            env.step([a])
        tq = [c[2]['forces'] for c in client.calls[n0:] if c[0] == 'setJointMotorControlArray']
        assert len(tq) == 10
        st['clip'].append(ple.sampled_data_idx); st['t0'].append(ple.time - 10 * 0.002); st['action'].append(a)
        st['q'].append(s[13:25]); st['qd'].append(s[25:37]); st['tau'].append(tq)
    out.update({'g8s_' + k: np.array(v) for k, v in st.items()})

    # ---- G10: the global NumPy stream with list-valued max_tau ------------------------------------------------------------
    g10 = dict(seed=[], max_tau=[], clip=[], t0=[], next_uniform=[])
    for seed in range(6):
        np.random.seed(seed)
        cfg_l = dict(G.PMC_CONFIG); cfg_l['max_tau'] = [10.0, 20.0]
        env_l = create_tracking_game(**cfg_l)
        rows_c, rows_t, rows_u = [], [], []
        for _ in range(3):
            with quiet():
                env_l.reset()
            rows_c.append(env_l.env.sampled_data_idx); rows_t.append(env_l.env.time)
            state = np.random.get_state()
            rows_u.append(np.random.uniform())                              # where the stream stands after the reset ...
            np.random.set_state(state)                                      # ... without moving it
        g10['seed'].append(seed); g10['max_tau'].append(float(env_l.env._legged_robot._max_taus[0]))
        g10['clip'].append(rows_c); g10['t0'].append(rows_t); g10['next_uniform'].append(rows_u)
    out.update({'g10_' + k: np.array(v) for k, v in g10.items()})

    path = os.path.join(HERE, 'pmc_config_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: getattr(v, 'shape', None) for k, v in out.items()})
    for c in client.calls[:0]:
        print(c)


if __name__ == '__main__':
    main()
