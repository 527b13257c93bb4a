"""Free-running EPMC / SEPMC environments made of the oracles only -- TEST INFRASTRUCTURE (never imported by the product).

epmc_oracle.py / sepmc_oracle.py restate what the reference's Python does around PyBullet; pmc_oracle.c restates this build's
physics spec.  Put together they are a complete float64 CPU implementation of the two envs: env logic (NumPy) + analytic rays
(cast_rays) + ten physics substeps per control step (orc_substep_terrain / orc_substep_pair) + this build's contact classes
(orc_touch).  bench.py times them as the `cpu_baseline` of --workload epmc / sepmc; tests/test_oracle_free_run.py runs them.
"""
import math

import numpy as np

from . import epmc_oracle as EO
from . import sepmc_oracle as SO
from . import oracle as orc

PLANE_FRICTION, BOX_FRICTION = 0.9, 0.5


def statics_to_records(rows):
    """epmc_oracle.gen_terrain rows (a box, then its two edge cylinders if any) -> the records the physics takes: x0 x1 y0 y1 z0 z1 rod r."""
    recs, i = [], 0
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 8)
    while i < len(rows):
        r = rows[i]
        if r[0] == 0 and r[4] > 0:
            rec = [r[1] - r[4], r[1] + r[4], r[2] - r[5], r[2] + r[5], r[3] - r[6], r[3] + r[6], 0.0, 0.0]
            if i + 1 < len(rows) and rows[i + 1][0] == 1:
                rec[6] = 1.0 if rows[i + 1][3] > r[3] else -1.0
                rec[7] = rows[i + 1][4]
                i += 2
            recs.append(rec)
        i += 1
    return np.array(recs, dtype=np.float64).reshape(-1, 8)


def near_records(rec, p, limit=8):
    if not len(rec):
        return rec, np.zeros(0, dtype=int)
    sel = np.nonzero((p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9))[0][:limit]
    return rec[sel], sel


def _batch(env_config, model_blob, mocap_table):
    cfg = orc.make_config(n_envs=1, control_freq=env_config.get('control_freq', 50.0), kp=env_config.get('kp', 50.0), kd=env_config.get('kd', 1.0),
                          max_tau=float(env_config.get('max_tau', 16.0)), prop_type=list(env_config['prop_type']))
    return orc.OracleBatch(cfg, model_blob, mocap_table)


class EpmcFreeRun(object):
    """One PlayGroundEnv on the CPU: EpmcOracleEnv + the C oracle's terrain substep."""

    def __init__(self, env_config, model_blob, mocap_table, init_state, seed=0):
        self.cfg = env_config
        self.env = EO.EpmcOracleEnv(env_config, init_state)
        self.B = _batch(env_config, model_blob, mocap_table)
        self.draws = EO.LiveDraws(seed)
        self.kp, self.kd, self.max_tau = env_config.get('kp', 50.0), env_config.get('kd', 1.0), float(env_config.get('max_tau', 16.0))

    def reset(self):
        obs = self.env.reset(self.draws)
        self.rec = statics_to_records(self.env.statics)
        return obs

    def step(self, action):
        env = self.env
        near, _ = near_records(self.rec, env.state[0:3])
        mu = env.foot_friction * PLANE_FRICTION

        def physics(k, tgt, force):
            s = env.state
            t = np.clip(tgt, -3.0, 3.0)                                             # LR:126-127
            tau = np.clip(self.kp * (t - s[13:25]) - self.kd * s[25:37], -self.max_tau, self.max_tau)
            s2, _, _ = self.B.substep_terrain(s, tau, mu, near, BOX_FRICTION / PLANE_FRICTION, force)
            return s2
        return env.step(action, self.draws, physics)


class SepmcFreeRun(object):
    """One ChaseTagGameEnv on the CPU: SepmcOracleEnv + the C oracle's two-robot substep + its contact classes."""

    def __init__(self, env_config, model_blob, mocap_table, init_state, seed=0):
        self.cfg = env_config
        self.model = SO.BlobModel(model_blob)
        self.env = SO.SepmcOracleEnv(env_config, init_state, self.model)
        self.B = _batch(env_config, model_blob, mocap_table)
        self.draws = SO.LiveDraws(seed)
        self.kp, self.kd, self.max_tau = env_config.get('kp', 50.0), env_config.get('kd', 1.0), float(env_config.get('max_tau', 18.0))
        self.touch = np.zeros((2, 3), dtype=np.int32)

    def _records(self):
        bx = self.env.all_boxes()                                                  # arena boxes, then the flag
        rec = np.c_[bx[:, 1] - bx[:, 4], bx[:, 1] + bx[:, 4], bx[:, 2] - bx[:, 5], bx[:, 2] + bx[:, 5], bx[:, 3] - bx[:, 6], bx[:, 3] + bx[:, 6], np.zeros(len(bx)), np.zeros(len(bx))]
        rec[0, 3] += 1.0; rec[1, 2] -= 1.0; rec[2, 1] += 1.0; rec[3, 0] -= 1.0      # the walls are solid outwards for contacts
        rec[0:2, 0] -= 1.0; rec[0:2, 1] += 1.0; rec[2:4, 2] -= 1.0; rec[2:4, 3] += 1.0      # ... and longer by the same at both ends: the corners are closed
        return rec

    def _contacts(self):
        """getContactPoints() in this build's order: plane / boxes, flag, other robot; empty before the first substep."""
        out = []
        for r in range(2):
            me, other = (SO.ROBOT0, SO.ROBOT1) if r == 0 else (SO.ROBOT1, SO.ROBOT0)
            if self.touch[r][0]:
                out.append((me, SO.STATIC, 1, -1))
            if self.touch[r][1]:
                out.append((me, SO.FLAG, 1, -1))
            if self.touch[r][2]:
                out.append((me, other, 1, -1))
        # per robot the first record decides (CTG:426-440); records of the two robots do not interact because a record names its
        # robot first and the other's body link is given as the trunk (-1)
        return out

    def reset(self):
        self.touch[:] = 0
        return self.env.reset(self.draws, contacts=lambda: [])

    def step(self, actions):
        env = self.env
        rec = self._records()
        near, flag_at = [], []
        for r in range(2):
            nr, sel = near_records(rec, env.states[r][0:3])
            near.append(nr); flag_at.append(int(np.nonzero(sel == len(rec) - 1)[0][0]) if (len(rec) - 1) in sel else -1)
        mu = env.foot_friction * PLANE_FRICTION
        n_sub = env.n_sub

        def physics(k, tgt, force):
            s = env.states
            tau = [np.clip(self.kp * (np.clip(tgt[r], -3.0, 3.0) - s[r][13:25]) - self.kd * s[r][25:37], -self.max_tau, self.max_tau) for r in range(2)]
            if k == n_sub - 1:
                self.touch = self.B.touch(s[0], s[1], near[0], flag_at[0], near[1], flag_at[1])
            push = force if force is not None else [None, None]
            s0, s1, _ = self.B.substep_pair(s[0], s[1], tau[0], tau[1], mu, near[0], near[1], BOX_FRICTION / PLANE_FRICTION, push[0], push[1])
            return [s0, s1]
        return env.step(actions, self.draws, physics, contacts=self._contacts)


def time_random_policy(runner, budget_s, n_act, rng, max_steps=10 ** 9):
    """Random-policy control steps (a ~ N(0, e^-2)) until the budget is spent; returns (steps, seconds, episodes)."""
    import time
    sigma = math.exp(-2.0)
    runner.reset()
    t0 = time.perf_counter()
    steps = episodes = 0
    while time.perf_counter() - t0 < budget_s and steps < max_steps:
        a = rng.normal(size=n_act) * sigma if isinstance(n_act, int) else [rng.normal(size=12) * sigma, rng.normal(size=12) * sigma]
        out = runner.step(a)
        steps += 1
        if out[2]:
            episodes += 1
            runner.reset()
    return steps, time.perf_counter() - t0, episodes


class SharedDraws(object):
    """Uniforms in [0, 1) rounded to float32, recorded in order, mapped to values exactly as the engines map them (epmc_step.hpp EpmcDraws):
    the oracle env draws from this object, the recorded uniforms are then handed to the engine (ll_*_reset h_draws, ll_*_set_step_draws),
    so both sides see the same random numbers."""

    def __init__(self, seed=0):
        self.g = np.random.default_rng(seed)
        self.rec = []

    def _u(self):
        u = float(np.float32(self.g.random() * 0.999))
        self.rec.append(u)
        return u

    def uniform(self, a, b):
        return a + (b - a) * self._u()

    def randint(self, a, b):
        return a + min(int((b - a) * self._u()), b - a - 1)

    def rand(self):
        return self._u()

    def take(self):
        out, self.rec = self.rec, []
        return out
