"""The reference's trained SEPMC policy (oracle/sepmc_policy.py: a NumPy restatement, test infrastructure) on BOTH robots of OUR chase-tag
arenas, closed-loop: the protocol of test_scripts/strategic_level/test_strategic_level_env.py (control_spd 1.0, friction 0.4 .. 1, pushes, no
arena elements, argmax) on N arenas at once.

    python tools/rollout_sepmc_policy.py 256 1000 [lib]

Reports how episodes end (catch / robot 0 fell / time-out), their length, the distance each robot covers, the closest approach."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def env_config(n, seed=0, lib_path=None):
    return {'arena_id': 'CTG', 'render': False, 'control_freq': 50.0,
            'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
            'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'friction_range': [0.4, 1.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2,
                                                              'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
                                     'control_spd': 1.0},
            'element_config': {'rand_cube': False, 'hurdle': False, 'hole': False},
            'num_envs': n, 'auto_reset': False, 'seed': seed, 'lib_path': lib_path}


def rollout(n, horizon, lib_path=None, seed=0, spec=None):
    import lifelike_agility_and_play_amd as lla
    from oracle.sepmc_policy import SepmcPolicy
    env = lla.create_chase_tag_game(**env_config(n, seed, lib_path))
    env.engine.set_spec(**{k: float(v) for k, v in (kv.split('=') for kv in os.environ.get('LL_SPEC', '').split(',') if kv)})   # e.g. LL_SPEC=friction_mode=0
    env.engine.set_spec(**(spec or {}))
    pol = SepmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'sepmc_policy.npz'), 2 * n)
    obs = env.reset()
    p0 = env.engine.state()[:, :, 0:2].copy()
    alive = np.ones(n, bool)
    steps, why = np.zeros(n, int), np.zeros(n, int)
    path = np.zeros((n, 2)); last = p0.copy(); closest = np.full(n, 1e9); fell1 = np.zeros(n, bool); touch_steps = 0
    for t in range(horizon):
        a = pol.act(obs.reshape(2 * n, -1)).reshape(n, 2, 12)
        obs, r, d, info = env.step(a)
        st = env.engine.state()
        pos = st[:, :, 0:2]
        path += np.where(alive[:, None], np.linalg.norm(pos - last, axis=2), 0.0); last = pos.copy()
        closest = np.where(alive, np.minimum(closest, np.linalg.norm(pos[:, 0] - pos[:, 1], axis=1)), closest)
        ep = env.engine.episode()
        touch_steps += int((alive & ((ep['who0'] == 4) | (ep['who_taker'] == 3) | (ep['who_taker'] == 4))).sum())
        steps += alive
        newly = alive & d
        why[newly] = info['done_reason'][newly]
        alive &= ~d
        if not alive.any():
            break
    env.close()
    return dict(steps=steps, why=why, alive=alive, path=path, closest=closest, touch_frac=touch_steps / max(1, steps.sum()))


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != '-' else None
    o = rollout(n, horizon, lib)
    why = o['why']
    print('SEPMC policy on both robots, %d arenas, horizon %d: caught %d, robot 0 fell %d, timed out %d, still running %d; mean length %.1f steps' % (
        n, horizon, int(((why & 8) != 0).sum()), int(((why & 1) != 0).sum()), int(((why & 2) != 0).sum()), int(o['alive'].sum()), o['steps'].mean()))
    print('  distance covered per episode: robot 0 %.2f m, robot 1 %.2f m (mean speed %.2f / %.2f m/s); closest approach median %.2f m; arena-steps with robot-robot contact %.4f' % (
        o['path'][:, 0].mean(), o['path'][:, 1].mean(), (o['path'][:, 0] / (o['steps'] * 0.02)).mean(), (o['path'][:, 1] / (o['steps'] * 0.02)).mean(),
        float(np.median(o['closest'])), o['touch_frac']))
