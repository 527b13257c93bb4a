"""The reference's trained EPMC policies (oracle/epmc_policy.py: a NumPy restatement, test infrastructure) driving OUR PlayGround env
closed-loop: the protocol of test_scripts/environmental_level/test_environmental_level_env.py (element per checkpoint, target speed 3 m/s,
friction 0.4 .. 1, pushes, no auxiliary cylinders, argmax code) on N envs at once.

    python tools/rollout_epmc_policy.py hurdle 256 1000 [lib] [gates] [forget_bias]

Reports per episode: how it ended (reached the target / fell / timed out), its length, distance covered along x, mean reward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

ELEMENT = {'hurdle': 1, 'hole': 2, 'cube': 3}


def env_config(element_id, n, seed=0, lib_path=None):
    return {'arena_id': 'Playground', 'render': False, 'control_freq': 50.0,
            'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
            'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'element_id': element_id, 'height_range': [0.0, 0.0], 'friction_range': [0.4, 1.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2,
                                                              'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
                                     'cmd_vary_freq_range': [9999, 10000], 'target_spd_range': [3.0, 3.0], 'auxiliary_radius': None,
                                     'hole_config': {'min_gap_height': 0.25, 'max_gap_height': 0.25}},
            'num_envs': n, 'auto_reset': False, 'seed': seed, 'lib_path': lib_path}


def rollout(which, n, horizon, lib_path=None, gates='ifou', forget_bias=1.0, seed=0, weights=None, blind=False):
    import lifelike_agility_and_play_amd as lla
    from oracle.epmc_policy import EpmcPolicy
    env = lla.create_playground_game(**env_config(ELEMENT[which], n, seed, lib_path))
    env.engine.set_spec(**{k: float(v) for k, v in (kv.split('=') for kv in os.environ.get('LL_SPEC', '').split(',') if kv)})   # e.g. LL_SPEC=friction_mode=0
    pol = EpmcPolicy(weights or os.path.join(ROOT, 'tests', 'golden', 'epmc_policy_%s.npz' % which), n, forget_bias=forget_bias, gates=gates)
    obs = env.reset()
    x0 = env.engine.state()[:, 0].copy()
    alive = np.ones(n, bool)
    steps, rsum, why, dist = np.zeros(n, int), np.zeros(n), np.zeros(n, int), np.zeros(n)
    codes = []
    for t in range(horizon):
        if blind:                            # control experiment: the policy is shown flat ground -- height map 0, nothing within the front rays' 3 m
            obs = obs.copy(); obs[:, 135:460] = 0.0; obs[:, 588:913] = 3.0
        a = pol.act(obs)
        codes.append(pol.last_code.copy())
        obs, r, d, info = env.step(a)
        rsum += np.where(alive, r, 0.0); steps += alive
        newly = alive & d
        why[newly] = info['done_reason'][newly]
        dist[newly] = env.engine.state()[newly, 0] - x0[newly]
        alive &= ~d
        if not alive.any():
            break
    dist[alive] = env.engine.state()[alive, 0] - x0[alive]
    env.close()
    return dict(steps=steps, rsum=rsum, why=why, dist=dist, alive=alive, codes=np.array(codes))


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'hurdle'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    lib = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != '-' else None
    gates = sys.argv[5] if len(sys.argv) > 5 else 'ifou'
    fb = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
    blind = len(sys.argv) > 7 and sys.argv[7] == 'blind'
    out = rollout(which, n, horizon, lib, gates, fb, blind=blind)
    st, why = out['steps'], out['why']
    print('%s policy%s, gates %s, forget bias %.1f, %d envs, horizon %d: mean episode length %.1f steps, distance along x %.2f m (median %.2f), '
          'reward per step %.3f' % (which, ' SHOWN FLAT GROUND (height map and front rays blanked)' if blind else '', gates, fb, n, horizon, st.mean(), out['dist'].mean(), np.median(out['dist']), out['rsum'].sum() / st.sum()))
    print('  ended by: reached the target %d, fell %d, timed out %d, still running at the horizon %d;  distinct codes used %d' % (
        int(((why & 4) != 0).sum()), int(((why & 1) != 0).sum()), int(((why & 2) != 0).sum()), int(out['alive'].sum()), len(np.unique(out['codes']))))
    print('  done-reason histogram', np.bincount(why, minlength=8).tolist())
