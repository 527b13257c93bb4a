"""What the physics choices this build made on its own are worth WHERE THE PMC TRACKING POLICY CANNOT SEE: terrain
contact under the reference's trained EPMC policies, robot-robot contact in chase-tag arenas.

    python tools/deviation_envs.py epmc  --episodes 64  [--engine-envs 2048]      (engine leg needs a GPU; without one it is skipped)
    python tools/deviation_envs.py sepmc --arenas 96 --steps 400 [--engine-arenas 2048 --engine-steps 1000]

The ORACLE legs run the float64 CPU envs (oracle/free_run.py: NumPy env logic + analytic rays + the C physics) with one audit switch of
include/llenv_model.h moved at a time, episodes spread over the host cores; the ENGINE leg runs the spec as shipped on the GPU at full size.
EPMC: the trained hurdle / cube policies (oracle/epmc_policy.py), protocol of test_environmental_level_env.py; scored by how episodes end
(reached the target / fell), distance, length.  SEPMC: the trained strategic_level policy on both robots (oracle/sepmc_policy.py): how games end (catch / robot 0 fell / time-out), how long a catch takes,
and the fraction of arena steps with robot-robot contact."""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np  # noqa: E402

EPMC_VARIANTS = [
    ('spec (as shipped)', {}),
    ('friction cone-coupled + manifold order', dict(friction_mode=2, row_order=1)),
    ('friction rows adjacent', dict(friction_mode=1)),
    ('self friction 0.25', dict(self_friction=0.25)),
    ('depenetration cap off', dict(max_depen_speed=1e30)),
    ('deepest-2 per leg', dict(max_contacts_per_leg=2)),
    ('no trunk-edge candidates', dict(trunk_edges=0)),
    ('max coordinate velocity 100', dict(max_coord_vel=100.0)),
]
SEPMC_VARIANTS = [
    ('spec (as shipped)', {}),
    ('pair friction 0.25', dict(pair_friction=0.25)),
    ('4 rows per robot pair', dict(max_pair=4)),
    ('4 rows + pair friction 0.25', dict(max_pair=4, pair_friction=0.25)),
    ('1 row per robot pair', dict(max_pair=1)),
    ('no robot-robot rows', dict(max_pair=0)),
    ('self friction 0.25', dict(self_friction=0.25)),
    ('friction cone-coupled + manifold order', dict(friction_mode=2, row_order=1)),
]


def _epmc_episode(args):
    which, spec, seed, horizon = args
    import rollout_epmc_policy as R
    from oracle import oracle as orc, free_run as FR, epmc_oracle as EO
    from oracle.epmc_policy import EpmcPolicy
    from lifelike_agility_and_play_amd import epmc_capi, mocap, urdf_model
    orc.reset_spec(); orc.set_spec(**spec)
    cfg = R.env_config(R.ELEMENT[which], 1)
    run = FR.EpmcFreeRun(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state(), seed=seed)
    pol = EpmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'epmc_policy_%s.npz' % which), 1)
    obs = run.reset()
    x0 = run.env.state[0]
    steps, why = 0, 0
    for t in range(horizon):
        a = pol.act(np.asarray(obs, np.float64).reshape(1, -1))[0]
        obs, r, done, info = run.step(a)
        steps += 1
        if done:
            s = run.env.state
            why = 1 if EO.check_fall(s[3:7]) else (4 if np.linalg.norm((run.env.target_pos - s[0:3])[:2]) < 0.5 else 2)
            break
    return steps, why, run.env.state[0] - x0


def _sepmc_episode(args):
    """one arena of the oracle env, BOTH robots driven by the reference's trained SEPMC policy (oracle/sepmc_policy.py), until max_steps
    arena-steps are spent: (length, reason, steps with robot-robot contact, closest approach) per episode"""
    spec, seed, max_steps = args
    import rollout_sepmc_policy as R
    from oracle import oracle as orc, free_run as FR, epmc_oracle as EO
    from oracle.sepmc_policy import SepmcPolicy
    from lifelike_agility_and_play_amd import epmc_capi, mocap, urdf_model
    orc.reset_spec(); orc.set_spec(**spec)
    cfg = R.env_config(1)
    run = FR.SepmcFreeRun(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state(), seed=seed)
    pol = SepmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'sepmc_policy.npz'), 2)
    out = []
    left = max_steps
    while left > 0:
        obs = run.reset()
        pol.reset()
        n, touch, why, closest = 0, 0, 0, 1e9
        while left > 0:
            a = pol.act(np.asarray(obs, np.float64).reshape(2, -1))
            o = run.step([a[0], a[1]])
            obs = o[0]
            n += 1; left -= 1
            touch += int(run.touch[0][2] or run.touch[1][2])
            closest = min(closest, float(np.linalg.norm(run.env.states[0][0:2] - run.env.states[1][0:2])))
            if o[2]:
                why = 1 if EO.check_fall(run.env.states[0][3:7]) else (2 if run.env.counter >= run.env.max_steps else 8)
                break
        out.append((n, why, touch, closest))
    return out


def _pool(fn, jobs, procs):
    import multiprocessing as mp
    with mp.get_context('fork').Pool(procs) as p:
        return p.map(fn, jobs, chunksize=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('env', choices=['epmc', 'sepmc'])
    ap.add_argument('--episodes', type=int, default=64); ap.add_argument('--horizon', type=int, default=420)
    ap.add_argument('--arenas', type=int, default=96); ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--engine-envs', type=int, default=2048); ap.add_argument('--engine-arenas', type=int, default=512); ap.add_argument('--engine-steps', type=int, default=1000)
    ap.add_argument('--procs', type=int, default=0); ap.add_argument('--only', default=''); ap.add_argument('--no-oracle', action='store_true')
    args = ap.parse_args()
    import bench
    procs = args.procs or bench.effective_cores()[0]
    have_gpu = False
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:       # noqa: BLE001
        pass
    if args.env == 'epmc':
        import rollout_epmc_policy as R
        print('# EPMC: the reference\'s trained hurdle / cube policies on our terrain, one spec switch moved at a time')
        print()
        print('Protocol of test_environmental_level_env.py (target speed 3 m/s, pushes, friction 0.4 .. 1, argmax code).  oracle = float64 CPU env '
              '(oracle/free_run.py), %d episodes per cell, horizon %d steps; engine = the HIP library as shipped, %d envs.  tools/deviation_envs.py.' % (args.episodes, args.horizon, args.engine_envs))
        print()
        print('| simulator / variant | hurdle: reached | fell | distance m | length | cube: reached | fell | distance m | length |')
        print('|---|---|---|---|---|---|---|---|---|')
        if have_gpu:
            cells = []
            for which, hz in (('hurdle', 500), ('cube', 700)):
                o = R.rollout(which, args.engine_envs, hz)
                cells += ['%.3f' % ((o['why'] & 4) != 0).mean(), '%.3f' % ((o['why'] & 1) != 0).mean(), '%.2f' % o['dist'].mean(), '%.1f' % o['steps'].mean()]
            print('| engine (float32), spec | %s |' % ' | '.join(cells), flush=True)
        for label, spec in EPMC_VARIANTS:
            if args.no_oracle or (args.only and args.only not in label):
                continue
            cells = []
            for which in ('hurdle', 'cube'):
                res = _pool(_epmc_episode, [(which, spec, 100 + i, args.horizon + (200 if which == 'cube' else 0)) for i in range(args.episodes)], procs)
                st, why, dist = np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res])
                cells += ['%.3f' % (why == 4).mean(), '%.3f' % (why == 1).mean(), '%.2f' % dist.mean(), '%.1f' % st.mean()]
            print('| oracle, %s | %s |' % (label, ' | '.join(cells)), flush=True)
    else:
        import rollout_sepmc_policy as RS
        print('# SEPMC: robot-robot contact choices on chase-tag games played by the reference\'s trained policy')
        print()
        print('Protocol of test_strategic_level_env.py (5 m arena, no elements, control_spd 1.0, friction 0.4 .. 1, pushes; the trained strategic_level '
              'policy on BOTH robots, argmax).  oracle = float64 CPU env, %d arenas x %d arena-steps per variant; engine = the HIP library as shipped, '
              '%d arenas, horizon %d.  tools/deviation_envs.py.' % (args.arenas, args.steps, args.engine_arenas, args.engine_steps))
        print()
        print('| simulator / variant | episodes | caught | robot 0 fell | timed out | mean length (steps to the catch) | arena-steps with robot-robot contact |')
        print('|---|---|---|---|---|---|---|')
        if have_gpu:
            o = RS.rollout(args.engine_arenas, args.engine_steps)
            why = o['why']; fin = why != 0; tot = max(1, int(fin.sum()))
            print('| engine (float32), spec | %d | %.4f | %.4f | %.4f | %.1f | %.4f |' % (tot, ((why & 8) != 0).sum() / tot, ((why & 1) != 0).sum() / tot, ((why & 2) != 0).sum() / tot,
                                                                                      o['steps'][fin].mean() if fin.any() else float('nan'), o['touch_frac']), flush=True)
        for label, spec in SEPMC_VARIANTS:
            if args.no_oracle or (args.only and args.only not in label):
                continue
            res = sum(_pool(_sepmc_episode, [(spec, i, args.steps) for i in range(args.arenas)], procs), [])
            n, why, touch = np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res])
            fin = why != 0
            tot = max(1, int(fin.sum()))
            print('| oracle, %s | %d | %.4f | %.4f | %.4f | %.1f | %.4f |' % (label, tot, (why == 8).sum() / tot, (why == 1).sum() / tot, (why == 2).sum() / tot,
                                                                             n[fin].mean() if fin.any() else float('nan'), touch.sum() / n.sum()), flush=True)


if __name__ == '__main__':
    main()
