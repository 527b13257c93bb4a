"""Which robot do the reference's five trained policies (all trained against PyBullet) recognise: the one with the URDF's <inertia> tensors
(rounds 1-3) or the one Bullet builds without URDF_USE_INERTIA_FROM_FILE -- AABB-box inertias of the collision shapes (urdf_model.py;
legged_robot.py:208-220)?  Same protocols as tools/deviation_table.py (PMC) and tools/deviation_envs.py (EPMC hurdle / cube / hole, SEPMC),
the spec as shipped, only the model blob differs (no kernel change: the model is data).

    python tools/inertia_table.py --engine                  (GPU: PMC 4096 episodes, EPMC 1024 per policy, SEPMC 512 arenas)
    python tools/inertia_table.py --oracle [--spec friction_mode=2,row_order=1]      (CPU, float64 oracle envs)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402

KINDS = ('file', 'collision_aabb')
HZ = {'hurdle': 500, 'cube': 700, 'hole': 600}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--engine', action='store_true'); ap.add_argument('--oracle', action='store_true')
    ap.add_argument('--pmc-episodes', type=int, default=0); ap.add_argument('--epmc-episodes', type=int, default=0)
    ap.add_argument('--arenas', type=int, default=0); ap.add_argument('--arena-steps', type=int, default=900)
    ap.add_argument('--spec', default='', help='spec overrides key=value,... (include/llenv_model.h LLM_SPEC_*); engine legs take the switches the engine carries')
    ap.add_argument('--kinds', default=','.join(KINDS)); ap.add_argument('--skip', default='')
    args = ap.parse_args()
    spec = {k: float(v) for k, v in (kv.split('=') for kv in args.spec.split(',') if kv)}
    skip = set(args.skip.split(','))
    import bench
    import deviation_envs as DE
    import deviation_table as DT
    from lifelike_agility_and_play_amd import mocap, urdf_model
    from oracle.pmc_policy import PmcPolicy
    pol = PmcPolicy(os.path.join(ROOT, 'lifelike_agility_and_play_amd', 'assets', 'pmc_policy.npz'))
    table = mocap.load_mocap('', 0.02)
    procs = bench.effective_cores()[0]
    print('| simulator | link inertias | PMC reward | PMC tracked | PMC length | hurdle reached / fell | cube reached / fell | hole reached / fell / under way | SEPMC games: caught / robot 0 fell / timed out (mean length) |')
    print('|---|---|---|---|---|---|---|---|---|')
    for kind in args.kinds.split(','):
        os.environ['LL_MODEL_INERTIA'] = kind
        blob = urdf_model.default_model_blob()
        assert np.array_equal(blob, urdf_model.model_blob(kind))
        if args.engine:
            import rollout_epmc_policy as R
            import rollout_sepmc_policy as RS
            t = time.time()
            os.environ['LL_SPEC'] = args.spec                   # (the EPMC / SEPMC rollouts read it)
            e = DT.run_engine(pol, blob, table, args.pmc_episodes or 4096, spec, 11)
            cells = ['%.4f' % e['reward'], '%.3f' % e['tracked'], '%.1f' % e['length']]
            for which in ('hurdle', 'cube', 'hole'):
                n = args.epmc_episodes or 1024
                o = R.rollout(which, n, HZ[which])
                c = '%d / %d' % (((o['why'] & 4) != 0).sum(), ((o['why'] & 1) != 0).sum())
                cells.append(c + (' / %d' % o['alive'].sum() if which == 'hole' else '') + ' of %d' % n)
            o = RS.rollout(args.arenas or 512, 1000)
            why = o['why']; fin = why != 0; tot = max(1, int(fin.sum()))
            cells.append('%.3f / %.3f / %.3f (%.0f) of %d' % (((why & 8) != 0).sum() / tot, ((why & 1) != 0).sum() / tot, ((why & 2) != 0).sum() / tot, o['steps'][fin].mean(), tot))
            print('| engine (float32 HIP)%s | %s | %s |' % (', ' + args.spec if args.spec else '', kind, ' | '.join(cells)), flush=True)
            print('engine legs under %s: %.0f s' % (kind, time.time() - t), file=sys.stderr)
        if args.oracle:
            t = time.time()
            cells = ['-', '-', '-']
            if 'pmc' not in skip:
                o = DT.run_oracle(pol, blob, table, args.pmc_episodes or 1024, spec, 11, procs)
                cells = ['%.4f' % o['reward'], '%.3f' % o['tracked'], '%.1f' % o['length']]
            for which in ('hurdle', 'cube', 'hole'):
                n = args.epmc_episodes or 128
                if 'epmc' in skip:
                    cells.append('-'); continue
                res = DE._pool(DE._epmc_episode, [(which, spec, 100 + i, HZ[which]) for i in range(n)], procs)
                why = np.array([r[1] for r in res])
                cells.append('%d / %d' % ((why == 4).sum(), (why == 1).sum()) + (' / %d' % (why == 0).sum() if which == 'hole' else '') + ' of %d' % n)
            if 'sepmc' in skip:
                cells.append('-')
            else:
                res = sum(DE._pool(DE._sepmc_episode, [(spec, i, args.arena_steps) for i in range(args.arenas or 64)], procs), [])
                n, why = np.array([r[0] for r in res]), np.array([r[1] for r in res])
                fin = why != 0; tot = max(1, int(fin.sum()))
                cells.append('%.3f / %.3f / %.3f (%.0f) of %d' % ((why == 8).sum() / tot, (why == 1).sum() / tot, (why == 2).sum() / tot, n[fin].mean(), tot))
            label = 'oracle (float64)' + (', ' + args.spec if args.spec else '')
            print('| %s | %s | %s |' % (label, kind, ' | '.join(cells)), flush=True)
            print('oracle legs under %s: %.0f s' % (kind, time.time() - t), file=sys.stderr)


if __name__ == '__main__':
    main()
