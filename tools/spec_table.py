"""Spec VARIANTS priced on the reference's five trained policies (all trained against PyBullet), on the float32 engine and the float64 oracle
alike -- the protocol of tools/inertia_table.py (PMC: tools/deviation_table.py; EPMC hurdle / cube / hole: tools/rollout_epmc_policy.py and
tools/deviation_envs.py; SEPMC: tools/rollout_sepmc_policy.py), one row per (simulator, variant).  Round 5: the joint-limit rule and the
penetration recovery (profiles/r05_limit_rows.md, r05_penetration_recovery.md).

    python tools/spec_table.py --engine  --variants 'spec:;bullet limits:limit_speculative=0'            (GPU: PMC 4096, EPMC 1024 per policy, SEPMC 512 arenas)
    python tools/spec_table.py --oracle  --variants '...' [--pmc-episodes 1024 --epmc-episodes 128 --arenas 64]   (CPU, float64 oracle envs)
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

HZ = {'hurdle': 500, 'cube': 700, 'hole': 600}
HEAD = ('| simulator | variant | PMC reward | PMC tracked | PMC length | hurdle reached / fell | cube reached / fell | hole reached / fell / under way | '
        'SEPMC games: caught / robot 0 fell / timed out (mean length) |\n|---|---|---|---|---|---|---|---|---|')


def parse_variants(text):
    out = []
    for item in text.split(';'):
        if not item.strip():
            continue
        name, _, kv = item.partition(':')
        out.append((name.strip(), {k.strip(): float(v) for k, v in (x.split('=') for x in kv.split(',') if x.strip())}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--engine', action='store_true'); ap.add_argument('--oracle', action='store_true')
    ap.add_argument('--variants', default='spec:')
    ap.add_argument('--pmc-episodes', type=int, default=0); ap.add_argument('--epmc-episodes', type=int, default=0)
    ap.add_argument('--arenas', type=int, default=0); ap.add_argument('--arena-steps', type=int, default=900)
    ap.add_argument('--skip', default='', help='comma list of pmc, hurdle, cube, hole, sepmc')
    ap.add_argument('--procs', type=int, default=0)
    args = ap.parse_args()
    skip = set(args.skip.split(','))
    import bench
    import deviation_envs as DE
    import deviation_table as DT
    from lifelike_agility_and_play_amd import mocap, urdf_model
    from oracle.pmc_policy import PmcPolicy
    pol = PmcPolicy(os.path.join(ROOT, 'lifelike_agility_and_play_amd', 'assets', 'pmc_policy.npz'))
    table = mocap.load_mocap('', 0.02)
    blob = urdf_model.default_model_blob()
    procs = args.procs or bench.effective_cores()[0]
    print(HEAD, flush=True)
    for name, spec in parse_variants(args.variants):
        spec_s = ','.join('%s=%g' % kv for kv in spec.items())
        if args.engine:
            import rollout_epmc_policy as R
            import rollout_sepmc_policy as RS
            t = time.time()
            os.environ['LL_SPEC'] = spec_s                       # (the EPMC / SEPMC rollouts read it)
            cells = ['-', '-', '-']
            if 'pmc' not in skip:
                e = DT.run_engine(pol, blob, table, args.pmc_episodes or 4096, spec, 11)
                cells = ['%.4f' % e['reward'], '%.3f' % e['tracked'], '%.1f' % e['length']]
            for which in ('hurdle', 'cube', 'hole'):
                if which in skip:
                    cells.append('-'); continue
                n = args.epmc_episodes or 1024
                o = R.rollout(which, n, HZ[which])
                c = '%d / %d' % (((o['why'] & 4) != 0).sum(), ((o['why'] & 1) != 0).sum())
                cells.append(c + (' / %d' % o['alive'].sum() if which == 'hole' else '') + ' of %d' % n)
            if 'sepmc' in skip:
                cells.append('-')
            else:
                o = RS.rollout(args.arenas or 512, 1000)
                why = o['why']; fin = why != 0; tot = max(1, int(fin.sum()))
                cells.append('%.3f / %.3f / %.3f (%.0f) of %d' % (((why & 8) != 0).sum() / tot, ((why & 1) != 0).sum() / tot, ((why & 2) != 0).sum() / tot, o['steps'][fin].mean(), tot))
            print('| engine (float32 HIP) | %s (%s) | %s |' % (name, spec_s or 'as shipped', ' | '.join(cells)), flush=True)
            print('engine legs of "%s": %.0f s' % (name, time.time() - t), file=sys.stderr)
        if args.oracle:
            t = time.time()
            cells = ['-', '-', '-']
            if 'pmc' not in skip:
                o = DT.run_oracle(pol, blob, table, args.pmc_episodes or 1024, spec, 11, procs)
                cells = ['%.4f' % o['reward'], '%.3f' % o['tracked'], '%.1f' % o['length']]
            for which in ('hurdle', 'cube', 'hole'):
                n = args.epmc_episodes or 128
                if which in skip:
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
            print('| oracle (float64) | %s (%s) | %s |' % (name, spec_s or 'as shipped', ' | '.join(cells)), flush=True)
            print('oracle legs of "%s": %.0f s' % (name, time.time() - t), file=sys.stderr)


if __name__ == '__main__':
    main()
