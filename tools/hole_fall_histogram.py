"""Where does the reference's bars ("hole") policy fall, relative to the bars?  (round-4 review, item 7: after the undersides of the hanging bars became solid --
bottom edges against the trunk, DESIGN.md 8 -- run that checkpoint once at 1024 episodes with a histogram of the fall positions.)
Protocol of tools/rollout_epmc_policy.py (test_environmental_level_env.py: target speed 3 m/s, pushes, friction 0.4 ... 1, argmax code).

    gpurun -- 'python tools/hole_fall_histogram.py 1024 600 > gpurun_out/hole_hist.txt'
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import rollout_epmc_policy as R  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    import lifelike_agility_and_play_amd as lla
    import epmc_parity_common as ec
    from oracle.epmc_policy import EpmcPolicy
    env = lla.create_playground_game(**R.env_config(R.ELEMENT['hole'], n, 0, os.environ.get('LL_LIB') or None))
    pol = EpmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'epmc_policy_hole.npz'), n)
    obs = env.reset()
    rows, cnt = env.engine.statics()
    bars = []
    for e in range(n):
        rec = ec.statics_to_records(rows[e, :cnt[e]].astype(np.float64))
        fl = rec[rec[:, 4] > 0.05]
        bars.append(np.sort(0.5 * (fl[:, 0] + fl[:, 1])))
    alive = np.ones(n, bool)
    why, xend, zmax_under = np.zeros(n, int), np.zeros(n), np.zeros(n)
    crossed = np.zeros(n, int)
    for t in range(horizon):
        a = pol.act(obs)
        obs, r, d, info = env.step(a)
        st = env.engine.state()
        newly = alive & d
        why[newly] = info['done_reason'][newly]
        xend[newly] = st[newly, 0]
        alive &= ~d
        if not alive.any():
            break
    st = env.engine.state()
    xend[alive] = st[alive, 0]
    env.close()
    fell = (why & 1) != 0
    reached = (why & 4) != 0
    print('bars policy, %d episodes, horizon %d, round-5 spec (bars solid from below): reached %d, fell %d, under way %d' % (n, horizon, reached.sum(), fell.sum(), alive.sum()))
    # position of a fall relative to the bars of its own course: distance to the next bar ahead (negative: that far in front of it) and bars already crossed
    rel, ncross = [], []
    for e in np.where(fell)[0]:
        b = bars[e]
        ahead = b[b > xend[e]]
        behind = b[b <= xend[e]]
        ncross.append(len(behind))
        nearest = b[np.argmin(np.abs(b - xend[e]))] if len(b) else np.nan
        rel.append(xend[e] - nearest)
    rel, ncross = np.array(rel), np.array(ncross)
    edges = [-1.5, -1.0, -0.7, -0.5, -0.3, -0.15, -0.05, 0.05, 0.15, 0.3, 0.5, 0.7, 1.0, 1.5]
    h, _ = np.histogram(rel, bins=edges)
    print('falls by x - x(nearest bar) [m] (negative: in front of it; a bar is 0.1 m long, the trunk 0.28 m):')
    for lo, hi, c in zip(edges[:-1], edges[1:], h):
        print('  %+5.2f .. %+5.2f : %4d  %s' % (lo, hi, c, '#' * int(60 * c / max(1, h.max()))))
    print('  outside the bins: %d' % (len(rel) - h.sum()))
    print('falls by number of bars already crossed:', {int(k): int((ncross == k).sum()) for k in np.unique(ncross)})
    under = np.abs(rel) < 0.19           # the trunk overlaps the bar in x
    print('falls with the trunk under a bar (|dx| < 0.19 m): %d of %d (%.1f %%); between two bars: %d' % (under.sum(), len(rel), 100.0 * under.mean(), (~under).sum()))


if __name__ == '__main__':
    main()
