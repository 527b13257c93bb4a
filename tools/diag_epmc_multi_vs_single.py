"""Which pair of EPMC launch paths disagrees on the reward of a step, and by how much (round 6 diagnosis: tests/epmc_parity_common.check_split_rays_equal_fused found
rewards 4 ulp apart between a fused multi-step launch and single steps with the rays split off, observation and state bit-equal)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import epmc_parity_common as ec
sg = float(np.exp(-2.0))
cfg = ec.env_config(1); cfg['max_steps'] = 15
def eng(mode):
    os.environ['LL_SPLIT_RAYS'] = mode
    E = ec.make_engine(cfg, 301, None, auto_reset=1, seed=4); E.reset(); return E
A, B, C = eng('0'), eng('0'), eng('1')
for t in range(12):
    A.step_random_n(sg, 4)                       # fused, ONE launch of four steps
    for _ in range(4): B.step_random_n(sg, 1)    # fused, four launches (actions drawn in the kernel, as A)
    for _ in range(4): C.step_random_n(sg, 1)    # rays split off, four launches
    ra, rb, rc = A.reward_done()[0], B.reward_done()[0], C.reward_done()[0]
    print('launch %2d: state A==B %s B==C %s | obs A==B %s B==C %s | reward A!=B %3d (max rel %.1e)  B!=C %3d' % (
        t, np.array_equal(A.state(), B.state()), np.array_equal(B.state(), C.state()), np.array_equal(A.obs(), B.obs()), np.array_equal(B.obs(), C.obs()),
        int((ra != rb).sum()), float(np.abs(ra - rb).max() / max(np.abs(rb).max(), 1e-30)), int((rb != rc).sum())))
    ea, eb = A.episode(), B.episode()
    bad = [k for k in ea if not np.array_equal(ea[k], eb[k])]
    if bad: print('   episode records that differ A vs B:', bad)
