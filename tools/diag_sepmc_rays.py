"""Diagnostic (GPU): where does the SEPMC one-wave-per-SIMD kernel built with seven rays per chunk (tools/ab.sh build s7 "-DLL_SEPMC_RAY_CHUNK=7")
part from the shipped build (three)?  Same seed, same actions, state re-synchronised every step; prints which entries of the observation differ."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import sepmc_parity_common as SC  # noqa: E402

NAMES = [(0, 135, 'prop | prop_a'), (135, 460, 'height grid'), (460, 588, 'fan'), (588, 913, 'front rays'), (913, 918, 'percept_vec'), (918, 948, 'oppo_info(+cheat)'),
         (948, 962, 'flag_info(+cheat)'), (962, 965, 'with_flag, spd')]
lib = os.path.join(ROOT, 'tools', '_build', 'ab_%s.so' % (sys.argv[1] if len(sys.argv) > 1 else 's7'))
cfg = SC.env_config(SC.ALL_ELEMENTS)
A = SC.make_engine(cfg, 201, None, auto_reset=1, seed=3)
B = SC.make_engine(cfg, 201, lib, auto_reset=1, seed=3)
A.reset(); B.reset()
print('after reset: obs equal', np.array_equal(A.obs(), B.obs()), 'state equal', np.array_equal(A.state(), B.state()))
rng = np.random.default_rng(0)
shown = 0
for t in range(200):
    act = (rng.normal(size=A.obs().shape[:-1] + (12,)) * 0.135).astype(np.float32)
    A.step_host(act); B.step_host(act)
    oa, ob = A.obs(), B.obs()
    sa, sb = A.state(), B.state()
    if not np.array_equal(sa, sb):
        print('step %d: STATES differ in %d entries (max %.3g)' % (t, (sa != sb).sum(), np.abs(sa - sb).max()))
        break
    if not np.array_equal(oa, ob):
        d = (oa != ob)
        rows = np.argwhere(d.any(-1))
        print('step %d: obs differ in %d robot rows; per field:' % (t, len(rows)), {n: int(d[..., a:b].sum()) for a, b, n in NAMES if d[..., a:b].any()})
        for (ar, rb) in rows[:3]:
            idx = np.nonzero(d[ar, rb])[0]
            print('   arena %d robot %d: %d entries, first %s: shipped %s  chunk-7 %s' % (ar, rb, len(idx), idx[:6], oa[ar, rb, idx[:6]], ob[ar, rb, idx[:6]]))
        ea, eb = A.episode(), B.episode()
        print('   episode records equal:', {k: bool(np.array_equal(ea[k], eb[k])) for k in ea if not np.array_equal(ea[k], eb[k])} or 'all')
        shown += 1
        if shown >= 4:
            break
print('done after', t + 1, 'steps')
