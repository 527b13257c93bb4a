"""At which source revision does the seven-rays-per-chunk build of the one-wave-per-SIMD chase-tag kernels part from the three-ray build?  For every commit given,
tools/_build/bis/libllenv_<commit>_c7.so against ..._c3.so (both built from that commit's csrc/ with today's flags): round 4's procedure (tools/diag_sepmc_rays.py then):
201 arenas, all elements, the same seed and actions, 40 control steps, every observation entry and every state entry compared bit for bit.
Today's python binding drives the old libraries (the ABI only gained functions since; missing ones are tolerated here and nowhere else).

    gpurun -- 'python tools/diag_sepmc_bisect.py 9b75351 ... > gpurun_out/bisect.txt 2>&1'
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


class TolerantCDLL(ctypes.CDLL):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if not name.startswith('ll_'):
                raise

            def missing(*a):
                raise RuntimeError(name + ' is not in this library')
            return missing


ctypes.CDLL = TolerantCDLL
import sepmc_parity_common as SC  # noqa: E402

NAMES = [(0, 135, 'prop'), (135, 460, 'height grid'), (460, 588, 'fan'), (588, 913, 'front rays'), (913, 918, 'percept_vec'), (918, 948, 'oppo_info(+cheat)'), (948, 962, 'flag_info(+cheat)'), (962, 965, 'with_flag, spd')]
N, STEPS = int(os.environ.get('LL_DIAG_N', '201')), int(os.environ.get('LL_DIAG_STEPS', '40'))
for h in sys.argv[1:]:
    libs = [os.path.join(ROOT, 'tools', '_build', 'bis', 'libllenv_%s_c%d.so' % (h, c)) for c in (3, 7)]
    try:
        cfg = SC.env_config(SC.ALL_ELEMENTS)
        from lifelike_agility_and_play_amd import sepmc_capi, urdf_model
        A, B = (sepmc_capi.SepmcEngine(sepmc_capi.make_sepmc_config(N, cfg, auto_reset=1, seed=3), urdf_model.default_model_blob(), lib_path=l) for l in libs)
        A.reset(); B.reset()
        rng = np.random.default_rng(0)
        steps_off, per_field, state_off, first = 0, {}, 0, None
        for t in range(STEPS):
            act = (rng.normal(size=(N, 2, 12)) * 0.135).astype(np.float32)
            A.step_host(act); B.step_host(act)
            oa, ob, sa, sb = A.obs(), B.obs(), A.state(), B.state()
            if not np.array_equal(sa, sb):
                state_off += 1
                B.set_state(sa.astype(np.float64))
            d = oa != ob
            if d.any():
                steps_off += 1
                for a, b, n in NAMES:
                    if d[..., a:b].any():
                        per_field[n] = per_field.get(n, 0) + int(d[..., a:b].any(-1).sum())
                if first is None:
                    ar, rb = np.argwhere(d.any(-1))[0]
                    idx = np.nonzero(d[ar, rb])[0][:4]
                    first = 'step %d arena %d robot %d entries %s: three rays %s seven rays %s' % (t, ar, rb, idx.tolist(), oa[ar, rb, idx].tolist(), ob[ar, rb, idx].tolist())
        print('%s: steps with a differing observation %d of %d, robots per field %s, steps with a differing state %d%s' % (h, steps_off, STEPS, per_field or '{}', state_off, ('; first: ' + first) if first else ''), flush=True)
        A.close(); B.close()
    except Exception as e:                                   # an old library this binding cannot drive
        print('%s: %s: %s' % (h, type(e).__name__, str(e)[:300]), flush=True)
