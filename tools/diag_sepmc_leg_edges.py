"""Round 6: which of this round's changes moves the two SEPMC nets (GPU build vs host build of the same source; one-wave-per-SIMD build vs the 256-register build)?
Runs both with the spec as shipped, with leg_edges = 0, and with the rays fused (LL_SPLIT_RAYS=0).  GPU.   python tools/diag_sepmc_leg_edges.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch  # noqa: F401
import sepmc_parity_common as SC
import epmc_parity_common as ec
from parity_common import quat_align

emul_dir = os.path.join(ROOT, 'tests', 'emul')
subprocess.check_call(['make', '-C', emul_dir, '-s'])
lib = os.path.join(emul_dir, '_build', 'libllenv_emul.so')


def budgets(spec):
    with ec.spec_variant(**spec):
        cfg = SC.env_config(SC.ALL_ELEMENTS)
        A = SC.make_engine(cfg, 32, None, seed=6); B = SC.make_engine(cfg, 2048 + 64, None, seed=6)
    A.reset(); B.reset()
    rows = A.state().reshape(-1, 37).shape[0]
    rng = np.random.default_rng(8)
    all_c, all_v = [], []
    for t in range(30):
        act = (rng.normal(size=B.obs().shape[:-1] + (12,)) * 0.2).astype(np.float32)
        A.step_host(act.reshape(-1, 12)[:rows].reshape(A.obs().shape[:-1] + (12,))); B.step_host(act)
        sa = A.state().reshape(-1, 37).astype(np.float64); sb_all = B.state(); sb = sb_all.reshape(-1, 37)[:rows].astype(np.float64)
        err = np.abs(np.stack([quat_align(sb[i], sa[i]) for i in range(rows)]) - sa)
        all_c.append(np.maximum(err[:, 0:7].max(1), err[:, 13:25].max(1)))
        all_v.append(np.maximum(err[:, 7:13].max(1), err[:, 25:37].max(1)) / (1.0 + np.abs(sa[:, 25:37]).max(1)))
        flat = sb_all.reshape(-1, 37); flat[:rows] = A.state().reshape(-1, 37); B.set_state(flat.reshape(sb_all.shape))
    c, v = np.concatenate(all_c), np.concatenate(all_v)
    out = (c >= 1e-4) | (v >= 1e-3)
    A.close(); B.close()
    return 'outside the bars %d of %d (worst config %.2e, velocity %.2e), median %.1e' % (out.sum(), len(c), c.max(), v.max(), np.median(c))


for name, spec, env in (('as shipped', {}, {}), ('leg_edges=0', dict(leg_edges=0), {}), ('rays fused', {}, {'LL_SPLIT_RAYS': '0'})):
    os.environ.pop('LL_SPLIT_RAYS', None); os.environ.update(env)
    o = SC.check_engine_against_emulation(lib, n_arenas=2048, steps=2, spec=spec, report_only=True)
    print('== %s: GPU vs host build: rough %d, left out %d, visibility ties %d, worst tail %.2e' % (name, o['rough'], o['left_out'], o.get('visibility_ties', 0), o['worst_tail']))
    for k, v in o['per_field'].items():
        print('   %-36s %s' % (k, v))
    print('   register budgets:', budgets(spec), flush=True)
