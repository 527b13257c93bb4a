"""Diagnostic (GPU): where do sepmc_step_kernel<1> and <2> part by more than the oracle bars in one control step?  Re-runs the loop of
tests/test_gpu_sepmc.py::test_both_register_budgets_compute_the_same_gpu and prints, for every row-step outside the bars, the joint that moved,
its angle against its URDF limits and its rates on both sides (LLM_LIMIT_GATE = 20 rad/s decides whether a limit row enters the solve)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import sepmc_parity_common as SC  # noqa: E402
from parity_common import quat_align  # noqa: E402
from lifelike_agility_and_play_amd import urdf_model as um  # noqa: E402

blob = um.default_model_blob()
lo, hi = blob[um.OFF_Q_LO:um.OFF_Q_LO + 12], blob[um.OFF_Q_HI:um.OFF_Q_HI + 12]
n_small, n_big = 32, 2048 + 64
cfg = SC.env_config(SC.ALL_ELEMENTS)
A = SC.make_engine(cfg, n_small, None, seed=6)
B = SC.make_engine(cfg, n_big, None, seed=6)
A.reset(); B.reset()
rows = A.state().reshape(-1, 37).shape[0]
rng = np.random.default_rng(8)
for t in range(30):
    act = (rng.normal(size=B.obs().shape[:-1] + (12,)) * 0.2).astype(np.float32)
    pre = A.state().reshape(-1, 37).astype(np.float64)
    A.step_host(act.reshape(-1, 12)[:rows].reshape(A.obs().shape[:-1] + (12,))); B.step_host(act)
    sa = A.state().reshape(-1, 37).astype(np.float64)
    sb_all = B.state()
    sb = sb_all.reshape(-1, 37)[:rows].astype(np.float64)
    err = np.abs(np.stack([quat_align(sb[i], sa[i]) for i in range(rows)]) - sa)
    c = np.maximum(err[:, 0:7].max(1), err[:, 13:25].max(1))
    v = np.maximum(err[:, 7:13].max(1), err[:, 25:37].max(1)) / (1.0 + np.abs(sa[:, 25:37]).max(1))
    for i in np.nonzero((c >= 1e-4) | (v >= 1e-3))[0]:
        j = int(np.argmax(err[i, 25:37]))
        print('step %d row %d: config %.2e vel(rel) %.2e | joint %d: q %.4f -> %.4f / %.4f  (limits %.4f .. %.4f), rate %.2f -> %.2f / %.2f | base z %.3f up_z %.2f | max |rate| before %.1f after %.1f'
              % (t, i, c[i], v[i], j, pre[i, 13 + j], sa[i, 13 + j], sb[i, 13 + j], lo[j], hi[j], pre[i, 25 + j], sa[i, 25 + j], sb[i, 25 + j], pre[i, 2],
                 1 - 2 * (pre[i, 3] ** 2 + pre[i, 4] ** 2), np.abs(pre[i, 25:37]).max(), np.abs(sa[i, 25:37]).max()))
    flat = sb_all.reshape(-1, 37)
    flat[:rows] = A.state().reshape(-1, 37)
    B.set_state(flat.reshape(sb_all.shape))
A.close(); B.close()
