"""Per-wave timeline of pmc_step_kernel from the PMC_TS wall-clock stamps of an ablation build (tools/ablate.sh build).

    LL_DEBUG_FLAGS=16 LL_LIB=tools/_build/libllenv_abl.so python tools/timeline.py [n_envs]
"""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import capi, mocap, urdf_model
RW = {'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05}
PT = ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob = urdf_model.default_model_blob(); table = mocap.load_mocap('', 0.02)
cfg = capi.make_config(n, control_freq=50.0, sim_freq=500.0, kd=0.5, reward_weights=RW, prop_type=PT, prioritized_sample_factor=3.0, auto_reset=1, seed=1)
E = capi.Engine(cfg, blob, table, lib_path=os.environ['LL_LIB'])
fn = E.lib.ll_debug_timestamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
E.reset()
NAMES = {8: 'obs row written', 9: 'state + ghost stored', 0: 'entry', 1: 'state loaded', 2: 'mocap gathered', 3: 'reward', 4: 'termination', 5: 'trajectory row', 6: 'episode end/re-seed', 7: 'obs + stores (end)'}
for k in range(10):
    NAMES[10 + k] = 'substep %d' % k
for k, nm in zip(range(21, 28), ['kinematics+inertias', 'base S factor', 'free accelerations', 'candidates+selection', 'limit rows', 'contact rows', 'PGS']):
    NAMES[k] = '  s5: ' + nm
acc = []
raw = []
POLICY = len(sys.argv) > 2 and sys.argv[2] == 'policy'      # the trained reference policy (fused HIP kernel) instead of the random one
if POLICY:
    from lifelike_agility_and_play_amd.pmc_policy_hip import HipPmcPolicy
    hp = HipPmcPolicy()
for it in range(260 if POLICY else 60):
    if POLICY:
        hp.act(E); E.step()
    else:
        E.step_random(math.exp(-2))
    if it < (240 if POLICY else 40):
        continue
    ts = np.zeros((n, 32), np.uint64)
    assert fn(E.h, ts.ctypes.data_as(C.c_void_p)) == 0
    done = E.reward_done()[1]
    raw.append(ts.astype(np.float64))
    t = ts.astype(np.float64) * 0.01          # us (100 MHz)
    t0 = t[:, 0].min()
    acc.append((t - t0, np.asarray(done).astype(bool)))
order = [0, 1] + list(range(10, 15)) + list(range(21, 28)) + list(range(15, 20)) + [2, 3, 4, 5, 6, 8, 9, 7]
print('n_envs %d: stamps relative to the first wave entry, us; mean over 20 steps of [mean | max over envs], split by reset' % n)
print('%-18s %8s %8s | %8s %8s (resetting envs)' % ('mark', 'mean', 'max', 'mean', 'max'))
for k in order:
    a = np.array([x[:, k].mean() for x, d in acc]).mean(); b = np.array([x[:, k].max() for x, d in acc]).mean()
    rs = [x[d, k] for x, d in acc if d.any()]
    c = np.mean([r.mean() for r in rs]) if rs else float('nan'); e = np.mean([r.max() for r in rs]) if rs else float('nan')
    print('%-18s %8.2f %8.2f | %8.2f %8.2f' % (NAMES[k], a, b, c, e))
occ = np.array([x[:, 30:32] for x in raw]); occ_s = np.array([x[:, 28] for x in raw])          # cumulative block counts per env
d = (occ[-1] - occ[0]) / (len(raw) - 1) / 10.0
print('active 4-turn blocks per substep (wave level): contact %.2f of 4, limit %.2f of 3' % (d[:, 0].mean(), d[:, 1].mean()))
print('substeps with a self-collision row somewhere in the wave: %.1f %%' % (100.0 * ((occ_s[-1] - occ_s[0]) / (len(raw) - 1) / 10.0).mean()))
print('resets per step: %.1f' % np.mean([d.sum() for x, d in acc]))
# what makes a wave slow: its duration (end stamp - entry stamp of its first env) against the number of its substeps that carried a
# self-collision row and whether one of its envs re-seeded, least squares over all (step, wave) samples
dur, nself, rese = [], [], []
for i in range(1, len(raw)):
    t = raw[i] * 0.01
    w = (t[:, 7] - t[:, 0]).reshape(-1, 4).max(1)
    dur.append(w); nself.append((raw[i][:, 28] - raw[i - 1][:, 28]).reshape(-1, 4).max(1)); rese.append(acc[i][1].reshape(-1, 4).any(1).astype(float))
per_step = np.stack(dur)                                   # [step][wave]
tot = per_step.sum(0)
print('the slowest wave, single-step launches vs one launch over the same %d steps (waves run on without waiting for each other):' % len(per_step))
print('  mean over steps of (max over waves) %.1f us = mean wave + %.1f %%;   (max over waves of the %d-step sum) / %d = %.1f us = mean wave + %.1f %%' % (
    per_step.max(1).mean(), 100 * (per_step.max(1).mean() / per_step.mean() - 1), len(per_step), len(per_step), tot.max() / len(per_step),
    100 * (tot.max() / tot.mean() - 1)))
dur, nself, rese = np.concatenate(dur), np.concatenate(nself), np.concatenate(rese)
A = np.stack([np.ones_like(dur), nself, rese], 1)
coef, res, _, _ = np.linalg.lstsq(A, dur, rcond=None)
print('wave duration (us): mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f' % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print('  = %.1f + %.2f x (substeps with a self-collision row, mean %.1f, max %d) + %.1f x (re-seeding wave, %.1f %% of waves); residual sd %.1f' % (
    coef[0], coef[1], nself.mean(), nself.max(), coef[2], 100 * rese.mean(), np.sqrt(np.mean((A @ coef - dur) ** 2))))
for k in (0, 5, 10):
    m = nself == k
    if m.any():
        print('  waves with %2d such substeps: %5.1f %% of waves, mean duration %.1f us' % (k, 100 * m.mean(), dur[m].mean()))
E.close()
