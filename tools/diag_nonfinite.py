"""Catch the PMC engine's rare non-finite resets (2 - 3e-8 per env-step under the random policy, profiles/r04_soak.txt) in the act: the state an env
had BEFORE the control step that blew up and the action it got, saved for a replay on the float64 oracle and the host build.
    python tools/diag_nonfinite.py [max_steps] [max_events]     (GPU; writes gpurun_out/nonfinite/events.npz)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import capi, mocap, urdf_model
from bench import PMC_PROP_TYPE, PMC_REWARD_WEIGHTS   # noqa: E402

max_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
max_events = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 4096
blob, table = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02)
cfg = capi.make_config(n, control_freq=50.0, kd=0.5, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0, reward_weights=PMC_REWARD_WEIGHTS, auto_reset=1, seed=77)
E = capi.Engine(cfg, blob, table)
E.reset()
rng = np.random.default_rng(5)
sig = math.exp(-2)
events = []
t0 = time.perf_counter()
for t in range(max_steps):
    pre = E.state()
    info = E.episode_info()
    act = (rng.normal(size=(n, 12)) * sig).astype(np.float32)
    E.step_host(act)
    _, d, why = E.reward_done()
    bad = np.nonzero((why & capi.LL_DONE_NONFINITE) != 0)[0]
    for e in bad:
        events.append(dict(step=t, env=int(e), pre=pre[e].copy(), act=act[e].copy(), clip=int(info['clip'][e]), time=float(info['time'][e]), ep_steps=int(info['steps'][e])))
        print('step %d env %d: non-finite; clip %d, time %.3f, episode step %d, pre-state finite %s, max |qd| %.1f, base z %.3f' % (
            t, e, events[-1]['clip'], events[-1]['time'], events[-1]['ep_steps'], np.isfinite(pre[e]).all(), np.abs(pre[e][25:37]).max(), pre[e][2]), flush=True)
    if len(events) >= max_events:
        break
print('%d steps x %d envs, %d events, %.1f s' % (t + 1, n, len(events), time.perf_counter() - t0))
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'nonfinite'), exist_ok=True)
if events:
    np.savez(os.path.join(ROOT, 'gpurun_out', 'nonfinite', 'events.npz'), pre=np.stack([e['pre'] for e in events]), act=np.stack([e['act'] for e in events]),
             clip=np.array([e['clip'] for e in events]), time=np.array([e['time'] for e in events]), step=np.array([e['step'] for e in events]), env=np.array([e['env'] for e in events]))
E.close()
