"""The PCIe-inclusive rate of the host-buffer route through the C ABI (ll_set_actions -> ll_step -> ll_get_obs / ll_get_reward_done): what a
caller that keeps its policy on the CPU would see.  bench.py's `value` is the device-resident rate; this one is only noted in DESIGN.md 7.
    python tools/host_path_rate.py [n_envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)
from lifelike_agility_and_play_amd import capi, mocap, urdf_model
from bench import PMC_PROP_TYPE, PMC_REWARD_WEIGHTS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = capi.make_config(n, control_freq=50.0, kd=0.5, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0, reward_weights=PMC_REWARD_WEIGHTS, auto_reset=1, seed=5)
E = capi.Engine(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02))
E.reset()
rng = np.random.default_rng(0)
acts = (rng.normal(size=(8, n, 12)) * 0.135).astype(np.float32)
for t in range(20):
    E.step_host(acts[t % 8]); E.obs(); E.reward_done()
t0 = time.perf_counter()
for t in range(steps):
    E.step_host(acts[t % 8])          # ll_set_actions (host -> device) + ll_step
    o = E.obs()                       # ll_get_obs (device -> host, synchronises)
    r, d, why = E.reward_done()
dt = (time.perf_counter() - t0) / steps
print('host-buffer route, %d envs: %.3f ms per step -> %.2f M env-steps/s (%.1f MB over PCIe per step)' % (n, dt * 1e3, n / dt / 1e6, n * (207 + 12 + 1) * 4 / 1e6 + n * 2 / 1e6))
E.close()
