"""Kernel time per control step as a function of the step index after a reset of all envs (single-step launches, HIP events per launch): what a
driver-style run of 20 steps after 5 sees against the steady state.   LL_LIB=tools/_build/libllenv_abl.so LL_DEBUG_FLAGS=... python tools/phase_profile.py"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import capi, mocap, urdf_model
from bench import PMC_PROP_TYPE, PMC_REWARD_WEIGHTS
cfg = capi.make_config(4096, control_freq=50.0, kd=0.5, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0, reward_weights=PMC_REWARD_WEIGHTS, auto_reset=1, seed=1234)
E = capi.Engine(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), lib_path=os.environ.get('LL_LIB'))
rows = []
for rep in range(3):
    E.reset()
    E.sync(); E.enable_kernel_timing(True)
    t = []
    for s in range(200):
        E.step_random(math.exp(-2))
        ms, n = E.kernel_time_ms()
        t.append(ms * 1e3)
    rows.append(t)
t = np.mean(rows, axis=0)
print('flags', os.environ.get('LL_DEBUG_FLAGS', '0'), 'kernel us per step, mean of 3 runs:')
for a, b in ((0, 5), (5, 15), (15, 25), (25, 40), (40, 60), (60, 100), (100, 200)):
    print('  steps %3d..%3d: %.1f' % (a, b - 1, t[a:b].mean()))
E.close()
