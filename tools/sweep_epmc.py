"""EPMC step-kernel time.  python tools/sweep_epmc.py "4096:0,4096:1,4096:3"   (n_envs:element_id)"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import epmc_capi, urdf_model

from env_configs import epmc_env_config as env_config  # noqa: E402
blob = urdf_model.default_model_blob()
for item in sys.argv[1].split(','):
    parts = [int(x) for x in item.split(':')]
    n, el = parts[0], parts[1]
    spl = parts[2] if len(parts) > 2 else 1            # control steps per launch (step_random_n)
    NT = 50 if spl == 1 else max(3, 256 // spl)

    def go():
        if spl == 1:
            E.fill_random_actions(math.exp(-2)); E.step()
        else:
            E.step_random_n(math.exp(-2), spl)
    E = epmc_capi.EpmcEngine(epmc_capi.make_epmc_config(n, env_config(el), auto_reset=1, seed=1), blob, lib_path=os.environ.get('LL_LIB'))
    E.set_spec(**{k: float(v) for k, v in (kv.split('=') for kv in os.environ.get('LL_SWEEP_SPEC', '').split(',') if kv)})   # e.g. LL_SWEEP_SPEC=friction_mode=2
    E.reset()
    for _ in range(max(2, 32 // spl)):
        go()
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(NT):
        go()
    E.sync(); dt = time.perf_counter() - t0
    ms, k, st = E.kernel_time_stats()
    c = E.counters()
    print('element %d spl %3d n_envs %6d kernel %.4f ms/step  wall/step %.4f ms  -> %.2f M env-steps/s   episodes %d' % (el, spl, n, ms * k / st, dt / st * 1e3, n * st / dt / 1e6, c['episodes']), flush=True)
    E.close()
