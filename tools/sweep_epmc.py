"""EPMC step-kernel time.  python tools/sweep_epmc.py "4096:0,4096:1,4096:3"   (n_envs:element_id)"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import epmc_capi, urdf_model

from env_configs import epmc_env_config as env_config  # noqa: E402
blob = urdf_model.default_model_blob()
for item in sys.argv[1].split(','):
    n, el = [int(x) for x in item.split(':')]
    E = epmc_capi.EpmcEngine(epmc_capi.make_epmc_config(n, env_config(el), auto_reset=1, seed=1), blob, lib_path=os.environ.get('LL_LIB'))
    E.reset()
    for _ in range(30):
        E.fill_random_actions(math.exp(-2)); E.step()
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(50):
        E.fill_random_actions(math.exp(-2)); E.step()
    E.sync(); dt = time.perf_counter() - t0
    ms, k = E.kernel_time_ms()
    c = E.counters()
    print('element %d n_envs %6d kernel %.3f ms  wall/step %.3f ms  -> %.2f M env-steps/s   episodes %d' % (el, n, ms, dt / 50 * 1e3, n * 50 / dt / 1e6, c['episodes']), flush=True)
    E.close()
