"""SEPMC step-kernel time.  python tools/sweep_sepmc.py "2048:0,2048:1,32768:1"   (n_arenas:all-elements 0/1)"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import sepmc_capi, urdf_model


from env_configs import sepmc_env_config as env_config  # noqa: E402
blob = urdf_model.default_model_blob()
for item in sys.argv[1].split(','):
    n, el = [int(x) for x in item.split(':')]
    E = sepmc_capi.SepmcEngine(sepmc_capi.make_sepmc_config(n, env_config(el), auto_reset=1, seed=1), blob, lib_path=os.environ.get('LL_LIB'))
    E.reset()
    for _ in range(30):
        E.fill_random_actions(math.exp(-2)); E.step()
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(50):
        E.fill_random_actions(math.exp(-2)); E.step()
    E.sync()
    wall = (time.perf_counter() - t0) / 50
    ms, cnt = E.kernel_time_ms()
    c = E.counters()
    print('elements %d n_arenas %6d (%6d robots) kernel %.3f ms  wall/step %.3f ms  -> %.2f M robot-steps/s   episodes %d' % (el, n, 2 * n, ms, wall * 1e3, 2 * n / wall / 1e6, c['episodes']))
    E.close()
