"""SEPMC step-kernel time.  python tools/sweep_sepmc.py "2048:0,2048:1,32768:1"   (n_arenas:all-elements 0/1)"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import sepmc_capi, urdf_model


from env_configs import sepmc_env_config as env_config  # noqa: E402
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
    E = sepmc_capi.SepmcEngine(sepmc_capi.make_sepmc_config(n, env_config(el), auto_reset=1, seed=1), blob, lib_path=os.environ.get('LL_LIB'))
    E.set_spec(**{k: float(v) for k, v in (kv.split('=') for kv in os.environ.get('LL_SWEEP_SPEC', '').split(',') if kv)})   # e.g. LL_SWEEP_SPEC=friction_mode=2
    E.reset()
    for _ in range(max(2, 32 // spl)):
        go()
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(NT):
        go()
    E.sync()
    ms, cnt, st = E.kernel_time_stats()
    wall = (time.perf_counter() - t0) / st
    c = E.counters()
    print('elements %d spl %3d n_arenas %6d (%6d robots) kernel %.4f ms/step  wall/step %.4f ms  -> %.2f M robot-steps/s   episodes %d' % (el, spl, n, 2 * n, ms * cnt / st, wall * 1e3, 2 * n / wall / 1e6, c['episodes']), flush=True)
    E.close()
