"""SEPMC step-kernel time.  python tools/sweep_sepmc.py "2048:0,2048:1,32768:1"   (n_arenas:all-elements 0/1)"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import sepmc_capi, urdf_model


def env_config(elements):
    return {'arena_id': 'CTG', 'render': False, 'control_freq': 50.0,
            'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
            'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'friction_range': [0.4, 3.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]}},
            'element_config': {'rand_cube': bool(elements), 'hurdle': bool(elements), 'hole': bool(elements)}}


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
