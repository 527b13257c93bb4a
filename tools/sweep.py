"""GPU sweep: step-kernel time per spec n_envs:epw:solver_iterations:substeps.  python tools/sweep.py "4096:4,8192:4:10:10,..." """
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import capi, mocap, urdf_model
RW = {'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05}
PT = ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']
blob = urdf_model.default_model_blob(); table = mocap.load_mocap('', 0.02)
for item in sys.argv[1].split(','):
    parts = [int(x) for x in item.split(':')]
    n, epw = parts[0], parts[1]
    iters = parts[2] if len(parts) > 2 else 10
    nsub = parts[3] if len(parts) > 3 else 10
    cfg = capi.make_config(n, control_freq=50.0, sim_freq=50.0 * nsub, kd=0.5, reward_weights=RW, prop_type=PT, prioritized_sample_factor=3.0, auto_reset=1, seed=1, solver_iterations=iters)
    E = capi.Engine(cfg, blob, table, lib_path=os.environ.get('LL_LIB'))
    E.reset()
    for _ in range(30):
        E.step_random(math.exp(-2))
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(50):
        E.step_random(math.exp(-2))
    E.sync(); dt = time.perf_counter() - t0
    ms, k = E.kernel_time_ms()
    print('iters %2d nsub %2d' % (iters, nsub), 'n_envs %6d epw %2d blocks %5d kernel %.3f ms  wall/step %.3f ms  -> %.2f M env-steps/s' % (n, epw, (n + 3) // 4, ms, dt / 50 * 1e3, n * 50 / dt / 1e6), flush=True)
    E.close()
