"""GPU sweep: step-kernel time per spec n_envs:epw:solver_iterations:substeps:steps_per_launch.  python tools/sweep.py "4096:4,8192:4:10:10:32,..."
LL_SWEEP_SPEC=friction_mode=2 : spec switches for every engine of the sweep (include/llenv_model.h LLM_SPEC_*)."""
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
    spl = parts[4] if len(parts) > 4 else 1
    NT = 50 if spl == 1 else max(4, 512 // spl)     # timed launches

    def go():
        if spl == 1:
            E.step_random(math.exp(-2))
        else:
            E.step_random_n(math.exp(-2), spl)
    cfg = capi.make_config(n, control_freq=50.0, sim_freq=50.0 * nsub, kd=0.5, reward_weights=RW, prop_type=PT, prioritized_sample_factor=3.0, auto_reset=1, seed=1, solver_iterations=iters)
    E = capi.Engine(cfg, blob, table, lib_path=os.environ.get('LL_LIB'))
    E.set_spec(**{k: float(v) for k, v in (kv.split('=') for kv in os.environ.get('LL_SWEEP_SPEC', '').split(',') if kv)})
    E.reset()
    for _ in range(max(2, 32 // spl)):
        go()
    E.sync(); E.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(NT):
        go()
    E.sync(); dt = time.perf_counter() - t0
    ms, k, st = E.kernel_time_stats()
    print('iters %2d nsub %2d spl %3d' % (iters, nsub, spl), 'n_envs %6d epw %2d blocks %5d kernel %.4f ms/step  wall/step %.4f ms  -> %.2f M env-steps/s' % (
        n, epw, (n + 3) // 4, ms * k / st, dt / st * 1e3, n * st / dt / 1e6), flush=True)
    E.close()
