"""bench.py -- env-steps/sec of the PMC tracking-env hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one 50 Hz control step (10 x 2 ms physics substeps + mocap lookup + obs + reward + termination +
in-kernel re-seed of finished episodes) of 4096 environments per GPU on all 62 mocap clips, flat terrain, with
the random policy a ~ N(0, e^-2) drawn on device.  The timed region runs ll_step_random_n: --steps-per-launch (default 32) complete
control steps per kernel launch -- the random policy needs nothing from the host between two steps, so every wavefront walks its own
envs through them without waiting for the slowest wave of each step (--steps-per-launch 1: one launch per control step).  Inputs are
resident in HBM before the timed region.  Weak scaling: every rank owns 4096 envs; for N > 1 the step kernel also writes every
transition into the env's unroll in the learner's wire format (X | A | neglogp | R | V | r | mask: 224 floats), TD(lambda) returns are
filled in per 128-step unroll, and rank 0 gathers the unrolls over RCCL, double-buffered against the next unroll's
steps, inside the timed region (--gather-mode blocking / none: the A/B legs of the overlap measurement).

Rank 0 prints ONE JSON line (see the driver contract), including
  roofline     : HBM roofline of the step kernel from HIP-event timings taken on the engine's launch stream
  cpu_baseline : the float64 CPU oracle timed on this box's host cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC: RCCL across processes needs it

import numpy as np  # noqa: E402

ENVS_PER_GPU = 4096
SIGMA = math.exp(-2.0)
ALGO_BYTES_PER_ENV_STEP = 2552          # SURVEY.md 8(d) / DESIGN.md "algorithmic bytes"
HBM_PEAK_GBPS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec
UNROLL = 128                            # example_pmc_train.sh:145 unroll_length
STEPS_PER_LAUNCH = 32                   # control steps per launch of the random-policy loop (ll_step_random_n); divides UNROLL
GAMMA, LAMBDA = 0.95, 0.95              # example_pmc_train.sh:21-22

PMC_REWARD_WEIGHTS = {'joint_pos': 0.3, 'joint_vel': 0.05, 'end_effector': 0.1, 'root_pose': 0.5, 'root_vel': 0.05}
PMC_PROP_TYPE = ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g']


def build_record():
    """hipcc's version and the sha256 of the gfx950 code object this process runs (written next to the library by __graft_entry__.build_hip; round-5 review #9)."""
    try:
        import __graft_entry__ as ge
        return ge.build_info()
    except Exception as e:               # noqa: BLE001
        return {'error': repr(e)}


def effective_cores():
    """Host threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container can see 256
    hardware threads and be allowed 8 cores' worth of time: 256 OpenMP threads then only take turns)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]                      # cgroup v2
        if q != 'max':
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())              # cgroup v1
            p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(math.ceil(quota))))
    return eff, n, quota


def cpu_baseline(blob, table, budget_s=8.0):
    """The oracle (a port, not the product; -O3 -march=native, OpenMP over envs) on bounded samples of the same workload, random policy:
    BASELINE config 1 as SURVEY 8d specifies it (ONE env, the walk clip alone, one thread), then config 2's mix (all clips) on one core
    and on every core this process may use.  The all-cores figure is the reported value."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import oracle as orc
    from lifelike_agility_and_play_amd import mocap
    cores, visible, quota = effective_cores()

    def run(n, threads, budget, tab):
        cfg = orc.make_config(n_envs=n, reward_weights=PMC_REWARD_WEIGHTS, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0)
        B = orc.OracleBatch(cfg, blob, tab)
        rng = np.random.default_rng(0)

        def reseed(i):
            c = int(rng.integers(0, B.n_clips))
            B.reset_env(i, c, float(rng.uniform(0, 1) * B.motion_duration(c)))
        for i in range(n):
            reseed(i)
        steps, eps, spent = 0, 0, 0.0                          # only the oracle's own time is counted (the Python glue between steps is not)
        while spent < budget:
            a = rng.normal(size=(n, 12)) * SIGMA
            t0 = time.perf_counter()
            _, _, d = B.step_all_mt(a, threads) if threads > 1 else B.step_all(a)
            spent += time.perf_counter() - t0
            steps += n
            for i in np.where(d)[0]:
                reseed(int(i)); eps += 1
        return steps, spent, eps
    walk = mocap.load_mocap('dog_quad_walkrun_001_ret.txt', 1.0 / 50.0)
    s0, t0, e0 = run(1, 1, budget_s / 2, walk)                  # config 1
    s1, t1, _ = run(64, 1, budget_s, table)
    sn, tn, _ = run(max(64, 16 * cores), cores, budget_s, table) if cores > 1 else (s1, t1, 0)
    return {'value': sn / tn, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'one_core_value': s1 / t1,
            'config1': {'value': s0 / t0, 'unit': 'env-steps/s', 'cores': 1,
                        'sample': '%d env-steps (%d episodes) of ONE env on dog_quad_walkrun_001_ret.txt alone, one thread, %.1f s' % (s0, e0, t0)},
            'host': {'threads_visible': visible, 'cgroup_cpu_quota': quota, 'threads_used': cores},
            'reference_cap': 'the reference env itself sleeps to real time: 50 env-steps/s per env (PLE:241-244)',
            'sample': '%d env-steps on %d threads (%d envs) + %d env-steps on 1 thread (64 envs), all clips, random policy, %.0f s + %.0f s '
                      'of the float64 oracle (oracle/pmc_oracle.c, -O3 -march=native, OpenMP over envs; Python glue between steps not timed)'
                      % (sn, cores, max(64, 16 * cores), s1, tn, t1)}


def committed_counters(kernel, units, spl=1):
    """PMC counters cannot be read from inside this process: `traffic` (HBM bytes per launch, FETCH_SIZE + WRITE_SIZE) and the
    issue-slot accounting of the dominant kernel come from the committed rocprofv3 --pmc passes of the same command at the same
    batch (tools/profile.sh -> tools/profile_summarize.py -> profiles/traffic.json and the per-kernel counters file it names)."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))[kernel]
        k = int(t.get('control_steps_per_launch', 1))                  # the committed passes ran launches of k control steps
        if t['units_per_launch'] != units or k != spl:
            return None, None, None
        c = json.load(open(os.path.join(ROOT, t['counters_file'])))
        n_inst = c['SQ_INSTS_VALU'] + c['SQ_INSTS_SALU'] + c['SQ_INSTS_LDS']
        issue = {'instructions_per_wave_per_control_step': n_inst / c['SQ_WAVES'] / k, 'issue_slots_per_wave_per_control_step': c['SQ_WAVE_CYCLES'] / c['SQ_WAVES'] / k,
                 'valu_per_wave_per_control_step': c['SQ_INSTS_VALU'] / c['SQ_WAVES'] / k,
                 'control_steps_per_launch': k, 'frac': n_inst / c['SQ_WAVE_CYCLES'],
                 'source': t['counters_file'] + ' (rocprofv3 --pmc SQ_INSTS_*, SQ_WAVE_CYCLES in quad-cycles)'}
        traffic = t['traffic_bytes'] + (t.get('percept_traffic_bytes') or 0.0)          # (the ray kernel behind a step kernel whose rays were split off: one launch pair = one control step)
        return traffic, issue, 'profiles/traffic.json <- %s (bytes per launch, FETCH_SIZE + WRITE_SIZE, uncorrected; see DESIGN.md 5.1)' % t['counters_file']
    except Exception:                    # noqa: BLE001
        return None, None, None


# ---- the N > 1 failure path (round-5 review #6: the first real 8-rank RCCL exchange will be the driver's, on a box nobody can rehearse on) ----------------
# Whatever goes wrong in a rank -- an exception, a peer that died (the launcher then terminates the others), a collective that never returns --
# ONE JSON line {"error", "rank", "phase", "rccl_version_line", ...} reaches stdout before the rank exits non-zero, the way the reference's actor
# logs before it reboots (bin/run_pg_actor.py:143-155).  There is no fallback for the contract line: a failed gather is reported, not retried without it.
METRIC = 'env-steps/sec (whole node), PMC tracking env, random policy'
_PHASE = ['init']                       # init | warmup | timed | gather | report
_BEAT = [time.monotonic()]
_FAIL = {'armed': False, 'rank': 0, 'world': 1, 'lock': None, 'rccl_log': None, 'stall_s': float(os.environ.get('LL_BENCH_STALL_S', '150'))}


def set_phase(p):
    _PHASE[0] = p
    _BEAT[0] = time.monotonic()


def _rccl_lines():
    ver, tail = None, []
    try:
        lines = [l.rstrip() for l in open(_FAIL['rccl_log'] or '').read().splitlines() if l.strip()]
        v = [l for l in lines if 'version' in l.lower()]
        ver, tail = (v[0].strip() if v else None), lines[-8:]
    except OSError:
        pass
    return ver, tail


def report_failure(what, code=1):
    """The one error line of a failed launch: the first rank to fail claims it (an exclusive lock file per launch), later ones only write to stderr."""
    ver, tail = _rccl_lines()
    line = json.dumps({'error': str(what)[-2000:], 'rank': _FAIL['rank'], 'world': _FAIL['world'], 'phase': _PHASE[0], 'rccl_version_line': ver, 'rccl_log_tail': tail,
                       'metric': METRIC, 'value': None, 'n_gpus': _FAIL['world']})
    first = True
    if _FAIL['lock']:
        try:
            os.close(os.open(_FAIL['lock'], os.O_CREAT | os.O_EXCL | os.O_WRONLY))
        except FileExistsError:
            first = False
        except OSError:
            pass
    print(line, file=sys.stdout if first else sys.stderr, flush=True)
    os._exit(code)                       # no destructors: a process group with a dead peer does not tear down


def arm_failure_path(rank, world):
    """A helper thread that can speak while the main thread sits in a collective or a stream wait: it hears SIGTERM (the launcher ending the ranks because a
    peer died) through the wake-up descriptor, and it notices a timed region that makes no progress for LL_BENCH_STALL_S seconds."""
    import signal
    import socket
    import tempfile
    import threading
    _FAIL.update(rank=rank, world=world, armed=True)
    if world > 1:
        _FAIL['lock'] = os.path.join(tempfile.gettempdir(), 'll_bench_err_%s_%d.lock' % (os.environ.get('MASTER_PORT', '0'), os.getppid()))
    r, w = socket.socketpair()
    w.setblocking(False)
    signal.signal(signal.SIGTERM, lambda *_: None)           # (the C-level handler writes the signal number to the descriptor at once, whatever the main thread is doing)
    signal.set_wakeup_fd(w.fileno(), warn_on_full_buffer=False)
    _FAIL['socks'] = (r, w)

    def watch():
        import select
        while _FAIL['armed']:
            ready, _, _ = select.select([r], [], [], 1.0)
            if ready and signal.SIGTERM in r.recv(64):
                report_failure('terminated by the launcher in phase %s: a peer rank failed or the launch was cut off' % _PHASE[0], 143)
            if _PHASE[0] in ('init', 'warmup', 'timed', 'gather') and time.monotonic() - _BEAT[0] > _FAIL['stall_s']:
                report_failure('no progress for %.0f s in phase %s (a collective or a stream wait that does not return)' % (_FAIL['stall_s'], _PHASE[0]), 3)
    threading.Thread(target=watch, daemon=True).start()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line, one per GPU, the way the driver's
    explicit launch line does (torch.distributed.run, rendezvous on 127.0.0.1, a free port), and pass rank 0's ONE JSON line through.  A launch that
    fails has printed one error line from the first rank that noticed (report_failure); if no rank got that far the launcher's exit code is reported here."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '1')                  # what torchrun would set (with a warning) anyway
    sys.stdout.flush()
    proc = subprocess.Popen(cmd, env=env)                   # the ranks inherit stdout: rank 0 prints the line
    rc = proc.wait()
    lock = os.path.join(tempfile.gettempdir(), 'll_bench_err_%d_%d.lock' % (port, proc.pid))
    reported = os.path.exists(lock)
    if reported:
        os.unlink(lock)
    if rc != 0:
        if not reported:
            print(json.dumps({'error': 'the launcher (torch.distributed.run) exited with code %d and no rank reported' % rc, 'rank': None, 'world': n, 'phase': 'launch',
                              'rccl_version_line': None, 'metric': METRIC, 'value': None, 'n_gpus': n}), flush=True)
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-single-step-leg', action='store_true', help='skip the secondary one-launch-per-control-step measurement (N = 1)')
    ap.add_argument('--workload', choices=['pmc', 'epmc', 'sepmc'], default='pmc',
                    help="pmc = BASELINE config 2 (the contract line); epmc = config 4 (PlayGroundEnv, DESIGN.md 8), sepmc = config 5 (ChaseTagGameEnv, 2048 arenas x 2 robots, DESIGN.md 8b), same JSON shape")
    ap.add_argument('--element', type=int, default=1, help='epmc only: env_randomize_config element_id (0 joystick, 1 hurdles, 2 holes, 3 cubes)')
    ap.add_argument('--steps-per-launch', type=int, default=STEPS_PER_LAUNCH,
                    help='pmc: control steps per kernel launch (ll_step_random_n; the random policy needs nothing from the host between two '
                         'steps).  1 = one launch per control step (ll_step_random).  Must divide the unroll length %d when N > 1' % UNROLL)
    ap.add_argument('--gather-mode', choices=['async', 'blocking', 'none', 'p2p'], default='async',
                    help='N > 1 only: async = the double-buffered gather overlapping the next unroll (the contract line); blocking = every '
                         'gather is waited for before the next step (A/B leg: what the overlap buys); none = no gather (A/B leg: the steps alone)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)
    if args.workload == 'epmc':
        return main_epmc(args)
    if args.workload == 'sepmc':
        return main_sepmc(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 or os.environ.get('LL_BENCH_FORCE_GATHER'):
        arm_failure_path(int(os.environ.get('RANK', '0')), world)
        try:
            main_pmc(args)
        except BaseException as e:       # noqa: BLE001  (SystemExit included: a rank that bows out is a failed launch)
            import traceback
            report_failure('%s: %s | %s' % (type(e).__name__, e, ' <- '.join(l.strip() for l in traceback.format_exc().splitlines()[-6:])), 1)
        _FAIL['armed'] = False
    else:
        main_pmc(args)


def main_pmc(args):
    import torch
    import torch.distributed as dist
    from lifelike_agility_and_play_amd import capi, mocap, urdf_model, gather

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py --gpus %d inside a %d-rank launch' % (args.gpus, world))
    tc = torch.cuda.is_available()      # torch is plumbing (streams, RCCL): a single-GPU run goes on without it if only torch fails to see the device
    if not tc and world > 1:
        raise SystemExit('bench.py --gpus > 1 needs torch.cuda for the RCCL gather')
    # test hooks (a 1-GPU box cannot host two RCCL ranks): LL_BENCH_BACKEND=gloo LL_BENCH_ONE_DEVICE=1 runs the N>1
    # control flow with every rank on device 0 and the gather staged through host memory
    backend = os.environ.get('LL_BENCH_BACKEND', 'nccl')
    if os.environ.get('LL_BENCH_ONE_DEVICE'):
        local_rank = 0
    if tc:
        torch.cuda.set_device(local_rank)
    # LL_BENCH_FORCE_GATHER=1 (test hook): the N > 1 control flow -- process group, unroll recording, gather, MAX over ranks -- with a
    # ONE-rank communicator, so that RCCL itself (backend "nccl") executes on a 1-GPU box
    multi = world > 1 or bool(os.environ.get('LL_BENCH_FORCE_GATHER'))
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s_.getsockname()[1]); s_.close()
        kw = {'device_id': torch.device('cuda', local_rank)} if backend == 'nccl' else {}
        rccl_log = None
        if backend == 'nccl':
            # RCCL's own log of THIS rank (NCCL_DEBUG=WARN: its version line at communicator creation, then whatever it warns about) goes to a file per rank; the version
            # line travels in the JSON so that a scaling record can be checked against the library that ran, the tail travels in the error line of a failed launch
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), 'll_bench_rccl_%s_r%d_%d.log' % (os.environ['MASTER_PORT'], rank, os.getpid()))
            os.environ.setdefault('NCCL_DEBUG', 'WARN')
            os.environ.setdefault('NCCL_DEBUG_FILE', rccl_log)
            _FAIL['rccl_log'] = os.environ['NCCL_DEBUG_FILE']
        import datetime
        # a rendezvous or a collective that a dead or wedged peer never joins ends after LL_BENCH_PG_TIMEOUT_S (120 s) instead of the backend's 10 - 30 minutes
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get('LL_BENCH_PG_TIMEOUT_S', '120'))), **kw)

    def die_here(phase):                                   # test hook (tests/test_gpu_env_api.py): rank LL_BENCH_KILL_RANK dies without a word at the start of a phase
        if os.environ.get('LL_BENCH_KILL_RANK') == str(rank) and os.environ.get('LL_BENCH_KILL_PHASE', 'timed') == phase:
            import signal
            os.kill(os.getpid(), signal.SIGKILL)

    def dev_sync():
        eng.sync()
        if tc:
            torch.cuda.synchronize()

    n = args.envs_per_gpu
    blob = urdf_model.default_model_blob()
    table = mocap.load_mocap('', 1.0 / 50.0)
    cfg = capi.make_config(n, control_freq=50.0, sim_freq=500.0, kp=50.0, kd=0.5, max_tau=18.0,
                           reward_weights=PMC_REWARD_WEIGHTS, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0,
                           auto_reset=1, seed=1234 + rank, device=local_rank)
    eng = capi.Engine(cfg, blob, table)                # (raises LL_ENODEV without a HIP device: there is no CPU path)
    if tc:
        gather.bind_torch_stream(eng)                  # step kernels, torch ops and the RCCL gather are ordered on one stream
    eng.reset()
    traj = gather.TrajectoryBuffer(eng, UNROLL, mode=args.gather_mode) if multi else None
    if traj is not None:
        traj.prepare(0)                                    # rank 0's world x 470 MB receive buffers exist before anything is timed
        # measurement hook (tools/simd_sharing.sh): with a ONE-rank communicator RCCL's gather is a 0.3 ms local copy; repeated k times it
        # stands in for the residency of an 8-rank gather (7 x 470 MB over xGMI: several ms) on the learner rank
        traj.extra_gathers = int(os.environ.get('LL_BENCH_GATHER_REPEAT', '0'))
        traj.p2p_no_cu = os.environ.get('LL_BENCH_P2P_NO_CU', '1') == '1'     # --gather-mode p2p: SDMA pulls (0: the runtime's default copy path)

    def measure_triad():
        """the nominal HBM figure next to a device triad measured on this box (a = b + s * c on 3 x 1 GiB, torch's own kernel: a
        calibration of the roofline's denominator, not part of the product)"""
        if not (rank == 0 and world == 1 and tc):
            return None
        try:
            nel = 1 << 28
            b_ = torch.ones(nel, device='cuda', dtype=torch.float32); c_ = torch.ones(nel, device='cuda', dtype=torch.float32); a_ = torch.empty_like(b_)
            for _ in range(3):
                torch.add(b_, c_, alpha=1.5, out=a_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            iters = int(os.environ.get('LL_BENCH_TRIAD_ITERS', '10'))
            for _ in range(iters):
                torch.add(b_, c_, alpha=1.5, out=a_)
            e1.record(); torch.cuda.synchronize()
            return iters * 3 * nel * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception:                # noqa: BLE001
            return None
    # The calibration runs BEFORE the warm-up and the timed region (LL_BENCH_TRIAD_FIRST=0: after): a driver-style run of 25 control steps is
    # 4 ms of GPU work from a cold device; with 40 ms of triad in front the same 20 timed steps run 2 % faster (profiles/r03_driver_style.txt)
    triad_first = os.environ.get('LL_BENCH_TRIAD_FIRST', '1') == '1'
    triad = measure_triad() if triad_first else None

    n_done = [0]                                           # control steps executed so far == the engine's step index
    spl = max(1, args.steps_per_launch)
    # the timed region is never ONE launch, however few steps the caller asks for (the driver's own invocation times 20): a third of them per launch at most
    spl_timed = max(1, min(spl, -(-args.steps // 3)))
    if traj is not None and UNROLL % spl:
        raise SystemExit('--steps-per-launch must divide the unroll length %d' % UNROLL)

    def run_steps(count, spl=spl):
        """`count` control steps of the random-policy loop: launches of up to `spl` steps each (ll_step_random_n draws a ~ N(0, sigma^2)
        on device and steps, `spl` times per launch), cut at unroll boundaries, where the finished unroll is handed to the gather."""
        left = count
        while left > 0:
            k = min(spl, left, UNROLL - n_done[0] % UNROLL) if traj is not None else min(spl, left)
            if k == 1:
                eng.step_random(SIGMA)
            else:
                eng.step_random_n(SIGMA, k)
            _BEAT[0] = time.monotonic()
            n_done[0] += k
            left -= k
            if traj is not None and n_done[0] % UNROLL == 0:   # the step kernel itself records the rows (ll_enable_unrolls);
                u = n_done[0] // UNROLL - 1
                traj.finish(u, GAMMA, LAMBDA)                  # TD(lambda) returns of the finished unroll (one small kernel, same stream)
                if args.gather_mode != 'none':
                    traj.gather_async(u, 0)                    # the gather of this unroll overlaps with the next unroll's steps

    set_phase('warmup')
    die_here('warmup')
    run_steps(args.warmup)
    if traj is not None:
        traj.wait()
        traj.stall_ms(); traj.host_stall_s = 0.0             # stall accounting covers the timed region only
    if multi:
        dist.barrier()
    dev_sync()
    set_phase('timed')
    die_here('timed')
    eng.enable_kernel_timing(True)
    t0 = time.perf_counter()
    run_steps(args.steps, spl_timed)
    set_phase('gather')
    if traj is not None:
        traj.wait()                                          # an in-flight gather belongs to the timed region
    dev_sync()
    if multi:
        dist.barrier()
    dev_sync()
    elapsed = time.perf_counter() - t0
    k_launch_ms, k_n, k_steps = eng.kernel_time_stats()      # HIP events on the launch stream: average per launch, launches, control steps
    k_ms = k_launch_ms * k_n / k_steps if k_steps else 0.0     # ... per control step
    eng.enable_kernel_timing(False)
    gather_check = None
    gather_stats = None
    if multi:
        # how long the step kernels stood still behind a gather (events around the stream-side wait) and how long the host was blocked,
        # inside the timed region; MAX over ranks like the elapsed time
        tt = torch.tensor([elapsed, traj.stall_ms(), traj.host_stall_s * 1e3], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        gather_stats = {'mode': args.gather_mode, 'backend': backend, 'stream_stall_ms_total': float(tt[1].item()), 'host_blocked_ms_total': float(tt[2].item()),
                        'bytes_per_rank_per_unroll': int(traj.buf[0].numel() * 4)}
        # who took part: every rank reports itself over the process group (rank, pid, device ordinal and name, envs it stepped, episodes it finished)
        me = {'rank': rank, 'pid': os.getpid(), 'device': local_rank, 'device_name': torch.cuda.get_device_name(local_rank) if tc else None,
              'envs': n, 'control_steps': int(n_done[0]), 'episodes': int(eng.counters()['episodes']), 'stale_reseeds': int(eng.table_sync())}
        seen = [None] * world
        dist.all_gather_object(seen, me)
        gather_stats['ranks_seen'] = seen
        gather_stats['receive_blocks'] = [len(traj.outs), len(traj.outs[0])] if (rank == 0 and traj.outs is not None) else None     # [2][world] on the learner rank
        if backend == 'nccl':
            try:
                gather_stats['rccl_version'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
            except Exception:            # noqa: BLE001
                gather_stats['rccl_version'] = None
            gather_stats['rccl_version_line'] = _rccl_lines()[0] if rank == 0 else None
        if os.environ.get('LL_BENCH_VERIFY') and traj.n_gathered > 0:
            # what rank 0 received for the last gathered unroll == what each rank's engine holds in that block (float64 checksums + probes)
            k_last = traj.n_gathered - 1
            mine = traj.half(k_last).double().cpu()
            sig = [float(mine.sum()), float(mine.abs().sum()), float(mine[0, 0, 0]), float(mine[-1, -1, -1]), float(mine[n // 2, UNROLL // 2, 207])]
            sigs = [None] * world
            dist.all_gather_object(sigs, sig)
            if rank == 0:
                got = [o.double().cpu() for o in traj.last]
                got_sig = [[float(g.sum()), float(g.abs().sum()), float(g[0, 0, 0]), float(g[-1, -1, -1]), float(g[n // 2, UNROLL // 2, 207])] for g in got]
                gather_check = 'ok' if got_sig == sigs and len({tuple(x) for x in sigs}) == world else 'MISMATCH %r vs %r' % (got_sig, sigs)
    set_phase('report')
    counters = eng.counters()
    stale_reseeds = eng.table_sync()                         # (before the single-step leg; 0 unless the chip was shared: ll_get_table_sync)
    ep_hist = [int(x) for x in eng.episode_histogram()]
    # The like-for-like leg (round-3 advice): the same loop as ONE launch per control step -- what an actor with a policy between the steps
    # runs, and what rounds 1-2 reported; the prioritized-sampling table is folded after every step, as PLE:235-240 does.  Measured after the
    # contract region on the same engine (N = 1 only; a short region of its own), reported beside `value`, never instead of it.
    single = None
    if not multi and spl > 1 and not args.no_single_step_leg:
        ks = max(1, min(args.steps, 200))
        for _ in range(5):
            eng.step_random(SIGMA)
        dev_sync()
        eng.enable_kernel_timing(True)
        ts = time.perf_counter()
        for _ in range(ks):
            eng.step_random(SIGMA)
        dev_sync()
        ts = time.perf_counter() - ts
        s_launch_ms, s_n, s_steps = eng.kernel_time_stats()
        eng.enable_kernel_timing(False)
        single = {'value': n * ks / ts, 'unit': 'env-steps/s', 'steps': ks, 'ms_per_step': ts / ks * 1e3, 'kernel_avg_ms': s_launch_ms,
                  'note': 'one kernel launch per control step (ll_step_random), the sampling table folded after every step: the loop an actor with a '
                          'policy between the steps runs'}
    if triad is None and not triad_first:
        triad = measure_triad()

    if rank == 0:
        total_env_steps = world * n * args.steps
        value = total_env_steps / elapsed
        # with the unroll recorded for the gather (N > 1) a step also writes its 224-float row (obs the policy acted on, action,
        # neglogp, R, V, r, mask) -- stated and added to the algorithmic bytes
        algo_bytes = ALGO_BYTES_PER_ENV_STEP + (4 * int(traj.buf.shape[-1]) if traj is not None else 0)
        achieved = (n * algo_bytes) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        traffic, issue, tsrc = committed_counters('pmc_step_kernel', n, spl)               # (the committed passes ran launches of `spl` steps)
        cspl = (k_steps / k_n) if k_n else float(spl_timed)                               # control steps per timed launch, as run
        if issue and k_ms > 0:
            # chip-level VALU utilisation: VALU instructions of all waves of a control step x 2 cycles (a SIMD's rate for a full-rate wave64
            # instruction with >= 2 resident waves: MI355X_MICROARCH.md, measured in profiles/r04_valu_issue.txt) over the SIMD-cycles the step took
            issue['chip_valu_frac'] = issue['valu_per_wave_per_control_step'] * (n / 4.0) * 2.0 / (k_ms * 1e-3 * 2.4e9 * 1024.0)
            issue['chip_valu_frac_note'] = ('one wave per SIMD issues at most one instruction per 4.4 cycles, and the DPP / v_med3 / packed forms this kernel '
                                            'is made of are half-rate at any occupancy (profiles/r04_valu_issue.txt): the ceiling of this mix is near 0.6, not 1')
        out = {
            'metric': METRIC,
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'float32 state, dynamics and solver (SURVEY 8d counts FP32 bytes); float64 only where the reference\'s own float64 matters: '
                          'the mocap time base and frame differences, the sampling table.  The reference computes in float64 (PyBullet)',
            'config': {'workload': 'PMC tracking env, %d parallel envs per MI355X, flat terrain, full mocap_data clip set '
                                   '(62 clips), random-policy actions N(0, e^-2), auto-reset%s' % (n, ', RCCL trajectory gather to rank 0 every %d steps' % UNROLL if multi else ''),
                       'envs_per_gpu': n, 'substeps_per_step': 10, 'solver_iterations': 10,
                       'physics_spec': {'link_inertias': urdf_model.default_inertia_source(), 'friction_mode': eng.get_spec('friction_mode'), 'limit_speculative': eng.get_spec('limit_speculative'),
                                        'contact_erp': eng.get_spec('erp'), 'limit_erp': eng.get_spec('limit_erp'), 'max_depen_speed': eng.get_spec('max_depen_speed'),
                                        'max_coord_vel': eng.get_spec('max_coord_vel')},
                       'steps_per_launch': spl_timed,
                       'steps_per_launch_note': 'the timed region runs ll_step_random_n: up to %d control steps of the random-policy loop per kernel launch, at least three '
                                                'launches (every step is complete: physics, mocap, obs, reward, termination, re-seed, unroll row, and the prioritized-sampling '
                                                'table folded after EVERY step -- a re-seed at step s of a launch draws from the table steps 0 .. s - 1 left, PLE:235-240: k steps in one '
                                                'launch are k launches bit for bit WHILE table_sync_stale_reseeds_rank0 IS 0, i.e. while the launch has the chip to itself; with another kernel resident a re-seed takes the newest '
                                                'version there is and is counted there -- LL_DETERMINISTIC=1 makes the engine run such calls as single launches); --steps-per-launch 1 is one launch per control step' % spl_timed,
                       'table_sync_stale_reseeds_rank0': stale_reseeds,
                       'episodes_finished_rank0': counters['episodes'], 'nonfinite_resets_rank0': counters['nonfinite'],
                       'mean_episode_length_steps_rank0': (counters['env_steps'] / counters['episodes']) if counters['episodes'] else None,
                       'episode_length_histogram_rank0': {'bucket_lower_edges_steps': [1 << b for b in range(16)], 'episodes': ep_hist},
                       **({'unrolls_gathered': traj.n_gathered, 'unroll_row_floats': int(traj.buf.shape[-1]), 'gather_check': gather_check,
                           'gather': gather_stats} if traj is not None else {})},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBPS) if achieved else None,
                         # per launch like `achieved`: the committed counter passes ran launches of `spl` control steps; this run's launches average cspl
                         'traffic': (traffic / spl * cspl) if traffic else None, 'traffic_per_control_step': (traffic / spl) if traffic else None,
                         'algorithmic_bytes_per_launch': n * algo_bytes * cspl, 'peak_measured_triad': triad, 'peak_measured_triad_when': 'before the warm-up steps' if triad_first else 'after the timed region',
                         'traffic_source': tsrc,
                         'kernel': 'pmc_step_kernel', 'kernel_avg_ms': k_ms, 'kernel_launches_timed': k_n,
                         'kernel_avg_launch_ms': k_launch_ms, 'control_steps_per_launch': (k_steps / k_n) if k_n else None,
                         'algorithmic_bytes_per_env_step': algo_bytes, 'single_wave_issue': issue,
                         'note': 'bound by single-wave instruction issue, not HBM (%s instructions per wave per step, four envs, vs 2.5 KB per env); see DESIGN.md 5.1' % ('%.1fe4' % (issue['instructions_per_wave_per_control_step'] / 1e4) if issue else 'about 8e4')},
        }
        out['build'] = build_record()
        if single is not None:
            out['single_step_launch'] = single
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(blob, table)
        print(json.dumps(out), flush=True)
    if traj is not None:
        if multi:
            dist.barrier()                                   # nobody unmaps what a peer may still be pulling
        traj.close()
    eng.close()
    if multi:
        dist.destroy_process_group()


EPMC_ALGO_BYTES_PER_ENV_STEP = 4 * (12 + 37 + 66 + 24 + 40 + 8 * 8) + 4 * (37 + 916 + 2 + 40)   # reads: action, state, older prop/action frames,
# per-env scalars, ~8 box records in reach; writes: state, obs (prop 99 + prop_a 36 + 778 rays + target 3), reward/done, scalars


def epmc_env_config(element_id):
    """train_scripts/example_epmc_train.sh:90-117 with the element of BASELINE config 4."""
    return {'arena_id': 'Playground', 'render': False, 'control_freq': 50.0, 'prop_type': list(PMC_PROP_TYPE), 'kp': 50.0, 'kd': 0.5, 'max_tau': 16,
            'max_steps': 1000, 'obs_randomization': {},
            'env_randomize_config': {'element_id': element_id, 'height_range': [0.0, 0.0], 'friction_range': [0.4, 3.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
                                     'cmd_vary_freq_range': [9999, 10000], 'target_spd_range': [0.5, 3.0], 'auxiliary_radius': 0.02,
                                     'hole_config': {'min_gap_height': 0.25, 'max_gap_height': 0.25}}}


def cpu_baseline_env(kind, env_config, budget_s=10.0):
    """The EPMC / SEPMC oracles put together as a CPU env (oracle/free_run.py: NumPy env logic + analytic rays + the float64 C physics), random
    policy, one core, a bounded sample.  A port (the checker), not the product and not a tuned CPU engine."""
    import numpy as np
    from oracle import oracle as orc, free_run as FR
    from lifelike_agility_and_play_amd import epmc_capi, mocap, urdf_model
    orc.build()
    blob, table, init = urdf_model.default_model_blob(), mocap.load_mocap('', 1.0 / 50.0), epmc_capi.default_init_state()
    rng = np.random.default_rng(0)
    if kind == 'epmc':
        run = FR.EpmcFreeRun(env_config, blob, table, init, seed=1)
        steps, secs, eps = FR.time_random_policy(run, budget_s, 12, rng)
        return {'value': steps / secs, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
                'sample': '%d env-steps (%d episodes) of one PlayGround env in %.1f s: oracle/epmc_oracle.py (NumPy) + oracle/pmc_oracle.c physics, random policy' % (steps, eps, secs)}
    run = FR.SepmcFreeRun(env_config, blob, table, init, seed=1)
    steps, secs, eps = FR.time_random_policy(run, budget_s, [12, 12], rng)
    return {'value': 2 * steps / secs, 'unit': 'robot-steps/s', 'cores': 1, 'kind': 'port',
            'sample': '%d arena-steps (%d episodes) of one chase-tag arena in %.1f s: oracle/sepmc_oracle.py (NumPy) + oracle/pmc_oracle.c two-robot physics, random policy' % (steps, eps, secs)}


def main_epmc(args):
    """Same measurement for the EPMC env (not the driver's contract line): env-steps/s of 4096 PlayGround envs per GPU."""
    import torch
    import torch.distributed as dist
    from lifelike_agility_and_play_amd import epmc_capi, urdf_model
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    tc = torch.cuda.is_available()
    if not tc and world > 1:
        raise SystemExit('bench.py --gpus > 1 needs torch.cuda')
    if tc:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('LL_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)
    n = args.envs_per_gpu
    cfg = epmc_capi.make_epmc_config(n, epmc_env_config(args.element), auto_reset=1, seed=1234 + rank, device=local_rank)
    eng = epmc_capi.EpmcEngine(cfg, urdf_model.default_model_blob())
    eng.reset()

    spl = max(1, args.steps_per_launch)

    def run_steps(count):                                   # launches of up to spl control steps (step_random_n draws the actions in the kernel)
        left = count
        while left > 0:
            k = min(spl, left)
            if k == 1:
                eng.fill_random_actions(SIGMA); eng.step()
            else:
                eng.step_random_n(SIGMA, k)
            left -= k
    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    eng.sync()
    if tc:
        torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    eng.sync()
    if tc:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k_launch_ms, k_n, k_steps = eng.kernel_time_stats()
    k_ms = k_launch_ms * k_n / k_steps if k_steps else 0.0    # per control step
    if world > 1:
        tt = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        achieved = n * EPMC_ALGO_BYTES_PER_ENV_STEP / (k_ms * 1e-3) / 1e9
        # how the engine ran a call of spl control steps (llenv.hip launch_epmc_step): ONE multi-step launch only with the rays fused and the grid at one wave per SIMD; otherwise
        # spl single-step launches, each followed by epmc_percept_kernel when the rays are split off (LL_SPLIT_RAYS, default 2 for this env) -- the engine's HIP events bracket the call
        ray_mode = int(os.environ.get('LL_SPLIT_RAYS', '2'))
        kspl = spl if (spl > 1 and ray_mode < 2 and n <= 4096) else 1
        split = ray_mode >= (1 if spl == 1 else 2)
        traffic, issue, tsrc = committed_counters('epmc_step_kernel', n, kspl)
        extra = {'cpu_baseline': cpu_baseline_env('epmc', epmc_env_config(args.element))} if (world == 1 and not args.no_cpu_baseline) else {}
        print(json.dumps({**extra, 'build': build_record(), **{
            'metric': 'env-steps/sec (whole node), EPMC PlayGround env, random policy', 'value': world * n * args.steps / elapsed, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'EPMC PlayGround env (BASELINE config 4), %d envs per MI355X, element_id %d, 778 rays per env-step, push forces, '
                                   'random-policy actions N(0, e^-2), auto-reset' % (n, args.element), 'envs_per_gpu': n, 'steps_per_launch': spl, 'episodes_finished_rank0': eng.counters()['episodes']},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'traffic_source': tsrc, 'traffic_note': 'counter traffic of a multi-step launch UNDER-counts HBM bytes: the rows a wave writes in one step of the launch and overwrites in the next (observation, state, bookkeeping) meet in its L2 / the Infinity Cache and need not reach HBM, so FETCH_SIZE + WRITE_SIZE can come out below the algorithmic bytes (0.8 - 0.9 x); single launches per step measure 1.1 - 1.5 x', 'single_wave_issue': issue,
                         'kernel': 'epmc_step_kernel' + (' + epmc_percept_kernel (the 778 rays per env, behind every step kernel)' if split else ''),
                         'kernel_avg_ms': k_ms, 'kernel_launches_timed': k_n if kspl > 1 else k_steps, 'kernel_avg_launch_ms': k_launch_ms if kspl > 1 else k_ms,
                         'control_steps_per_launch': kspl, 'control_steps_per_call': spl,
                         'launch_note': ('one multi-step launch per call' if kspl > 1 else 'a call of %d control steps runs as %d single-step launches%s; HIP events bracket the call, so the per-launch '
                                         'figure is the call\'s time / its steps (launch gaps included)' % (spl, spl, ', each followed by the ray kernel' if split else '')),
                         'algorithmic_bytes_per_env_step': EPMC_ALGO_BYTES_PER_ENV_STEP,
                         'algorithmic_bytes_per_launch': n * EPMC_ALGO_BYTES_PER_ENV_STEP * kspl,
                         'note': 'bound by single-wave instruction issue, not HBM; see DESIGN.md 8'}}}), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def sepmc_env_config():
    """train_scripts/example_sepmc_train.sh:93-117 (BASELINE config 5: no arena elements)."""
    return {'arena_id': 'CTG', 'render': False, 'control_freq': 50.0, 'prop_type': list(PMC_PROP_TYPE), 'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': 1000,
            'obs_randomization': {},
            'env_randomize_config': {'friction_range': [0.4, 3.0],
                                     'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]}},
            'element_config': {'rand_cube': False, 'hurdle': False, 'hole': False}}


SEPMC_ALGO_BYTES_PER_ROBOT_STEP = 37 * 4 * 2 + 965 * 4 + 135 * 4 + 12 * 4 + 40 * 4 * 2 + 13 * 8 * 4 + 16   # state r/w, obs w, history r, action, scalars r/w, boxes r, reward/done


def main_sepmc(args):
    """Same measurement for the SEPMC env (not the driver's contract line): robot-steps/s of envs_per_gpu / 2 chase-tag arenas per GPU."""
    import torch
    import torch.distributed as dist
    from lifelike_agility_and_play_amd import sepmc_capi, urdf_model
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    tc = torch.cuda.is_available()
    if not tc and world > 1:
        raise SystemExit('bench.py --gpus > 1 needs torch.cuda')
    if tc:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('LL_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)
    n_arenas = args.envs_per_gpu // 2                              # 4096 robots = the 2048 arenas of BASELINE config 5
    cfg = sepmc_capi.make_sepmc_config(n_arenas, sepmc_env_config(), auto_reset=1, seed=1234 + rank, device=local_rank)
    eng = sepmc_capi.SepmcEngine(cfg, urdf_model.default_model_blob())
    eng.reset()

    spl = max(1, args.steps_per_launch)

    def run_steps(count):                                   # launches of up to spl control steps (step_random_n draws the actions in the kernel)
        left = count
        while left > 0:
            k = min(spl, left)
            if k == 1:
                eng.fill_random_actions(SIGMA); eng.step()
            else:
                eng.step_random_n(SIGMA, k)
            left -= k
    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    eng.sync()
    if tc:
        torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    eng.sync()
    if tc:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k_launch_ms, k_n, k_steps = eng.kernel_time_stats()
    k_ms = k_launch_ms * k_n / k_steps if k_steps else 0.0    # per control step
    if world > 1:
        tt = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        achieved = 2 * n_arenas * SEPMC_ALGO_BYTES_PER_ROBOT_STEP / (k_ms * 1e-3) / 1e9
        ray_mode = int(os.environ.get('LL_SPLIT_RAYS', '1'))          # (see main_epmc; this env's default leaves multi-step calls fused -- up to one wave per SIMD)
        if ray_mode == 1 and 2 * n_arenas > 4096:
            ray_mode = 2
        one_wave = os.environ.get('LL_SEPMC_ONE_WAVE', '1') == '1' and os.environ.get('LL_SHARE_SIMDS', '0') != '1'      # (llenv.hip: the one-wave-per-SIMD build at every batch size)
        kspl = spl if (spl > 1 and ray_mode < 2 and (2 * n_arenas <= 4096 or one_wave)) else 1
        split = ray_mode >= (1 if spl == 1 else 2)
        traffic, issue, tsrc = committed_counters('sepmc_step_kernel', 2 * n_arenas, kspl)
        extra = {'cpu_baseline': cpu_baseline_env('sepmc', sepmc_env_config())} if (world == 1 and not args.no_cpu_baseline) else {}
        print(json.dumps({**extra, 'build': build_record(), **{
            'metric': 'robot-steps/sec (whole node), SEPMC chase-tag env, random policy', 'value': world * 2 * n_arenas * args.steps / elapsed, 'unit': 'robot-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'SEPMC ChaseTagGameEnv (BASELINE config 5), %d arenas x 2 robots per MI355X, 2 x 778 perception rays + 21 visibility rays per '
                                   'arena-step, two-robot push schedule, robot-robot contact, random-policy actions N(0, e^-2), auto-reset' % n_arenas,
                       'arenas_per_gpu': n_arenas, 'steps_per_launch': spl, 'episodes_finished_rank0': eng.counters()['episodes']},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'traffic_source': tsrc, 'traffic_note': 'counter traffic of a multi-step launch UNDER-counts HBM bytes: the rows a wave writes in one step of the launch and overwrites in the next (observation, state, bookkeeping) meet in its L2 / the Infinity Cache and need not reach HBM, so FETCH_SIZE + WRITE_SIZE can come out below the algorithmic bytes (0.8 - 0.9 x); single launches per step measure 1.1 - 1.5 x', 'single_wave_issue': issue,
                         'kernel': 'sepmc_step_kernel' + (' + epmc_percept_kernel (the 778 rays per robot, behind every step kernel)' if split else ''),
                         'kernel_avg_ms': k_ms, 'kernel_launches_timed': k_n if kspl > 1 else k_steps, 'kernel_avg_launch_ms': k_launch_ms if kspl > 1 else k_ms,
                         'control_steps_per_launch': kspl, 'control_steps_per_call': spl,
                         'algorithmic_bytes_per_robot_step': SEPMC_ALGO_BYTES_PER_ROBOT_STEP,
                         'algorithmic_bytes_per_launch': 2 * n_arenas * SEPMC_ALGO_BYTES_PER_ROBOT_STEP * kspl,
                         'note': 'bound by single-wave instruction issue, not HBM; see DESIGN.md 8b'}}}), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
