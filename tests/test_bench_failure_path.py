"""bench.py --gpus N on a box nobody can rehearse on (round-5 review #6): whatever goes wrong in a rank, ONE JSON line
{"error", "rank", "phase", "rccl_version_line", ...} reaches stdout and the exit code is non-zero.  CPU legs: a launch whose ranks fail in
`init` (no HIP device here), a rank the launcher terminates while it sits in a blocking call, a timed region that makes no progress.  The GPU
leg (a rank killed after warm-up in an 8-rank launch) is tests/test_gpu_env_api.py::test_bench_eight_ranks_one_rank_killed."""
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith('{')]


def test_failed_launch_prints_one_error_line():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    env['HIP_VISIBLE_DEVICES'] = ''                      # (also on a GPU box: the ranks must fail in init)
    env['CUDA_VISIBLE_DEVICES'] = ''
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2'], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout[-2000:]
    j = lines[0]
    assert j['phase'] == 'init' and j['rank'] in (0, 1) and j['world'] == 2 and j['value'] is None and 'error' in j and 'rccl_version_line' in j


HARNESS = r'''
import os, sys, time
sys.path.insert(0, %r)
import bench
bench.arm_failure_path(3, 1)
bench.set_phase('timed')
print('armed', flush=True)
time.sleep(60)          # a blocking call the main thread does not come back from (stands in for a collective with a dead peer)
'''


def test_terminated_rank_still_reports():
    p = subprocess.Popen([sys.executable, '-c', HARNESS % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    assert p.stdout.readline().strip() == 'armed'
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=30)
    lines = _json_lines(out)
    assert p.returncode == 143 and len(lines) == 1
    assert lines[0]['phase'] == 'timed' and lines[0]['rank'] == 3 and 'terminated by the launcher' in lines[0]['error']


def test_stalled_rank_reports_and_exits():
    p = subprocess.run([sys.executable, '-c', HARNESS % ROOT], env=dict(os.environ, LL_BENCH_STALL_S='2'), capture_output=True, text=True, cwd=ROOT, timeout=60)
    lines = _json_lines(p.stdout)
    assert p.returncode == 3 and len(lines) == 1
    assert lines[0]['phase'] == 'timed' and 'no progress' in lines[0]['error']
