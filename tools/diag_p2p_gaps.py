"""Where do the launch gaps of the p2p hand-off come from (profiles/r05_p2p_no_cu.txt: + 4 % wall that neither the device-side wait nor the copy explains)?
One rank, 4096 envs, bench.py --gather-mode p2p with one element of the hand-off left out at a time (gather.py LL_P2P_DIAG):
    nothing   no hand-off call at all (only the unroll recording and TD(lambda): = --gather-mode none)
    norecord  barrier + helper threads, but no interprocess event is recorded and nothing is pulled
    nopull    events recorded and waited for, no copy
    nowait    everything except the engine stream's wait for "block copied"
    (empty)   the hand-off as shipped
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for diag in ('nothing', 'norecord', 'nopull', 'nowait', ''):
        env = dict(os.environ, LL_BENCH_FORCE_GATHER='1', LL_BENCH_BACKEND='gloo', LL_P2P_DIAG=diag, HSA_ENABLE_IPC_MODE_LEGACY='0')
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1024', '--warmup', '256', '--no-cpu-baseline', '--gather-mode', 'p2p'], env=env, cwd=ROOT, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(diag or 'as shipped', 'FAILED', out.stderr[-300:]); continue
        j = json.loads(line[0])
        print('%-10s wall %.4f ms per step, kernel %.4f, gaps %.4f' % (diag or 'as shipped', j['ms_per_step'], j['roofline']['kernel_avg_ms'], j['ms_per_step'] - j['roofline']['kernel_avg_ms']), flush=True)
