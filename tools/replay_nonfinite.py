"""Replay of the events tools/diag_nonfinite.py caught (gpurun_out/nonfinite/events.npz) on the float64 oracle -- without the velocity clip (max_coord_vel=1e30) and with the spec's
(LLM_MAX_COORD_VEL = 100) -- and on the host build of the kernel source; prints the mocap frames around the first event."""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, conftest
from conftest import make_oracle_batch
from oracle import oracle as orc
from lifelike_agility_and_play_amd import mocap, urdf_model
import parity_common as pc
ev = np.load('/root/repo/gpurun_out/nonfinite/events.npz')
blob = urdf_model.default_model_blob(); table = mocap.load_mocap('', 0.02)
for k in range(len(ev['clip'])):
    pre = ev['pre'][k].astype(np.float64); act = ev['act'][k].astype(np.float64)
    print('event', k, 'clip', ev['clip'][k], 'time', ev['time'][k], 'qd', np.round(pre[25:37], 1))
    for spec in ({'max_coord_vel': 1e30}, {}):
        orc.reset_spec(); orc.set_spec(**spec)
        B = make_oracle_batch(orc, blob, table, n_envs=1)
        B.reset_env(0, int(ev['clip'][k]), float(ev['time'][k])); B.set_state(0, pre)
        o, r, d = B.step_env(0, act)
        s = B.get_state(0)
        print('   oracle', spec, 'finite', np.isfinite(s).all(), 'done', d, 'max |qd| after %.3g' % np.abs(s[25:37]).max(), 'max|q| %.3g' % np.abs(s[13:25]).max(), 'z %.3g' % s[2])
    orc.reset_spec()
    # the emulated kernel from the same state
    E = pc.make_engine(blob, table, 4, '/root/repo/tests/emul/_build/libllenv_emul.so')
    E.reset(clip=[int(ev['clip'][k])] * 4, t0=[float(ev['time'][k])] * 4)
    st = E.state(); print('   emul reset state qd', np.round(st[0][25:37], 1)[:6], '... equals pre:', np.allclose(st[0], ev['pre'][k], atol=1e-3))
    E.step_host(np.tile(ev['act'][k], (4, 1)))
    r, d, why = E.reward_done(); print('   emul after step: done', d[0], 'why', why[0], 'finite', np.isfinite(E.state()[0]).all())
    E.close()
# the mocap frames around the discontinuity
c = int(ev['clip'][0]); t = float(ev['time'][0])
fs = table.frame_step; i = int(t / fs)
rows = table.frames[table.clip_off[c] + i - 2: table.clip_off[c] + i + 4]
print('frame step', fs, 'frames around: joint cols'); print(np.round(rows[:, 7:19], 3))
