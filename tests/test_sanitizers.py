"""the CPU restatement (oracle/pmc_oracle.c) and the host build of the kernel body (tests/emul) run under AddressSanitizer +
UndefinedBehaviorSanitizer: reset, free-running steps with falls and re-seeds, contact-rich and self-colliding states, the terrain and
two-robot substeps, the unroll buffers.  Each runs in a subprocess with the sanitizer runtime preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.check_output(['gcc', '-print-file-name=libasan.so']).decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run(code, extra_env):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip('no libasan.so next to gcc')
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1',
               PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'), OMP_NUM_THREADS='2', **extra_env)
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    bad = [l for l in out.stderr.splitlines() if 'AddressSanitizer' in l or 'runtime error' in l or 'UndefinedBehaviorSanitizer' in l]
    assert out.returncode == 0 and not bad, (out.returncode, out.stderr[-3000:])
    assert 'sanitized run ok' in out.stdout, out.stdout[-1000:]


ORACLE_CODE = r'''
import ctypes as C, numpy as np, os
from oracle import oracle as orc
orc.LIB = os.path.join(os.path.dirname(orc.LIB), 'libpmc_oracle_asan.so')
orc.build = lambda force=False: orc.LIB                      # the sanitized build, made by the test
from conftest import PMC_PROP_TYPE, PMC_REWARD_WEIGHTS
from lifelike_agility_and_play_amd import mocap, urdf_model
blob, table = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02)
B = orc.OracleBatch(orc.make_config(n_envs=6, reward_weights=PMC_REWARD_WEIGHTS, prop_type=PMC_PROP_TYPE, prioritized_sample_factor=3.0, set_obstacle=True,
                                    obstacle_height=0.2), blob, table)
rng = np.random.default_rng(0)
for i in range(6):
    B.reset_env(i, int(rng.integers(0, B.n_clips)), 0.3)
for spec in ({}, dict(self_friction=0.25, warm_start=0.85), dict(max_contacts_per_leg=2, limit_gate=1e30, max_depen_speed=1e30)):
    orc.reset_spec(); orc.set_spec(**spec)
    for t in range(12):
        obs, r, d = B.step_all_mt(rng.normal(size=(6, 12)) * 0.6, 2)
        for i in np.where(d)[0]:
            B.reset_env(int(i), int(rng.integers(0, B.n_clips)), 0.1)
orc.reset_spec()
s = B.get_state(0); s[2] = 0.05; s[13:25] = rng.uniform(-1, 1, 12)        # pressed into the ground, legs folded: contacts, limits, self-collision
B.set_state(0, s)
for t in range(5):
    B.step_env(0, np.zeros(12))
print('sanitized run ok')
'''

EMUL_CODE = r'''
import numpy as np, os
import parity_common as pc
from lifelike_agility_and_play_amd import mocap, urdf_model
lib = os.environ['LL_ASAN_EMUL']
blob, table = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02)
E = pc.make_engine(blob, table, 5, lib, auto_reset=1, seed=2, keep_terminal_obs=True)
E.reset()
ptr, w = E.enable_unrolls(4, 2)
rng = np.random.default_rng(1)
for t in range(14):
    E.step_host((rng.normal(size=(5, 12)) * 0.7).astype(np.float32))
E.finish_unroll(0, 0.95, 0.95)
E.step_random(0.3)
s = E.state(); s[1, 2] = 0.05; s[1, 13:25] = rng.uniform(-1, 1, 12).astype(np.float32); s[2, 0] = np.nan
E.set_state(s)
E.step_host(np.zeros((5, 12), np.float32))
assert E.counters()['nonfinite'] == 1
E.probe_pd_torque(rng.normal(size=(7, 12)), rng.normal(size=(7, 12)), rng.normal(size=(7, 12)))
E.reset(env_ids=[0, 3])
E.close()
O = pc.make_engine(blob, table, 3, lib, set_obstacle=True, obstacle_height=0.2, auto_reset=1, seed=3)
O.reset()
for t in range(6):
    O.step_random(0.4)
O.close()
print('sanitized run ok')
'''


def test_oracle_under_asan_ubsan():
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    _run(ORACLE_CODE, {})


def test_kernel_body_host_build_under_asan_ubsan():
    emul = os.path.join(ROOT, 'tests', 'emul')
    subprocess.check_call(['make', '-C', emul, '-s', 'asan'])
    _run(EMUL_CODE, {'LL_ASAN_EMUL': os.path.join(emul, '_build', 'libllenv_emul_asan.so')})
