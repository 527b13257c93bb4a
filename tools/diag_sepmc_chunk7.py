"""Does the full-observation net (tests/sepmc_parity_common.check_engine_against_emulation) catch the failure it was built for?  Runs it on a build of the
HIP library with SEVEN rays per chunk in the one-wave-per-SIMD chase-tag kernels (hipcc ... -DLL_SEPMC_RAY_CHUNK=7 -o tools/_build/libllenv_chunk7.so) -- the
build that wrote garbage into flag_info of re-seeding arenas in round 4 (profiles/r04_sepmc_chunk7_diag.txt) -- and on the shipped library.

    gpurun -- 'python tools/diag_sepmc_chunk7.py > gpurun_out/chunk7.txt 2>&1'
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sepmc_parity_common as SC  # noqa: E402

emul_dir = os.path.join(ROOT, 'tests', 'emul')
subprocess.check_call(['make', '-C', emul_dir, '-s'])
emul = os.environ.get('LL_DIAG_EMUL') or os.path.join(emul_dir, '_build', 'libllenv_emul.so')
def fresh(lib):
    """A library older than any kernel source is refused: this round's first diagnosis compared a stale seven-ray build with a fresh host build (HISTORY.md)."""
    if lib is None or os.environ.get('LL_DIAG_LIB7'):
        return True
    src = os.path.join(ROOT, 'lifelike_agility_and_play_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith(('.hip', '.hpp', '.inc')))
    return os.path.exists(lib) and os.path.getmtime(lib) >= newest


for label, lib in (('shipped (three rays per chunk)', os.environ.get('LL_DIAG_LIB3') or None), ('seven rays per chunk', os.environ.get('LL_DIAG_LIB7') or os.path.join(ROOT, 'tools', '_build', 'libllenv_chunk7.so'))):
    if not fresh(lib):
        print(label, 'SKIPPED:', lib, 'is missing or older than csrc/ -- rebuild it (hipcc ... -DLL_SEPMC_RAY_CHUNK=7 -o tools/_build/libllenv_chunk7.so)'); continue
    for spec in ({}, {'friction_mode': 0}):
        try:
            print(label, spec or 'cone friction', 'PASS', SC.check_engine_against_emulation(emul, n_arenas=2048, steps=2, spec=spec, gpu_lib=lib, report_only=True), flush=True)
        except AssertionError as e:
            print(label, spec or 'cone friction', 'FAIL', str(e)[:1500], flush=True)
