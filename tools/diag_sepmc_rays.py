"""Diagnostic (GPU): which ray-phase variant (tools/_build/ab_<v>.so) keeps the SEPMC free-running invariants?"""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sepmc_parity_common as SC  # noqa: E402
for v in sys.argv[1:]:
    lib = None if v == 'default' else os.path.join(ROOT, 'tools', '_build', 'ab_%s.so' % v)
    try:
        print(v, 'free running 201 x 200:', SC.check_free_running(lib, n_arenas=201, steps=200), flush=True)
        print(v, 'multi-step launch:', SC.check_multi_step_launch(lib, sizes=(35,), k=7, n_launches=3), flush=True)
    except Exception as e:      # noqa: BLE001
        print(v, 'FAILED:', str(e).strip().split('\n')[0:6], flush=True)
