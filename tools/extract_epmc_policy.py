"""Extract the policy branch of the reference's trained EPMC checkpoints (DATA: float32 arrays 0, 1, 47..101 of 102) into
tests/golden/epmc_policy_<element>.npz for the environmental-level trained-policy sanity run (oracle/epmc_policy.py).  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_policy import _U, _S, checked_find_class  # noqa: E402,F401


def load_checkpoint(path):
    """The reference's checkpoints come in two container formats: a plain pickle (hurdle, cube, primitive_level, strategic_level) and
    joblib.dump's (hole: NumpyArrayWrapper records with the raw array bytes inline in the stream -- pickle.Unpickler stops at the first one
    with 'invalid load key').  joblib's own unpickler reads the second; the tleague classes are stubbed in both."""
    try:
        return _U(open(path, 'rb')).load().model
    except Exception:       # noqa: BLE001
        from joblib.numpy_pickle import NumpyUnpickler

        class _J(NumpyUnpickler):
            def find_class(self, module, name):
                return checked_find_class(super().find_class, module, name)
        with open(path, 'rb') as fh:
            return _J(path, fh, ensure_native_byte_order=True).load().model

if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for el in ('hurdle', 'hole', 'cube'):
        try:
            m = load_checkpoint('/root/reference/data/models/environmental_level_%s.model' % el)
        except Exception as e:      # noqa: BLE001
            print('cannot read the %s checkpoint: %r' % (el, e))
            continue
        assert len(m) == 102
        keep = [0, 1] + list(range(47, 102))
        dst = os.path.join(root, 'tests', 'golden', 'epmc_policy_%s.npz' % el)
        np.savez_compressed(dst, **{'w%d' % i: np.asarray(m[i], dtype=np.float32) for i in keep})
        print('wrote', dst, os.path.getsize(dst))
    m = _U(open('/root/reference/data/models/strategic_level.model', 'rb')).load().model
    assert len(m) == 152
    dst = os.path.join(root, 'tests', 'golden', 'sepmc_policy.npz')
    np.savez_compressed(dst, **{'w%d' % i: np.asarray(m[i], dtype=np.float32) for i in [0, 1] + list(range(51, 152))})
    print('wrote', dst, os.path.getsize(dst))
