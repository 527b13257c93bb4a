"""Extract the policy branch of the reference's trained EPMC checkpoints (DATA: float32 arrays 0, 1, 47..101 of 102) into
tests/golden/epmc_policy_<element>.npz for the environmental-level trained-policy sanity run (oracle/epmc_policy.py).  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_policy import _U  # noqa: E402

if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for el in ('hurdle', 'hole', 'cube'):
        try:
            m = _U(open('/root/reference/data/models/environmental_level_%s.model' % el, 'rb')).load().model
        except Exception as e:      # noqa: BLE001  (environmental_level_hole.model of this snapshot does not unpickle: 'invalid load key')
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
