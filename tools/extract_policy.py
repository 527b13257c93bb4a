"""Extract the trained PMC policy weights (DATA: 28 float32 arrays) from the reference's pickle into
lifelike_agility_and_play_amd/assets/pmc_policy.npz.  Build container only.  Used for the trained-policy sanity run:
the policy was trained against PyBullet, so if it tracks mocap clips in OUR simulator, our physics is Bullet-like."""
import pickle
import sys

import numpy as np


class _Stub(type):
    def __getattr__(cls, name):
        return _Stub(name, (object,), {})


class _S(object, metaclass=_Stub):
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {'state': st})


class _U(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('tleague'):
            return _S
        if module.startswith('numpy.core'):
            module = module.replace('numpy.core', 'numpy._core')
        return super().find_class(module, name)


if __name__ == '__main__':
    src = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/data/models/primitive_level.model'
    dst = sys.argv[2] if len(sys.argv) > 2 else 'lifelike_agility_and_play_amd/assets/pmc_policy.npz'
    m = _U(open(src, 'rb')).load().model
    assert len(m) == 28
    np.savez_compressed(dst, **{'w%02d' % i: np.asarray(a, dtype=np.float32) for i, a in enumerate(m)})
    print('wrote', dst)
