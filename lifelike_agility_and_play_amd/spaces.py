"""gym.spaces when gym is installed, otherwise structural stand-ins with the attributes TLeague reads
(.shape / .spaces / .dtype).  The reference builds Box(0, 0, shape=...) placeholders (PLE:117-124)."""
from collections import OrderedDict

import numpy as np

try:                                       # pragma: no cover - gym is absent in the build container
    from gym.spaces import Box, Dict, Discrete, Tuple  # noqa: F401
    HAVE_GYM = True
except Exception:                          # noqa: BLE001
    HAVE_GYM = False

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def __repr__(self):
            return 'Box(%r)' % (self.shape,)

    class Discrete(object):
        def __init__(self, n):
            self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

        def __repr__(self):
            return 'Discrete(%d)' % self.n

    class Dict(object):
        def __init__(self, spaces):
            self.spaces = OrderedDict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def __repr__(self):
            return 'Dict(%r)' % (list(self.spaces.items()),)

    class Tuple(object):
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

        def __getitem__(self, i):
            return self.spaces[i]

        def __len__(self):
            return len(self.spaces)

        def __repr__(self):
            return 'Tuple(%r)' % (self.spaces,)
