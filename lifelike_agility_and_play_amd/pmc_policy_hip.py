"""ctypes binding of include/llenv_policy.h: the trained PMC policy as ONE fused MFMA kernel inside libllenv.so, evaluated
straight on an engine's device buffers.  `oracle/pmc_policy.py` (NumPy) and `pmc_policy_torch.TorchPmcPolicy` (library GEMMs)
state the same forward pass."""
import ctypes as C
import os

import numpy as np

from . import capi

DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'pmc_policy.npz')
LLP_N_FLOATS = 358647

_SIGS = {
    'll_policy_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'll_policy_destroy': (C.c_int, [C.c_void_p]),
    'll_policy_act': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'll_policy_act_pg': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]),
    'll_policy_enable_timing': (C.c_int, [C.c_void_p, C.c_int]),
    'll_policy_time_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)
_bound = {}


def load_library(path=None):
    lib = capi.load_library(path)
    if id(lib) not in _bound:
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound[id(lib)] = True
    return lib


def pack_weights(npz_path=DEFAULT_WEIGHTS):
    z = np.load(npz_path)
    flat = np.concatenate([z['w%02d' % i].astype(np.float32).ravel() for i in range(28)])
    assert flat.size == LLP_N_FLOATS
    return np.ascontiguousarray(flat)


class HipPmcPolicy(object):
    def __init__(self, npz_path=DEFAULT_WEIGHTS, device=0, lib_path=None):
        self.lib = load_library(lib_path)
        w = pack_weights(npz_path)
        self._pid = os.getpid()
        self.h = C.c_void_p()
        self._chk(self.lib.ll_policy_create(w.ctypes.data_as(C.c_void_p), int(w.size), int(device), C.byref(self.h)))

    def _chk(self, rc):
        if rc != 0:
            raise capi.LLError(rc, self.lib.ll_last_error().decode())

    def act_ptr(self, d_obs, d_actions, n_envs, stream=None, d_code=None):
        self._chk(self.lib.ll_policy_act(self.h, C.c_void_p(int(d_obs)), C.c_void_p(int(d_actions)), C.c_void_p(int(d_code)) if d_code else None,
                                         int(n_envs), C.c_void_p(int(stream)) if stream else None))

    def act(self, engine, d_code=None):
        """obs buffer of `engine` -> its action buffer, queued on the engine's stream (then engine.step() applies it)."""
        p = engine.device_ptrs()
        self.act_ptr(p.obs, p.actions, p.n_envs, p.stream, d_code)

    def act_pg(self, engine, seed, step, sample=True, d_code=None):
        """The actor's forward pass (ll_policy_act_pg): a ~ pi(.|obs) into the engine's action buffer, -log p(a|obs) and V(obs) into its
        ll_pg_ptrs buffers -- from where the next engine.step() copies them into the unroll it records."""
        p = engine.device_ptrs()
        nl, v = engine.pg_ptrs()
        self._chk(self.lib.ll_policy_act_pg(self.h, C.c_void_p(int(p.obs)), C.c_void_p(int(p.actions)), C.c_void_p(int(d_code)) if d_code else None,
                                            C.c_void_p(int(nl)), C.c_void_p(int(v)), int(p.n_envs), int(seed), int(step), 1 if sample else 0,
                                            C.c_void_p(int(p.stream)) if p.stream else None))
        engine.pg_mark_current()               # the value buffer now belongs to the current observation (ll_finish_unroll checks the stamp)

    def enable_timing(self, on=True):
        self._chk(self.lib.ll_policy_enable_timing(self.h, 1 if on else 0))

    def time_ms(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.lib.ll_policy_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            if getattr(self, '_pid', None) == os.getpid():      # (a fork()ed child inherits the object, not the HIP context: it must not destroy it)
                self.lib.ll_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:     # noqa: BLE001
            pass
