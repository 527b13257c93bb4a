"""ctypes binding of include/llenv_xfer.h: HIP IPC handles and CU-free (SDMA) device-to-device pulls -- the transport of
gather.TrajectoryBuffer(mode='p2p'), in which the learner rank pulls the other ranks' finished unroll blocks instead of every rank
pushing them through an RCCL collective (distill_actor.py:159-167 is what both replace)."""
import ctypes as C

from . import capi

HANDLE_BYTES = 64


class Handle(C.Structure):
    _fields_ = [('bytes', C.c_ubyte * HANDLE_BYTES)]


_SIGS = {
    'll_xfer_export_mem': (C.c_int, [C.c_int, C.c_void_p, C.POINTER(Handle), C.POINTER(C.c_uint64)]),
    'll_xfer_open_mem': (C.c_int, [C.c_int, C.POINTER(Handle), C.c_uint64, C.POINTER(C.c_void_p)]),
    'll_xfer_close_mem': (C.c_int, [C.c_int, C.c_void_p]),
    'll_xfer_event_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(Handle)]),
    'll_xfer_event_open': (C.c_int, [C.c_int, C.POINTER(Handle), C.POINTER(C.c_void_p)]),
    'll_xfer_event_destroy': (C.c_int, [C.c_void_p]),
    'll_xfer_event_record': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_xfer_event_synchronize': (C.c_int, [C.c_void_p]),
    'll_xfer_stream_wait': (C.c_int, [C.c_void_p, C.c_void_p]),
    'll_xfer_stream_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'll_xfer_stream_destroy': (C.c_int, [C.c_void_p]),
    'll_xfer_stream_synchronize': (C.c_int, [C.c_void_p]),
    'll_xfer_pull': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    'll_xfer_can_wait_value': (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    'll_xfer_signal_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'll_xfer_signal_destroy': (C.c_int, [C.c_int, C.c_void_p]),
    'll_xfer_stream_wait_value': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    'll_xfer_stream_write_value': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    'll_xfer_set_device': (C.c_int, [C.c_int]),
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


def _chk(lib, rc):
    if rc != 0:
        raise capi.LLError(rc, lib.ll_last_error().decode())


def export_mem(device, ptr, lib=None):
    """(handle bytes, offset) of the allocation holding device pointer `ptr` -- picklable, to be sent to the pulling process"""
    lib = lib or load_library()
    h, off = Handle(), C.c_uint64()
    _chk(lib, lib.ll_xfer_export_mem(int(device), C.c_void_p(int(ptr)), C.byref(h), C.byref(off)))
    return bytes(h.bytes), int(off.value)


def open_mem(device, handle_bytes, offset, lib=None):
    """device pointer (int) in THIS process of another process's exported allocation + offset; keep (ptr - offset) for close_mem"""
    lib = lib or load_library()
    h = Handle(); C.memmove(h.bytes, handle_bytes, HANDLE_BYTES)
    p = C.c_void_p()
    _chk(lib, lib.ll_xfer_open_mem(int(device), C.byref(h), C.c_uint64(offset), C.byref(p)))
    return int(p.value)


def close_mem(device, base_ptr, lib=None):
    lib = lib or load_library()
    _chk(lib, lib.ll_xfer_close_mem(int(device), C.c_void_p(int(base_ptr))))


class IpcEvent(object):
    """An interprocess HIP event: created (owner) or opened from the owner's handle."""

    def __init__(self, device, handle_bytes=None, lib=None):
        self.lib = lib or load_library()
        ev = C.c_void_p()
        if handle_bytes is None:
            h = Handle()
            _chk(self.lib, self.lib.ll_xfer_event_create(int(device), C.byref(ev), C.byref(h)))
            self.handle = bytes(h.bytes)
        else:
            h = Handle(); C.memmove(h.bytes, handle_bytes, HANDLE_BYTES)
            _chk(self.lib, self.lib.ll_xfer_event_open(int(device), C.byref(h), C.byref(ev)))
            self.handle = bytes(handle_bytes)
        self.ev = ev

    def record(self, stream_handle):
        _chk(self.lib, self.lib.ll_xfer_event_record(self.ev, C.c_void_p(int(stream_handle or 0))))

    def synchronize(self):
        _chk(self.lib, self.lib.ll_xfer_event_synchronize(self.ev))

    def make_stream_wait(self, stream_handle):
        _chk(self.lib, self.lib.ll_xfer_stream_wait(C.c_void_p(int(stream_handle or 0)), self.ev))

    def close(self):
        if self.ev:
            self.lib.ll_xfer_event_destroy(self.ev)
            self.ev = None


class CopyStream(object):
    def __init__(self, device, lib=None):
        self.lib = lib or load_library()
        s = C.c_void_p()
        _chk(self.lib, self.lib.ll_xfer_stream_create(int(device), C.byref(s)))
        self.handle = int(s.value)

    def pull(self, dst_ptr, src_ptr, nbytes, no_cu=True):
        _chk(self.lib, self.lib.ll_xfer_pull(C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), C.c_size_t(int(nbytes)), 1 if no_cu else 0, C.c_void_p(self.handle)))

    def synchronize(self):
        _chk(self.lib, self.lib.ll_xfer_stream_synchronize(C.c_void_p(self.handle)))

    def close(self):
        if self.handle:
            self.lib.ll_xfer_stream_destroy(C.c_void_p(self.handle))
            self.handle = 0


def set_device(device, lib=None):
    """hipSetDevice for the calling THREAD (a new thread starts on device 0)"""
    lib = lib or load_library()
    _chk(lib, lib.ll_xfer_set_device(int(device)))


def can_wait_value(device, lib=None):
    lib = lib or load_library()
    v = C.c_int()
    _chk(lib, lib.ll_xfer_can_wait_value(int(device), C.byref(v)))
    return bool(v.value)


class SignalWord(object):
    """One word of signal memory: a stream can be made to wait ON THE DEVICE until it has reached a value (hipStreamWaitValue32) and another stream
    raises it in stream order (hipStreamWriteValue32) -- the launching thread blocks on neither."""

    def __init__(self, device, lib=None):
        self.lib = lib or load_library()
        self.device = int(device)
        p = C.c_void_p()
        _chk(self.lib, self.lib.ll_xfer_signal_create(self.device, C.byref(p)))
        self.ptr = int(p.value)

    def make_stream_wait(self, stream_handle, value):
        _chk(self.lib, self.lib.ll_xfer_stream_wait_value(C.c_void_p(int(stream_handle or 0)), C.c_void_p(self.ptr), C.c_uint32(int(value))))

    def write(self, stream_handle, value):
        _chk(self.lib, self.lib.ll_xfer_stream_write_value(C.c_void_p(int(stream_handle or 0)), C.c_void_p(self.ptr), C.c_uint32(int(value))))

    def close(self):
        if self.ptr:
            self.lib.ll_xfer_signal_destroy(self.device, C.c_void_p(self.ptr))
            self.ptr = 0
