"""Process-wide device context (pmk_ctx) of the package."""
import ctypes as C

from . import _lib

_CTX = None
_DEVICE = 0


class Context:
    def __init__(self, device=0):
        self.L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.L.pmk_ctx_create(device, C.byref(h)), "pmk_ctx_create")
        self.h = h
        self.device = device

    def set_stream(self, stream_ptr):
        _lib.check(self.L.pmk_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)), "pmk_ctx_set_stream")

    def synchronize(self):
        _lib.check(self.L.pmk_ctx_synchronize(self.h), "pmk_ctx_synchronize")

    def enable_timers(self, on=True):
        _lib.check(self.L.pmk_ctx_enable_timers(self.h, 1 if on else 0))

    def timer_ms(self, stage):
        ms = C.c_double()
        _lib.check(self.L.pmk_ctx_timer_ms(self.h, stage.encode(), C.byref(ms)), "pmk_ctx_timer_ms")
        return ms.value

    def close(self):
        if self.h:
            self.L.pmk_ctx_destroy(self.h)
            self.h = None


def set_device(device):
    """choose the GPU of the default context (before first use)"""
    global _DEVICE, _CTX
    if _CTX is not None and _CTX.device != device:
        _CTX.close()
        _CTX = None
    _DEVICE = device


def default_context():
    global _CTX
    if _CTX is None:
        _CTX = Context(_DEVICE)
    return _CTX
