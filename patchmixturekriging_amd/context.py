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
        """launch on a caller-owned stream; 0 / None = the context's own (non-blocking) stream"""
        _lib.check(self.L.pmk_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)), "pmk_ctx_set_stream")

    def set_stream_null(self):
        """launch on the device's legacy default stream (handle 0: what a framework's default stream is)"""
        _lib.check(self.L.pmk_ctx_set_stream_null(self.h), "pmk_ctx_set_stream_null")

    def synchronize(self):
        _lib.check(self.L.pmk_ctx_synchronize(self.h), "pmk_ctx_synchronize")

    def enable_timers(self, on=True):
        _lib.check(self.L.pmk_ctx_enable_timers(self.h, 1 if on else 0))

    def timer_ms(self, stage):
        ms = C.c_double()
        _lib.check(self.L.pmk_ctx_timer_ms(self.h, stage.encode(), C.byref(ms)), "pmk_ctx_timer_ms")
        return ms.value

    def set_pipeline(self, on):
        """pipelined kernel-matrix build of fit (default off: measured slower on MI355X)"""
        _lib.check(self.L.pmk_ctx_set_pipeline(self.h, 1 if on else 0), "pmk_ctx_set_pipeline")

    def shader_clock(self, which):
        """GHz seen by workgroup 0 of the last fit's step launches (0) / of the last strip kernel (1); 0.0 if none ran"""
        g = C.c_double()
        _lib.check(self.L.pmk_ctx_shader_clock(self.h, int(which), C.byref(g)), "pmk_ctx_shader_clock")
        return g.value

    def close(self):
        if self.h:
            self.L.pmk_ctx_destroy(self.h)
            self.h = None


COMM_ID_BYTES = 128


def comm_unique_id():
    """pmk_comm_unique_id: 128 opaque bytes made by ONE rank, to be shipped to the others by any host channel"""
    L = _lib.lib()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _lib.check(L.pmk_comm_unique_id(buf), "pmk_comm_unique_id")
    return buf.raw


class Comm:
    """pmk_comm: the library's own RCCL communicator (one rank per GPU); creation is collective"""

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("the communicator id is %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _lib.check(ctx.L.pmk_comm_create(ctx.h, self.rank, self.world, buf, C.byref(h)), "pmk_comm_create")
        self.h = h

    def last_bytes(self):
        """payload bytes this rank (sent, received) in the exchange of its last predict step"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self.ctx.L.pmk_comm_last_bytes(self.h, C.byref(a), C.byref(b)), "pmk_comm_last_bytes")
        return a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.pmk_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def shard_segments(region_offsets, world):
    """pmk_shard_segments: (first, count) of every rank's segment of a region-sorted item list"""
    import numpy as np
    L = _lib.lib()
    off = np.ascontiguousarray(region_offsets, dtype=np.int64)
    first, count = np.empty(world, dtype=np.int64), np.empty(world, dtype=np.int64)
    ip = C.POINTER(C.c_int64)
    _lib.check(L.pmk_shard_segments(off.ctypes.data_as(ip), len(off) - 1, int(world), first.ctypes.data_as(ip),
                                    count.ctypes.data_as(ip)), "pmk_shard_segments")
    return list(zip(first.tolist(), count.tolist()))


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
