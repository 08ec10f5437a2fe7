"""ctypes binding of viennaray_amd/libviennaray_amd_rccl.so (include/viennaray_amd_rccl.h): the RCCL
(xGMI) all-reduce callback for Trace.applySharded / vr_apply_sharded."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libviennaray_amd_rccl.so")
UNIQUE_ID_BYTES = 128
SYMBOLS = ("vr_rccl_unique_id", "vr_rccl_init_rank", "vr_rccl_destroy", "vr_rccl_allreduce", "vr_rccl_last_error")
_lib = None


def load():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.vr_rccl_unique_id.argtypes = [C.c_char_p]
        L.vr_rccl_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int]
        L.vr_rccl_destroy.argtypes = [C.c_void_p]
        L.vr_rccl_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class Communicator:
    """One rank of an RCCL communicator.  `unique_id` (bytes) comes from rank 0's `unique_id()` and is
    handed to the other ranks by whatever the application already has (MPI, a file, a socket)."""

    def __init__(self, rank, world, unique_id=None):
        L = load()
        if unique_id is None:
            unique_id = Communicator.unique_id()
        self._h = C.c_void_p()
        if L.vr_rccl_init_rank(C.byref(self._h), unique_id, rank, world) != 0:
            raise RuntimeError("vr_rccl_init_rank: " + L.vr_rccl_last_error().decode())
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        if load().vr_rccl_unique_id(buf) != 0:
            raise RuntimeError("vr_rccl_unique_id: " + load().vr_rccl_last_error().decode())
        return buf.raw

    @property
    def allreduce(self):
        return load().vr_rccl_allreduce

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            load().vr_rccl_destroy(self._h)
            self._h = C.c_void_p()
