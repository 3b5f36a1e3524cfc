"""A direct RCCL communicator for the gradient all-reduces that are captured INTO the training step's HIP graph.

torch.distributed's ProcessGroupNCCL cannot be used inside a stream capture on this stack: its watchdog thread polls the end event
of every collective it has issued (hipEventQuery), and an event last recorded in a capturing stream answers hipErrorCapturedEvent —
the process group terminates the process (tools/rccl_single_rank_probe.py, profiles/r05_rccl_single_rank.log).  RCCL itself captures
fine: ncclAllReduce on a capturing stream becomes kernel nodes of the graph.  So the captured collectives go through RCCL's C API
(the librccl.so that torch already loaded), on a communicator of our own whose unique id travels through the existing process group;
everything else (barriers, broadcasts, the eager path, the gloo CPU tests) stays on torch.distributed.

    comm = RcclComm(device)                       # collective over the default process group: every rank must call it
    comm.all_reduce_(flat, avg=True, stream=s)    # in place, on HIP stream `s` (capturable)
"""
import ctypes
import os

import torch
import torch.distributed as dist

__all__ = ["RcclComm", "available"]

_DTYPES = {torch.float32: 7, torch.float16: 6, torch.bfloat16: 9, torch.float64: 8, torch.int32: 2, torch.int64: 4, torch.uint8: 1}
_SUM, _AVG = 0, 4


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]


_lib = None


def _rccl():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)            # the copy torch.distributed uses (already mapped: same handle)
        lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = ctypes.c_int, [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.restype = ctypes.c_int
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.restype = ctypes.c_int
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = ctypes.c_int, [ctypes.c_void_p]
        lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [ctypes.c_int]
        _lib = lib
    return _lib


def available():
    try:
        _rccl()
        return True
    except OSError:
        return False


def _check(code, what):
    if code != 0:
        raise RuntimeError(f"{what}: RCCL error {code}: {_rccl().ncclGetErrorString(code).decode()}")


class RcclComm:
    def __init__(self, device):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm needs an initialised torch.distributed process group (it carries the unique id)")
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        lib = _rccl()
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # 128 bytes from rank 0 to everybody, on whatever the process group's backend moves (RCCL: device tensors; gloo: host)
        on = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(uid), 128)), dtype=torch.uint8).clone().to(on)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tobytes())
        ctypes.memmove(ctypes.addressof(uid), raw, 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, t, avg=True, stream=None):
        """In-place sum / mean of `t` over the ranks, enqueued on `stream` (default: the current stream).  Capturable."""
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
            raise ValueError("RcclComm.all_reduce_: contiguous GPU tensor of a supported dtype expected")
        st = stream if stream is not None else torch.cuda.current_stream(t.device)
        with torch.cuda.device(t.device):
            _check(_rccl().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], _AVG if avg else _SUM, self._comm,
                                         ctypes.c_void_p(st.cuda_stream)), "ncclAllReduce")
        return t

    def close(self):
        if getattr(self, "_comm", None):
            _rccl().ncclCommDestroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
