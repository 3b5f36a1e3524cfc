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

__all__ = ["RcclComm", "CollectiveWatchdog", "available"]

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
        err = None
        try:
            import contextlib
            with (torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()):
                _check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        except RuntimeError as e:
            err, self._comm = e, None
        # all ranks have a communicator or none keeps one: a rank that went on alone would sit in its first collective for good
        flag = torch.tensor([0.0 if err is not None else 1.0], device=on)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0.0:
            self.close()
            raise RuntimeError(f"RcclComm: communicator setup failed on {'this rank: ' + str(err) if err is not None else 'another rank'}")

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
        if getattr(self, "_comm", None) is not None and self._comm:
            _rccl().ncclCommDestroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CollectiveWatchdog:
    """Nothing watches a collective that was captured into a HIP graph (the process group's watchdog never saw it): if a peer dies, the
    survivors sit in the replayed all-reduce for good.  This thread does the watching: after every replay the trainer records an event
    behind the graph (`arm`, outside any capture) and the thread polls it; an event that has not completed after `timeout_s` makes
    the rank EXIT NON-ZERO (`os._exit`, no cleanup: the GPU side is wedged) — starting a fresh process is the launcher's job
    (torchrun --max-restarts), never a re-exec of a process that has touched the GPU.  VMASR_RCCL_TIMEOUT_S (default 300)."""

    EXIT_CODE = 3

    def __init__(self, timeout_s=None, poll_s=0.05, on_timeout=None):
        import queue
        import threading
        self.timeout_s = float(os.environ.get("VMASR_RCCL_TIMEOUT_S", "300")) if timeout_s is None else float(timeout_s)
        self.poll_s = poll_s
        self.on_timeout = on_timeout or self._die
        self._q = queue.Queue()
        self._thread = threading.Thread(target=self._run, name="vmasr-rccl-watchdog", daemon=True)
        self._thread.start()

    def arm(self, event, what="captured all-reduce"):
        """`event`: anything with .query() -> bool (a torch.cuda.Event recorded behind the replayed graph)."""
        import time
        self._q.put((event, time.monotonic(), what))

    def _die(self, what, waited):
        import sys
        sys.stderr.write(f"[vm_asr_amd] {what} has not completed after {waited:.0f} s (a peer rank is gone or the fabric is wedged): "
                         f"exiting with code {self.EXIT_CODE}; restart the job from the last checkpoint\n")
        sys.stderr.flush()
        os._exit(self.EXIT_CODE)

    def _run(self):
        import time
        while True:
            event, t0, what = self._q.get()
            while True:
                try:
                    done = bool(event.query())
                except Exception:      # noqa: BLE001  (a query that throws — device lost — is a timeout too)
                    done = False
                if done:
                    break
                waited = time.monotonic() - t0
                if waited > self.timeout_s:
                    self.on_timeout(what, waited)
                    break
                time.sleep(self.poll_s)
