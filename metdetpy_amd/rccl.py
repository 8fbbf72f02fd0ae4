"""librccl through ctypes: the collectives of the row-band path, enqueued on the detector library's OWN streams
(`Context.stream_handle`), so that a pass is queued without a host round trip -- no torch tensor, no `torch.cuda` stream, no
synchronisation between the band kernels, the exchange and the Hough stage (include/metdet_hip.h, "WITHOUT host round trips").

Only what the band path needs: a communicator (the 128-byte unique id travels through whatever control plane the launcher has:
`torch.distributed.broadcast_object_list` under torchrun), an in-place all-reduce of int64 sums, and an equal-split all-to-all of
u64 bit-plane pieces as one group of ncclSend / ncclRecv.  xGMI is point to point: the all-to-all is exactly one send and one
receive per peer, `piece` words each (8 GPUs, 8K, 256 frames per pass: 32 frames x 540 rows x 120 words x 8 B = 16.6 MB per peer).

RCCL refuses two ranks on one device, so on a one-GPU box this module can only be exercised with a communicator of ONE rank (the
all-reduce is then the identity and the all-to-all a send to self): tests/test_bands.py does that under `-m gpu`; the 2-process
tests keep the gloo / host-buffer path.
"""
from __future__ import annotations

import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128
ncclSum, ncclInt64, ncclUint64 = 0, 4, 5


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


class Config(C.Structure):
    """ncclConfig_t (nccl.h, 2.27): size / magic / version header, then the fields; NCCL_CONFIG_INITIALIZER's values."""
    _fields_ = [("size", C.c_size_t), ("magic", C.c_uint), ("version", C.c_uint), ("blocking", C.c_int), ("cgaClusterSize", C.c_int),
                ("minCTAs", C.c_int), ("maxCTAs", C.c_int), ("netName", C.c_char_p), ("splitShare", C.c_int), ("trafficClass", C.c_int),
                ("commName", C.c_char_p), ("collnetEnable", C.c_int), ("CTAPolicy", C.c_int), ("shrinkShare", C.c_int),
                ("nvlsCTAs", C.c_int), ("pad", C.c_int * 8)]


class RcclError(RuntimeError):
    pass


class RcclTimeout(RcclError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MDH_RCCL_LIB", "librccl.so")
    try:
        L = C.CDLL(path)
    except OSError:
        L = C.CDLL(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librccl.so"))
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(vp), i32, UniqueId, i32]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclAllReduce.argtypes = [vp, vp, sz, i32, i32, vp, vp]
    L.ncclSend.argtypes = [vp, sz, i32, i32, vp, vp]
    L.ncclRecv.argtypes = [vp, sz, i32, i32, vp, vp]
    L.ncclGetErrorString.argtypes = [i32]
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclCommCount.argtypes = [vp, C.POINTER(i32)]
    L.ncclCommUserRank.argtypes = [vp, C.POINTER(i32)]
    L.ncclCommCuDevice.argtypes = [vp, C.POINTER(i32)]
    L.ncclCommGetAsyncError.argtypes = [vp, C.POINTER(i32)]
    L.ncclCommAbort.argtypes = [vp]
    L.ncclCommInitRankConfig.argtypes = [C.POINTER(vp), i32, UniqueId, i32, C.POINTER(Config)]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {load().ncclGetErrorString(rc).decode()}")


def new_unique_id() -> bytes:
    uid = UniqueId()
    _check(load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES))


class Communicator:
    """One rank of an RCCL communicator on HIP device `device` (selected here, before ncclCommInitRank binds to "the current
    device": the caller may sit on another thread or another torch.cuda device than the detector context).

    `timeout_s`: ncclCommInitRank blocks until every rank has arrived; with a timeout the communicator is created in RCCL's
    non-blocking mode and polled, and a rank whose peers never turn up aborts with RcclTimeout instead of hanging.  After
    creation the communicator's own size / rank / device are checked against what the launcher said."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int | None = None, timeout_s: float | None = None):
        if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of ncclGetUniqueId")
        self.L = load()
        self.rank, self.world, self.device = int(rank), int(world), device
        if device is not None:
            from . import _lib
            _lib.set_device(int(device))
        uid = UniqueId()
        C.memmove(C.addressof(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        if timeout_s is None:
            _check(self.L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        else:
            import time
            INT_MIN = -2147483648
            cfg = Config(C.sizeof(Config) - 8 * C.sizeof(C.c_int), 0xcafebeef, 22707, 0, INT_MIN, INT_MIN, INT_MIN, None, INT_MIN, INT_MIN, None,
                         INT_MIN, INT_MIN, INT_MIN, INT_MIN)
            rc = self.L.ncclCommInitRankConfig(C.byref(self.comm), self.world, uid, self.rank, C.byref(cfg))
            if rc not in (0, 7):  # ncclSuccess, ncclInProgress
                _check(rc, "ncclCommInitRankConfig")
            t0, state = time.monotonic(), C.c_int(7)
            while True:
                _check(self.L.ncclCommGetAsyncError(self.comm, C.byref(state)), "ncclCommGetAsyncError")
                if state.value == 0:
                    break
                if state.value != 7:
                    self.abort()
                    _check(state.value, "ncclCommInitRankConfig (async)")
                if time.monotonic() - t0 > timeout_s:
                    self.abort()
                    raise RcclTimeout(f"rank {rank}: the communicator of {world} ranks did not form within {timeout_s} s (a peer is missing)")
                time.sleep(0.01)
        n, r = C.c_int(-1), C.c_int(-1)
        _check(self.L.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        _check(self.L.ncclCommUserRank(self.comm, C.byref(r)), "ncclCommUserRank")
        if n.value != self.world or r.value != self.rank:
            self.abort()
            raise RcclError(f"the communicator holds {n.value} ranks (this one is {r.value}); the launcher said {self.world} / {self.rank}")

    def describe(self) -> dict:
        """What the communicator itself says it is (tools/run_8gpu.sh prints it per rank)."""
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        self.L.ncclCommCount(self.comm, C.byref(n))
        self.L.ncclCommUserRank(self.comm, C.byref(r))
        self.L.ncclCommCuDevice(self.comm, C.byref(d))
        return {"ranks": n.value, "rank": r.value, "hip_device": d.value}

    def check_async(self):
        """Raise if RCCL has recorded an asynchronous error (a peer that died mid-collective)."""
        state = C.c_int(0)
        _check(self.L.ncclCommGetAsyncError(self.comm, C.byref(state)), "ncclCommGetAsyncError")
        if state.value not in (0, 7):
            _check(state.value, "asynchronous RCCL error")

    def abort(self):
        """Tear the communicator down without waiting for its peers (after a timeout: the collectives queued on the library's
        streams are released, so a wait that timed out does not turn into a hang at destroy time)."""
        if self.comm:
            self.L.ncclCommAbort(self.comm)
            self.comm = C.c_void_p()

    @classmethod
    def from_torch_dist(cls, dist, rank: int, world: int, device: int | None = None, timeout_s: float | None = None):
        """The id is made on rank 0 and broadcast through the process group the launcher set up (control plane only)."""
        box = [new_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0], device=device, timeout_s=timeout_s)

    def all_reduce_sum_i64(self, ptr: int, count: int, stream: int):
        """In place, on `stream` (a hipStream_t as an integer)."""
        if count:
            _check(self.L.ncclAllReduce(ptr, ptr, count, ncclInt64, ncclSum, self.comm, stream), "ncclAllReduce")

    def all_to_all_u64(self, send: int, recv: int, piece_words: int, stream: int):
        """Equal split: words [r * piece, (r + 1) * piece) of `send` go to rank r, the same range of `recv` comes from rank r."""
        _check(self.L.ncclGroupStart(), "ncclGroupStart")
        try:
            for r in range(self.world):
                off = r * piece_words * 8
                _check(self.L.ncclSend(send + off, piece_words, ncclUint64, r, self.comm, stream), "ncclSend")
                _check(self.L.ncclRecv(recv + off, piece_words, ncclUint64, r, self.comm, stream), "ncclRecv")
        except BaseException:
            self.L.ncclGroupEnd()  # (close the group, but the send / receive error is the one to report)
            raise
        _check(self.L.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
