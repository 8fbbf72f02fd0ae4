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


class RcclError(RuntimeError):
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
    """One rank of an RCCL communicator on the current HIP device (the caller's context has selected it)."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of ncclGetUniqueId")
        self.L = load()
        self.rank, self.world = int(rank), int(world)
        uid = UniqueId()
        C.memmove(C.addressof(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        _check(self.L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    @classmethod
    def from_torch_dist(cls, dist, rank: int, world: int):
        """The id is made on rank 0 and broadcast through the process group the launcher set up (control plane only)."""
        box = [new_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

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
        finally:
            _check(self.L.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
