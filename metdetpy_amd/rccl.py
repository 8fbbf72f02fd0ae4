"""librccl through ctypes: the collectives of the row-band path, enqueued on the detector library's OWN streams
(`Context.stream_handle`), so that a pass is queued without a host round trip -- no torch tensor, no `torch.cuda` stream, no
synchronisation between the band kernels, the exchange and the Hough stage (include/metdet_hip.h, "WITHOUT host round trips").

Only what the sharded modes need: a communicator (the 128-byte unique id travels through the control plane of rendezvous.py: a
key/value store rank 0 serves), an in-place all-reduce of int64 sums, an equal-split all-to-all of u64 bit-plane pieces as one
group of ncclSend / ncclRecv, and an all-gather of bytes for the final detection gather.  xGMI is point to point: the all-to-all is exactly one send and one
receive per peer, `piece` words each (8 GPUs, 8K, 256 frames per pass: 32 frames x 540 rows x 120 words x 8 B = 16.6 MB per peer).

RCCL refuses two ranks on one device, so on a one-GPU box this module can only be exercised with a communicator of ONE rank (the
all-reduce is then the identity and the all-to-all a send to self): tests/test_bands.py does that under `-m gpu`; the 2-process
tests keep the gloo / host-buffer path.
"""
from __future__ import annotations

import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128
ncclSum, ncclMax = 0, 2
ncclInt8, ncclUint8, ncclInt64, ncclUint64, ncclFloat64 = 0, 1, 4, 5, 8


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


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
    L.ncclAllGather.argtypes = [vp, vp, sz, i32, vp, vp]
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

    `timeout_s`: ncclCommInitRank blocks until every rank has arrived.  The communicator stays a BLOCKING one (every later
    ncclAllReduce / ncclSend / ncclRecv / ncclGroupEnd returns ncclSuccess or a real error, never ncclInProgress: round 4 created
    it with blocking = 0 to be able to poll, and every data-path call on such a communicator may then return ncclInProgress);
    the bound on the wait comes from running the init call on a helper thread and joining it with a timeout -- a rank whose
    peers never turn up raises RcclTimeout instead of hanging (the helper thread stays parked in RCCL; the process is expected to
    exit).  After creation the communicator's own size / rank / device are checked against what the launcher said."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int | None = None, timeout_s: float | None = None):
        if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of ncclGetUniqueId")
        self.L = load()
        self.rank, self.world, self.device = int(rank), int(world), device
        uid = UniqueId()
        C.memmove(C.addressof(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()

        def init():
            if device is not None:  # (the HIP device is per thread: selected on the thread that calls ncclCommInitRank)
                from . import _lib
                _lib.set_device(int(device))
            return self.L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)

        if timeout_s is None:
            _check(init(), "ncclCommInitRank")
        else:
            import threading
            box = {}

            def run():
                try:
                    box["rc"] = init()
                except BaseException as e:  # noqa: BLE001
                    box["exc"] = e

            th = threading.Thread(target=run, name="mdh-rccl-init", daemon=True)
            th.start()
            th.join(float(timeout_s))
            if th.is_alive():
                self.comm = C.c_void_p()  # (still being written by the parked thread: never handed to RCCL again)
                raise RcclTimeout(f"rank {rank}: the communicator of {world} ranks did not form within {timeout_s} s (a peer is missing)")
            if "exc" in box:
                raise box["exc"]
            _check(box["rc"], "ncclCommInitRank")
        if device is not None:
            from . import _lib
            _lib.set_device(int(device))
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
    def from_group(cls, group, device: int | None = None, timeout_s: float | None = None):
        """The id is made on rank 0 and broadcast through the control plane (`rendezvous.Group`, or anything with its
        rank / world / broadcast_bytes)."""
        # The broadcast is UNCONDITIONAL: a rank 0 that cannot make the id (librccl missing, ncclGetUniqueId failing) still takes
        # part and sends the reason instead -- raised before the collective, rank 0 alone would have stayed one collective behind
        # the others, and the "RCCL is not usable: go on over the store" agreement that follows (dist.Group) would have met a
        # time-out instead of an answer.
        blob = None
        if group.rank == 0:
            try:
                blob = b"U" + new_unique_id()
            except Exception as e:  # noqa: BLE001
                blob = b"E" + f"{type(e).__name__}: {e}".encode("utf-8", "replace")
        blob = group.broadcast_bytes(blob, 0)
        if blob[:1] != b"U":
            raise RcclError("rank 0 could not make the ncclUniqueId: " + blob[1:].decode("utf-8", "replace"))
        return cls(group.rank, group.world, blob[1:], device=device, timeout_s=timeout_s)

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

    def all_gather_bytes(self, send: int, recv: int, nbytes: int, stream: int):
        """`nbytes` from every rank, rank order, into `recv` (world * nbytes); device pointers, on `stream`."""
        _check(self.L.ncclAllGather(send, recv, nbytes, ncclUint8, self.comm, stream), "ncclAllGather")

    def all_reduce_max_f64(self, ptr: int, count: int, stream: int):
        _check(self.L.ncclAllReduce(ptr, ptr, count, ncclFloat64, ncclMax, self.comm, stream), "ncclAllReduce")

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
