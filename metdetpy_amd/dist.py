"""Multi-GPU plumbing: one process per GPU, RCCL only where the path has an exchange.  No torch anywhere.

The hot path has no exchange step between independent streams: each rank owns its detector state (ring,
binaries, noise EMA, collector), so the only collective of that mode is the gather of the finished detection records -- a
few hundred bytes per stream, done once at the end ("RCCL only for the final detection gather", BASELINE.json north_star).

`init_from_env()` joins the job a launcher described with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
(`python -m torch.distributed.run` exports exactly these; the launcher is the driver's, nothing of torch is imported here):

  * the control plane is the key/value store of rendezvous.py (rank 0 serves it): barrier, the 128-byte ncclUniqueId, and
    the collectives of ranks that have NO communicator;
  * when every rank has a device of its own the data collectives are RCCL calls (librccl through ctypes, rccl.py) on a
    private HIP stream: `ncclAllGather` for the gather of the detection records / counts, `ncclAllReduce(max)` for the
    max-over-ranks time.  RCCL refuses two ranks on one device, so ranks that SHARE a device (the one-GPU build boxes:
    world 2 or 8 on GPU 0) fall back to the store for those few bytes -- `Group.backend` says which ("rccl" | "store").
    MDH_DIST_BACKEND=store forces the fall-back.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct

from . import rendezvous


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class Group:
    """This rank's view of the job: control plane (always) + an RCCL communicator with a stream of its own (when possible)."""

    def __init__(self, rank: int, local: int, world: int, backend: str | None = None, timeout_s: float | None = None):
        self.rank, self.local, self.world = int(rank), int(local), int(world)
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("MDH_RCCL_TIMEOUT", "120"))
        self.cp = rendezvous.Group(self.rank, self.world, timeout_s=self.timeout_s)
        self.comm = None
        self.device = self.local
        self._stream = self._dbuf = None
        self._dcap = 0
        want = backend or os.environ.get("MDH_DIST_BACKEND") or "auto"
        if want not in ("auto", "rccl", "store"):
            raise ValueError("backend must be auto | rccl | store")
        # (asked whatever the backend: with the store forced -- tools/run_8gpu.sh --dry-run -- the ranks still spread over the
        # devices there are, round-robin, instead of piling onto device 0)
        from . import _lib
        try:
            ndev = int(_lib.load().mdh_device_count())
        except (_lib.MetDetHipError, OSError):
            ndev = 0
        # every rank must take the same branch: rank 0 decides from what all ranks report (devices visible, local rank)
        reports = self.cp.all_gather_object((ndev, self.local))
        local_world = 1 + max(r[1] for r in reports)
        own_device = all(r[0] >= local_world for r in reports)
        if want == "rccl" and not own_device:
            self.cp.close()
            raise RuntimeError(f"backend rccl: {local_world} ranks on this node but {min(r[0] for r in reports)} HIP device(s) visible "
                               "-- RCCL refuses two ranks on one device")
        self.rccl_note = None
        if want != "store" and own_device:
            # RCCL with more than one rank has never run on the build's one-GPU boxes: whatever goes wrong here -- the communicator
            # does not form, the first collective does not return what it must -- is reported on stderr and the job goes on over
            # the store (every rank takes the same decision: the outcomes are exchanged through the control plane first) unless the
            # caller insisted on RCCL.
            from . import _lib, rccl
            err = None
            try:
                self.comm = rccl.Communicator.from_group(self.cp, device=self.local, timeout_s=self.timeout_s)
                self._L = _lib.load()
                self._stream = self._L.mdh_stream_create(self.local)
                if not self._stream:
                    raise _lib.MetDetHipError(f"mdh_stream_create failed: {_lib.last_error()}")
                got = self._all_gather_fixed(struct.pack("<q", 1000 + self.rank))  # the first collective: every rank's number, in order
                if [struct.unpack("<q", b)[0] for b in got] != [1000 + r for r in range(self.world)]:
                    raise rccl.RcclError(f"ncclAllGather returned {got!r}")
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            outcomes = self.cp.all_gather_object(err)
            if any(o is not None for o in outcomes):
                first = next(o for o in outcomes if o is not None)
                if want == "rccl":
                    self.cp.close()
                    raise RuntimeError(f"backend rccl: {first}")
                import sys
                if self.rank == 0:
                    print(f"[metdetpy_amd.dist] RCCL is not usable ({first}); the final gather goes through the rendezvous store", file=sys.stderr, flush=True)
                if self.comm is not None:
                    try:
                        self.comm.abort()
                    except Exception:  # noqa: BLE001
                        pass
                    self.comm = None
                self.rccl_note = first
        else:
            self.device = self.local % max(ndev, 1)  # ranks share devices round-robin
        self.backend = "rccl" if self.comm is not None else "store"

    # ---- device staging of the RCCL branch: a few bytes in, world x a few bytes out -------------
    def _staging(self, nbytes: int) -> int:
        from . import _lib
        if self._dcap < nbytes:
            if self._dbuf:
                _lib.check(self._L.mdh_dev_free(self.local, self._dbuf), "mdh_dev_free")
            self._dcap = max(4096, 2 * nbytes)
            self._dbuf = self._L.mdh_dev_malloc(self.local, self._dcap)
            if not self._dbuf:
                raise _lib.MetDetHipError(f"mdh_dev_malloc failed: {_lib.last_error()}")
        return self._dbuf

    def _sync(self):
        from . import _lib
        rc = self._L.mdh_stream_sync(self.local, self._stream, self.timeout_s * 1e3)
        if rc:
            if self.comm is not None:
                self.comm.abort()  # (releases the collective a dead peer left hanging)
            _lib.check(rc, "mdh_stream_sync (a collective did not complete: a peer is missing)")

    def _all_gather_fixed(self, blob: bytes) -> list:
        """RCCL all-gather of equally sized byte strings."""
        from . import _lib
        n = len(blob)
        buf = self._staging((self.world + 1) * n)
        send, recv = buf, buf + n
        src = C.create_string_buffer(blob, n)
        _lib.check(self._L.mdh_dev_copy(self.local, send, C.addressof(src), n, 1, self._stream), "mdh_dev_copy")
        self.comm.all_gather_bytes(send, recv, n, self._stream)
        self._sync()
        out = C.create_string_buffer(self.world * n)
        _lib.check(self._L.mdh_dev_copy(self.local, C.addressof(out), recv, self.world * n, 0, self._stream), "mdh_dev_copy")
        return [out.raw[r * n:(r + 1) * n] for r in range(self.world)]

    # ---- collectives -----------------------------------------------------------------------------
    def barrier(self):
        self.cp.barrier()

    def all_gather_bytes(self, blob: bytes) -> list:
        """The final detection gather's primitive.  RCCL: one all-gather of the sizes, one of the padded payloads."""
        if self.world == 1:
            return [blob]
        if self.comm is None:
            return self.cp.all_gather_bytes(blob)
        sizes = [struct.unpack("<q", b)[0] for b in self._all_gather_fixed(struct.pack("<q", len(blob)))]
        cap = max(max(sizes), 1)
        cap = (cap + 15) // 16 * 16
        parts = self._all_gather_fixed(blob + b"\0" * (cap - len(blob)))
        return [p[:s] for p, s in zip(parts, sizes)]

    def all_gather_i64(self, values) -> list:
        """[[int64 ...] per rank]: the detection counts of the timed harness."""
        vals = [int(v) for v in values]
        blob = struct.pack(f"<{len(vals)}q", *vals)
        got = [blob] if self.world == 1 else (self._all_gather_fixed(blob) if self.comm is not None else self.cp.all_gather_bytes(blob))
        return [list(struct.unpack(f"<{len(vals)}q", b)) for b in got]

    def max_f64(self, value: float) -> float:
        if self.world == 1:
            return float(value)
        if self.comm is None:
            return max(struct.unpack("<d", b)[0] for b in self.cp.all_gather_bytes(struct.pack("<d", float(value))))
        from . import _lib
        buf = self._staging(8)
        src = C.c_double(float(value))
        _lib.check(self._L.mdh_dev_copy(self.local, buf, C.addressof(src), 8, 1, self._stream), "mdh_dev_copy")
        self.comm.all_reduce_max_f64(buf, 1, self._stream)
        self._sync()
        out = C.c_double(0.0)
        _lib.check(self._L.mdh_dev_copy(self.local, C.addressof(out), buf, 8, 0, self._stream), "mdh_dev_copy")
        return float(out.value)

    def all_gather_object(self, obj) -> list:
        if self.comm is None or self.world == 1:
            return self.cp.all_gather_object(obj)  # (sealed and verified before unpickling: rendezvous.Group)
        import pickle  # over RCCL the bytes travel between this job's own device buffers
        return [pickle.loads(b) for b in self.all_gather_bytes(pickle.dumps(obj, protocol=4))]

    def gather_object(self, obj, dst: int = 0):
        """Everything to rank `dst` (None elsewhere).  Over RCCL it is an all-gather whose result only `dst` unpacks: the payloads
        are detection records, a few hundred bytes per stream."""
        if self.comm is None and self.world > 1:
            return self.cp.gather_object(obj, dst)
        got = self.all_gather_object(obj)
        return got if self.rank == dst else None

    # ---- host-buffer data path for ranks WITHOUT a communicator (shared device; the CPU engines of the tests) ----
    def all_reduce_sum_i64(self, arr):
        """numpy int64 array -> elementwise sum over the ranks (exact)."""
        import numpy as np
        a = np.ascontiguousarray(arr, dtype=np.int64)
        if self.world == 1 or a.size == 0:
            return a
        parts = self.cp.all_gather_bytes(a.tobytes())
        return np.sum([np.frombuffer(p, np.int64) for p in parts], axis=0).reshape(a.shape)

    def all_to_all_u64(self, send):
        """send: (world, ...) uint64 array, slice r goes to rank r; returns the (world, ...) array of what the ranks sent here."""
        import numpy as np
        s = np.ascontiguousarray(send, dtype=np.uint64)
        if self.world == 1:
            return s
        got = self.cp.all_to_all_bytes([s[r].tobytes() for r in range(self.world)])
        return np.stack([np.frombuffer(b, np.uint64).reshape(s.shape[1:]) for b in got])

    def describe(self) -> dict:
        d = {"rank": self.rank, "world": self.world, "device": self.device, "backend": self.backend}
        if self.rccl_note:
            d["rccl_not_usable"] = self.rccl_note
        if self.comm is not None:
            d["communicator"] = self.comm.describe()
        return d

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._dbuf:
            self._L.mdh_dev_free(self.local, self._dbuf)
            self._dbuf = None
        if self._stream:
            self._L.mdh_stream_destroy(self.local, self._stream)
            self._stream = None
        self.cp.close()


def init_from_env(backend: str | None = None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (rank, local_rank, world, group or None)."""
    rank, local, world = env_world()
    if world <= 1:
        return rank, local, world, None
    return rank, local, world, Group(rank, local, world, backend=backend)


def shard_streams(n_streams: int, rank: int, world: int) -> list[int]:
    """Stream ids owned by `rank`: round-robin, so any n_streams / world combination is covered exactly once."""
    return [s for s in range(n_streams) if s % world == rank]


def gather_records(local: dict, group) -> dict | None:
    """Gather {stream_id: [record dicts]} from every rank to rank 0 (returns None elsewhere): the final detection gather.
    Payload: JSON bytes; over RCCL one all-gather for the sizes and one for the padded payloads (Group.all_gather_bytes)."""
    if group is None:
        return dict(local)
    blob = json.dumps({str(k): v for k, v in local.items()}).encode("utf-8")
    parts = group.all_gather_bytes(blob)
    if group.rank != 0:
        return None
    merged = {}
    for b in parts:
        for k, v in json.loads(b.decode("utf-8")).items():
            if int(k) in merged:
                raise ValueError(f"stream {k} reported by two ranks")
            merged[int(k)] = v
    return merged


def max_over_ranks(value: float, group) -> float:
    return float(value) if group is None else group.max_f64(value)
