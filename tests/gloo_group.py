"""TEST infrastructure: a `dist.Group` look-alike over torch.distributed's gloo backend (CPU tensors).

The product package imports no torch (its control plane is metdetpy_amd/rendezvous.py, its data collectives RCCL through ctypes);
the multi-process CPU tests run every sharded mode twice -- over the product's own store-backed group and over this independent
gloo-backed one -- so that the exchange logic is checked against a second, well-known implementation of the collectives.
"""
import os
import pickle

import numpy as np


class GlooGroup:
    backend = "gloo"
    comm = None

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.d = torch, dist
        dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = 0

    def barrier(self):
        self.d.barrier()

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.d.all_gather_object(out, obj)
        return out

    def gather_object(self, obj, dst=0):
        out = [None] * self.world if self.rank == dst else None
        self.d.gather_object(obj, out, dst=dst)
        return out

    def all_gather_bytes(self, blob: bytes):
        return [pickle.loads(b) for b in self.all_gather_object(pickle.dumps(blob))]

    def all_gather_i64(self, values):
        t = self.torch.tensor([int(v) for v in values], dtype=self.torch.int64)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.d.all_gather(out, t)
        return [o.tolist() for o in out]

    def max_f64(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.d.all_reduce(t, op=self.d.ReduceOp.MAX)
        return float(t.item())

    def all_reduce_sum_i64(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.int64)
        if a.size == 0:
            return a
        t = self.torch.from_numpy(a.copy())
        self.d.all_reduce(t)
        return t.numpy()

    def all_to_all_u64(self, send):
        s = self.torch.from_numpy(np.ascontiguousarray(send, dtype=np.uint64).view(np.int64))
        r = self.torch.empty_like(s)
        self.d.all_to_all_single(r, s)
        return r.numpy().view(np.uint64)

    def close(self):
        self.d.barrier()
        self.d.destroy_process_group()


def make_group(kind: str):
    """kind: "store" (the product's dist.Group over the rendezvous store; RCCL when every rank has a device) or "gloo"."""
    if kind == "gloo":
        return GlooGroup()
    from metdetpy_amd import dist as md
    rank, local, world = md.env_world()
    return md.Group(rank, local, world)
