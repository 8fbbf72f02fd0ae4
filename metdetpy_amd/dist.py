"""Multi-GPU plumbing: one process per GPU, independent streams per rank, RCCL only for the final gather.

The hot path has no exchange step between streams (mode "independent streams"): each rank owns
its detector state (ring, binaries, noise EMA, collector), so the only collective is the gather of the finished
detection records (a few hundred bytes per stream) -- latency-bound, done once at the end.
`torch.distributed` is used for the rendezvous and the collectives only (backend "nccl" is RCCL on ROCm,
"gloo" on CPU for the tests); no tensor of the data path goes through torch.
"""
from __future__ import annotations

import json
import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns
    (rank, local_rank, world, dist_module_or_None)."""
    rank, local, world = env_world()
    if world <= 1:
        return rank, local, world, None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, local, world, dist


def shard_streams(n_streams: int, rank: int, world: int) -> list[int]:
    """Stream ids owned by `rank`: round-robin, so any n_streams / world combination is covered exactly once."""
    return [s for s in range(n_streams) if s % world == rank]


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_records(local: dict, dist) -> dict | None:
    """Gather {stream_id: [record dicts]} from every rank to rank 0 (returns None elsewhere).
    Payload: JSON bytes in uint8 tensors; one all_gather for the sizes, one for the padded payloads."""
    if dist is None:
        return dict(local)
    import torch
    dev = _device(dist)
    world = dist.get_world_size()
    blob = json.dumps({str(k): v for k, v in local.items()}).encode("utf-8")
    size = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
    if blob:
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for s, b in zip(sizes, bufs):
        part = json.loads(bytes(b[:int(s.item())].cpu().numpy().tobytes()).decode("utf-8"))
        for k, v in part.items():
            if int(k) in merged:
                raise ValueError(f"stream {k} reported by two ranks")
            merged[int(k)] = v
    return merged


def max_over_ranks(value: float, dist) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
