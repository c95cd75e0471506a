"""One process per GPU; utterances shard with no data-path collective.

The reference has no distributed code at all ("run the script again in another
container on another GPU", README.md:41-43).  Each utterance is an independent forward
over frozen weights, so the only exchanges are
  C1  one broadcast of the frozen weights from rank 0 (RCCL over xGMI when the backend
      is "nccl"; the same code runs on gloo/CPU in the tests), as a single flat fp32 bucket;
  C2  a few integers (the compat layer index, end-of-run counters).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

StateDict = Dict[str, torch.Tensor]


def env() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None, force: bool = False) -> Tuple[int, int, int]:
    """``force``: create the process group even for a single rank (tests/test_gpu_dist.py runs every collective of this file on
    the RCCL backend with world size 1 -- one GPU cannot host two RCCL ranks)."""
    rank, world, local_rank = env()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_files(files: Sequence[str], sizes: Sequence[int], rank: int, world: int) -> List[str]:
    """Longest-first round-robin deal of the file list: every rank gets the same amount of audio
    (within one file) and its own list stays sorted by length, so consecutive files make batches of
    similar lengths.  Deterministic on every rank (ties broken by name), disjoint, exhaustive."""
    order = sorted(range(len(files)), key=lambda i: (-int(sizes[i]), files[i]))
    return [files[i] for i in order[rank::world]]


def broadcast_int(value: int, src: int = 0, device: Optional[torch.device] = None) -> int:
    if not dist.is_initialized():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device or _comm_device())
    dist.broadcast(t, src=src)
    return int(t.item())


def _comm_device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_state_dict(sd: Optional[StateDict], src: int = 0) -> Tuple[StateDict, float, int]:
    """C1.  ``sd`` is the real state dict on rank ``src`` and may be None elsewhere.
    Returns (state dict of fp32 tensors, seconds spent in the broadcast, bytes).  The tensors are VIEWS of the one
    flat bucket that was broadcast and stay where the collective left them -- in HBM on the "nccl" (RCCL) backend,
    on every rank including ``src`` -- so the weights cross xGMI once and never bounce through host memory; the
    encoder constructors fold / split them on the device and keep COPIES of the small fp32 vectors (engine._dev_f32), so
    dropping the returned dict after ``build_encoder`` frees the bucket: weight memory is 1x, not fp32 + 16-bit planes."""
    if not dist.is_initialized():
        assert sd is not None
        return sd, 0.0, 0
    rank = dist.get_rank()
    manifest = [[(k, tuple(sd[k].shape)) for k in sorted(sd)]] if rank == src else [None]
    dist.broadcast_object_list(manifest, src=src)
    entries = manifest[0]
    # every entry starts on a 16-byte boundary of the bucket (the kernels read biases / LayerNorm vectors with 16-byte loads
    # and the encoder constructors may pass a view straight to a launcher)
    starts, total = [], 0
    for _, shape in entries:
        starts.append(total)
        total += (int(torch.Size(shape).numel()) + 3) // 4 * 4
    dev = _comm_device()
    if rank == src:
        flat = torch.zeros(total, dtype=torch.float32, device=dev)      # filled tensor by tensor: no second host copy of the checkpoint
        for (k, shape), o in zip(entries, starts):
            n = int(torch.Size(shape).numel())
            flat[o:o + n].copy_(sd[k].reshape(-1).to(torch.float32), non_blocking=False)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=dev)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)              # one flat bucket: per-link-bound ring over xGMI
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == src and dev.type == "cpu":
        return sd, dt, total * 4                   # gloo: rank src already holds CPU tensors
    out: StateDict = {}
    for (k, shape), o in zip(entries, starts):
        n = int(torch.Size(shape).numel())
        out[k] = flat[o:o + n].view(shape)         # no copy: a view into the bucket, which lives until the last view is dropped
    return out, dt, total * 4


def max_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float) -> float:
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
