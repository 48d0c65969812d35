"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Frames are independent on this path (LogoScan.hpp:1577-1580, :1132-1158), so every pass shards by contiguous
frame range with no data-path collective.  What has to be exchanged is tiny and happens once per pass:
  * the all-frames logo scan gathers its per-frame {corr0,corr1} records so that rank 0 can run selectLogo /
    writeResult over the whole clip            -> all_gather   (8 B per frame per logo)
  * logo generation all-reduces the exact int64 accumulators and agrees on which frames are inside the
    numMaxFrames quota ("first N valid frames in stream order", LogoScan.hpp:885)  -> all_gather + all_reduce
  * erase needs analysis records of 8 frames either side of a shard for CalcFade2 (:1265-1285): recomputed
    locally as a halo, nothing is sent.
Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

FADE_HALO = 8   # CalcFade2 looks at frames n-8 .. n+8


def shard_range(num_frames: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous [first, last) of `rank`; earlier ranks take the remainder"""
    base, rem = divmod(num_frames, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def halo_range(first: int, last: int, num_frames: int, halo: int = FADE_HALO) -> tuple[int, int]:
    return max(0, first - halo), min(num_frames, last + halo)


def gather_frame_records(local: torch.Tensor, num_frames: int, group=None) -> torch.Tensor:
    """local: this rank's [n_local, ...] records (frames shard_range(num_frames, rank, world)).  Returns the
    [num_frames, ...] records of the whole clip on every rank.  Ragged shards are padded to the largest."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(num_frames, r, world) for r in range(world)]
    nmax = max(b - a for a, b in sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)


def stream_order_quota(local_valid: int, max_valid: int, group=None) -> int:
    """How many of this rank's valid frames fall inside the first `max_valid` valid frames of the whole stream."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = torch.tensor([local_valid], dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = t.to(dev)
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t, group=group)
    before = sum(int(b.item()) for b in bufs[:rank])
    return max(0, min(local_valid, max_valid - before))


def allreduce_scan_sums(sums: torch.Tensor, plane_sums: torch.Tensor, nframes: int, group=None):
    """Exact (int64) all-reduce of the logo-scan accumulators; returns (sums, plane_sums, nframes) of the whole clip."""
    if dist.get_world_size(group) == 1:
        return sums, plane_sums, nframes
    n = torch.tensor([nframes], dtype=torch.int64, device=sums.device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(plane_sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return sums, plane_sums, int(n.item())
