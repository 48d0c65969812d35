"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Frames are independent on this path (LogoScan.hpp:1577-1580, :1132-1158), so every pass shards by contiguous
frame range with no data-path collective.  What has to be exchanged is tiny and happens once per pass:
  * the all-frames logo scan gathers its per-frame {corr0,corr1} records so that rank 0 can run selectLogo /
    writeResult over the whole clip            -> all_gather   (8 B per frame per logo)
  * logo generation all-reduces the exact int64 accumulators and agrees on which frames are inside the
    numMaxFrames quota ("first N valid frames in stream order", LogoScan.hpp:885)  -> all_gather + all_reduce
  * erase needs analysis records of 8 frames either side of a shard for CalcFade2 (:1265-1285): recomputed
    locally as a halo, nothing is sent.
  * the self-specified CM / KFM frame metrics compare a frame with the one before it: a shard brings the frame before its
    range along as a one-frame halo (decoded locally, nothing is sent), the 64-byte metric records are all-gathered and the
    cadence / scene-change decisions run replicated on every rank             -> all_gather   (64 B per frame)
Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import binding

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
    counts = torch.cat(bufs).tolist()                 # one device -> host transfer for all ranks' counts
    before = sum(counts[:rank])
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


# ------------------------------------------------------------------------------------------------------------------
# The C ABI's sharded drivers (amtgpu_scanlogo_sharded, amtgpu_logoframe_allgather_results) take two host-memory
# collectives as callbacks; this is their torch.distributed implementation (RCCL with the nccl backend, gloo on CPU).
# ------------------------------------------------------------------------------------------------------------------
class TorchCollectives:
    """AmtGpuCollectives over a torch.distributed process group.  Keep the object alive while the C side may call it."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device = device
        self.error = None

        def allgather(user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), (nbytes,))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), (nbytes * self.world,))
                t = torch.from_numpy(src.copy()).to(self.device)
                out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(out, t, group=self.group)
                dst[:] = out.cpu().numpy()
                return 1
            except Exception as e:       # no exception may cross the C boundary
                self.error = e
                return 0

        def allreduce(user, buf, count):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int64)), (count,))
                t = torch.from_numpy(a.copy()).to(self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)       # exact: int64
                a[:] = t.cpu().numpy()
                return 1
            except Exception as e:
                self.error = e
                return 0

        self._ag, self._ar = binding.ALLGATHER_CB(allgather), binding.ALLREDUCE_CB(allreduce)
        self.struct = binding.Collectives(self.rank, self.world, self._ag, self._ar, None)

    def ref(self):
        return C.byref(self.struct)


def logoframe_allgather(lf, first: int, nlocal: int, coll: TorchCollectives):
    """after lf.scan_batch over this rank's frames [first, first+nlocal): every rank gets the whole clip's records"""
    lf.ctx.check(lf.ctx.lib.amtgpu_logoframe_allgather_results(lf.h, coll.ref(), first, nlocal))


def framestats_allgather(fs, local_metrics, first: int, num_frames: int, coll: TorchCollectives):
    """amtgpu_framestats_allgather: this rank's (nlocal, 8) uint64 records of frames [first, first + nlocal) -> the whole clip's
    (num_frames, 8) records on every rank; the cadence / scene-change decisions then run replicated (fs.cadence, fs.scene_changes)."""
    local = np.ascontiguousarray(local_metrics, np.uint64).reshape(-1, 8)
    out = np.zeros((num_frames, 8), np.uint64)
    from .api import _p
    fs.ctx.check(fs.ctx.lib.amtgpu_framestats_allgather(fs.h, coll.ref() if coll is not None else None, _p(local), first, int(local.shape[0]),
                                                        num_frames, _p(out)))
    return out


def framestats_sharded(fs, Y, first: int, num_frames: int, coll: TorchCollectives, prevY=None):
    """amtgpu_framestats_sharded: Y = this rank's frames [first, first + n) resident in HBM, prevY = frame first - 1 (the one-frame
    halo; None on the rank that starts the clip) -> (num_frames, 8) uint64 records of the whole clip on every rank."""
    from .api import _p
    es = 1 if fs.bits <= 8 else 2
    out = np.zeros((num_frames, 8), np.uint64)
    fs.ctx.check(fs.ctx.lib.amtgpu_framestats_sharded(fs.h, coll.ref() if coll is not None else None, _p(Y), int(Y.stride(0)) * es, int(Y.stride(1)),
                                                      _p(prevY), first, int(Y.shape[0]), num_frames, _p(out)))
    return out


def scan_logo_sharded(ctx, clip_local, serviceid, dstpath, imgx, imgy, w, h, thy, numMaxFrames, coll: TorchCollectives, cb=None):
    """ScanLogo (LogoScan.hpp:1083-1098) over a stream whose frames are sharded by contiguous range: clip_local holds this
    rank's frames.  Rank 0 writes dstpath; returns True/False like the reference's export."""
    from .api import _p
    cbf = binding.CB(cb) if cb else binding.CB(lambda p, a, b, c: 1)
    ok = ctx.lib.amtgpu_scanlogo_sharded(ctx.h, coll.ref(), _p(clip_local.Y), _p(clip_local.U), _p(clip_local.V), clip_local.strideY,
                                         clip_local.strideUV, clip_local.pitchY, clip_local.pitchUV, clip_local.width, clip_local.height,
                                         clip_local.num_frames, serviceid, str(dstpath).encode() if dstpath else None, imgx, imgy, w, h,
                                         thy, numMaxFrames, cbf)
    return bool(ok)
