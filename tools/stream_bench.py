#!/usr/bin/env python3
"""Frames that are NOT resident: host (pageable) -> pinned ring -> hipMemcpyAsync on the side stream, overlapped with the
analysis of the previous batch on the compute stream (double-buffered device batches).  Prints the PCIe-inclusive rate next to
ingest alone and compute alone.  This is the number DESIGN.md quotes beside `value`; bench.py's `value` is HBM-resident.
  python tools/stream_bench.py --batch 256 --batches 12
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import numpy as np
import torch

import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, FrameStats, Logo, LogoFrame

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--batches", type=int, default=12)
a = ap.parse_args()
W, H, PY, PUV, LW, LH, X, Y0 = 1440, 1080, 1472, 768, 256, 128, 1120, 64
B = a.batch
dev = torch.device("cuda:0")
ctx = Context(0)
data, alpha, alphaUV = S.make_logo(LW, LH)
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
# one batch of distinct frames generated on the GPU, copied to pageable host memory: the "decoder output"
g = S.make_clip_torch(B, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, pitchY=PY, pitchUV=PUV)
hY = g["Y"].cpu().numpy().copy()                      # Y plane only is analysed here; a full frame is 1.5x these bytes
del g
ybytes = hY.nbytes
dbuf = [torch.empty((B, H, PY), dtype=torch.uint8, device=dev) for _ in range(2)]
lf = LogoFrame(ctx, [logo], 0.35)
lf.begin(W, H, 8, B * a.batches)
an = AMTAnalyzeLogo(ctx, logo, 0.35)
fs = FrameStats(ctx, W, H, 8)
d_an = torch.empty((B, 33), dtype=torch.float32, device=dev)
d_st = torch.empty((B, 8), dtype=torch.int64, device=dev)


def upload(k):
    ctx.check(ctx.lib.amtgpu_frames_upload(ctx.h, dbuf[k & 1].data_ptr(), hY.ctypes.data, ybytes))


def compute(k):
    ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))          # compute stream waits for the copies issued so far
    lf.scan_batch(dbuf[k & 1], 8, k * B, B)
    an.analyze_device(dbuf[k & 1], 8, d_an)
    fs.run_device(dbuf[k & 1], d_st)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def ingest_only():
    for k in range(a.batches):
        upload(k)
    ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))


def compute_only():
    for k in range(a.batches):
        compute(k)


def pipelined():
    done = [torch.cuda.Event(), torch.cuda.Event()]            # compute on a buffer finished
    upload(0)
    for k in range(a.batches):
        compute(k)                                              # launches are asynchronous ...
        done[k & 1].record()
        if k + 1 < a.batches:
            if k >= 1:
                done[(k + 1) & 1].synchronize()                 # batch k-1 has released the buffer batch k+1 goes into
            upload(k + 1)                                       # ... so this host copy + DMA runs beside batch k's kernels


for fn in (ingest_only, compute_only, pipelined):
    fn()                                                        # warm up
n = B * a.batches
ti, tc, tp = timed(ingest_only), timed(compute_only), timed(pipelined)
print(f"{n} frames of 1440x1080 (Y plane, {ybytes / B / 1e6:.2f} MB each), batches of {B}:")
print(f"  ingest alone   {n / ti:9.0f} frames/s  ({ybytes * a.batches / ti / 1e9:.1f} GB/s host -> HBM)")
print(f"  compute alone  {n / tc:9.0f} frames/s  (scan 1 logo + analysis + frame metrics, resident)")
print(f"  pipelined      {n / tp:9.0f} frames/s  (takes {100 * tp / max(ti, tc):.0f} % of the time of the slower stage alone; 100 % = perfect overlap)")
