#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: host frames -> pinned ring -> hipMemcpyAsync on the side stream (amtgpu_frames_upload)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from amatsukaze_amd import Context

ctx = Context(0)
frame = 1472 * 1080 + 2 * 768 * 540
n = 512
host = np.random.randint(0, 255, n * frame, dtype=np.uint8)
d = ctx.lib.amtgpu_device_alloc(ctx.h, host.nbytes)
for _ in range(2):
    t0 = time.perf_counter()
    ctx.check(ctx.lib.amtgpu_frames_upload(ctx.h, d, host.ctypes.data, host.nbytes))
    ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))
    ctx.synchronize()
    dt = time.perf_counter() - t0
print(f"ingest: {host.nbytes / dt / 1e9:.1f} GB/s = {n / dt:.0f} frames/s (1440x1080 YUV420 8-bit, pitch 1472, pageable host -> pinned ring -> HBM)")
back = np.empty(4096, np.uint8)
ctx.check(ctx.lib.amtgpu_download(ctx.h, back.ctypes.data, d, 4096))
assert np.array_equal(back, host[:4096])
ctx.lib.amtgpu_device_free(ctx.h, d)
