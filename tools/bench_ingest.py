"""PCIe-inclusive ingest rates for bench.py's `ingest` object (never `value`): frames that are NOT resident in HBM.

Two hand-offs from the decoder (AMTSource::GetFrame, AMTSource.hpp:721-780; frame buffers :428-442) are measured:
  * pageable frames -> amtgpu_frames_upload: staged through the ring of four pinned 16 MiB slots, the staging memcpy shared out over
    the context's upload threads, hipMemcpyAsync on the side stream;
  * frames in a pool the host has registered once (amtgpu_frames_register = hipHostRegister): one DMA copy straight out of the
    caller's memory, no staging.
Each alone and pipelined against the analysis of the previous batch, for whole Y planes (the frame metrics need every sample) and
for the logo rectangle's rows only (all the logo passes read).  `link` = tools/ubench/pcie_ceiling on the same box: what plain
pinned hipMemcpyAsync carries, so that the rates above can be read as a fraction of the link.
"""
from __future__ import annotations

import json
import os
import subprocess
import time

import numpy as np


def link_ceiling(root):
    exe = os.path.join(root, "tools", "ubench", "pcie_ceiling")
    src = exe + ".cpp"
    try:
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-o", exe, src, "-pthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                           # the ceiling is context, never a reason to lose the line
        return {"error": f"{type(e).__name__}: {e}"}


def measure(ctx, logos, alpha, alphaUV, dev, K, B=256, batches=10, threads=None, with_link=True):
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, FrameStats, LogoFrame
    W, H, PITCH_Y, PITCH_UV, IMGX, IMGY, LH, MASKRATIO = (K[k] for k in ("W", "H", "PITCH_Y", "PITCH_UV", "IMGX", "IMGY", "LH", "MASKRATIO"))
    g = S.make_clip_torch(B, W, H, 0x5EED0002, alpha, alphaUV, IMGX, IMGY, dev, pitchY=PITCH_Y, pitchUV=PITCH_UV, chroma=False)
    hY = g["Y"].cpu().numpy().copy()                                  # the "decoder output": pageable host memory
    del g
    ybytes = hY.nbytes
    # rectangle-only variant for the logo passes: the rows [IMGY, IMGY+LH) of the Y plane at full pitch (the kernels address
    # (imgx, imgy) inside a frame, so the upload keeps the pitch and drops the rows nobody reads): LH*pitch bytes per frame
    hR = np.ascontiguousarray(hY[:, IMGY:IMGY + LH, :])
    rbytes = hR.nbytes
    dbuf = [torch.empty((B, H, PITCH_Y), dtype=torch.uint8, device=dev) for _ in range(2)]
    lf = LogoFrame(ctx, [logos[0]], MASKRATIO)
    lf.begin(W, H, 8, B * batches)
    an = AMTAnalyzeLogo(ctx, logos[0], MASKRATIO)
    fs = FrameStats(ctx, W, H, 8)
    d_an = torch.empty((B, 33), dtype=torch.float32, device=dev)
    d_st = torch.empty((B, 8), dtype=torch.int64, device=dev)
    if threads is not None:
        ctx.check(ctx.lib.amtgpu_context_set_upload_threads(ctx.h, threads))

    def upload(k, rect):
        if rect:   # rows [IMGY, IMGY+LH) of every frame: B pieces of LH*pitch bytes, one frame stride apart on the device
            ctx.check(ctx.lib.amtgpu_frames_upload_strided(ctx.h, dbuf[k & 1].data_ptr() + IMGY * PITCH_Y, H * PITCH_Y, hR.ctypes.data,
                                                           LH * PITCH_Y, LH * PITCH_Y, B))
        else:
            ctx.check(ctx.lib.amtgpu_frames_upload(ctx.h, dbuf[k & 1].data_ptr(), hY.ctypes.data, ybytes))

    def compute(k, rect):
        ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))          # compute stream waits for the copies issued so far
        lf.scan_batch(dbuf[k & 1], 8, k * B, B)
        an.analyze_device(dbuf[k & 1], 8, d_an)
        if not rect:
            fs.run_device(dbuf[k & 1], d_st)

    def timed(fn, *a):
        fn(*a)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            fn(*a)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    def ingest_only(rect):
        for k in range(batches):
            upload(k, rect)
        ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))

    def compute_only(rect):
        for k in range(batches):
            compute(k, rect)

    def pipelined(rect):
        done = [torch.cuda.Event(), torch.cuda.Event()]
        upload(0, rect)
        for k in range(batches):
            compute(k, rect)
            done[k & 1].record()
            if k + 1 < batches:
                if k >= 1:
                    done[(k + 1) & 1].synchronize()                   # batch k-1 has released the buffer batch k+1 goes into
                upload(k + 1, rect)

    n = B * batches

    def block(rect, nbytes):
        ti, tc, tp = timed(ingest_only, rect), timed(compute_only, rect), timed(pipelined, rect)
        return {"bytes_per_frame": nbytes // B, "ingest_only_fps": n / ti, "ingest_GBs": nbytes * batches / ti / 1e9, "compute_only_fps": n / tc,
                "pipelined_fps": n / tp, "passes": "scan 1 logo + analysis (logo passes only)" if rect else "scan 1 logo + analysis + frame metrics"}

    out = {"what": "PCIe-inclusive rates (never `value`): frames start in host memory and the previous batch is analysed meanwhile.  y_plane / "
                   "logo_rectangle_rows: pageable frames through amtgpu_frames_upload (ring of four pinned 16 MiB slots, staging memcpy on "
                   "`upload_threads` threads, hipMemcpyAsync on the side stream); *_registered: the frames sit in a pool the host registered once "
                   "(amtgpu_frames_register), one DMA copy straight from the caller's memory",
           "batch_frames": B, "batches": batches}
    out["y_plane"] = block(False, ybytes)
    out["logo_rectangle_rows"] = block(True, rbytes)
    # ---- the decoder's pool page-locked in place ----
    t0 = time.perf_counter()
    ctx.check(ctx.lib.amtgpu_frames_register(ctx.h, hY.ctypes.data, ybytes))
    ctx.check(ctx.lib.amtgpu_frames_register(ctx.h, hR.ctypes.data, rbytes))
    reg_s = time.perf_counter() - t0
    try:
        out["y_plane_registered"] = block(False, ybytes)
        out["logo_rectangle_rows_registered"] = block(True, rbytes)
        # what arrived is what was sent
        upload(0, False)
        ctx.check(ctx.lib.amtgpu_frames_upload_wait(ctx.h))
        torch.cuda.synchronize()
        out["registered_upload_verified"] = bool(np.array_equal(dbuf[0][B - 1].cpu().numpy(), hY[B - 1]))
    finally:
        ctx.check(ctx.lib.amtgpu_frames_unregister(ctx.h, hY.ctypes.data))
        ctx.check(ctx.lib.amtgpu_frames_unregister(ctx.h, hR.ctypes.data))
    out["register_GBs"] = (ybytes + rbytes) / reg_s / 1e9
    if with_link:
        out["link"] = link_ceiling(K["ROOT"])
        peak = out["link"].get("pinned_h2d_GBs_256MB")
        if peak:
            out["fraction_of_link"] = {"y_plane": out["y_plane"]["ingest_GBs"] / peak, "y_plane_registered": out["y_plane_registered"]["ingest_GBs"] / peak}
    return out
