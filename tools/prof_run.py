#!/usr/bin/env python3
"""Small driver for rocprofv3: runs one component of the hot path a few times on synthetic 1440x1080 frames.
  python tools/prof_run.py --what analyze --frames 1024 --iters 3
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch

import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, FrameStats, Logo, LogoFrame

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="analyze")
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--logos", type=int, default=1, help="candidate logos of the LogoFrame scan (bench.py: 3)")
ap.add_argument("--mode", default="exact", help="analysis mode: exact | linear")
a = ap.parse_args()
W, H, LW, LH, X, Y0 = 1440, 1080, 256, 128, 1120, 64
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(a.frames, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, pitchY=1472, pitchUV=768)
dclip = DeviceClip(clip["Y"], clip["U"], clip["V"], W, H, 8)
ctx = Context(0)
if os.environ.get("AMT_STATS_MASK_CUS"):         # launches on the first N CUs only (hipExtStreamCreateWithCUMask; bits interleave over the 8 XCDs)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)()
    for i in range(int(os.environ["AMT_STATS_MASK_CUS"])):
        words[i // 32] |= 1 << (i % 32)
    mst = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(mst), 8, words) == 0
    ctx.check(ctx.lib.amtgpu_context_set_stream(ctx.h, mst))
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
out = torch.empty((a.frames, 33), dtype=torch.float32, device=dev)
st = torch.empty((a.frames, 8), dtype=torch.int64, device=dev)
an = AMTAnalyzeLogo(ctx, logo, 0.35, mode=a.mode)
cands = [logo] + [Logo.from_planes(ctx, S.make_logo(LW, LH, seed=0x10600002 + k, strength=0.5 + 0.1 * k)[0], LW, LH, W, H, X, Y0)
                  for k in range(a.logos - 1)]
lf = LogoFrame(ctx, cands, 0.35)
lf.begin(W, H, 8, a.frames)
fs = FrameStats(ctx, W, H, 8)
ctx.profile(True)
for _ in range(a.iters):
    if a.what in ("analyze", "all"):
        an.analyze_device(dclip.Y, 8, out)
    if a.what in ("scan", "all"):
        lf.scan_batch(dclip.Y, 8, 0, a.frames)
    if a.what in ("stats", "all"):
        fs.run_device(dclip.Y, st)
torch.cuda.synchronize()
for k, (calls, ms) in ctx.profile_report().items():
    print(f"{k}: calls={calls} avg_ms={ms / max(1, calls):.4f} -> {a.frames * a.iters / (ms * 1e-3) if ms else 0:.0f} frames/s in this kernel")
