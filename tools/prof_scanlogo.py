#!/usr/bin/env python3
"""Per-kernel timing of the logo-generation path (ScanLogo, LogoScan.hpp:794-1098) on a synthetic 1440x1080 clip whose logo
rectangle has flat-background frames:  python tools/prof_scanlogo.py --frames 2048"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch

import amt_synth as S
from amatsukaze_amd import Context, DeviceClip, ScanLogo

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2048)
a = ap.parse_args()
W, H, LW, LH, X, Y0 = 1440, 1080, 256, 128, 1120, 64
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(LW, LH)
base = S.make_clip_np(64, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, period=16, fade=4, flat_every=2, pitchY=1472, pitchUV=768)
rep = (a.frames + 63) // 64                          # the CPU generator is slow: 64 distinct frames, tiled
dclip = DeviceClip(*(torch.from_numpy(base[k]).to(dev).repeat(rep, 1, 1)[: a.frames].contiguous() for k in "YUV"), W, H, 8)
ctx = Context(0)
out = os.path.join(tempfile.mkdtemp(), "gen.lgd")
ctx.profile(True)
t0 = time.perf_counter()
ok = ScanLogo(ctx, dclip, 1, out, X, Y0, LW, LH, 12, 100000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ScanLogo ok={ok} {a.frames} frames in {dt * 1e3:.1f} ms wall ({a.frames / dt:.0f} frames/s), lgd {os.path.getsize(out) if ok else 0} B")
for k, (calls, ms) in ctx.profile_report().items():
    print(f"  {k}: calls={calls} total_ms={ms:.3f} avg_ms={ms / max(1, calls):.4f}")
