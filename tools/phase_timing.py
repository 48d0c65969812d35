#!/usr/bin/env python3
"""Per-phase cycle counts of one workgroup of the fused evaluation kernel (instrumented build, -DAMT_FUSED_TIMING).

  here (CPU box):   python tools/phase_timing.py --build
  on the GPU box:   AMTGPU_LIB=amatsukaze_amd/libamt_gpu_timing.so python tools/phase_timing.py --what analyze
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--what", default="analyze")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--logos", type=int, default=1, help="scan: candidate logos (copies of one)")
ap.add_argument("--mode", default="exact", help="analysis mode: exact | linear_unguarded")
a = ap.parse_args()
if a.build:
    from amatsukaze_amd import build as b
    print(b.build_variant("timing", ["AMT_FUSED_TIMING", "AMT_PAIR_TIMING", "AMT_LIN_TIMING"]))
    sys.exit(0)

import torch

import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, DeviceClip, Logo, LogoFrame

W, H, LW, LH, X, Y0 = 1440, 1080, 256, 128, 1120, 64
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(a.frames, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, pitchY=1472, pitchUV=768)
dclip = DeviceClip(clip["Y"], clip["U"], clip["V"], W, H, 8)
ctx = Context(0)
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
if a.what == "scan":
    # the kernel dumps its counters past the results of the frames it was given: declare a longer clip, scan its head
    import numpy as np
    lf = LogoFrame(ctx, [logo] * a.logos, 0.35)
    lf.begin(W, H, 8, a.frames + 128)
    for _ in range(2):
        lf.scan_batch(dclip.Y[: a.frames], 8, 0, a.frames)
    torch.cuda.synchronize()
    r = np.ascontiguousarray(lf.evalResults.reshape(-1)[a.frames * 2 * a.logos:a.frames * 2 * a.logos + 2 * (16 * 8 + 16)])
    tall = r.view(np.int64)
    hwid = tall[16 * 8:16 * 8 + 16]
    tall = tall[:16 * 8].reshape(16, 8)
    nw = int((tall.sum(1) > 0).sum())
    t = tall[[0, min(3, nw - 2), nw - 2, nw - 1]]
    print("waves of the middle workgroup: SIMD (HW_ID bits 5:4) / evaluation phase cycles / barrier wait:")
    print("  " + "  ".join(f"w{w}:simd{(int(hwid[w]) >> 4) & 3}/{int(tall[w, 2])}/{int(tall[w, 5])}" for w in range(nw)))
else:
    out = torch.zeros((a.frames + 8, 33), dtype=torch.float32, device=dev)     # the kernel dumps its counters past the results
    an = AMTAnalyzeLogo(ctx, logo, 0.35, mode=a.mode)
    for _ in range(2):
        an.analyze_device(dclip.Y[: a.frames], 8, out)
    torch.cuda.synchronize()
    t = out[a.frames:].reshape(-1)[:64].contiguous().view(torch.int64).cpu().numpy().reshape(4, 8)
if a.what == "scan":
    pn = ["convert raw -> {s,bg} / ordered sum", "next tile + request raw samples", "window reads + evaluation", "previous terms, scale gathers",
          "band end: last terms, next pixel + taps", "wait at the barrier", "loop bookkeeping", "wait for the raw samples"]
    tot = t.sum(1)
    print("pair kernel: cycles (s_memtime ticks) of the middle workgroup; three evaluation waves (first, fourth, last) and the summing wave:")
    for k in range(8):
        print(f"  {pn[k]:40s} " + "  ".join(f"{t[w, k]:10d} ({100.0 * t[w, k] / max(1, tot[w]):4.1f}%)" for w in range(4)))
    print("  total                                    " + "  ".join(f"{tot[w]:10d}        " for w in range(4)))
    sys.exit(0)
if a.what != "scan" and a.mode != "exact":
    ln = ["loop bookkeeping", "convert next frame's tile, request the one after", "window reads + evaluation of s and bg", "bins of the 11 fades + fix-up",
          "previous terms, quad sums -> running sums", "new tile's pixel + taps, scale gathers", "-", "-"]
    tot = t.sum(1)
    print("linear kernel: cycles (s_memtime ticks) of a workgroup of the deint logo, its four waves (the timing build spills: indicative only):")
    for k in range(8):
        print(f"  {ln[k]:48s} " + "  ".join(f"{t[w, k]:10d} ({100.0 * t[w, k] / max(1, tot[w]):4.1f}%)" for w in range(4)))
    print("  total                                " + "  ".join(f"{tot[w]:10d}        " for w in range(4)))
    sys.exit(0)
names = ["band prologue (slot, taps)", "staging loads+LDS writes", "ordered sum (one wave)", "wait B1", "window reads", "fade loop",
         "wait B0", "-"]
tot = t[:, :7].sum(1)
print("cycles (s_memtime ticks) of the middle workgroup, per wave:")
for k in range(7):
    print(f"  {names[k]:28s} " + "  ".join(f"{t[w, k]:10d} ({100.0 * t[w, k] / max(1, tot[w]):4.1f}%)" for w in range(4)))
print("  total                        " + "  ".join(f"{tot[w]:10d}        " for w in range(4)))
