#!/usr/bin/env python3
"""diagnostic: the linear analysis at the bench's launch geometry against the exact mode (error, frames the guard flags)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, DeviceClip, Logo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W, H, LW, LH, X, Y0 = 1440, 1080, 256, 128, 1120, 64
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, period=900, fade=12, pitchY=1472, pitchUV=768, chroma=False)
ctx = Context(0)
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
Y = clip["Y"]
outs = {}
for mode in ("exact", "linear_unguarded", "linear"):
    an = AMTAnalyzeLogo(ctx, logo, 0.35, mode=mode)
    o = torch.empty((N, 33), dtype=torch.float32, device=dev)
    an.analyze_device(Y, 8, o)
    torch.cuda.synchronize()
    outs[mode] = o.cpu().numpy()
    if mode == "linear":
        print("refined", an.last_refined(), "bounds", [an.error_bound(k, 8) for k in range(3)])
e = np.abs(outs["linear_unguarded"] - outs["exact"])
print("unguarded: max err", e.max(), "nan", np.isnan(outs["linear_unguarded"]).sum(), "frames with err>1e-4", (e.max(1) > 1e-4).sum())
bad = np.nonzero(e.max(1) > 1e-4)[0]
print("bad frames head", bad[:20], "per group k", [(e[:, 11*k:11*k+11].max()) for k in range(3)])
x = outs["exact"]
for k in range(3):
    srt = np.sort(x[:, 11*k:11*k+11], axis=1); m = srt[:, 1] - srt[:, 0]
    print("group", k, "margin min/median", m.min(), np.median(m), "frac < 0.006", (m < 0.006).mean())
print("guarded == exact bytes:", outs["linear"].tobytes() == outs["exact"].tobytes(), "max diff", np.abs(outs["linear"] - outs["exact"]).max())
