#!/usr/bin/env python3
"""diagnostic: unguarded linear vs exact on a small shape: python tools/diag_linear2.py LW LH maskratio"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, DeviceClip, Logo
LW, LH, mr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
W, H, X, Y0, N = 352, 240, 222, 18, 9
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, period=4, fade=2, chroma=False)
ctx = Context(0)
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
outs = {}
for mode in ("exact", "linear_unguarded"):
    an = AMTAnalyzeLogo(ctx, logo, mr, mode=mode)
    o = torch.empty((N, 33), dtype=torch.float32, device=dev)
    an.analyze_device(clip["Y"], 8, o)
    torch.cuda.synchronize()
    outs[mode] = o.cpu().numpy()
e = np.abs(outs["linear_unguarded"] - outs["exact"])
print(f"{LW}x{LH} mr={mr}: max err per group", [float(e[:, 11*k:11*k+11].max()) for k in range(3)], "counts", [logo.mask_tables(k, mr)["count"] for k in range(3)])
