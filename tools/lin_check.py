"""The unguarded linear analysis of one small clip against the exact GPU kernel (max abs error, NaN count): a quick health check of a
library variant (AMTGPU_LIB).  python tools/lin_check.py [bits]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0); torch.cuda.init()
ctx = Context(0)
W, H, LW, LH, X, Y0, N = 1440, 1080, 256, 128, 1120, 64, 700
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, period=37, fade=5, chroma=False, bits=bits)
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
out = torch.empty((N, 33), dtype=torch.float32, device=dev); out2 = torch.empty_like(out)
AMTAnalyzeLogo(ctx, logo, 0.35).analyze_device(clip["Y"], bits, out)
AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear_unguarded").analyze_device(clip["Y"], bits, out2)
torch.cuda.synchronize()
ref = out.cpu().numpy(); lin = out2.cpu().numpy()
d = np.abs(lin - ref)
print("bits", bits, "max abs err", float(np.nanmax(d)), "nan", int(np.isnan(lin).sum()), "bad frames", int((np.nan_to_num(d, nan=1.0).max(1) > 1e-4).sum()), "of", N,
      "per group", [float(np.nanmax(d[:, 11 * k:11 * k + 11])) for k in range(3)])
