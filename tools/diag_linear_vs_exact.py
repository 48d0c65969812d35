"""The unguarded linear analysis kernel against the exact GPU kernel on a small clip, frame by frame: max |difference|, which frames
and which of the 33 scores exceed 1e-4.  What located this round's two bugs (an inline-asm row selector read too early by the MFMA
behind it; v_readlane of a lane the divergent branch had switched off).  Honours AMTGPU_LIB (instrumented builds).
    python tools/diag_linear_vs_exact.py [frames]        on the GPU box, from the repo root"""
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
W, H, LW, LH, X, Y0, N = 352, 240, 96, 48, 224, 18, int(sys.argv[1]) if len(sys.argv)>1 else 2801
dev=torch.device('cuda',0)
ctx=Context(0)
data, alpha, alphaUV = S.make_logo(LW, LH)
clip = S.make_clip_torch(N, W, H, 0x5EED0009, alpha, alphaUV, X, Y0, dev, period=40, fade=6, chroma=False)
Yd=clip["Y"]
logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
out = torch.empty((N, 33), dtype=torch.float32, device=dev)
AMTAnalyzeLogo(ctx, logo, 0.35).analyze_device(Yd, 8, out)
torch.cuda.synchronize()
an = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear_unguarded")
out2 = torch.empty((N, 33), dtype=torch.float32, device=dev)
an.analyze_device(Yd, 8, out2); torch.cuda.synchronize()
d=(out2-out).abs().cpu().numpy()
print("max", d.max(), "bad frames", (d.max(1)>1e-4).sum(), "of", N)
bad=np.nonzero(d.max(1)>1e-4)[0]
print(bad[:40])
for fr in bad[:3]:
    print(fr, np.round(out[fr].cpu().numpy(),3)); print(np.round(out2[fr].cpu().numpy(),3))
print("per column bad:", (d>1e-4).sum(0))
