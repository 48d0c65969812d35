"""The unguarded linear analysis against the exact GPU kernel over several logo shapes, strengths and bit depths (worst relative error
with a floor of 1).  python tools/stress_linear.py   on the GPU box, from the repo root.  Round 3: worst 4.4e-6."""
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
dev=torch.device('cuda',0); torch.cuda.init()
ctx=Context(0)
worst=0
for i,(W,H,LW,LH,X,Y0,N,bits) in enumerate([(352,240,96,48,224,18,900,8),(720,480,130,64,500,30,600,8),(1440,1080,256,128,1120,64,400,8),(640,360,64,96,100,100,700,10),(1920,1080,320,100,1500,40,300,8),(352,240,98,50,222,16,800,12)]):
    for strength in (0.3, 0.8, 1.0):
        data, alpha, alphaUV = S.make_logo(LW, LH, seed=0x10600001+i, strength=strength)
        clip = S.make_clip_torch(N, W, H, 0x5EED0100+i, alpha, alphaUV, X, Y0, dev, period=37, fade=5, chroma=False, bits=bits)
        Yd=clip["Y"]
        logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
        out = torch.empty((N, 33), dtype=torch.float32, device=dev); out2=torch.empty_like(out)
        AMTAnalyzeLogo(ctx, logo, 0.35).analyze_device(Yd, bits, out)
        AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear_unguarded").analyze_device(Yd, bits, out2)
        torch.cuda.synchronize()
        ref=out.cpu().numpy(); d=np.abs(out2.cpu().numpy()-ref)/np.maximum(1.0,np.abs(ref))
        worst=max(worst,d.max())
        print(i, bits, strength, "max rel err", d.max(), "nan", np.isnan(d).sum())
print("WORST", worst)
