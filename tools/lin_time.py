"""The linear kernel's time on the bench batch, a few repetitions (python tools/lin_time.py [frames] [bits])."""
import sys, numpy as np, torch, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import amt_synth as S, bench
from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda', 0); torch.cuda.init()
ctx = Context(0)
logos_np, alpha, alphaUV = bench.make_logos()
W, H, X = (bench.W, bench.H, bench.IMGX) if bits == 8 else (1920, 1080, 1600)
kw = dict(pitchY=bench.PITCH_Y, pitchUV=bench.PITCH_UV) if bits == 8 else {}
clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, bench.IMGY, dev, period=900, fade=12, chroma=False, bits=bits, **kw)
logo = Logo.from_planes(ctx, logos_np[0], bench.LW, bench.LH, W, H, X, bench.IMGY)
an = AMTAnalyzeLogo(ctx, logo, bench.MASKRATIO, mode="linear")
out = torch.empty((N, 33), dtype=torch.float32, device=dev)
an.analyze_device(clip["Y"], bits, out); torch.cuda.synchronize()
res = []
for rep in range(3):
    ctx.profile(False); ctx.profile(True)
    for _ in range(5): an.analyze_device(clip["Y"], bits, out)
    torch.cuda.synchronize()
    rep_ = {k: round(ms / c, 4) for k, (c, ms) in ctx.profile_report().items() if c}
    res.append(rep_.get("logo_eval_linear_kernel.analysis"))
print("bits", bits, "N", N, "linear ms", res, "refined", an.last_refined(), "sha", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12])
