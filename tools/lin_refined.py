"""How many frames of the bench batch the guard hands to the exact kernel, and the linear kernel's time
(python tools/lin_refined.py [frames [queue entries ...]])."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import amt_synth as S, bench
from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
QS = [int(a) for a in sys.argv[2:]] or [64, 128, 192, 256, 384, 512, 768, 1024]
dev = torch.device('cuda', 0); torch.cuda.init()
ctx = Context(0)
logos_np, alpha, alphaUV = bench.make_logos()
clip = S.make_clip_torch(N, bench.W, bench.H, 0x5EED0002, alpha, alphaUV, bench.IMGX, bench.IMGY, dev, period=900, fade=12, pitchY=bench.PITCH_Y, pitchUV=bench.PITCH_UV, chroma=False)
logo = Logo.from_planes(ctx, logos_np[0], bench.LW, bench.LH, bench.W, bench.H, bench.IMGX, bench.IMGY)
an = AMTAnalyzeLogo(ctx, logo, bench.MASKRATIO, mode="linear")
out = torch.empty((N, 33), dtype=torch.float32, device=dev)
an.analyze_device(clip["Y"], 8, out); torch.cuda.synchronize()
ctx.profile(True)
for _ in range(3): an.analyze_device(clip["Y"], 8, out)
torch.cuda.synchronize()
print("refined", an.last_refined(), {k: round(ms / c, 4) for k, (c, ms) in ctx.profile_report().items() if c})
for q in QS:
    an.set_fixup_queue(q)
    an.analyze_device(clip["Y"], 8, out); torch.cuda.synchronize()
    ctx.profile(False); ctx.profile(True)
    for _ in range(3): an.analyze_device(clip["Y"], 8, out)
    torch.cuda.synchronize()
    print("queue", q, "refined", an.last_refined(), {k: round(ms / c, 4) for k, (c, ms) in ctx.profile_report().items() if c})
