"""The scan (pair) kernel's time on the bench batch (3 logos), a few repetitions, and a hash of the records (python tools/scan_time.py [frames] [bits])."""
import sys, numpy as np, torch, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import amt_synth as S, bench
from amatsukaze_amd import Context, Logo, LogoFrame
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda', 0); torch.cuda.init()
ctx = Context(0)
logos_np, alpha, alphaUV = bench.make_logos()
W, H, X = (bench.W, bench.H, bench.IMGX) if bits == 8 else (1920, 1080, 1600)
kw = dict(pitchY=bench.PITCH_Y, pitchUV=bench.PITCH_UV) if bits == 8 else {}
clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, bench.IMGY, dev, period=900, fade=12, chroma=False, bits=bits, **kw)
logos = [Logo.from_planes(ctx, d, bench.LW, bench.LH, W, H, X, bench.IMGY) for d in logos_np]
lf = LogoFrame(ctx, logos, bench.MASKRATIO); lf.begin(W, H, bits, N)
lf.scan_batch(clip["Y"], bits, 0, N); torch.cuda.synchronize()
res = []
for rep in range(3):
    ctx.profile(False); ctx.profile(True)
    for _ in range(5): lf.scan_batch(clip["Y"], bits, 0, N)
    torch.cuda.synchronize()
    r = {k: round(ms / c, 4) for k, (c, ms) in ctx.profile_report().items() if c}
    res.append(r.get("logo_eval_pair_kernel.scan"))
print("bits", bits, "N", N, "scan ms", res, "sha", hashlib.sha256(np.ascontiguousarray(lf.evalResults).tobytes()).hexdigest()[:12])
