"""Can the HBM-bound frame metrics run BESIDE the VALU-bound logo kernels?  The two do not co-schedule as launched (the logo kernels take
every CU's registers / LDS), so the device is partitioned: the frame metrics on a stream created with a CU mask (a fraction of every XCD's
CUs), the analysis + scan on the complementary mask.  Reports each alone (full device, masked) and both together.
Run on the GPU box from the repo root:  python tools/overlap_probe.py > gpurun_out/overlap_probe.json"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import amt_synth as S
from amatsukaze_amd import AMTAnalyzeLogo, Context, FrameStats, Logo, LogoFrame

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init()
N, W, H, P = 10000, 1440, 1080, 1472
data, alpha, alphaUV = S.make_logo(256, 128)
Y = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, 1120, 64, dev, pitchY=P, chroma=False)["Y"]
d_an = torch.empty((N, 33), dtype=torch.float32, device=dev)
d_st = torch.empty((N, 8), dtype=torch.int64, device=dev)
torch.cuda.synchronize()


def masked_stream(pred):
    """stream whose kernels run only on CUs i with pred(i) (256 CUs = 8 words)"""
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st


def make(stream):
    ctx = Context(0)
    if stream is not None:
        ctx.check(ctx.lib.amtgpu_context_set_stream(ctx.h, stream))
    logos = [Logo.from_planes(ctx, d, 256, 128, W, H, 1120, 64) for d in (data, S.make_logo(256, 128, seed=0x10600002, strength=0.5)[0],
                                                                         S.make_logo(256, 128, seed=0x10600003, strength=0.8)[0])]
    lf = LogoFrame(ctx, logos, 0.35); lf.begin(W, H, 8, N)
    an = AMTAnalyzeLogo(ctx, logos[0], 0.35, mode="linear")
    fs = FrameStats(ctx, W, H, 8)
    return dict(ctx=ctx, lf=lf, an=an, fs=fs, logos=logos)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
full = make(None)
logo = lambda o: (o["an"].analyze_device(Y, 8, d_an), o["lf"].scan_batch(Y, 8, 0, N))
stats = lambda o: o["fs"].run_device(Y, d_st)
out["full_device"] = {"logo_ms": timed(lambda: logo(full)), "stats_ms": timed(lambda: stats(full)), "sequential_ms": timed(lambda: (logo(full), stats(full)))}
print("full", out["full_device"], file=sys.stderr, flush=True)
# mask bits interleave over the 8 XCDs (bit i -> XCD i % 8): a contiguous range of F bits is F / 8 CUs of every XCD (tools/cumask_probe.py)
for F in (48, 56, 64, 72, 80):       # the frame metrics get the first F CUs, the logo kernels the other 256 - F
    a = make(masked_stream(lambda i: i >= F))
    b = make(masked_stream(lambda i: i < F))
    r = {"stats_cus": F, "logo_alone_ms": timed(lambda: logo(a)), "stats_alone_ms": timed(lambda: stats(b)),
         "together_ms": timed(lambda: (logo(a), stats(b))), "together_stats_first_ms": timed(lambda: (stats(b), logo(a)))}
    out[f"stats_on_{F}_cus"] = r
    print(F, r, file=sys.stderr, flush=True)
print(json.dumps(out))
