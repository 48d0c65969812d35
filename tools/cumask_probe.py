"""how hipExtStreamCreateWithCUMask's bits map to CUs on this device: frame metrics (per-CU bandwidth bound) under different 64-CU masks"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import amt_synth as S
from amatsukaze_amd import Context, FrameStats
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0"); torch.cuda.init()
N, W, H, P = 4000, 1440, 1080, 1472
Y = S.make_clip_torch(N, W, H, 0x5EED0002, None, None, 0, 0, dev, pitchY=P, chroma=False)["Y"]
d_st = torch.empty((N, 8), dtype=torch.int64, device=dev)
def masked(pred, nwords=8):
    words = (C.c_uint32 * nwords)()
    n = 0
    for i in range(32 * nwords):
        if pred(i):
            words[i // 32] |= 1 << (i % 32); n += 1
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), nwords, words)
    return (st if rc == 0 else None), n, rc
def run(st):
    ctx = Context(0)
    if st is not None:
        ctx.check(ctx.lib.amtgpu_context_set_stream(ctx.h, st))
    fs = FrameStats(ctx, W, H, 8)
    fs.run_device(Y, d_st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        fs.run_device(Y, d_st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 4 * 1e3
pats = {"all": None, "every4th": lambda i: i % 4 == 0, "first64": lambda i: i < 64, "first8_of_each32": lambda i: i % 32 < 8, "blocks_of_8_every4th": lambda i: (i // 8) % 4 == 0,
        "every3rd": lambda i: i % 3 == 0, "first128": lambda i: i < 128, "every2nd": lambda i: i % 2 == 0, "first32": lambda i: i < 32, "every8th": lambda i: i % 8 == 0,
        "first64_of_304bits": (lambda i: i < 64, 10)}
out = {}
for name, p in pats.items():
    if p is None:
        out[name] = {"cus": 256, "ms": run(None)}
    else:
        pred, nw = (p if isinstance(p, tuple) else (p, 8))
        st, n, rc = masked(pred, nw)
        out[name] = {"bits": n, "rc": rc, "ms": run(st) if st is not None else None}
    print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out))
