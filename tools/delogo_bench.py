"""delogo_kernel variants (-DAMT_DELOGO_ROWS / _FRAMES) on the bench's shape: 10 000 frames of 1440x1080 8-bit, the fades of the bench's clip
(54 % of the frames have a non-zero fade), time per launch and a hash of the erased rectangles.
    python tools/delogo_bench.py --build          (where hipcc is)
    python tools/delogo_bench.py > gpurun_out/delogo_bench.json     (on the GPU box)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VARIANTS = {"r32_f8": ["AMT_DELOGO_ROWS=32"], "r64_f8": ["AMT_DELOGO_ROWS=64"], "r8_f8": ["AMT_DELOGO_ROWS=8"], "r16_f16": ["AMT_DELOGO_FRAMES=16"],
            "r16_f4": ["AMT_DELOGO_FRAMES=4"], "r32_f4": ["AMT_DELOGO_ROWS=32", "AMT_DELOGO_FRAMES=4"], "r64_f4": ["AMT_DELOGO_ROWS=64", "AMT_DELOGO_FRAMES=4"],
            "r256_f2": ["AMT_DELOGO_ROWS=256", "AMT_DELOGO_FRAMES=2"], "r256_f4": ["AMT_DELOGO_ROWS=256", "AMT_DELOGO_FRAMES=4"]}
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, defs in VARIANTS.items():
        print(name, B.build_variant("delogo_" + name, defs))
    sys.exit(0)
if "--child" in sys.argv:
    import numpy as np, torch
    import amt_synth as S
    import bench
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, Logo
    N = int(os.environ.get("AMT_DELOGO_N", "10000"))
    dev = torch.device("cuda:0")
    ctx = Context(0)
    logos_np, alpha, alphaUV = bench.make_logos()
    W, H, LW, LH, X, Y0 = bench.W, bench.H, bench.LW, bench.LH, bench.IMGX, bench.IMGY
    clip = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, Y0, dev, period=900, fade=12, pitchY=bench.PITCH_Y, pitchUV=bench.PITCH_UV)
    dclip = DeviceClip(clip["Y"], clip["U"], clip["V"], W, H, 8)
    logo = Logo.from_planes(ctx, logos_np[0], LW, LH, W, H, X, Y0)
    an = AMTAnalyzeLogo(ctx, logo, bench.MASKRATIO, mode="linear")
    er = AMTEraseLogo(ctx, logo, "", 0, 16)
    d_an = torch.empty((N, 33), dtype=torch.float32, device=dev)
    d_f = torch.empty((N, 2), dtype=torch.float32, device=dev)
    an.analyze_device(dclip.Y, 8, d_an)
    er.calc_fades_device(d_an, N, out=d_f)
    keep = [t.clone() for t in (dclip.Y[:, Y0:Y0 + LH, X:X + LW], dclip.U[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2], dclip.V[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2])]
    def restore():
        dclip.Y[:, Y0:Y0 + LH, X:X + LW] = keep[0]; dclip.U[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2] = keep[1]; dclip.V[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2] = keep[2]
    er.erase_device_fades(dclip, d_f); restore(); torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(10):
        er.erase_device_fades(dclip, d_f)
        restore()
    torch.cuda.synchronize()
    c, ms = ctx.profile_report()["delogo_kernel"]
    ctx.profile(False)
    er.erase_device_fades(dclip, d_f); torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (dclip.Y[:, Y0:Y0 + LH, X:X + LW], dclip.U[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2], dclip.V[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2]):
        h.update(t.contiguous().cpu().numpy().tobytes())
    share = float((d_f.abs().sum(dim=1) != 0).float().mean())
    byts = 2 * (LW * LH + 2 * (LW // 2) * (LH // 2)) * share * N
    print(json.dumps({"ms": ms / c, "GBs": byts / (ms / c * 1e-3) / 1e9, "frac_hbm": byts / (ms / c * 1e-3) / 8e12, "nonzero_share": share, "sha": h.hexdigest()[:16]}))
    sys.exit(0)
res = {}
for name in ["default"] + list(VARIANTS):
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_delogo_{name}.so")
        if not os.path.exists(so):
            continue
        env["AMTGPU_LIB"] = so
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    try:
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res[name] = {"error": (r.stderr or r.stdout)[-300:]}
    print(name, json.dumps(res[name]), file=sys.stderr, flush=True)
print(json.dumps(res))
