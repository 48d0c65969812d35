"""logo_eval_fused_kernel variants (build.py build_variant) on the bench's exact analysis: ms per 10 000-frame launch, and the records
against the default build's (bytes).  `--build` where hipcc is; run on the GPU box: python tools/fused_bench.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VARIANTS = {"bg_lds": ["AMT_FUSED_BG_LDS=1"]}
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, defs in VARIANTS.items():
        print(name, B.build_variant("fused_" + name, defs))
    sys.exit(0)
if "--child" in sys.argv:
    import hashlib
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo
    ctx = Context(0); dev = torch.device("cuda:0")
    out = {}
    for tag, (W, H, bits, pitch, N, X) in {"1440x1080_8bit": (1440, 1080, 8, 1472, 10000, 1120), "1920x1080_10bit": (1920, 1080, 10, 1920, 3000, 1600)}.items():
        data, alpha, alphaUV = S.make_logo(256, 128)
        Y = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, 64, dev, bits=bits, pitchY=pitch, chroma=False)["Y"]
        logo = Logo.from_planes(ctx, data, 256, 128, W, H, X, 64)
        an = AMTAnalyzeLogo(ctx, logo, 0.35, mode="exact")
        o = torch.zeros((N, 33), dtype=torch.float32, device=dev)
        an.analyze_device(Y, bits, o); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(4):
            an.analyze_device(Y, bits, o)
        torch.cuda.synchronize()
        c, ms = ctx.profile_report()["logo_eval_fused_kernel.analysis"]
        ctx.profile(False)
        out[tag] = {"ms": ms / c, "sha": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]}
        del Y
    print(json.dumps(out)); sys.exit(0)
res = {}
for name in ["default"] + list(VARIANTS):
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_fused_{name}.so")
        if not os.path.exists(so):
            continue
        env["AMTGPU_LIB"] = so
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    try:
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res[name] = {"error": (r.stderr or r.stdout)[-300:]}
    print(name, json.dumps(res[name]), flush=True)
