"""Tile-kernel variants (build.py build_variant) on the bench's scan (logo_eval_pair_kernel, 3 logos) and linear analysis
(logo_eval_linear_kernel): ms per 10 000-frame launch and a hash of the records (the scan's must equal the default build's: bit-exact
kernel; the linear mode's may differ in the last bits when its summation order changes).  `--build` where hipcc is; run on the GPU
box: python tools/tile_bench.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = {
    "w15_g5_occ4": ["AMT_PAIR_OCC=4", "AMT_TILE_WAVES=15", "AMT_TILE_G=5"],
    "w7_g6_occ4": ["AMT_PAIR_OCC=4", "AMT_TILE_WAVES=7", "AMT_TILE_G=6"],
    "w7_g8": ["AMT_TILE_WAVES=7", "AMT_TILE_G=8"],
    # the 16-bit linear kernel runs two workgroups per CU: room for more frames per workgroup
    "lin16_g8": ["AMT_LIN_G16=8"], "lin16_g8_1k": ["AMT_LIN_G16=8", "AMT_LIN_WGS_MIN16=1024"],
    "lin16_g10_1k": ["AMT_LIN_G16=10", "AMT_LIN_WGS_MIN16=1024"], "lin16_g12_1k": ["AMT_LIN_G16=12", "AMT_LIN_WGS_MIN16=1024"],
    "lin16_g4": ["AMT_LIN_G16=4"],
    # listed re-evaluation of the linear mode's guard: fades per workgroup (0 = all eleven in one workgroup)
    "listed0": ["AMT_LISTED_FADE_CHUNK=0"], "listed2": ["AMT_LISTED_FADE_CHUNK=2"], "listed4": ["AMT_LISTED_FADE_CHUNK=4"],
}
ONLY = [a for a in sys.argv[1:] if not a.startswith("--")]
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in ONLY}
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, defs in VARIANTS.items():
        print(name, B.build_variant("tile_" + name, defs))
    sys.exit(0)
if "--child" in sys.argv:
    import hashlib
    import torch
    import amt_synth as S
    import bench
    from amatsukaze_amd import AMTAnalyzeLogo, Context, Logo, LogoFrame
    ctx = Context(0); dev = torch.device("cuda:0")
    out = {}
    logos_np, alpha, alphaUV = bench.make_logos()
    for tag, (W, H, bits, pitch, N, X) in {"1440x1080_8bit": (1440, 1080, 8, 1472, 10000, 1120), "1920x1080_10bit": (1920, 1080, 10, 1920, 4096, 1600)}.items():
        Y = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, X, 64, dev, bits=bits, pitchY=pitch, chroma=False, period=300, fade=12)["Y"]
        logos = [Logo.from_planes(ctx, d, bench.LW, bench.LH, W, H, X, 64) for d in logos_np]
        lf = LogoFrame(ctx, logos, 0.35); lf.begin(W, H, bits, N)
        an = AMTAnalyzeLogo(ctx, logos[0], 0.35, mode="linear")
        o = torch.zeros((N, 33), dtype=torch.float32, device=dev)
        an.analyze_device(Y, bits, o); lf.scan_batch(Y, bits, 0, N); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(4):
            an.analyze_device(Y, bits, o); lf.scan_batch(Y, bits, 0, N)
        torch.cuda.synchronize()
        rep = ctx.profile_report(); ctx.profile(False)
        r = {(k.split(".")[0] if "refine" not in k else "refine"): ms / c for k, (c, ms) in rep.items() if c and ("pair" in k or "linear" in k or "refine" in k)}
        r["scan_sha"] = hashlib.sha256(lf.evalResults.tobytes()).hexdigest()[:16]
        r["analysis_sha"] = hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]
        out[tag] = r
        del Y
    print(json.dumps(out)); sys.exit(0)
res = {}
names = ["default"] + list(VARIANTS)
for name in names:
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_tile_{name}.so")
        if not os.path.exists(so):
            continue
        env["AMTGPU_LIB"] = so
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    try:
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res[name] = {"error": (r.stderr or r.stdout)[-400:]}
    print(name, json.dumps(res[name]), flush=True)
