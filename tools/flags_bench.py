"""Compiler-option variants of single translation units on the bench's launches: per-kernel time (HIP events) and a hash of every output.
    python tools/flags_bench.py --build        (where hipcc is)
    python tools/flags_bench.py > gpurun_out/flags_bench.json     (on the GPU box)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
NOSLP = ["-fno-slp-vectorize"]
VARIANTS = {"noslp_all": {f: NOSLP for f in ("eval_linear_kernels.hip", "eval_pair_kernels.hip", "eval_fused_kernels.hip", "stats_kernels.hip", "erase_scan_kernels.hip")},
            "noslp_linear": {"eval_linear_kernels.hip": NOSLP}, "noslp_fused": {"eval_fused_kernels.hip": NOSLP},
            "lin_maxilp": {"eval_linear_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]},
            "lin_maxmem": {"eval_linear_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]},
            "pair_maxilp": {"eval_pair_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}}
DEFS = {"stats_copying": ["AMT_STATS_PINGPONG=0"],
        "lin_g6": ["AMT_LIN_G=6"], "lin_g5": ["AMT_LIN_G=5"], "lin_g4": ["AMT_LIN_G=4"], "lin_occ3": ["AMT_LIN_OCC=3", "AMT_LIN_OCC16=3"],
        "wg_plain_map": ["AMT_WG_PLAIN_MAP"],
        # ablations (wrong results by design: what a part of a kernel costs)
        "abl_lin_raw_sameframe": ["AMT_LIN_RAW_SAMEFRAME"], "abl_lin_no_flush": ["AMT_LIN_NO_FLUSH"], "abl_lin_no_fixup": ["AMT_LIN_NO_FIXUP"],
        "abl_lin_no_convert": ["AMT_LIN_NO_CONVERT"], "abl_lin_no_eval": ["AMT_LIN_NO_EVAL"],
        "abl_pair_raw_sameframe": ["AMT_PAIR_RAW_SAMEFRAME"], "abl_pair_no_sum": ["AMT_PAIR_NO_SUM"], "abl_pair_no_gather": ["AMT_PAIR_NO_GATHER"],
        "abl_pair_no_flush": ["AMT_PAIR_NO_FLUSH"], "abl_pair_no_convert": ["AMT_PAIR_NO_CONVERT"], "abl_pair_no_eval": ["AMT_PAIR_NO_EVAL"]}
for k in DEFS:
    VARIANTS.setdefault(k, {})
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, ff in VARIANTS.items():
        if only_build := [a for a in sys.argv[1:] if not a.startswith("--")]:
            if name not in only_build:
                continue
        print(name, B.build_variant("flags_" + name, DEFS.get(name, []), file_flags=ff))
    sys.exit(0)
if "--child" in sys.argv:
    import numpy as np, torch
    import amt_synth as S
    import bench
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, FrameStats, Logo, LogoFrame
    N = int(os.environ.get("AMT_FLAGS_N", "10000"))
    dev = torch.device("cuda:0")
    ctx = Context(0)
    logos_np, alpha, alphaUV = bench.make_logos()
    W, H, LW, LH, X, Y0 = bench.W, bench.H, bench.LW, bench.LH, bench.IMGX, bench.IMGY
    out = {}
    for bits, (Wk, Hk, Xk, pY, pUV, n) in {8: (W, H, X, bench.PITCH_Y, bench.PITCH_UV, N), 10: (1920, 1080, 1600, None, None, max(256, N // 4))}.items():
        kw = dict(pitchY=pY, pitchUV=pUV) if pY else {}
        clip = S.make_clip_torch(n, Wk, Hk, 0x5EED0002, alpha, alphaUV, Xk, Y0, dev, period=900, fade=12, bits=bits, **kw)
        dclip = DeviceClip(clip["Y"], clip["U"], clip["V"], Wk, Hk, bits)
        logos = [Logo.from_planes(ctx, d, LW, LH, Wk, Hk, Xk, Y0) for d in logos_np]
        lf = LogoFrame(ctx, logos, bench.MASKRATIO); lf.begin(Wk, Hk, bits, n)
        anl = AMTAnalyzeLogo(ctx, logos[0], bench.MASKRATIO, mode="linear")
        anx = AMTAnalyzeLogo(ctx, logos[0], bench.MASKRATIO, mode="exact")
        er = AMTEraseLogo(ctx, logos[0], "", 0, 16)
        fs = FrameStats(ctx, Wk, Hk, bits)
        d_l = torch.empty((n, 33), dtype=torch.float32, device=dev); d_x = torch.empty_like(d_l)
        d_s = torch.empty((n, 8), dtype=torch.int64, device=dev); d_f = torch.empty((n, 2), dtype=torch.float32, device=dev)
        def run():
            anl.analyze_device(dclip.Y, bits, d_l); anx.analyze_device(dclip.Y, bits, d_x); lf.scan_batch(dclip.Y, bits, 0, n); fs.run_device(dclip.Y, d_s)
        run(); torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        rep = {k: ms / c for k, (c, ms) in ctx.profile_report().items() if c}
        ctx.profile(False)
        er.calc_fades_device(d_x, n, out=d_f); er.erase_device_fades(dclip, d_f); torch.cuda.synchronize()
        h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
        ry, rx = slice(Y0, Y0 + LH), slice(Xk, Xk + LW)
        out[f"{bits}bit"] = {"ms": {k: round(v, 4) for k, v in rep.items()},
                             "sha": {"linear": h(d_l.cpu().numpy()), "exact": h(d_x.cpu().numpy()), "scan": h(lf.evalResults), "stats": h(d_s.cpu().numpy()),
                                     "erased": h(dclip.Y[:, ry, rx].contiguous().cpu().numpy())}}
        del clip, dclip, lf, anl, anx, er, fs
        torch.cuda.empty_cache()
    print(json.dumps(out))
    sys.exit(0)
res = {}
only = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in ["default"] + [v for v in VARIANTS if not only or v in only]:
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_flags_{name}.so")
        if not os.path.exists(so):
            continue
        env["AMTGPU_LIB"] = so
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
    try:
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res[name] = {"error": (r.stderr or r.stdout)[-400:]}
    print(name, json.dumps(res[name]), file=sys.stderr, flush=True)
print(json.dumps(res))
