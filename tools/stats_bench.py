"""frame_stats_kernel variants (amatsukaze_amd/build.py build_variant: -DAMT_STATS_VG / _ROWS / _RUN) on the bench's shapes: time per
10 000-frame launch and equality of every record with the default build's.  Build where hipcc is (`--build`), run on the GPU box:
    python tools/stats_bench.py > gpurun_out/stats_bench.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VARIANTS = {"r03_form": ["AMT_STATS_LEAN=0"], "lean_no_nt": ["AMT_STATS_NT=0"], "lean_nt3": ["AMT_STATS_NT=3"], "lean_vg4": ["AMT_STATS_VG=4"],
            "lean_8B_columns": ["AMT_STATS_COLB=8"], "lean_8B_prefetch": ["AMT_STATS_COLB=8", "AMT_STATS_PREFETCH=1"],
            "lean_8rows_prefetch": ["AMT_STATS_ROWS=8", "AMT_STATS_PREFETCH=1"], "lean_run64": ["AMT_STATS_RUN=64"],
            # taller tiles: the halo rows are 2 / ROWS of the traffic
            "rows24": ["AMT_STATS_ROWS=24"], "rows24_8B": ["AMT_STATS_ROWS=24", "AMT_STATS_COLB=8"], "rows32_8B": ["AMT_STATS_ROWS=32", "AMT_STATS_COLB=8"],
            "rows32_8B_run64": ["AMT_STATS_ROWS=32", "AMT_STATS_COLB=8", "AMT_STATS_RUN=64"],
            "w3_r16_r12": ["AMT_STATS_WAVES=3", "AMT_STATS_ROWS8=16", "AMT_STATS_ROWS=12"], "w3_r12_r12": ["AMT_STATS_WAVES=3", "AMT_STATS_ROWS8=12", "AMT_STATS_ROWS=12"],
            "w3_r16_r8": ["AMT_STATS_WAVES=3", "AMT_STATS_ROWS8=16", "AMT_STATS_ROWS=8"], "w4_r8_r8": ["AMT_STATS_WAVES=4", "AMT_STATS_ROWS8=8", "AMT_STATS_ROWS=8"],
            "w3_r14_r10": ["AMT_STATS_WAVES=3", "AMT_STATS_ROWS8=14", "AMT_STATS_ROWS=10"],
            "deal2": ["AMT_STATS_DEAL=2"], "deal2_rows16": ["AMT_STATS_DEAL=2", "AMT_STATS_ROWS8=16"], "deal2_rows20": ["AMT_STATS_DEAL=2", "AMT_STATS_ROWS8=20"],
            "deal1": ["AMT_STATS_DEAL=1"], "deal1_no_nt": ["AMT_STATS_DEAL=1", "AMT_STATS_NT=0"], "deal1_run64": ["AMT_STATS_DEAL=1", "AMT_STATS_RUN=64"],
            "deal1_rows16": ["AMT_STATS_DEAL=1", "AMT_STATS_ROWS8=16"], "run64": ["AMT_STATS_RUN=64"], "no_nt": ["AMT_STATS_NT=0"],
            "rows8bit_16": ["AMT_STATS_ROWS8=16"], "rows8bit_20": ["AMT_STATS_ROWS8=20"], "rows8bit_28": ["AMT_STATS_ROWS8=28"]}
ONLY = [a for a in sys.argv[1:] if not a.startswith("--")]
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in ONLY}
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, defs in VARIANTS.items():
        if os.path.exists(os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_stats_{name}.so")) and "--force" not in sys.argv:
            continue
        print(name, B.build_variant("stats_" + name, defs))
    sys.exit(0)
if "--child" in sys.argv:
    import hashlib, time
    import torch
    import amt_synth as S
    from amatsukaze_amd import Context, FrameStats
    ctx = Context(0)
    dev = torch.device("cuda:0")
    ncu = int(os.environ.get("AMT_STATS_MASK_CUS", "0"))
    if ncu:                                  # the kernel on the first `ncu` CUs only (hipExtStreamCreateWithCUMask, contiguous bits)
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        words = (C.c_uint32 * 8)()
        for i in range(ncu):
            words[i // 32] |= 1 << (i % 32)
        st = C.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words) == 0
        ctx.check(ctx.lib.amtgpu_context_set_stream(ctx.h, st))
    out = {}
    shapes = {"1440x1080_8bit": (1440, 1080, 8, 1472, 10000), "1920x1080_8bit": (1920, 1080, 8, 1920, 6000), "1920x1080_10bit": (1920, 1080, 10, 1920, 3000)}
    if os.environ.get("AMT_STATS_FRAMES"):        # (counter passes: fewer frames)
        shapes = {k: v[:4] + (min(v[4], int(os.environ["AMT_STATS_FRAMES"])),) for k, v in shapes.items()}
    if os.environ.get("AMT_STATS_PITCHES"):      # the same 1440-wide frames at other pitches: 1472 = AviSynth's 64-byte alignment (odd rows start mid-line)
        shapes = {f"1440x1080_8bit_pitch{p}": (1440, 1080, 8, p, int(os.environ.get("AMT_STATS_FRAMES", "10000"))) for p in (1472, 1536, 1440, 1408 + 128 + 64)}
    for tag, (W, H, bits, pitch, N) in shapes.items():
        Y = S.make_clip_torch(N, W, H, 0x5EED0002, None, None, 0, 0, dev, bits=bits, pitchY=pitch, chroma=False)["Y"]
        fs = FrameStats(ctx, W, H, bits)
        o = torch.zeros((N, 8), dtype=torch.int64, device=dev)
        fs.run_device(Y, o)
        torch.cuda.synchronize()
        ctx.profile(True)
        for _ in range(8):
            fs.run_device(Y, o)
        torch.cuda.synchronize()
        c, ms = ctx.profile_report()["frame_stats_kernel"]
        ctx.profile(False)
        es = 1 if bits <= 8 else 2
        torch.cuda.synchronize()
        out[tag] = {"ms": ms / c, "alg_TBs": W * H * es * N / (ms / c * 1e-3) / 1e12, "sha": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]}
        if ncu:
            out[tag]["GBs_per_cu"] = W * H * es * N / (ms / c * 1e-3) / 1e9 / ncu
        del Y
    print(json.dumps(out))
    sys.exit(0)
res = {}
for name in ["default"] + list(VARIANTS):
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_stats_{name}.so")
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
