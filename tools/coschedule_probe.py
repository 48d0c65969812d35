"""Do the frame metrics run BESIDE the scan kernel when their waves are small enough to share its compute units?
logo_eval_pair_kernel holds 3 x 136 VGPRs per SIMD and 135 KB of LDS per CU: 104 registers per SIMD and every wave slot but three are
left.  A frame_stats build with 8-byte columns (<= 96 VGPRs, no LDS) fits in there: launched on a second, lower-priority stream once
the scan is under way, its waves fill the holes -- if the dispatcher lets them.  `--build` where hipcc is; on the GPU box:
    python tools/coschedule_probe.py > gpurun_out/coschedule.json"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = {"slim96": ["AMT_STATS_COLB=8", "AMT_STATS_WAVES=5"],
            "slim96_lds25k": ["AMT_STATS_COLB=8", "AMT_STATS_WAVES=5", "AMT_STATS_LDS_BYTES=25600"],
            "slim96_lds12k": ["AMT_STATS_COLB=8", "AMT_STATS_WAVES=5", "AMT_STATS_LDS_BYTES=12800"],
            "slim96_lds8k": ["AMT_STATS_COLB=8", "AMT_STATS_WAVES=5", "AMT_STATS_LDS_BYTES=8192"],
            "slim56_lds12k": ["AMT_STATS_COLB=8", "AMT_STATS_ROWS=8", "AMT_STATS_WAVES=8", "AMT_STATS_LDS_BYTES=12800"]}
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    for name, defs in VARIANTS.items():
        print(name, B.build_variant("cosched_" + name, defs))
    sys.exit(0)
if "--child" in sys.argv:
    import torch
    import amt_synth as S
    import bench
    from amatsukaze_amd import AMTAnalyzeLogo, Context, FrameStats, Logo, LogoFrame
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda:0"); torch.cuda.init()
    N, W, H, P = 10000, 1440, 1080, 1472
    logos_np, alpha, alphaUV = bench.make_logos()
    Y = S.make_clip_torch(N, W, H, 0x5EED0002, alpha, alphaUV, 1120, 64, dev, pitchY=P, chroma=False, period=300, fade=12)["Y"]
    d_an = torch.empty((N, 33), dtype=torch.float32, device=dev)
    d_st = torch.zeros((N, 8), dtype=torch.int64, device=dev)
    d_st2 = torch.zeros((N, 8), dtype=torch.int64, device=dev)
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))        # lo = least priority (numerically largest), hi = greatest

    def stream(prio):
        st = C.c_void_p()
        assert hip.hipStreamCreateWithPriority(C.byref(st), 1, prio) == 0
        return st

    def make(st):
        ctx = Context(0)
        ctx.check(ctx.lib.amtgpu_context_set_stream(ctx.h, st))
        logos = [Logo.from_planes(ctx, d, bench.LW, bench.LH, W, H, 1120, 64) for d in logos_np]
        lf = LogoFrame(ctx, logos, 0.35); lf.begin(W, H, 8, N)
        return dict(ctx=ctx, st=st, lf=lf, an=AMTAnalyzeLogo(ctx, logos[0], 0.35, mode="linear"), fs=FrameStats(ctx, W, H, 8))

    sA, sB = stream(hi.value), stream(lo.value)
    A, B = make(sA), make(sB)
    ev = [C.c_void_p() for _ in range(3)]
    for e in ev:
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0         # hipEventDisableTiming
    rec = lambda e, s: hip.hipEventRecord(e, s)
    wait = lambda s, e: hip.hipStreamWaitEvent(s, e, 0)

    def timed(fn, reps=6):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    analysis = lambda o: o["an"].analyze_device(Y, 8, d_an)
    scan = lambda o: o["lf"].scan_batch(Y, 8, 0, N)
    stats = lambda o, out=d_st: o["fs"].run_device(Y, out)

    def beside_scan():            # analysis -> [scan || stats]: the metrics are enqueued behind the scan's launch
        analysis(A); rec(ev[0], sA); scan(A)
        wait(sB, ev[0]); stats(B, d_st2); rec(ev[1], sB); wait(sA, ev[1])

    def beside_scan_stats_first():
        analysis(A); rec(ev[0], sA)
        wait(sB, ev[0]); stats(B, d_st2); rec(ev[1], sB)
        scan(A); wait(sA, ev[1])

    def beside_all():             # [analysis -> scan] || stats
        rec(ev[0], sA); wait(sB, ev[0]); analysis(A); scan(A); stats(B, d_st2); rec(ev[1], sB); wait(sA, ev[1])

    out = {"stream_priorities": [hi.value, lo.value],
           "analysis_ms": timed(lambda: analysis(A)), "scan_ms": timed(lambda: scan(A)), "stats_ms": timed(lambda: stats(A)),
           "sequential_ms": timed(lambda: (analysis(A), scan(A), stats(A))),
           "analysis_then_scan_beside_stats_ms": timed(beside_scan),
           "analysis_then_stats_enqueued_first_ms": timed(beside_scan_stats_first),
           "everything_beside_stats_ms": timed(beside_all)}
    torch.cuda.synchronize()
    out["metrics_equal"] = bool(torch.equal(d_st, d_st2))
    print(json.dumps(out)); sys.exit(0)
for name in ["default"] + list(VARIANTS):
    env = dict(os.environ)
    if name != "default":
        so = os.path.join(ROOT, "amatsukaze_amd", f"libamt_gpu_cosched_{name}.so")
        if not os.path.exists(so):
            continue
        env["AMTGPU_LIB"] = so
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    try:
        print(json.dumps({name: json.loads(r.stdout.strip().splitlines()[-1])}), flush=True)
    except Exception:
        print(json.dumps({name: {"error": (r.stderr or r.stdout)[-600:]}}), flush=True)
