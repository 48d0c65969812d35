"""BASELINE configs[4] end to end: CMAnalyze's all-frames logo scan + AMTAnalyzeLogo -> CalcFade -> AMTEraseLogo + the CM / KFM frame
metrics and their decisions over a long 1920x1080i 10-bit stream, frames sharded over the ranks of one node.

Imported by bench.py (`--workload e2e10`, and attached to the default line as configs.e2e_1080p10): measurement code, not product.

The stream is a function of the absolute frame index (tools/amt_synth.py), so every rank generates its own contiguous range
[f0, f1) on its GPU, CHUNK frames at a time, the way a rank would decode its own part of the transport stream; 8 more frames
either side of a chunk are generated as well -- the analysis halo CalcFade2 reads (LogoScan.hpp:1265-1285, n-8 .. n+8) and, in
it, the frame before the chunk that the frame metrics compare the chunk's first frame with.  Per chunk, on the device, in
stream order and without a host synchronise:

    AMTAnalyzeLogo over chunk + halo  ->  CalcFade (amtgpu_erase_calc_fades_device)  ->  LogoFrame scan (3 logos) and frame
    metrics of the chunk's own frames  ->  AMTEraseLogo in place with the device-resident fades.

Only these calls are inside the timed region (inputs resident in HBM when it starts; generation is fenced off).  After the last
chunk come the exchanges of DESIGN.md section 8 -- scan records (amtgpu_logoframe_allgather_results), metric records
(amtgpu_framestats_allgather), fades -- and the decisions, replicated on every rank: selectLogo + logoframe text, cadence per
frame, scene changes.  `decisions_sha256` hashes {fades, logoframe text, cadence + phase per frame, scene-change list, per-frame
checksum of the erased rectangles}: it must not depend on the number of ranks or on the chunk size.  Sampled blocks (clip start,
across every rank boundary, clip end) are compared with the CPU oracle as bytes by the ranks that own them.
"""
from __future__ import annotations

import hashlib
import os
import time

import numpy as np

FULL_FRAMES = 431568            # 4 h at 30000/1001 fps (BASELINE configs[4])
SHARE_FRAMES = FULL_FRAMES // 8  # one GPU's share of it at N = 8: the default clip
W, H, BITS = 1920, 1080, 10
LX, LY = 1600, 64               # logo rectangle origin
SEG = 1800                      # cadence segments: 24p / 30i / 30p alternating (the generator's labels)
HALO = 8


def _generate(S, torch, dev, a0, a1, alpha, alphaUV, seed=0x5EED0006):
    """frames [a0, a1) of the stream as (Y, U, V) int16 tensors; cadence changes every SEG frames"""
    order = ("24p", "30i", "30p")
    Ys, Us, Vs = [], [], []
    n = a0
    while n < a1:
        e = min(a1, (n // SEG + 1) * SEG)
        c = S.make_clip_torch(e - n, W, H, seed, alpha, alphaUV, LX, LY, dev, bits=BITS, period=300, fade=12, start=n,
                              cadence=order[(n // SEG) % 3], noise="soft")
        Ys.append(c["Y"]); Us.append(c["U"]); Vs.append(c["V"])
        n = e
    cat = lambda l: l[0] if len(l) == 1 else torch.cat(l, 0)
    return cat(Ys), cat(Us), cat(Vs)


def run(E, nt=SHARE_FRAMES, chunk=4096, verify=True, mode="linear", scaling="strong", verify_budget_s=300.0):
    """E: namespace with torch, dist, rank, world, dev, ctx, logos_np, alpha, alphaUV, fence(), max_over_ranks(x), OracleLogos, rccl"""
    torch, dist, rank, world, dev, ctx = E.torch, E.dist, E.rank, E.world, E.dev, E.ctx
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, DeviceClip, FrameStats, Logo, LogoFrame
    from amatsukaze_amd import sharding as SH
    LW, LH = E.alpha.shape[1], E.alpha.shape[0]
    logos = [Logo.from_planes(ctx, d, LW, LH, W, H, LX, LY) for d in E.logos_np]
    f0, f1 = SH.shard_range(nt, rank, world)
    nloc = f1 - f0
    lf = LogoFrame(ctx, logos, E.maskratio)
    lf.begin(W, H, BITS, nt)
    an = AMTAnalyzeLogo(ctx, logos[0], E.maskratio, mode=mode)
    er = AMTEraseLogo(ctx, logos[0], "", 0, 16)
    st = FrameStats(ctx, W, H, BITS)
    coll = SH.TorchCollectives() if world > 1 else None
    d_an = torch.empty((chunk + 2 * HALO, 33), dtype=torch.float32, device=dev)
    d_fades = torch.empty((max(1, nloc), 2), dtype=torch.float32, device=dev)
    d_stats = torch.empty((max(1, nloc), 8), dtype=torch.int64, device=dev)
    d_ansave = torch.empty((max(1, nloc), 33), dtype=torch.float32, device=dev)      # kept for the sampled oracle comparison only
    erased_sum = torch.zeros(max(1, nloc), dtype=torch.int64, device=dev)

    # probe blocks for the oracle comparison: 40 frames at the clip start, across every rank boundary (or chunk boundaries at
    # world 1), at the clip end; a frame is checked by the rank that owns it
    PB = 40
    bounds = [SH.shard_range(nt, r, world)[0] for r in range(1, world)] or [min(nt, chunk), min(nt, 2 * chunk)]
    # (the end block starts on a multiple of 8: CalcFade2 addresses the analyze clip as (k >> 3, k & 7), LogoScan.hpp:1271-1276, and
    # clamps the frame number at the clip's end -- the oracle run on the block as a clip of its own sees the same groups of 8 only then)
    if nt >= 2 * PB:
        starts = sorted({0, (nt - PB) // 8 * 8} | {max(0, min((nt - PB) // 8 * 8, b - PB // 2)) for b in bounds})
        probes = [(p, (nt - p) if p == (nt - PB) // 8 * 8 else PB) for p in starts]
    else:
        probes = [(0, nt)]
    stash = {}                                         # frame index -> erased (Y, U, V) of probe frames this rank owns
    is_probe = lambda n: any(p <= n < p + k for p, k in probes)
    # ... and EVERY frame of the rank's share, chunk by chunk right behind the chunk's timed calls (tools/bench_verify.py: the oracle on this
    # rank's part of the host's cores, a pristine copy of the chunk + halo kept on the device for it).  `verify_budget_s` bounds the wall
    # time a rank spends in it (the whole 53 946-frame share of one GPU takes about a minute of a 256-core host; eight ranks sharing that
    # host take eight times as long for the 4-hour stream): once it is used up the remaining chunks are left to the probe blocks, and the
    # line says how many frames were compared.
    import bench_verify as BV
    vstate = {"spent": 0.0, "frames": 0, "max_an": 0.0, "ok": {k: True for k in ("scan", "analysis", "fades", "erase", "metrics")}, "mismatches": [],
              "stage": {}, "ol": None, "skipped_chunks": 0}
    vthreads = max(1, BV.host_threads() // max(1, world))
    # warm-up, untimed (the headline's warm-up steps): the first use of every object builds its tile plans and tables, uploads them and
    # loads the 16-bit kernels -- per-logo set-up like the reference's constructors (CreateLogoMask, LogoScan.hpp:1164-1201), ~9 ms.
    # Everything it writes is overwritten by the rank's first chunk.
    if nloc > 0:
        wn = min(nloc, 32, chunk)
        wa0, wa1 = max(0, f0 - HALO), min(nt, f0 + wn + HALO)
        Yw, Uw, Vw = _generate(S, torch, dev, wa0, wa1, E.alpha, E.alphaUV)
        wclip = DeviceClip(Yw[f0 - wa0:f0 - wa0 + wn], Uw[f0 - wa0:f0 - wa0 + wn], Vw[f0 - wa0:f0 - wa0 + wn], W, H, BITS)
        an.analyze_device(Yw, BITS, d_an[:wa1 - wa0])
        er.calc_fades_device(d_an[:wa1 - wa0], nt, f0, wn, analysis_first=wa0, out=d_fades[:wn])
        lf.scan_batch(wclip.Y, BITS, f0, wn)
        st.run_device(wclip.Y, d_stats[:wn], prevY=Yw[f0 - wa0 - 1] if f0 > 0 else None)
        er.erase_device_fades(wclip, d_fades[:wn])
        torch.cuda.synchronize()
        del Yw, Uw, Vw, wclip
    gen_s, timed_s, kern = 0.0, 0.0, {}
    # AMT_E2E_PHASES=1: a diagnostic run that fences every call of the chunk loop and reports where the chunk time goes (its `value`
    # then carries the fences -- not a bench figure)
    phases = {} if os.environ.get("AMT_E2E_PHASES") else None
    lap_t = [0.0]

    def lap(name):
        if phases is None:
            return
        torch.cuda.synchronize()
        t = time.perf_counter()
        if name != "begin":
            phases[name] = phases.get(name, 0.0) + (t - lap_t[0]) * 1e3
        lap_t[0] = t
    ctx.profile(True)
    for c0 in range(f0, f1, chunk):
        c1 = min(f1, c0 + chunk)
        a0, a1 = max(0, c0 - HALO), min(nt, c1 + HALO)
        t0 = time.perf_counter()
        Y, U, V = _generate(S, torch, dev, a0, a1, E.alpha, E.alphaUV)
        torch.cuda.synchronize()
        gen_s += time.perf_counter() - t0
        vchunk = verify and vstate["spent"] < verify_budget_s
        if vchunk:
            pY, pU, pV = Y.clone(), U.clone(), V.clone()          # what the oracle is fed: the chunk + halo before the erase rewrites it
        elif verify:
            vstate["skipped_chunks"] += 1
        own = DeviceClip(Y[c0 - a0:c1 - a0], U[c0 - a0:c1 - a0], V[c0 - a0:c1 - a0], W, H, BITS)
        nown = c1 - c0
        # ---------------- timed: the hot path over one resident chunk ----------------
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec = d_an[:a1 - a0]
        lap("begin")
        an.analyze_device(Y, BITS, rec)                                                   # a11 over chunk + halo
        lap("analysis")
        er.calc_fades_device(rec, nt, c0, nown, analysis_first=a0, out=d_fades[c0 - f0:c1 - f0])   # a12 CalcFade, on the device
        lap("calc_fades")
        lf.scan_batch(own.Y, BITS, c0, nown)                                              # a9: 3 logos x 2 fades
        lap("scan")
        st.run_device(own.Y, d_stats[c0 - f0:c1 - f0], prevY=Y[c0 - a0 - 1] if c0 > 0 else None)   # CM / KFM metrics, 1-frame halo
        lap("frame_metrics")
        er.erase_device_fades(own, d_fades[c0 - f0:c1 - f0])                              # a12 Delogo in place
        lap("erase")
        torch.cuda.synchronize()
        timed_s += time.perf_counter() - t0
        # ---------------- untimed: what the encoder would consume, reduced to a checksum; probe frames kept ----------------
        d_ansave[c0 - f0:c1 - f0] = rec[c0 - a0:c1 - a0]
        ry, rx = slice(LY, LY + LH), slice(LX, LX + LW)
        cy, cx = slice(LY // 2, (LY + LH) // 2), slice(LX // 2, (LX + LW) // 2)
        erased_sum[c0 - f0:c1 - f0] = (own.Y[:, ry, rx].sum(dim=(1, 2), dtype=torch.int64) * 3 + own.U[:, cy, cx].sum(dim=(1, 2), dtype=torch.int64) * 5
                                       + own.V[:, cy, cx].sum(dim=(1, 2), dtype=torch.int64) * 7)
        if verify:
            for n in range(c0, c1):
                if is_probe(n):
                    stash[n] = tuple(t[n - c0].cpu().numpy().view(np.uint16) for t in (own.Y, own.U, own.V))
        if vchunk:
            tv = time.perf_counter()
            if vstate["ol"] is None:
                vstate["ol"] = E.OracleLogos(E.logos_np, W, H, LX, LY, BITS)

            def pristine(lo, hi, a0=a0, a1=a1):
                if a0 <= lo and hi <= a1:
                    return pY[lo - a0:hi - a0], pU[lo - a0:hi - a0], pV[lo - a0:hi - a0]
                return _generate(S, torch, dev, lo, hi, E.alpha, E.alphaUV)       # (the few frames the group-of-8 alignment reaches back)
            rr = BV.verify_range(torch, vstate["ol"], BITS, nt, c0, c1, pristine,
                                 lambda lo, hi: (own.Y[lo - c0:hi - c0], own.U[lo - c0:hi - c0], own.V[lo - c0:hi - c0]),
                                 lf.evalResults[c0:c1], rec[c0 - a0:c1 - a0].cpu().numpy(), d_fades[c0 - f0:c1 - f0].cpu().numpy(),
                                 d_stats[c0 - f0:c1 - f0].cpu().numpy().view(np.uint64), base=c0, tol=1e-4 if mode == "linear" else 0.0,
                                 chunk=256, threads=vthreads, stage_cache=vstate["stage"])
            for k in vstate["ok"]:
                vstate["ok"][k] &= rr[k]
            vstate["frames"] += rr["frames"]
            vstate["max_an"] = max(vstate["max_an"], rr["analysis_max_abs_err"])
            vstate["mismatches"] += rr.get("mismatches", [])[:4]
            del pY, pU, pV
            vstate["spent"] += time.perf_counter() - tv
        del Y, U, V, own
    prof = ctx.profile_report()
    ctx.profile(False)
    for k, (c, ms) in prof.items():
        if c:
            kern[k] = {"launches": c, "total_ms": ms}

    # ---------------- timed: exchanges + replicated decisions ----------------
    E.fence()
    t0 = time.perf_counter()
    lap("begin")
    fades_loc = d_fades[:nloc].cpu()
    stats_loc = d_stats[:nloc].cpu().numpy().view(np.uint64)
    if world > 1:
        SH.logoframe_allgather(lf, f0, nloc, coll)                                        # 8 B per frame per logo
        metrics = SH.framestats_allgather(st, stats_loc, f0, nt, coll)                    # 64 B per frame
        fades = SH.gather_frame_records(fades_loc.to(dev) if dist.get_backend() == "nccl" else fades_loc, nt).cpu().numpy()
    else:
        metrics, fades = stats_loc, fades_loc.numpy()
    lap("tail_download_and_exchange")
    lf.selectLogo(len(logos))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        lf.writeResult(os.path.join(td, "logof.txt"))
        logof_text = open(os.path.join(td, "logof.txt"), "rb").read()
    lap("tail_select_logo_and_text")
    cad, ph = st.cadence(metrics)
    lap("tail_cadence")
    sc = st.scene_changes(metrics)
    lap("tail_scene_changes")
    torch.cuda.synchronize()
    tail_s = time.perf_counter() - t0
    total_s = E.max_over_ranks(timed_s + tail_s)

    # ---------------- untimed: hash, accuracy against the generator's labels, sampled oracle comparison ----------------
    es = erased_sum[:nloc]
    an_loc = d_ansave[:nloc]
    if world > 1:
        gl = (lambda t: SH.gather_frame_records(t if dist.get_backend() == "nccl" else t.cpu(), nt).cpu())
        es_all, an_all = gl(es).numpy(), gl(an_loc).numpy()
    else:
        es_all, an_all = es.cpu().numpy(), an_loc.cpu().numpy()
    ev = lf.evalResults
    h = hashlib.sha256()
    for part in (np.ascontiguousarray(fades, np.float32).tobytes(), logof_text, cad.tobytes(), ph.tobytes(),
                 np.asarray(sc, np.int32).tobytes(), np.ascontiguousarray(es_all, np.int64).tobytes()):
        h.update(hashlib.sha256(part).digest())
    ok = {"scan": True, "analysis": True, "fades": True, "erase": True, "metrics": True, "frames": 0}
    max_an = 0.0
    if verify:
        ol = E.OracleLogos(E.logos_np, W, H, LX, LY, BITS)
        tol = 1e-4 if mode == "linear" else 0.0
        for p0, pn in probes:
            mine = [n for n in range(p0, p0 + pn) if f0 <= n < f1]
            if not mine:
                continue
            Yb, Ub, Vb = (t.cpu().numpy().view(np.uint16) for t in _generate(S, torch, dev, p0, p0 + pn, E.alpha, E.alphaUV))
            prevY = _generate(S, torch, dev, p0 - 1, p0, E.alpha, E.alphaUV)[0][0].cpu().numpy().view(np.uint16) if p0 > 0 else None
            a = ol.analyze(Yb, pn).reshape(pn, 33)
            o_scan = ol.scan(Yb, pn).reshape(pn, -1)
            o_met = ol.metrics(Yb, pn, prevY)
            lo, hi = (0 if p0 == 0 else HALO), (pn if p0 + pn == nt else pn - HALO)
            for n in mine:
                i = n - p0
                ok["frames"] += 1
                ok["scan"] &= o_scan[i].tobytes() == np.ascontiguousarray(ev[n]).tobytes()
                ok["metrics"] &= o_met[i].tobytes() == np.ascontiguousarray(metrics[n]).tobytes()
                d = float(np.abs(a[i] - an_all[n]).max())
                max_an = max(max_an, d)
                ok["analysis"] &= (d <= tol) if tol else (a[i].tobytes() == np.ascontiguousarray(an_all[n]).tobytes())
                if lo <= i < hi:
                    ft, fb = ol.fade(a.reshape(-1), pn, i)
                    ok["fades"] &= (np.float32(ft).tobytes() + np.float32(fb).tobytes()) == np.ascontiguousarray(fades[n]).tobytes()
                    ol.erase(Yb, Ub, Vb, i, ft, fb)
                    eY, eU, eV = stash[n]
                    ok["erase"] &= bool(np.array_equal(Yb[i], eY) and np.array_equal(Ub[i], eU) and np.array_equal(Vb[i], eV))
        for k in vstate["ok"]:
            ok[k] &= vstate["ok"][k]
        max_an = max(max_an, vstate["max_an"])
        vstate["stage"].clear()
    flags = torch.tensor([int(ok[k]) for k in ("scan", "analysis", "fades", "erase", "metrics")] + [ok["frames"], vstate["frames"], vstate["skipped_chunks"]],
                         dtype=torch.int64)
    mx = torch.tensor([max_an, vstate["spent"]], dtype=torch.float64)
    gen_t = torch.tensor([gen_s, timed_s, tail_s], dtype=torch.float64)
    if world > 1:
        tdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        flags, mx, gen_t = flags.to(tdev), mx.to(tdev), gen_t.to(tdev)
        fl_min = flags.clone(); dist.all_reduce(fl_min, op=dist.ReduceOp.MIN)
        fl_sum = flags.clone(); dist.all_reduce(fl_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(gen_t, op=dist.ReduceOp.MAX)
        flags = torch.cat([fl_min[:5], fl_sum[5:]]).cpu()
        mx, gen_t = mx.cpu(), gen_t.cpu()
    if rank != 0:
        return None
    # accuracy of the self-specified detectors against the generator's labels (parity unpinned; this is what they are worth)
    code = {"30i": 0, "24p": 1, "30p": 2}
    order = ("24p", "30i", "30p")
    truth = np.array([code[order[(n // SEG) % 3]] for n in range(nt)], np.uint8)
    cuts = set(range(97, nt, 97))
    det = set(int(x) for x in sc.tolist())
    verified = dict(zip(("scan", "analysis", "fades", "erase", "metrics"), (bool(v) for v in flags[:5].tolist())))
    verified.update({"frames_compared_with_cpu_oracle": int(flags[6]), "frames_total": nt, "whole_stream": bool(int(flags[6]) == nt),
                     "probe_frames_compared_again": int(flags[5]), "probe_blocks": [[p, k] for p, k in probes],
                     "chunks_left_to_the_probe_blocks_by_the_time_budget": int(flags[7]), "verify_budget_s_per_rank": verify_budget_s,
                     "verify_seconds_max_over_ranks": float(mx[1]), "oracle_threads_per_rank": vthreads,
                     "how": "every chunk of every rank's share right behind its timed calls: scan records, frame metrics, fades and the erased Y/U/V "
                            "planes as bytes, analysis records " + ("within 1e-4" if mode == "linear" else "as bytes") + " (tools/bench_verify.py)",
                     "analysis_max_abs_err": float(mx[0]), "analysis_compare": "abs <= 1e-4" if mode == "linear" else "bytes",
                     **({"mismatches_rank0": vstate["mismatches"][:8]} if vstate["mismatches"] else {}),
                     "ok": bool(all(flags[:5].tolist())) if verify else None})
    byts = W * H * 2
    fsk = kern.get("frame_stats_kernel")
    # ---- roofline of the 16-bit kernels (rank 0's launches, HIP events on the launch stream; algorithmic work as in bench.py: 101 flop per
    #      mask-pixel evaluation + 6 per rectangle pixel per evaluation; HBM traffic from the committed PMC passes over this very workload) ----
    import json
    try:
        pmc16 = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_pmc_traffic16.json")))
    except Exception:
        pmc16 = {}
    an_tab = [logos[0].mask_tables(k, E.maskratio)["count"] for k in (0, 1, 2)]
    scan_tab = [l.mask_tables(0, E.maskratio)["count"] for l in logos]
    fl_an = 101 * 11 * sum(an_tab) + 6 * 11 * (LW * LH + 2 * LW * (LH // 2))
    fl_sc = 101 * 2 * sum(scan_tab) + 6 * 2 * 3 * LW * LH
    nchunks = -(-nloc // chunk) if nloc else 0
    fr_an = nloc + 2 * HALO * nchunks                        # (chunk + halo per analysis launch; a little less at the clip's ends)
    roof16 = {}
    for name, (kind, per_frame, frames, algb) in {"logo_eval_linear_kernel.analysis": ("fp32-valu", fl_an, fr_an, 2 * LW * LH + 132),
                                                  "logo_eval_pair_kernel.scan": ("fp32-valu", fl_sc, nloc, 3 * 2 * LW * LH + 24),
                                                  "frame_stats_kernel": ("hbm", byts, nloc, byts),
                                                  "delogo_kernel": ("hbm", 2 * 2 * (LW * LH + 2 * (LW // 2) * (LH // 2)), nloc, 2 * 2 * (LW * LH + 2 * (LW // 2) * (LH // 2)))}.items():
        k = kern.get(name) or kern.get(name.split(".")[0])
        if not k or not k["total_ms"]:
            continue
        if name == "delogo_kernel":                                    # frames whose fades are both 0 are skipped on the device: no traffic
            frames = frames * float((np.abs(fades[f0:f1]).sum(axis=1) != 0).mean()) if nloc else 0
        rate = per_frame * frames / (k["total_ms"] * 1e-3)
        peak = 157.3e12 if kind == "fp32-valu" else 8000e9
        tr = pmc16.get({"logo_eval_linear_kernel.analysis": "logo_eval_linear_kernel16"}.get(name, name.split(".")[0]), {}).get("hbm_bytes_per_frame")
        roof16[name + (" (16-bit samples)" if "linear" in name or "pair" in name else "<16-bit>")] = {"bound": kind, "achieved": rate / (1e12 if kind == "fp32-valu" else 1e9), "unit": "TFLOP/s" if kind == "fp32-valu" else "GB/s",
                        "peak": peak / (1e12 if kind == "fp32-valu" else 1e9), "frac": rate / peak, "launches": k["launches"], "total_ms": k["total_ms"],
                        "algorithmic_bytes_per_frame": algb, "traffic_bytes_per_frame": tr,
                        "traffic_source": "profiles/r06_pmc_traffic16.json" if tr else None}
    return {
        "workload": f"BASELINE configs[4]: end-to-end logo scan + AMTAnalyzeLogo + CalcFade + AMTEraseLogo + CM/KFM frame metrics and decisions on "
                    f"{nt} frames ({nt / 29.97 / 3600:.2f} h) of 1920x1080i 10-bit (16-bit containers), frames sharded over {world} GPU(s) by contiguous "
                    f"range, {chunk}-frame chunks generated on the device with an {HALO}-frame halo either side" +
                    (f" -- {SHARE_FRAMES} frames per GPU = one GPU's share of the 4-hour stream ({FULL_FRAMES} frames) at N = 8" if nt == SHARE_FRAMES * world else ""),
        "frames_total": nt, "frames_per_gpu": nloc, "n_gpus": world, "chunk_frames": chunk, "analysis_mode": mode,
        "value": nt / total_s, "unit": "frames/sec", "timed_s": total_s, "scaling": scaling,
        "timed_region": "per chunk: analysis (chunk + halo) -> device CalcFade -> scan -> frame metrics -> erase, inputs resident in HBM; plus the "
                        "final exchanges and the replicated decisions; max over ranks.  Untimed: stream generation, and one 32-frame warm-up "
                        "pass (first use of every object: tile plans, tables, code objects -- per-logo set-up)",
        "rank0_seconds": {"chunks": float(timed_s), "exchange_and_decisions": float(tail_s), "generation_untimed_max_over_ranks": float(gen_t[0])},
        "kernels_rank0": kern,
        **({"chunk_phases_ms_fenced_diagnostic": phases} if phases is not None else {}),
        "frame_stats_hbm": ({"achieved_gbs": byts * nloc / (fsk["total_ms"] * 1e-3) / 1e9, "frac": byts * nloc / (fsk["total_ms"] * 1e-3) / 1e9 / 8000.0}
                            if fsk else None),
        "roofline16": roof16,
        "decisions_sha256": h.hexdigest(),
        "decisions_hashed": "sha256 over the sha256 of: fades (float32 pairs), logoframe text, cadence per frame, 3:2 phase per frame, scene-change "
                            "list (int32), per-frame checksum of the erased logo rectangles (3*sum Y + 5*sum U + 7*sum V) -- identical at every N and "
                            "chunk size or the sharded run is wrong",
        "best_logo": lf.getBestLogo(), "logo_ratio": lf.getLogoRatio(), "scene_changes": len(det),
        "frames_with_nonzero_fade_share": float((np.abs(fades).sum(axis=1) != 0).mean()),
        "accuracy_vs_generator_labels": {"cadence_agreement_all_frames": float((cad == truth).mean()),
                                         "scene_cut_precision": len(det & cuts) / max(1, len(det)), "scene_cut_recall": len(det & cuts) / max(1, len(cuts))},
        "verified": verified,
        "exchanges": ["amtgpu_logoframe_allgather_results (8 B/frame/logo)", "amtgpu_framestats_allgather (64 B/frame)", "fades all-gather (8 B/frame)"]
                     if world > 1 else [],
    }
