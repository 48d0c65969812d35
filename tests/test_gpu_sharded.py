"""The frame-sharded drivers on the HIP path, world_size 2: one process per rank with its own context, the kernels of both
ranks on real hardware.  With >= 2 GPUs every rank takes its own device and the collectives run over RCCL (backend nccl);
on a 1-GPU box both ranks share device 0 and the collectives run over gloo -- the sharded code path (quota, all-gather,
exact all-reduce, identical regression on every rank, halo analysis) is the same.  Expected values are the REAL
reference's outputs (tests/golden/, no oracle in the loop): the sharded result must be the single-GPU result."""
import os
import socket

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmpdir, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    ndev = torch.cuda.device_count()
    devidx = rank % ndev
    torch.cuda.set_device(devidx)
    backend = "nccl" if ndev >= world else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", devidx))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, Logo, LogoFrame
        from amatsukaze_amd import sharding as SH
        dev = torch.device("cuda", devidx)
        ctx = Context(devidx)
        coll = SH.TorchCollectives()
        g = G.load()
        W, H, LW, LH, X, Y0, N = (g[k] for k in ("W", "H", "LW", "LH", "X", "Y0", "N"))
        res = {"rank": rank, "backend": backend}

        # ---- ScanLogo, sharded: the .lgd must be the reference's own file ----
        Y2, U2, V2 = G.frames(g, "scanlogo_crop_y", "scanlogo_crop_u", "scanlogo_crop_v")
        n2 = Y2.shape[0]
        a, b = SH.shard_range(n2, rank, world)
        loc = DeviceClip(torch.from_numpy(Y2[a:b]).to(dev), torch.from_numpy(U2[a:b]).to(dev), torch.from_numpy(V2[a:b]).to(dev), W, H)
        dst = os.path.join(tmpdir, "sharded.lgd")
        ok = SH.scan_logo_sharded(ctx, loc, 1041, dst if rank == 0 else None, X, Y0, LW, LH, 12, 25, coll)
        res["scanlogo_ok"] = bool(ok) and coll.error is None
        if rank == 0:
            res["scanlogo_lgd_equal"] = open(dst, "rb").read() == g["scanlogo_lgd"].tobytes()
        # a quota that ends inside rank 0's shard (rank 1 contributes nothing) and one nobody reaches
        for cap, tag in ((3, "cap3"), (10000, "capall")):
            d2 = os.path.join(tmpdir, f"sharded_{tag}.lgd")
            ok = SH.scan_logo_sharded(ctx, loc, 1041, d2 if rank == 0 else None, X, Y0, LW, LH, 12, cap, coll)
            if rank == 0:
                from amatsukaze_amd import ScanLogo
                full = DeviceClip(torch.from_numpy(Y2).to(dev), torch.from_numpy(U2).to(dev), torch.from_numpy(V2).to(dev), W, H)
                d1 = os.path.join(tmpdir, f"single_{tag}.lgd")
                ok1 = ScanLogo(ctx, full, 1041, d1, X, Y0, LW, LH, 12, cap)
                res[f"scanlogo_{tag}"] = (bool(ok) == bool(ok1)) and (not ok1 or open(d1, "rb").read() == open(d2, "rb").read())

        # ---- a failure on ONE rank (here: rank 1's rectangle lies outside the frame) must end the run on BOTH, not strand rank 0 in a
        #      collective: the status rides along every exchange (ShardGuard, amt_gpu_erase_scan.hip) ----
        bad = SH.scan_logo_sharded(ctx, loc, 1041, None, X if rank == 0 else W, Y0, LW, LH, 12, 25, coll)
        res["failsafe_scanlogo_failed"] = not bad
        res["failsafe_scanlogo_msg"] = ctx.lib.amtgpu_last_error(ctx.h).decode(errors="replace")

        # ---- LogoFrame all-frames scan, sharded (ragged: 39 frames -> 20 + 19) + all-gather on every rank ----
        Y, U, V = G.frames(g, pitch_pad=32)
        NS = N - 1
        f0, f1 = SH.shard_range(NS, rank, world)
        logos = [Logo.from_planes(ctx, g[k], LW, LH, W, H, X, Y0) for k in ("logo0", "logo1")]
        lf = LogoFrame(ctx, logos, 0.35)
        lf.begin(W, H, 8, NS)
        lf.scan_batch(torch.from_numpy(Y[f0:f1]).to(dev), 8, f0, f1 - f0)
        try:                                               # rank 1 reports a range outside the clip: both ranks get an error, nobody hangs
            SH.logoframe_allgather(lf, f0 if rank == 0 else NS, f1 - f0, coll)
            res["failsafe_allgather_failed"] = False
        except Exception as e:
            res["failsafe_allgather_failed"] = True
            res["failsafe_allgather_msg"] = str(e)
        SH.logoframe_allgather(lf, f0, f1 - f0, coll)
        res["logoframe_equal"] = lf.evalResults.tobytes() == g["logoframe_evals"][:NS].tobytes()
        lf.selectLogo(2)
        res["best"] = lf.getBestLogo()

        # ---- analysis + erase of a shard with an 8-frame analysis halo either side (CalcFade2 reads n-8 .. n+8) ----
        e0, e1 = SH.shard_range(N, rank, world)
        h0, h1 = SH.halo_range(e0, e1, N)
        an = np.zeros((N, 33), np.float32)
        hal = DeviceClip(torch.from_numpy(Y[h0:h1]).to(dev), torch.from_numpy(U[h0:h1]).to(dev), torch.from_numpy(V[h0:h1]).to(dev), W, H)
        an[h0:h1] = AMTAnalyzeLogo(ctx, logos[0], 0.35).analyze(hal)
        er = AMTEraseLogo(ctx, logos[0], "", 0, 16)
        fades = er.calc_fades(an, N, e0, e1 - e0)          # touches only records inside [e0-8, e1+8)
        res["fades_equal"] = fades.tobytes() == g["erase_nolf_fades"][e0:e1].tobytes()
        own = DeviceClip(hal.Y[e0 - h0:e1 - h0], hal.U[e0 - h0:e1 - h0], hal.V[e0 - h0:e1 - h0], W, H)
        er.erase(own, fades)
        ctx.synchronize()
        cy, cu, cv = G.crops(g, own.Y.cpu().numpy(), own.U.cpu().numpy(), own.V.cpu().numpy())
        res["erase_equal"] = bool(np.array_equal(cy, g["erase_nolf_Y"][e0:e1]) and np.array_equal(cu, g["erase_nolf_U"][e0:e1])
                                  and np.array_equal(cv, g["erase_nolf_V"][e0:e1]))
        # ---- the same shard with the decision on the DEVICE: halo analysis records stay in HBM, fades computed and consumed there ----
        d_an = torch.from_numpy(an[h0:h1]).to(dev)
        d_f = er.calc_fades_device(d_an, N, e0, e1 - e0, analysis_first=h0)
        res["device_fades_equal"] = d_f.cpu().numpy().tobytes() == g["erase_nolf_fades"][e0:e1].tobytes()
        own2 = DeviceClip(torch.from_numpy(Y[e0:e1]).to(dev), torch.from_numpy(U[e0:e1]).to(dev), torch.from_numpy(V[e0:e1]).to(dev), W, H)
        er.erase_device_fades(own2, d_f)
        ctx.synchronize()
        res["device_erase_equal"] = bool(torch.equal(own2.Y, own.Y) and torch.equal(own2.U, own.U) and torch.equal(own2.V, own.V))

        # ---- self-specified CM / KFM frame metrics, sharded: contiguous range + the frame before it as halo,
        #      all-gather of the 64-byte records, decisions replicated; against the numpy oracle over the whole clip ----
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import frame_stats_oracle as FS
        from amatsukaze_amd import FrameStats
        fs = FrameStats(ctx, W, H, 8)
        Yd = torch.from_numpy(Y).to(dev)                      # (this rank only touches its own rows of it below)
        m = SH.framestats_sharded(fs, Yd[e0:e1], e0, N, coll, prevY=Yd[e0 - 1] if e0 > 0 else None)
        whole = FS.frame_metrics(Y[:, :, :W])
        res["metrics_equal"] = bool(np.array_equal(m, whole))
        cad, ph = fs.cadence(m)
        ocad, oph = FS.classify_cadence(whole, W, H)
        res["decisions_equal"] = bool(np.array_equal(cad, ocad) and np.array_equal(ph, oph) and fs.scene_changes(m).tolist() == FS.scene_changes(whole, W, H))
        try:                                                  # a shard that does not start the clip must bring its halo frame
            SH.framestats_sharded(fs, Yd[e0:e1], e0, N, coll, prevY=None)
            res["metrics_halo_required"] = False
        except Exception as e:
            res["metrics_halo_required"] = True
            res["metrics_halo_msg"] = str(e)
        q.put(res)
    except Exception as e:      # surface the failure in the parent instead of a bare timeout
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
    finally:
        dist.destroy_process_group()


def test_sharded_hip_path_world2(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert "error" not in r, r["error"]
    r0, r1 = res
    assert r0["scanlogo_ok"] and r1["scanlogo_ok"]
    assert r0["scanlogo_lgd_equal"], "sharded ScanLogo .lgd differs from the reference's"
    assert r0["scanlogo_cap3"] and r0["scanlogo_capall"]
    # one rank's failure ends the sharded call on every rank, each with a message that says whose it was
    assert r0["failsafe_scanlogo_failed"] and r1["failsafe_scanlogo_failed"]
    assert "another rank failed" in r0["failsafe_scanlogo_msg"] and "outside the frame" in r1["failsafe_scanlogo_msg"]
    assert r0["failsafe_allgather_failed"] and r1["failsafe_allgather_failed"]
    assert "another rank failed" in r0["failsafe_allgather_msg"] and "outside the clip" in r1["failsafe_allgather_msg"]
    for r in res:
        assert r["logoframe_equal"] and r["best"] == int(G.load()["logoframe_best"])
        assert r["fades_equal"] and r["erase_equal"]
        assert r["device_fades_equal"] and r["device_erase_equal"]
        assert r["metrics_equal"] and r["decisions_equal"]
        assert r["metrics_halo_required"]                    # rank 1 lacks its halo: BOTH ranks fail, with their own message
    assert "needs the frame before it" in r1["metrics_halo_msg"] and "another rank failed" in r0["metrics_halo_msg"]
    assert all(p.exitcode == 0 for p in procs)


def _bench_lines(r):
    """(the compact line bench.py prints last on stdout -- what the driver parses --, the full detail it writes to stderr)"""
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(r.stdout.strip().splitlines()[-1]) < 6144
    full = next(json.loads(l)["bench_detail"] for l in reversed(r.stderr.splitlines()) if l.startswith('{"bench_detail"'))
    for k in ("value", "n_gpus", "scaling", "steps"):
        assert (line[k] == full[k]) or abs(line[k] - full[k]) <= 1e-5 * abs(full[k]), k
    return line, full


def test_bench_multi_rank_control_flow_dry_run():
    """`bench.py --gpus 2` on whatever box runs this: the launcher re-executes itself under torch.distributed.run, the ranks shard
    the work and exchange their records.  On a 1-GPU box the ranks share device 0 and talk over gloo (AMT_BENCH_SHARED_GPU=1: a
    dry run of the control flow, its numbers mean nothing); with >= 2 devices it is a real RCCL run."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["AMT_BENCH_SHARED_GPU"] = "1"
    common = ["--gpus", "2", "--strong-frames", "4096", "--strong-steps", "2"]
    for extra, scaling in ((["--steps", "2", "--warmup", "1", "--frames", "512", "--no-ingest", "--cpu-frames", "0", "--no-alt-mode"], "weak"),
                           (["--scaling", "strong"], "strong")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        compact, line = _bench_lines(r)
        assert compact["n_gpus"] == 2 and compact["scaling"] == scaling and compact["value"] > 0 and compact["strong_scan"]["value"] > 0
        assert compact["collectives"]["world_size_observed"] == 2 and compact["collectives"]["backend"] in ("rccl", "gloo")
        ss = line["strong_scan"]
        assert ss["n_gpus"] == 2 and ss["verified"]["sharded_equals_single_launch"] and ss["verified"]["equals_cpu_oracle"]
        sl = ss["scanlogo"]                              # the sharded "full LogoScan" of the same stream: quota hand-out + 3 all-reduces
        assert sl["n_gpus"] == 2 and len(sl["lgd_sha256"]) == 64
        lgd_hashes = locals().setdefault("lgd_hashes", set())
        lgd_hashes.add(sl["lgd_sha256"])
        if scaling == "weak":
            assert line["verified"]["ok"] and line["verified"]["tolerance_ok"] and line["verified"]["fades_equal_all"]
            assert compact["verified"]["ok"] and compact["verified"]["frames"] == 512 and compact["roofline"]["frac"] > 0
    assert len(lgd_hashes) == 1          # the weak line's attached strong scan and the strong line scanned the same stream


def test_bench_eight_rank_control_flow_dry_run():
    """The driver's 8-GPU invocation (`bench.py --gpus 8`) has never met hardware here: run its whole control flow -- launcher, ragged
    shards (107 892 = 8 x 13 486 + 4 at full size; 4 099 frames here), the quota hand-out of the sharded ScanLogo, every exchange, the
    e2e10 stream with chunk halos across seven rank boundaries -- with eight ranks on whatever the box has (all on device 0 over gloo
    when it has fewer than eight: AMT_BENCH_SHARED_GPU=1, numbers meaningless).  The hashes of the records and decisions must equal
    the single-rank run's."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if torch.cuda.device_count() < 8:
        env["AMT_BENCH_SHARED_GPU"] = "1"

    def run(extra, gpus):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus)] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return _bench_lines(r)[1]

    # (numMaxFrames 512 of ~1 025 valid frames in 4 099: the stream-order quota of the sharded ScanLogo, LogoScan.hpp:885, closes the stream
    #  in the fifth rank's shard -- ranks 0-3 keep all their valid frames, rank 4 a part, ranks 5-7 none)
    strong = ["--scaling", "strong", "--strong-frames", "4099", "--strong-steps", "1", "--scanlogo-max-frames", "512"]
    s8, s1 = run(strong, 8), run(strong, 1)
    assert s8["n_gpus"] == 8 and s8["collectives"]["world_size_observed"] == 8
    v8 = s8["strong_scan"]["verified"]
    assert v8["sharded_equals_single_launch"] and v8["equals_cpu_oracle"] and v8["frames"] == 4099      # every record of every rank's shard
    assert s8["strong_scan"]["records_sha256"] == s1["strong_scan"]["records_sha256"]
    assert s8["strong_scan"]["scanlogo"]["lgd_sha256"] == s1["strong_scan"]["scanlogo"]["lgd_sha256"]
    for s_ in (s8, s1):                                   # ... and the whole stream's .lgd against the CPU oracle's, quota reached
        sv = s_["strong_scan"]["scanlogo"]["verified"]
        assert sv["lgd_equals_cpu_oracle"] and sv["quota_hit"] and sv["valid_frames"] == 512 and sv["frames"] == 4099
    e2e = ["--workload", "e2e10", "--e2e-frames", "1500", "--e2e-chunk", "96"]
    e8, e1 = run(e2e, 8), run(e2e + ["--e2e-chunk", "512"], 1)
    assert e8["n_gpus"] == 8 and e8["e2e10"]["verified"]["ok"] and e1["e2e10"]["verified"]["ok"]
    for e_ in (e8, e1):                                   # every frame of every rank's share against the CPU oracle
        assert e_["e2e10"]["verified"]["whole_stream"] and e_["e2e10"]["verified"]["frames_compared_with_cpu_oracle"] == 1500
    assert e8["e2e10"]["decisions_sha256"] == e1["e2e10"]["decisions_sha256"]
    weak = run(["--steps", "1", "--warmup", "1", "--frames", "256", "--no-ingest", "--cpu-frames", "0", "--no-alt-mode", "--no-configs", "--no-strong",
                "--exact-steps", "0"], 8)
    assert weak["n_gpus"] == 8 and weak["scaling"] == "weak" and weak["verified"]["ok"] and weak["collectives"]["world_size_observed"] == 8
