"""The sharded drivers of the C ABI over RCCL, driven from C++ (tests/cpp/sharded_rccl_test.cpp with
include/amt_rccl_collectives.hpp: ncclAllGather / ncclAllReduce(int64) staged through HBM), one rank per visible GPU, on the
clip of the real reference's golden outputs: the sharded .lgd and the gathered LogoFrame records must be the reference's own
(tests/golden/logo_path_v1.npz, LogoScan.hpp:885, 917-1036, 1577-1584).  On a 1-GPU box the world is one rank -- the
collectives still run through RCCL; with N GPUs it is an N-rank run."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_sharded_drivers_over_rccl(tmp_path):
    import torch
    from amatsukaze_amd import binding
    subprocess.check_call(["make", "-C", CPP, "sharded_rccl_test"], stdout=subprocess.DEVNULL)
    lib = binding.load()
    g = G.load()
    W, H, LW, LH, X, Y0 = (g[k] for k in ("W", "H", "LW", "LH", "X", "Y0"))
    # the two candidate logos as .lgd files (amtgpu_logo_from_planes / _save need no GPU)
    paths = []
    for i, k in enumerate(("logo0", "logo1")):
        h = lib.amtgpu_logo_from_planes(None, LW, LH, 1, 1, W, H, X, Y0, np.ascontiguousarray(g[k], np.float32).ctypes.data)
        p = str(tmp_path / f"l{i}.lgd")
        assert h and lib.amtgpu_logo_save(None, h, p.encode(), b"golden", 1041) == 1
        lib.amtgpu_logo_destroy(h)
        paths.append(p)
    out = tmp_path / "out"
    ranks = None
    for tag, (ky, ku, kv), cap in (("scanlogo", ("scanlogo_crop_y", "scanlogo_crop_u", "scanlogo_crop_v"), 25),):
        Y, U, V = G.frames(g, ky, ku, kv)
        raw = tmp_path / f"{tag}.raw"
        with open(raw, "wb") as f:
            f.write(np.array([W, H, 8, Y.shape[0], Y.shape[2], U.shape[2]], np.int32).tobytes())
            for a in (Y, U, V):
                f.write(a.tobytes())
        out.mkdir(exist_ok=True)
        r = subprocess.run([os.path.join(CPP, "sharded_rccl_test"), str(raw), paths[0], paths[1], str(out), str(X), str(Y0), str(LW), str(LH), str(cap)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        last = r.stdout.strip().splitlines()[-1]               # (RCCL prints its version banner first)
        assert last.startswith("ok world=") and "RCCL version" in r.stdout
        ranks = int(last.split("world=")[1].split()[0])
        assert ranks == max(1, torch.cuda.device_count())
        # the reference's own .lgd for this clip and quota
        assert (out / "sharded.lgd").read_bytes() == g["scanlogo_lgd"].tobytes()
    # the all-frames scan on the logoframe clip: every rank's gathered records are the reference's
    Y, U, V = G.frames(g)
    NS = Y.shape[0]
    raw = tmp_path / "lf.raw"
    with open(raw, "wb") as f:
        f.write(np.array([W, H, 8, NS, Y.shape[2], U.shape[2]], np.int32).tobytes())
        for a in (Y, U, V):
            f.write(a.tobytes())
    r = subprocess.run([os.path.join(CPP, "sharded_rccl_test"), str(raw), paths[0], paths[1], str(out), str(X), str(Y0), str(LW), str(LH), "1000"],
                       capture_output=True, text=True, timeout=600)
    # (this clip has no flat-border frames: ScanLogo reports "Insufficient logo frames" on it, sharded and single alike -- the scan part
    #  of the run is what is checked here)
    want = g["logoframe_evals"][:NS].astype(np.float32)
    for k in range(ranks):
        p = out / f"eval_rank{k}.bin"
        if r.returncode == 0:
            assert np.fromfile(p, np.float32).tobytes() == want.tobytes()
    if r.returncode != 0:
        assert "Insufficient logo frames" in r.stderr, r.stderr
