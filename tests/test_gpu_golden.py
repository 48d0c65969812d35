"""The HIP path against the committed golden vectors of the REAL reference (tests/golden/, tools/make_golden.py):
no oracle in the loop."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


def test_gpu_reproduces_reference_golden(tmp_path):
    import torch
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, Logo, LogoFrame, LogoScan, ScanLogo
    g = G.load()
    W, H, LW, LH, X, Y0, N = (g[k] for k in ("W", "H", "LW", "LH", "X", "Y0", "N"))
    dev = torch.device("cuda:0")
    ctx = Context(0)
    Y, U, V = G.frames(g, pitch_pad=32)
    mk = lambda: DeviceClip(torch.from_numpy(Y).to(dev), torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev), W, H)
    logos = [Logo.from_planes(ctx, g[k], LW, LH, W, H, X, Y0) for k in ("logo0", "logo1")]
    p = tmp_path / "l.lgd"
    logos[0].save(p, "golden", 1041)
    assert p.read_bytes() == g["lgd0"].tobytes()
    lf = LogoFrame(ctx, logos, 0.35)
    lf.scanFrames(mk())
    assert lf.evalResults.tobytes() == g["logoframe_evals"].tobytes()
    lf.selectLogo(2)
    assert lf.getBestLogo() == int(g["logoframe_best"]) and np.float32(lf.getLogoRatio()) == g["logoframe_ratio"]
    out = tmp_path / "lf.txt"
    lf.writeResult(out, 0)
    assert out.read_bytes() == g["logoframe_text"].tobytes()
    an = AMTAnalyzeLogo(ctx, logos[0], 0.35).analyze(mk())
    assert an.tobytes() == g["analysis"].tobytes()
    for tag in ("nolf", "lf"):
        dc = mk()
        er = AMTEraseLogo(ctx, logos[0], g["logof_text"].tobytes().decode() if tag == "lf" else "", 0, 16)
        fades = er.calc_fades(an, N)
        assert fades.tobytes() == g[f"erase_{tag}_fades"].tobytes()
        er.erase(dc, fades)
        cy, cu, cv = G.crops(g, dc.Y.cpu().numpy(), dc.U.cpu().numpy(), dc.V.cpu().numpy())
        assert np.array_equal(cy, g[f"erase_{tag}_Y"]) and np.array_equal(cu, g[f"erase_{tag}_U"]) and np.array_equal(cv, g[f"erase_{tag}_V"])
    scan = LogoScan(ctx, LW, LH, 12)
    valid, nacc = scan.add_batch(mk(), X, Y0)
    assert valid.tolist() == g["scan_valid"].tolist()
    s, pl = scan.sums()
    ref = g["scan_sums"]
    s = s.reshape(-1, 3)
    assert np.array_equal(s[:, 0], ref[:, 0].astype(np.int64)) and np.array_equal(s[:, 1], ref[:, 2].astype(np.int64)) \
        and np.array_equal(s[:, 2], ref[:, 4].astype(np.int64))
    assert scan.get_logo(255, False, W, H, X, Y0).planes.tobytes() == g["scan_logo_raw"].tobytes()
    assert scan.get_logo(255, True, W, H, X, Y0).planes.tobytes() == g["scan_logo_clean"].tobytes()
    Y2, U2, V2 = G.frames(g, "scanlogo_crop_y", "scanlogo_crop_u", "scanlogo_crop_v")
    dc2 = DeviceClip(torch.from_numpy(Y2).to(dev), torch.from_numpy(U2).to(dev), torch.from_numpy(V2).to(dev), W, H)
    dst = tmp_path / "scan.lgd"
    assert ScanLogo(ctx, dc2, 1041, dst, X, Y0, LW, LH, 12, 25)
    assert dst.read_bytes() == g["scanlogo_lgd"].tobytes()


@pytest.mark.parametrize("name", ["sd10", "sd12", "hd8", "fhd10"])
def test_gpu_reproduces_reference_golden_hibit_hd(name):
    """> 8-bit containers (10/12-bit in uint16) and HD frame sizes against outputs of the REAL reference."""
    import torch
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, Logo, LogoFrame
    cases, logof_text = G.load_v2()
    c = cases[name]
    W, H, bits, LW, LH, X, Y0, N = (c[k] for k in ("W", "H", "bits", "LW", "LH", "X", "Y0", "N"))
    dev = torch.device("cuda:0")
    ctx = Context(0)
    Y, U, V = G.frames_v2(c, pitch_pad=32)
    tt = (lambda a: torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(dev))
    mk = lambda: DeviceClip(tt(Y), tt(U), tt(V), W, H, bits)
    logo = Logo.from_planes(ctx, c["logo"], LW, LH, W, H, X, Y0)
    an = AMTAnalyzeLogo(ctx, logo, 0.35).analyze(mk())
    assert an.tobytes() == c["analysis"].tobytes()
    for tag in ("nolf", "lf"):
        if f"erase_{tag}_fades" not in c:
            continue
        dc = mk()
        er = AMTEraseLogo(ctx, logo, logof_text.decode() if tag == "lf" else "", 0, 16)
        fades = er.calc_fades(an, N)
        assert fades.tobytes() == c[f"erase_{tag}_fades"].tobytes()
        er.erase(dc, fades)
        ctx.synchronize()
        gy, gu, gv = (t.cpu().numpy().view(Y.dtype) for t in (dc.Y, dc.U, dc.V))
        assert np.array_equal(gy[:, Y0:Y0 + LH, X:X + LW], c[f"erase_{tag}_Y"])
        assert np.array_equal(gu[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2], c[f"erase_{tag}_U"])
        assert np.array_equal(gv[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2], c[f"erase_{tag}_V"])
    if "logoframe_evals" in c:
        lf = LogoFrame(ctx, [logo], 0.35)
        lf.scanFrames(mk())
        assert lf.evalResults.tobytes() == c["logoframe_evals"].tobytes()
    if "logoframe_evals_bytepitch" in c:
        # LogoFrame::ScanFrame<uint16_t>'s byte-pitch stride (LogoScan.hpp:1547,1561) reproduced by passing a doubled pitch
        Yq = tt(G.quirk_frames_v2(c))
        lf = LogoFrame(ctx, [logo], 0.35)
        lf.begin(W, H, bits, N)
        ctx.check(ctx.lib.amtgpu_logoframe_scan_batch(lf.h, Yq.data_ptr(), int(Yq.stride(0)) * 2, 2 * int(Yq.stride(1)), 0, N))
        assert lf.evalResults.tobytes() == c["logoframe_evals_bytepitch"].tobytes()


def test_gpu_scanlogo_file_export_reproduces_reference_lgd(tmp_path):
    """amtgpu_scanlogo_file has the reference's ScanLogo argument list (LogoScan.hpp:1083-1098: srcpath, serviceid, workfile,
    dstpath, imgx, imgy, w, h, thy, numMaxFrames, cb); on the raw clip the reference's shim decoder read when the golden file was
    made it must write the very same .lgd.  Frames stream from the file in chunks; the callback can cancel."""
    from amatsukaze_amd import Context, ScanLogoFile
    from amtlib import write_raw_clip
    g = G.load()
    W, H, LW, LH, X, Y0 = (g[k] for k in ("W", "H", "LW", "LH", "X", "Y0"))
    Y2, U2, V2 = G.frames(g, "scanlogo_crop_y", "scanlogo_crop_u", "scanlogo_crop_v")
    raw = tmp_path / "clip.raw"
    write_raw_clip(raw, Y2, U2, V2, W, H)
    ctx = Context(0)
    calls = []
    dst = tmp_path / "out.lgd"
    assert ScanLogoFile(ctx, raw, 1041, tmp_path / "work.dat", dst, X, Y0, LW, LH, 12, 25, lambda p, nread, total, ngather: calls.append((p, nread, ngather)) or 1)
    assert dst.read_bytes() == g["scanlogo_lgd"].tobytes()
    assert calls and calls[-1][0] == 1.0 and max(c[2] for c in calls) == 25 and not (tmp_path / "work.dat").exists()
    # cancel from the callback -> 0 and the reference's message
    assert not ScanLogoFile(ctx, raw, 1041, tmp_path / "work.dat", tmp_path / "x.lgd", X, Y0, LW, LH, 12, 25, lambda *a: 0)
    assert b"Cancel requested" in ctx.lib.amtgpu_last_error(ctx.h)
    # too few flat-bordered frames -> "Insufficient logo frames" like GetLogo (:380-395 through :1061-1064)
    assert not ScanLogoFile(ctx, raw, 1041, "", tmp_path / "y.lgd", X, Y0, LW, LH, 12, 1)
    assert not ScanLogoFile(ctx, tmp_path / "missing.raw", 1, "", tmp_path / "z.lgd", X, Y0, LW, LH, 12, 25)
