"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact everywhere -- the float logo scores too (the kernels reproduce the reference's fp32
evaluation order, see amatsukaze_amd/csrc/exact_math.h), which is stricter than the 1e-4 the north star
allows and is what keeps every downstream integer decision identical.
"""
import ctypes as C

import numpy as np
import pytest

import amt_synth as S
from amtlib import ROOT, Oracle, _ptr

pytestmark = pytest.mark.gpu

SMALL = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=40, period=16, fade=6, flat=3)
HD = dict(W=1440, H=1080, LW=256, LH=128, IMGX=1120, IMGY=64, N=12, period=6, fade=3, flat=3)


@pytest.fixture(scope="module")
def gpu():
    import torch
    from amatsukaze_amd import Context
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return dict(torch=torch, ctx=Context(0), dev=torch.device("cuda:0"))


def make_case(gpu, cfg, bits=8, pitch_pad=0, seed=0x5EED0001):
    from amatsukaze_amd import DeviceClip, Logo
    torch = gpu["torch"]
    data, alpha, alphaUV = S.make_logo(cfg["LW"], cfg["LH"])
    W, H = cfg["W"], cfg["H"]
    pY = W + pitch_pad
    pUV = W // 2 + pitch_pad // 2
    clip = S.make_clip_np(cfg["N"], W, H, seed, alpha, alphaUV, cfg["IMGX"], cfg["IMGY"], bits=bits, period=cfg["period"],
                          fade=cfg["fade"], flat_every=cfg["flat"], pitchY=pY, pitchUV=pUV)
    tdt = torch.uint8 if bits <= 8 else torch.int16
    dclip = DeviceClip(*(torch.from_numpy(clip[k].view(np.uint8 if bits <= 8 else np.int16)).to(gpu["dev"]).to(tdt) for k in "YUV"),
                       width=W, height=H, bits=bits)
    logo = Logo.from_planes(gpu["ctx"], data, cfg["LW"], cfg["LH"], W, H, cfg["IMGX"], cfg["IMGY"])
    orc = Oracle()
    lo = orc.make_logo(data, cfg["LW"], cfg["LH"], W, H, cfg["IMGX"], cfg["IMGY"])
    return dict(cfg=cfg, clip=clip, dclip=dclip, logo=logo, orc=orc, lo=lo, data=data, alpha=alpha, alphaUV=alphaUV, bits=bits)


def oracle_eval_logos(orc, lo, maskratio=0.35):
    d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, maskratio, 1)
    t = orc.lib.orc_logo_field(lo, 0); orc.lib.orc_logo_create_mask(t, maskratio, 1)
    b = orc.lib.orc_logo_field(lo, 1); orc.lib.orc_logo_create_mask(b, maskratio, 1)
    return d, t, b


@pytest.mark.parametrize("cfgname,bits,pad", [("small", 8, 0), ("small", 8, 32), ("small", 10, 0), ("hd", 8, 32)])
def test_logoframe_scan_bit_exact(gpu, tmp_path, cfgname, bits, pad):
    from amatsukaze_amd import Logo, LogoFrame
    cfg = SMALL if cfgname == "small" else HD
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=pad)
    orc, ctx = cs["orc"], gpu["ctx"]
    # candidate 2: weaker look-alike; candidate 3: a logo made for another frame size (scores {0,-1})
    d2, _, _ = S.make_logo(cfg["LW"], cfg["LH"], seed=0x10600002, strength=0.5)
    logo2 = Logo.from_planes(ctx, d2, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
    logo3 = Logo.from_planes(ctx, d2, cfg["LW"], cfg["LH"], cfg["W"] + 16, cfg["H"], cfg["IMGX"], cfg["IMGY"])
    lf = LogoFrame(ctx, [cs["logo"], logo2, logo3], 0.35)
    lf.scanFrames(cs["dclip"], batch=17)          # ragged batches
    got = lf.evalResults
    lo2 = orc.make_logo(d2, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
    lo3 = orc.make_logo(d2, cfg["LW"], cfg["LH"], cfg["W"] + 16, cfg["H"], cfg["IMGX"], cfg["IMGY"])
    hs = []
    for l in (cs["lo"], lo2, lo3):
        d = orc.lib.orc_logo_deint(l); orc.lib.orc_logo_create_mask(d, 0.35, 1); hs.append(d)
    Y = cs["clip"]["Y"]
    n = Y.shape[0]
    want = np.zeros(n * 3 * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 3)(*hs), 3, _ptr(Y), Y.strides[0], Y.shape[2], bits, cfg["W"], cfg["H"], n, _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()
    assert np.all(got[:, 2, 0] == 0) and np.all(got[:, 2, 1] == -1)
    # decisions: selectLogo over the first two, logoframe text of the best
    lf.selectLogo(2)
    best, ratio = C.c_int(), C.c_float()
    orc.lib.orc_logoframe_select(_ptr(want), n, 3, 2, C.byref(best), C.byref(ratio))
    assert lf.getBestLogo() == best.value == 0
    assert np.float32(lf.getLogoRatio()).tobytes() == np.float32(ratio.value).tobytes()
    out = tmp_path / "logof.txt"
    lf.writeResult(out)
    buf = C.create_string_buffer(1 << 16)
    ln = orc.lib.orc_logoframe_write_result(_ptr(want), n, 3, 0, 30000, 1001, buf, len(buf))
    assert out.read_bytes() == buf.raw[:ln]
    if cfgname == "small" and bits == 8:
        assert ln > 0
    # the same decisions from the records alone, on the host (what a rank does with gathered records)
    from amatsukaze_amd.api import logoframe_decide_host
    hbest, hratio, htext = logoframe_decide_host(got, 30000, 1001, numCandidates=2)
    assert hbest == best.value and np.float32(hratio).tobytes() == np.float32(ratio.value).tobytes() and htext == buf.raw[:ln]
    # LogoFrame::dumpResult (LogoScan.hpp:1632-1643): "<base><logo>", one "%f,%f" line {corr0, corr1} per frame
    lf.dumpResult(tmp_path / "dump_")
    for i in range(3):
        assert (tmp_path / f"dump_{i}").read_text() == "".join("%f,%f\n" % (float(a), float(b)) for a, b in got[:, i])


def test_logoframe_scan_kernel_choice(gpu):
    """The scan's two fades {0, 1} run on the pair kernel (s and bg evaluated as the two halves of one packed instruction
    stream) when every logo coefficient is finite and below 1e30, and on the generic blend kernel otherwise; both give the
    oracle's bytes.  A coefficient of 1e31 keeps bg finite (so the oracle's result is an ordinary number) and forces the generic one."""
    from amatsukaze_amd import Logo, LogoFrame
    cs = make_case(gpu, SMALL)
    cfg, orc, ctx = cs["cfg"], cs["orc"], gpu["ctx"]
    big = cs["data"].copy()
    big.reshape(-1)[cfg["LW"] * 2 + 5] = 1e31               # A plane of Y, row 2
    Y = cs["clip"]["Y"]
    n = Y.shape[0]
    for data, kernel in ((cs["data"], "logo_eval_pair_kernel.scan"), (big, "logo_eval_fused_kernel.scan")):
        logo = Logo.from_planes(ctx, data, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
        lf = LogoFrame(ctx, [logo], 0.35)
        ctx.profile(True)
        lf.scanFrames(cs["dclip"])
        got = lf.evalResults
        used = [k for k, (calls, _) in ctx.profile_report().items() if calls]
        ctx.profile(False)
        assert used == [kernel], used
        lo = orc.make_logo(data, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
        d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, 0.35, 1)
        want = np.zeros(n * 2, np.float32)
        orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], 8, cfg["W"], cfg["H"], n, _ptr(want))
        assert got.reshape(-1).tobytes() == want.tobytes()
        assert np.isfinite(want).all()


@pytest.mark.parametrize("cfgname,bits", [("small", 8), ("small", 12), ("hd", 8)])
def test_analyze_logo_bit_exact(gpu, cfgname, bits):
    from amatsukaze_amd import AMTAnalyzeLogo
    cfg = SMALL if cfgname == "small" else HD
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=64)
    got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(cs["dclip"])
    orc = cs["orc"]
    d, t, b = oracle_eval_logos(orc, cs["lo"])
    Y = cs["clip"]["Y"]
    n = Y.shape[0]
    want = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, n, _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


def test_known_answers_flat_frames(gpu):
    """(ii)/(iii)/(v) of SURVEY.md section 8c: flat-16 frame with the logo composited scores exactly 1.0 at fade 0
    and ~0 at fade 1; Delogo with fade 0 is the identity."""
    from amatsukaze_amd import AMTEraseLogo, DeviceClip, LogoFrame
    torch = gpu["torch"]
    cfg = SMALL
    cs = make_case(gpu, cfg)
    W, H, LW, LH, X, Y0 = cfg["W"], cfg["H"], cfg["LW"], cfg["LH"], cfg["IMGX"], cfg["IMGY"]
    # composite exactly like AddLogo (LogoScan.hpp:320-333) on a flat-16 background, unrounded is impossible
    # for u8 frames, so check the float property through the oracle's EvaluateLogo and the GPU on rounded data
    a = cs["data"][:LW * LH].reshape(LH, LW)
    b = cs["data"][LW * LH:2 * LW * LH].reshape(LH, LW)
    frame = np.full((1, H, W), 16, np.uint8)
    comp = np.clip(np.rint((16.0 - b * 255.0) / a), 0, 255).astype(np.uint8)
    frame[0, Y0:Y0 + LH, X:X + LW] = comp
    U = np.full((1, H // 2, W // 2), 128, np.uint8)
    dclip = DeviceClip(torch.from_numpy(frame).to(gpu["dev"]), torch.from_numpy(U).to(gpu["dev"]), torch.from_numpy(U.copy()).to(gpu["dev"]), W, H)
    lf = LogoFrame(gpu["ctx"], [cs["logo"]], 0.35)
    lf.scanFrames(dclip)
    r = lf.evalResults[0, 0]
    assert 0.7 < r[0] <= 1.05 and abs(r[1]) < 0.2
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    before = dclip.Y.clone()
    er.erase(dclip, np.zeros((1, 2), np.float32))
    assert torch.equal(before, dclip.Y)


@pytest.mark.parametrize("with_logof", [False, True])
@pytest.mark.parametrize("bits", [8, 10])
def test_erase_logo_bit_exact(gpu, with_logof, bits):
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo
    cs = make_case(gpu, SMALL, bits=bits, pitch_pad=32)
    orc, clip, cfg = cs["orc"], cs["clip"], cs["cfg"]
    n = clip["Y"].shape[0]
    text = ("    14 S 0 ALL     12     17\n    20 E 0 ALL     18     23\n    30 S 0 ALL     29     33\n"
            "    39 E 0 ALL     38     39\n") if with_logof else ""
    analysis = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(cs["dclip"])
    er = AMTEraseLogo(gpu["ctx"], cs["logo"], text, 0, 16)
    fades = er.calc_fades(analysis, n)
    er.erase(cs["dclip"], fades)
    # oracle
    d, t, b = oracle_eval_logos(orc, cs["lo"])
    Y, U, V = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    an = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, n, _ptr(an))
    fr = np.zeros(n, np.int32)
    if with_logof:
        assert orc.lib.orc_read_logoframe(text.encode(), n, _ptr(fr)) == 0
    want_f = np.zeros((n, 2), np.float32)
    for i in range(n):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(_ptr(fr), 1 if with_logof else 0, 16, _ptr(an), n, i, C.byref(ft), C.byref(fb))
        want_f[i] = (ft.value, fb.value)
        orc.lib.orc_erase_frame(cs["lo"], _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], bits, ft.value, fb.value)
    assert fades.tobytes() == want_f.tobytes()
    view = (lambda x: x.cpu().numpy().view(np.uint16)) if bits > 8 else (lambda x: x.cpu().numpy())
    assert np.array_equal(view(cs["dclip"].Y), Y)
    assert np.array_equal(view(cs["dclip"].U), U)
    assert np.array_equal(view(cs["dclip"].V), V)


def test_calc_fades_abrupt_and_edges(gpu):
    """CalcFade2's abrupt-change branch (per-field fades) and the n+2i sampling near the clip ends
    (LogoScan.hpp:1263-1315) on crafted analysis records."""
    from amatsukaze_amd import AMTEraseLogo
    cs = make_case(gpu, SMALL)
    orc = cs["orc"]
    rng = np.random.RandomState(11)
    n = 45
    an = rng.rand(n, 33).astype(np.float32) + 0.5
    best = np.zeros(n, np.int64)
    best[:20] = 0; best[20:] = 10            # logo switches on abruptly at frame 20
    for i in range(n):
        an[i, best[i]] = 0.01
        an[i, 11 + (3 if i == 20 else best[i])] = 0.001      # top field of the switching frame half way
        an[i, 22 + (9 if i == 20 else best[i])] = 0.002
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    got = er.calc_fades(an, n)
    want = np.zeros((n, 2), np.float32)
    for i in range(n):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(None, 0, 16, _ptr(an), n, i, C.byref(ft), C.byref(fb))
        want[i] = (ft.value, fb.value)
    assert got.tobytes() == want.tobytes()
    assert any(a != b for a, b in want.tolist())            # the abrupt branch fired
    # partial ranges give the same answers
    assert er.calc_fades(an, n, first=17, nframes=9).tobytes() == want[17:26].tobytes()


def test_erase_field_mode_odd_chroma_rows(gpu):
    """field mode with an odd number of chroma rows leaves the last chroma row untouched (LogoScan.hpp:1392-1396)."""
    from amatsukaze_amd import AMTEraseLogo
    cfg = dict(SMALL, LH=50, IMGY=16, N=3)
    cs = make_case(gpu, cfg)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    fades = np.array([[0.3, 0.8], [1.0, 0.0], [0.5, 0.5]], np.float32)
    er.erase(cs["dclip"], fades)
    Y, U, V = (cs["clip"][k].copy() for k in "YUV")
    for i in range(3):
        cs["orc"].lib.orc_erase_frame(cs["lo"], _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], 8, float(fades[i, 0]), float(fades[i, 1]))
    assert np.array_equal(cs["dclip"].Y.cpu().numpy(), Y)
    assert np.array_equal(cs["dclip"].U.cpu().numpy(), U)
    assert np.array_equal(cs["dclip"].V.cpu().numpy(), V)


def test_erase_fade0_clamps_out_of_range_10bit_samples(gpu):
    """Delogo with fade 0 is the identity only for samples <= maxv: min(tmp + 0.5, maxv) (LogoScan.hpp:1258) clamps container values
    above 1023 of a 10-bit clip, so fade-0 frames may be skipped at 8 and 16 bits only."""
    from amatsukaze_amd import AMTEraseLogo
    torch = gpu["torch"]
    cfg = dict(SMALL, N=3)
    cs = make_case(gpu, cfg, bits=10)
    X, Y0, LW, LH = cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"]
    Y, U, V = (cs["clip"][k].copy() for k in "YUV")
    Y[:, Y0 + 3:Y0 + 9, X + 5:X + 40] = 3000                   # out-of-range container values inside the rectangle
    U[:, Y0 // 2 + 2, X // 2 + 4:X // 2 + 20] = 60000
    d = cs["dclip"]
    d.Y.copy_(torch.from_numpy(Y.view(np.int16)).to(gpu["dev"]))
    d.U.copy_(torch.from_numpy(U.view(np.int16)).to(gpu["dev"]))
    fades = np.array([[0.0, 0.0], [0.4, 0.0], [0.0, 0.0]], np.float32)
    AMTEraseLogo(gpu["ctx"], cs["logo"]).erase(d, fades)
    for i in range(3):
        cs["orc"].lib.orc_erase_frame(cs["lo"], _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], 10, float(fades[i, 0]), float(fades[i, 1]))
    assert Y[0, Y0 + 3, X + 5] == 1023                         # the reference did clamp
    view = lambda x: x.cpu().numpy().view(np.uint16)
    assert np.array_equal(view(d.Y), Y) and np.array_equal(view(d.U), U) and np.array_equal(view(d.V), V)


@pytest.mark.parametrize("bits,imgy", [(8, 16), (10, 18)])
def test_erase_rectangle_only_planes(gpu, bits, imgy):
    """amtgpu_erase_rect_batch on planes that hold only the logo rectangle == the oracle's Delogo on whole frames, including the
    chroma row parity that depends on the rectangle's position in the frame (LogoScan.hpp:1374-1397); the strided upload /
    download pair moves only the rectangle's rows between a host frame and the device."""
    import ctypes as C
    from amatsukaze_amd import AMTEraseLogo
    torch = gpu["torch"]
    cfg = dict(SMALL, IMGY=imgy, N=4)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=32)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    X, Y0, LW, LH = cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"]
    assert er.rect == (X, Y0, LW, LH, 1)
    fades = np.array([[0.3, 0.8], [1.0, 1.0], [0.0, 0.0], [0.5, 0.0]], np.float32)
    Y, U, V = (cs["clip"][k].copy() for k in "YUV")
    for i in range(4):
        cs["orc"].lib.orc_erase_frame(cs["lo"], _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], bits, float(fades[i, 0]), float(fades[i, 1]))
    d = cs["dclip"]
    rY = d.Y[:, Y0:Y0 + LH, X:X + LW].contiguous()
    rU = d.U[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2].contiguous()
    rV = d.V[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2].contiguous()
    er.erase_rect(rY, rU, rV, bits, fades)
    view = (lambda x: x.cpu().numpy().view(np.uint16)) if bits > 8 else (lambda x: x.cpu().numpy())
    assert np.array_equal(view(rY), Y[:, Y0:Y0 + LH, X:X + LW])
    assert np.array_equal(view(rU), U[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2])
    assert np.array_equal(view(rV), V[:, Y0 // 2:(Y0 + LH) // 2, X // 2:(X + LW) // 2])
    # host frame -> rectangle rows up, erase, rows back: the per-frame path of include/amt_filters.hpp
    lib, ctx = gpu["ctx"].lib, gpu["ctx"]
    es = 1 if bits <= 8 else 2
    host = cs["clip"]["Y"][0].copy()
    pitch = host.strides[0]
    dev = torch.zeros(LW * LH * es, dtype=torch.uint8, device=gpu["dev"])
    off = Y0 * pitch + X * es
    ctx.check(lib.amtgpu_frames_upload_strided(ctx.h, C.c_void_p(dev.data_ptr()), LW * es, C.c_void_p(host.ctypes.data + off), pitch, LW * es, LH))
    ctx.check(lib.amtgpu_frames_upload_wait(ctx.h))
    ctx.synchronize()
    got = dev.cpu().numpy().view(host.dtype).reshape(LH, LW)
    assert np.array_equal(got, cs["clip"]["Y"][0][Y0:Y0 + LH, X:X + LW])
    dev.fill_(7)
    ctx.check(lib.amtgpu_download_strided(ctx.h, C.c_void_p(host.ctypes.data + off), pitch, C.c_void_p(dev.data_ptr()), LW * es, LW * es, LH))
    want = cs["clip"]["Y"][0].copy()
    want[Y0:Y0 + LH, X:X + LW] = 0x0707 if es == 2 else 7
    assert np.array_equal(host, want)
    assert not lib.amtgpu_download_strided(ctx.h, C.c_void_p(host.ctypes.data), 4, C.c_void_p(dev.data_ptr()), LW * es, LW * es, 2)


def test_logoscan_sums_and_logo_bit_exact(gpu):
    from amatsukaze_amd import LogoScan
    cs = make_case(gpu, SMALL, pitch_pad=32)
    cfg, orc, clip = cs["cfg"], cs["orc"], cs["clip"]
    LW, LH, X, Y0 = cfg["LW"], cfg["LH"], cfg["IMGX"], cfg["IMGY"]
    scan = LogoScan(gpu["ctx"], LW, LH, 12)
    valid, nacc = scan.add_batch(cs["dclip"], X, Y0)
    so = orc.lib.orc_scan_create(LW, LH, 1, 1, 12)
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    want_valid = []
    for i in range(Y.shape[0]):
        want_valid.append(orc.lib.orc_scan_add_frame_u8(so, Y[i, Y0:, X:].ctypes.data, U[i, Y0 // 2:, X // 2:].ctypes.data,
                                                        V[i, Y0 // 2:, X // 2:].ctypes.data, Y.shape[2], U.shape[2]))
    assert valid.tolist() == want_valid and nacc == sum(want_valid) == scan.nframes and 3 <= nacc < len(want_valid)
    npx = LW * LH + 2 * (LW // 2) * (LH // 2)
    osum = np.zeros(npx * 5)
    orc.lib.orc_scan_sums(so, _ptr(osum))
    osum = osum.reshape(npx, 5)
    s, p = scan.sums()
    s = s.reshape(npx, 3)
    assert np.array_equal(s[:, 0], osum[:, 0].astype(np.int64))      # sumF
    assert np.array_equal(s[:, 1], osum[:, 2].astype(np.int64))      # sumF2
    assert np.array_equal(s[:, 2], osum[:, 4].astype(np.int64))      # sumFB
    assert p[0] == int(osum[0, 1]) and p[1] == int(osum[0, 3])
    assert p[2] == int(osum[LW * LH, 1]) and p[4] == int(osum[LW * LH + (LW // 2) * (LH // 2), 1])
    for clean in (False, True):
        lg = scan.get_logo(255, clean, cfg["W"], cfg["H"], X, Y0)
        lo = orc.lib.orc_scan_get_logo(so, 255, 1 if clean else 0, cfg["W"], cfg["H"], X, Y0)
        assert lg.planes.tobytes() == orc.logo_arrays(lo)[0].tobytes()
    # max_valid / use_mask semantics: stream order cut-off, masked frames never offered
    scan2 = LogoScan(gpu["ctx"], LW, LH, 12)
    v2, n2 = scan2.add_batch(cs["dclip"], X, Y0, max_valid=2)
    assert n2 == 2 and v2.tolist() == [1 if (x and sum(want_valid[:i + 1]) <= 2) else 0 for i, x in enumerate(want_valid)]
    mask = np.array([i % 2 for i in range(len(want_valid))], np.uint8)
    scan3 = LogoScan(gpu["ctx"], LW, LH, 12)
    v3, n3 = scan3.add_batch(cs["dclip"], X, Y0, use_mask=mask)
    assert v3.tolist() == [int(a and b) for a, b in zip(want_valid, mask)]
    # sharding: sums of two halves add up exactly (all-reduce in the multi-GPU path)
    from amatsukaze_amd import DeviceClip
    half = len(want_valid) // 2
    dc = cs["dclip"]
    parts = []
    for sl in (slice(0, half), slice(half, None)):
        sc = LogoScan(gpu["ctx"], LW, LH, 12)
        sc.add_batch(DeviceClip(dc.Y[sl], dc.U[sl], dc.V[sl], dc.width, dc.height), X, Y0)
        parts.append((sc.sums(), sc.nframes))
    assert np.array_equal(parts[0][0][0] + parts[1][0][0], s.reshape(-1))
    assert np.array_equal(parts[0][0][1] + parts[1][0][1], p) and parts[0][1] + parts[1][1] == nacc


def test_scanlogo_pipeline_lgd_identical(gpu, tmp_path):
    from amatsukaze_amd import DeviceClip, ScanLogo
    torch = gpu["torch"]
    cfg = SMALL
    data, alpha, alphaUV = S.make_logo(cfg["LW"], cfg["LH"])
    W, H = cfg["W"], cfg["H"]
    clip = S.make_clip_np(60, W, H, 0x5EED0004, alpha, alphaUV, cfg["IMGX"], cfg["IMGY"], period=20, fade=4, flat_every=2)
    dclip = DeviceClip(*(torch.from_numpy(clip[k]).to(gpu["dev"]) for k in "YUV"), width=W, height=H)
    dst = tmp_path / "gpu.lgd"
    calls = []
    assert ScanLogo(gpu["ctx"], dclip, 1041, dst, cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"], 12, 25,
                    cb=lambda p, a, b, c: calls.append(p) or 1)
    orc = Oracle()
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    nvalid = C.c_int()
    lo = orc.lib.orc_scanlogo(_ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], Y.shape[2], U.shape[2], W, H, Y.shape[0],
                              cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"], 12, 25, 1, C.byref(nvalid), None)
    assert lo and nvalid.value == 25
    want = tmp_path / "orc.lgd"
    assert orc.lib.orc_logo_save(lo, str(want).encode(), b"No Name", 1041) == 1
    assert dst.read_bytes() == want.read_bytes()
    assert calls and calls[-1] == 1.0
    # cancel and "insufficient frames" follow the reference's error convention: 0 + message on the context
    assert not ScanLogo(gpu["ctx"], dclip, 1, tmp_path / "x.lgd", cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"], 12, 25, cb=lambda *a: 0)
    assert b"Cancel" in gpu["ctx"].lib.amtgpu_last_error(gpu["ctx"].h)
    assert not ScanLogo(gpu["ctx"], dclip, 1, tmp_path / "x.lgd", cfg["IMGX"], cfg["IMGY"], cfg["LW"], cfg["LH"], -1, 25)
    assert b"Insufficient logo frames" in gpu["ctx"].lib.amtgpu_last_error(gpu["ctx"].h)


def test_empty_and_error_paths(gpu):
    from amatsukaze_amd import AMTAnalyzeLogo, AmtError, DeviceClip, Logo, LogoFrame
    torch = gpu["torch"]
    cs = make_case(gpu, SMALL)
    ctx = gpu["ctx"]
    lf = LogoFrame(ctx, [cs["logo"]], 0.35)
    empty = DeviceClip(cs["dclip"].Y[:0], cs["dclip"].U[:0], cs["dclip"].V[:0], SMALL["W"], SMALL["H"])
    lf.scanFrames(empty)
    assert lf.evalResults.shape == (0, 1, 2)
    with pytest.raises(AmtError):
        AMTAnalyzeLogo(ctx, "/nonexistent/logo.lgd")
    with pytest.raises(AmtError):
        Logo.load(ctx, "/nonexistent/logo.lgd")
    lf2 = LogoFrame(ctx, ["/nonexistent/a.lgd"], 0.35)      # unreadable files are ignored (LogoScan.hpp:1612-1614)
    lf2.scanFrames(cs["dclip"])
    assert np.all(lf2.evalResults[:, 0, 1] == -1)
    with pytest.raises(AmtError):
        lf.begin(SMALL["W"], SMALL["H"], 8, 4)
        lf.scan_batch(cs["dclip"].Y, 8, 2)                  # range outside the declared clip
    # zero frames through the other passes: nothing is launched, nothing faults
    from amatsukaze_amd import AMTEraseLogo, FrameStats, LogoScan
    for mode in ("exact", "linear"):
        assert AMTAnalyzeLogo(ctx, cs["logo"], 0.35, mode=mode).analyze(empty).shape == (0, 33)
    er = AMTEraseLogo(ctx, cs["logo"])
    assert er.calc_fades(np.zeros((0, 33), np.float32), 0).shape == (0, 2)
    er.erase(empty, np.zeros((0, 2), np.float32))
    assert FrameStats(ctx, SMALL["W"], SMALL["H"], 8).run(empty).shape == (0, 8)
    scan = LogoScan(ctx, SMALL["LW"], SMALL["LH"], 12)
    valid, nacc = scan.add_batch(empty, SMALL["IMGX"], SMALL["IMGY"])
    assert len(valid) == 0 and nacc == 0 and scan.nframes == 0
    ctx.synchronize()


def test_score_bin_edge_means(gpu):
    """CorrelationScore picks its scale row by (int)avg clamped to [0,255] >> 3 (LogoScan.hpp:304) with x86 conversion
    semantics; window means outside [0,256) -- negative, huge, infinite -- come from absurd but legal logo planes.  The kernel's
    5-instruction bin must land on the same row as the oracle for all of them (whole outputs compared, NaNs included)."""
    from amatsukaze_amd import AMTAnalyzeLogo, Logo
    cfg = dict(SMALL, N=6)
    for scale_a, scale_b in ((1.0, -40.0), (1.0, 300.0), (3.0e7, 0.0), (1.0, 3.0e37), (-1.0e30, 1.0e30)):
        cs = make_case(gpu, cfg, bits=8)
        data = cs["data"].copy()
        n = cfg["LW"] * cfg["LH"]
        data[:n] *= scale_a                      # A plane of luma
        data[n:2 * n] = data[n:2 * n] * 0 + scale_b / 255.0 * (1 + np.arange(n, dtype=np.float32) % 7)   # B plane
        logo = Logo.from_planes(gpu["ctx"], data, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
        got = AMTAnalyzeLogo(gpu["ctx"], logo, 0.35).analyze(cs["dclip"])
        orc = cs["orc"]
        lo = orc.make_logo(data, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
        d, t, b = oracle_eval_logos(orc, lo)
        Y = cs["clip"]["Y"]
        want = np.zeros(cfg["N"] * 33, np.float32)
        orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, cfg["N"], _ptr(want))
        assert got.reshape(-1).tobytes() == want.tobytes(), (scale_a, scale_b)
        # the scan's two fades on the same logo (pair kernel while the coefficients stay below 1e30, generic kernel beyond)
        from amatsukaze_amd import LogoFrame
        lf = LogoFrame(gpu["ctx"], [logo], 0.35)
        lf.scanFrames(cs["dclip"])
        want2 = np.zeros(cfg["N"] * 2, np.float32)
        orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], 8, cfg["W"], cfg["H"], cfg["N"], _ptr(want2))
        assert lf.evalResults.reshape(-1).tobytes() == want2.tobytes(), (scale_a, scale_b)


def test_experiment_environment_variables_are_ignored(gpu, monkeypatch):
    """round-1 builds let AMTGPU_DBG skip kernel phases (wrong results, return code ok); the release library no longer reads it"""
    from amatsukaze_amd import AMTAnalyzeLogo
    for k, v in (("AMTGPU_DBG", "7"), ("AMTGPU_LDSPAD", "40000"), ("AMTGPU_FPI", "1"), ("AMTGPU_G", "3")):
        monkeypatch.setenv(k, v)
    cs = make_case(gpu, SMALL)
    orc = cs["orc"]
    d, t, b = oracle_eval_logos(orc, cs["lo"])
    Y = cs["clip"]["Y"]
    n = Y.shape[0]
    want = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, n, _ptr(want))
    got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(cs["dclip"])
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("cfgname,bits", [("small", 8), ("small", 10), ("small", 12), ("hd", 8)])
def test_analyze_linear_guarded_mode(gpu, cfgname, bits):
    """AMTGPU_ANALYZE_LINEAR_GUARDED: the 11 fades formed from one evaluation of the source and one of the background window.
    Gate (VERDICT r1 #3 ii): every score within 1e-4 of the oracle's AND inside the library's own error bound, CalcFade outputs
    IDENTICAL (the guard re-evaluates exactly every frame whose argmin is not safe), erased frames identical."""
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo
    cfg = SMALL if cfgname == "small" else HD
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=32)
    orc, ctx = cs["orc"], gpu["ctx"]
    d, t, b = oracle_eval_logos(orc, cs["lo"])
    Y = cs["clip"]["Y"]
    n = Y.shape[0]
    want = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, n, _ptr(want))
    want = want.reshape(n, 33)
    # the linear evaluation by itself (no guard: nothing is re-evaluated exactly, so this is the kernel's own accuracy)
    raw = AMTAnalyzeLogo(ctx, cs["logo"], 0.35, mode="linear_unguarded").analyze(cs["dclip"])
    assert np.abs(raw - want).max() <= 1e-4, float(np.abs(raw - want).max())
    an = AMTAnalyzeLogo(ctx, cs["logo"], 0.35, mode="linear")
    got = an.analyze(cs["dclip"])
    refined = an.last_refined()
    assert 0 <= refined <= n // 4, refined                    # the guard is for the rare close calls, not a crutch
    err = np.abs(got - want)
    bounds = [an.error_bound(k, bits) for k in range(3)]
    assert all(0 < e < 0.05 for e in bounds), bounds
    assert err.max() <= 1e-4, (err.max(), refined)
    for k in range(3):
        assert err[:, 11 * k:11 * k + 11].max() <= bounds[k]
    # integer decisions: identical fades from both records, hence identical erased frames
    er = AMTEraseLogo(ctx, cs["logo"], "", 0, 16)
    f_lin, f_ref = er.calc_fades(got, n), er.calc_fades(want, n)
    assert f_lin.tobytes() == f_ref.tobytes()
    # frames the guard re-evaluated carry the exact record
    amb = np.zeros(n, bool)
    for k in range(3):
        srt = np.sort(want[:, 11 * k:11 * k + 11], axis=1)
        amb |= (srt[:, 1] - srt[:, 0]) <= 1.0 * bounds[k]          # certainly inside the guard's 2x margin
    assert (got[amb] == want[amb]).all()
    exact = AMTAnalyzeLogo(ctx, cs["logo"], 0.35).analyze(cs["dclip"])
    assert exact.tobytes() == want.tobytes()
    print(f"linear mode {cfgname}/{bits}: max err {err.max():.2e}, bounds {bounds}, refined {refined}/{n}")


@pytest.mark.gpu
def test_context_before_torch_touches_the_gpu():
    """A fresh process that creates a Context BEFORE torch has initialised HIP: the library and torch both link libamdhip64 and must end
    up on ONE runtime (torch's, loaded first) -- the other order left torch reporting "No HIP GPUs are available" (api.Context)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from amatsukaze_amd import Context\n"
            "c = Context(0)\n"
            "import torch\n"
            "x = torch.ones(4, device='cuda:0'); torch.cuda.synchronize(); c.synchronize(); print(int(x.sum().item()))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("4"), r.stderr[-800:]


def test_one_context_shared_by_two_host_threads(gpu):
    """AviSynth's MT_NICE_FILTER runs GetFrame of filter instances that share one context on several threads
    (include/amt_filters.hpp); the context serialises them (engine.hpp: the lock taken by every entry point).  Two threads hammer
    the pinned upload ring, the analysis and the scan of one context at once (ctypes releases the GIL during the calls); every
    result must equal the single-threaded one."""
    import ctypes as C
    import threading
    from amatsukaze_amd import AMTAnalyzeLogo, LogoFrame
    torch = gpu["torch"]
    ctx, lib = gpu["ctx"], gpu["ctx"].lib
    cs = make_case(gpu, SMALL)
    an = AMTAnalyzeLogo(ctx, cs["logo"], 0.35)
    want_an = an.analyze(cs["dclip"]).copy()
    lf = LogoFrame(ctx, [cs["logo"]], 0.35)
    lf.scanFrames(cs["dclip"])
    want_scan = lf.evalResults.copy()
    rng = np.random.default_rng(7)
    host = [rng.integers(0, 256, 3 << 20, dtype=np.uint8) for _ in range(2)]      # 3 MB each: several ring slots per upload
    errors = []

    def worker(k):
        try:
            dev = torch.zeros(host[k].size, dtype=torch.uint8, device=gpu["dev"])
            mine_an = AMTAnalyzeLogo(ctx, cs["logo"], 0.35) if k else an
            mine_lf = LogoFrame(ctx, [cs["logo"]], 0.35)
            for it in range(6):
                ctx.check(lib.amtgpu_frames_upload(ctx.h, C.c_void_p(dev.data_ptr()), C.c_void_p(host[k].ctypes.data), host[k].size))
                ctx.check(lib.amtgpu_frames_upload_wait(ctx.h))
                got = mine_an.analyze(cs["dclip"])
                mine_lf.scanFrames(cs["dclip"])
                ctx.synchronize()
                if got.tobytes() != want_an.tobytes() or mine_lf.evalResults.tobytes() != want_scan.tobytes():
                    errors.append((k, it, "records differ"))
                if not np.array_equal(dev.cpu().numpy(), host[k]):
                    errors.append((k, it, "uploaded bytes differ"))
        except Exception as e:      # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
