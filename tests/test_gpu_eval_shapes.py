"""GPU parity of the fused evaluation kernel on the shapes its fast paths special-case: logo widths that are not a
multiple of 4 (ragged last lane of the staging), wider than 256 (second column group), rectangle origins that are even
but not 4-byte aligned (unaligned 4-sample loads), 16-bit samples, several frames per workgroup with a short last group.
Bit-exact against the CPU oracle, like tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

import amt_synth as S
from amtlib import _ptr
from test_gpu_parity import gpu, make_case, oracle_eval_logos  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu

SHAPES = {
    # name: (W, H, LW, LH, IMGX, IMGY, N)
    "w98_ragged": (352, 240, 98, 44, 222, 18, 9),
    "w322_two_groups": (704, 240, 322, 40, 362, 26, 7),
    "origin_mod4_2": (352, 240, 96, 48, 226, 22, 9),
    "tall_narrow": (352, 288, 36, 200, 300, 40, 5),
}


@pytest.mark.parametrize("bits", [8, 10])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_analyze_shapes_bit_exact(gpu, shape, bits):
    from amatsukaze_amd import AMTAnalyzeLogo
    W, H, LW, LH, X, Y0, N = SHAPES[shape]
    cfg = dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=32)
    got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(cs["dclip"])
    d, t, b = oracle_eval_logos(cs["orc"], cs["lo"])
    Y = cs["clip"]["Y"]
    want = np.zeros(N * 33, np.float32)
    cs["orc"].lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, N, _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


def test_many_frames_per_workgroup(gpu):
    """Batches large enough that a workgroup owns several frames (G = min(8, frames * logos / 2048) = 4 here, the last group is
    short): the taps of a band are loaded once and re-used across the group's frames, the linear kernel keeps the band's logo
    coefficients in LDS across them and the scan stages two frames per iteration.  Exact analysis and scan: bytes; linear analysis
    (unguarded): within 1e-4."""
    import ctypes as C
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, DeviceClip, Logo, LogoFrame
    from amtlib import Oracle
    W, H, LW, LH, X, Y0, N = 352, 240, 96, 48, 224, 18, 2801
    data, alpha, alphaUV = S.make_logo(LW, LH)
    clip = S.make_clip_torch(N, W, H, 0x5EED0009, alpha, alphaUV, X, Y0, gpu["dev"], period=40, fade=6, chroma=False)
    Yd = clip["Y"]
    Y = Yd.cpu().numpy()
    ctx = gpu["ctx"]
    logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
    orc = Oracle()
    lo = orc.make_logo(data, LW, LH, W, H, X, Y0)
    d, t, b = oracle_eval_logos(orc, lo)
    want = np.zeros(N * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, N, _ptr(want))
    want = want.reshape(N, 33)
    out = torch.empty((N, 33), dtype=torch.float32, device=gpu["dev"])
    AMTAnalyzeLogo(ctx, logo, 0.35).analyze_device(Yd, 8, out)
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == want.tobytes()
    AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear_unguarded").analyze_device(Yd, 8, out)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-4
    guarded = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear")
    guarded.analyze_device(Yd, 8, out)
    assert guarded.last_refined() <= N // 20
    # the all-frames scan with three logos (G = 4, two frames per iteration)
    others = [S.make_logo(LW, LH, seed=0x10600002 + k, strength=0.5 + 0.3 * k)[0] for k in range(2)]
    logos = [logo] + [Logo.from_planes(ctx, o, LW, LH, W, H, X, Y0) for o in others]
    lf = LogoFrame(ctx, logos, 0.35)
    lf.begin(W, H, 8, N)
    lf.scan_batch(Yd, 8, 0, N)
    hs = [d]
    for o in others:
        h = orc.lib.orc_logo_deint(orc.make_logo(o, LW, LH, W, H, X, Y0)); orc.lib.orc_logo_create_mask(h, 0.35, 1); hs.append(h)
    wscan = np.zeros(N * 3 * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 3)(*hs), 3, _ptr(Y), Y.strides[0], Y.shape[2], 8, W, H, N, _ptr(wscan))
    assert lf.evalResults.tobytes() == wscan.tobytes()


@pytest.mark.parametrize("group,bits", [(4, 8), (5, 10)])
def test_scan_in_ragged_batches_bit_exact(gpu, group, bits):
    """LogoFrame::scanFrames fed in batches of 23 frames (40 = 23 + 17) with two logos; frames-per-workgroup sizes above 1 are
    covered by test_many_frames_per_workgroup (the AMTGPU_G knob set here only exists in instrumented builds)."""
    import ctypes as C
    import amt_synth as S
    from amatsukaze_amd import Logo, LogoFrame
    cfg = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=40, period=16, fade=6, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=0)
    d2, _, _ = S.make_logo(cfg["LW"], cfg["LH"], seed=0x10600002, strength=0.5)
    logo2 = Logo.from_planes(gpu["ctx"], d2, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
    os.environ["AMTGPU_G"] = str(group)
    try:
        lf = LogoFrame(gpu["ctx"], [cs["logo"], logo2], 0.35)
        lf.scanFrames(cs["dclip"], batch=23)
        got = lf.evalResults
    finally:
        del os.environ["AMTGPU_G"]
    orc = cs["orc"]
    lo2 = orc.make_logo(d2, cfg["LW"], cfg["LH"], cfg["W"], cfg["H"], cfg["IMGX"], cfg["IMGY"])
    hs = []
    for l in (cs["lo"], lo2):
        d = orc.lib.orc_logo_deint(l); orc.lib.orc_logo_create_mask(d, 0.35, 1); hs.append(d)
    Y = cs["clip"]["Y"]
    want = np.zeros(cfg["N"] * 2 * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 2)(*hs), 2, _ptr(Y), Y.strides[0], Y.shape[2], bits, cfg["W"], cfg["H"], cfg["N"], _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


@pytest.mark.parametrize("bits", [8, 10])
@pytest.mark.parametrize("shape,maskratio", [("w98_ragged", 0.35), ("origin_mod4_2", 0.35), ("tall_narrow", 0.35), ("w322_two_groups", 0.35),
                                             ("w98_ragged", 1.0), ("origin_mod4_2", 0.02)])
def test_scan_shapes_bit_exact(gpu, shape, maskratio, bits):
    """LogoFrame scan on the shapes the pair kernel's tile staging special-cases (a last unit moved left at a ragged right edge,
    rectangle origins that are even but not 4-byte aligned, bands of many short rows, dense and sparse masks, a logo wider than
    one wave's lanes).  Records are the oracle's bytes."""
    import ctypes as C
    from amatsukaze_amd import LogoFrame
    W, H, LW, LH, X, Y0, N = SHAPES[shape]
    cfg = dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=32)
    ctx = gpu["ctx"]
    lf = LogoFrame(ctx, [cs["logo"]], maskratio)
    ctx.profile(True)
    lf.scanFrames(cs["dclip"])
    got = lf.evalResults
    used = [k for k, (calls, _) in ctx.profile_report().items() if calls]
    ctx.profile(False)
    assert used == ["logo_eval_pair_kernel.scan"], used
    orc = cs["orc"]
    d = orc.lib.orc_logo_deint(cs["lo"]); orc.lib.orc_logo_create_mask(d, maskratio, 1)
    Y = cs["clip"]["Y"]
    want = np.zeros(N * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], bits, W, H, N, _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


def test_logo_without_mask_pixels(gpu):
    """maskratio 0 selects no mask pixel: every score is 0 / blackScore = 0 / 0.  The one-pixel-per-thread kernels (linear analysis,
    pair scan) have no band to walk and hand such logos to the generic kernel; nothing faults and the records are NaN like the
    oracle's."""
    import ctypes as C
    from amatsukaze_amd import AMTAnalyzeLogo, LogoFrame
    cfg = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=5, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg)
    orc = cs["orc"]
    d = orc.lib.orc_logo_deint(cs["lo"]); orc.lib.orc_logo_create_mask(d, 0.0, 1)
    Y = cs["clip"]["Y"]
    want = np.zeros(cfg["N"] * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], 8, cfg["W"], cfg["H"], cfg["N"], _ptr(want))
    assert np.isnan(want).all()
    lf = LogoFrame(gpu["ctx"], [cs["logo"]], 0.0)
    lf.scanFrames(cs["dclip"])
    assert np.isnan(lf.evalResults).all()
    for mode in ("exact", "linear"):
        got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.0, mode=mode).analyze(cs["dclip"])
        assert got.shape == (cfg["N"], 33) and np.isnan(got).all()


@pytest.mark.parametrize("case", ["w576_whole_rows", "w680_column_bands", "w1200_maskratio1", "w680_10bit", "w900_scan"])
def test_logos_of_any_width(gpu, case):
    """The exact kernel's bands hold the 5 rows of a window in an LDS plane of 3072 floats: up to 576 columns a band stages whole
    rows (row pitch 584, three 256-column staging groups); wider logos -- the reference takes any even w x h (LogoScan.hpp:69) -- get
    bands that stage only the columns their windows touch.  Bytes against the oracle, for AMTAnalyzeLogo (exact, the library default)
    and for the generic kernel on the scan (field logos / fades other than {0, 1} take it)."""
    import ctypes as C
    from amatsukaze_amd import AMTAnalyzeLogo, LogoFrame
    W, H, LW, LH, X, Y0, N, bits, ratio = {"w576_whole_rows": (704, 96, 576, 24, 100, 30, 5, 8, 0.35),
                                           "w680_column_bands": (800, 96, 680, 24, 60, 30, 4, 8, 0.35),
                                           "w1200_maskratio1": (1440, 64, 1200, 16, 120, 20, 3, 8, 1.0),
                                           "w680_10bit": (800, 96, 680, 24, 62, 30, 3, 10, 0.35),
                                           "w900_scan": (1024, 80, 900, 20, 64, 28, 4, 8, 0.35)}[case]
    cfg = dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=0)
    Y = cs["clip"]["Y"]
    d, t, b = oracle_eval_logos(cs["orc"], cs["lo"], ratio)
    if case == "w900_scan":
        lf = LogoFrame(gpu["ctx"], [cs["logo"]], ratio)
        lf.scanFrames(cs["dclip"])
        want = np.zeros(N * 2, np.float32)
        cs["orc"].lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], bits, W, H, N, _ptr(want))
        assert lf.evalResults.reshape(-1).tobytes() == want.tobytes()
        return
    got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], ratio).analyze(cs["dclip"])
    want = np.zeros(N * 33, np.float32)
    cs["orc"].lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, N, _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


@pytest.mark.parametrize("maskratio", [0.02, 1.0])
def test_mask_ratio_extremes(gpu, maskratio):
    """maskratio 1.0 selects every pixel (mask pixels = w*h, LogoScan.hpp:172; runs are whole rows); 0.02 leaves a few dozen."""
    from amatsukaze_amd import AMTAnalyzeLogo
    cfg = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=6, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=8, pitch_pad=0)
    got = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], maskratio).analyze(cs["dclip"])
    d, t, b = oracle_eval_logos(cs["orc"], cs["lo"], maskratio)
    Y = cs["clip"]["Y"]
    want = np.zeros(cfg["N"] * 33, np.float32)
    cs["orc"].lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, cfg["N"], _ptr(want))
    assert got.reshape(-1).tobytes() == want.tobytes()


@pytest.mark.parametrize("bits", [8, 10, 16])
@pytest.mark.parametrize("shape,maskratio", [("w98_ragged", 0.35), ("origin_mod4_2", 0.35), ("tall_narrow", 0.35), ("w98_ragged", 1.0),
                                             ("origin_mod4_2", 0.02)])
def test_analyze_linear_mode_shapes(gpu, shape, maskratio, bits):
    """The guarded linear mode on the shapes its staging special-cases: a ragged right edge (w % 4 == 2: the last lane group is
    shifted left), an origin that is not 4-byte aligned, a narrow tall logo (short LDS rows hold the LDS-direct raw rows, bands
    capped at 16 rows), every pixel / a few dozen pixels in the mask, 16-bit samples.  Scores within 1e-4 and inside the
    library's bound; fades identical."""
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo
    W, H, LW, LH, X, Y0, N = SHAPES[shape]
    cfg = dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=32)
    an = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], maskratio, mode="linear")
    got = an.analyze(cs["dclip"])
    raw = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], maskratio, mode="linear_unguarded").analyze(cs["dclip"])
    d, t, b = oracle_eval_logos(cs["orc"], cs["lo"], maskratio)
    Y = cs["clip"]["Y"]
    want = np.zeros(N * 33, np.float32)
    cs["orc"].lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, N, _ptr(want))
    want = want.reshape(N, 33)
    err = np.abs(got - want)
    tol = 1e-4 * np.maximum(1.0, np.abs(want))            # scores of a 16-bit clip evaluated on the 8-bit scale run to hundreds
    assert (err <= tol).all(), (float(err.max()), an.last_refined())
    assert (np.abs(raw - want) <= tol).all(), float(np.abs(raw - want).max())     # the kernel's own accuracy, nothing re-evaluated
    for k in range(3):
        assert err[:, 11 * k:11 * k + 11].max() <= an.error_bound(k, bits) * max(1.0, float(np.abs(want).max()))
    er = AMTEraseLogo(gpu["ctx"], cs["logo"], "", 0, 16)
    assert er.calc_fades(got, N).tobytes() == er.calc_fades(want, N).tobytes()


def test_linear_mode_takes_wide_logos(gpu):
    """a logo wider than one wave's lanes (the tile plans have no width limit): linear mode within its own error bound of the
    exact mode's records, which are the oracle's bytes"""
    from amatsukaze_amd import AMTAnalyzeLogo
    W, H, LW, LH, X, Y0, N = SHAPES["w322_two_groups"]
    cs = make_case(gpu, dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3), bits=8, pitch_pad=0)
    an = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35, mode="linear_unguarded")
    got = an.analyze(cs["dclip"])
    want = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(cs["dclip"])
    assert got.shape == want.shape == (N, 33)
    for k in range(3):
        assert np.abs(got - want)[:, 11 * k:11 * k + 11].max() <= an.error_bound(k, 8) * max(1.0, float(np.abs(want).max()))


def test_linear_mode_hands_out_of_range_samples_to_the_exact_kernel(gpu):
    """A 10-bit clip in 16-bit containers may carry values above 1023; the linear mode's error bound (and the margins of its bin test and
    decision guard) assume samples <= maxv, so frames whose logo rectangle holds such a sample are re-evaluated by the exact kernel
    whatever their scores say: their records are the exact mode's bytes, and the guard counts them."""
    from amatsukaze_amd import AMTAnalyzeLogo, DeviceClip
    torch = gpu["torch"]
    cfg = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=24, period=6, fade=3, flat=3)
    cs = make_case(gpu, cfg, bits=10, pitch_pad=32)
    dc = cs["dclip"]
    Y = dc.Y.clone()
    dirty = [3, 4, 11, 23]
    for n in dirty:                                   # one sample inside the rectangle above maxv (and one outside it, which nothing reads)
        Y[n, cfg["IMGY"] + 5 + n % 7, cfg["IMGX"] + 9 + 2 * n] = 3000 + n
    Y[7, 2, 3] = 4000
    clip = DeviceClip(Y, dc.U, dc.V, dc.width, dc.height, 10)
    exact = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35).analyze(clip)
    an = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35, mode="linear")
    got = an.analyze(clip)
    assert an.last_refined() >= len(dirty)
    for n in dirty:
        assert got[n].tobytes() == exact[n].tobytes(), n
    clean = [n for n in range(cfg["N"]) if n not in dirty]
    assert np.abs(got[clean] - exact[clean]).max() <= 1e-4
    # an 8-bit clip never pays for the pre-pass and a clean 10-bit clip is not sent to the exact kernel wholesale
    an2 = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35, mode="linear")
    an2.analyze(cs["dclip"])
    assert an2.last_refined() < cfg["N"] // 2


@pytest.mark.parametrize("bits,maskratio", [(8, 0.35), (16, 1.0), (10, 0.6)])
def test_linear_mode_under_adversarial_coefficients(gpu, bits, maskratio):
    """Logos the bound has to work hardest for: alpha up to 0.9 (|a| + |b| ~ 19 instead of ~ 4: window values twenty times the
    sample range, the bin test's fixed point loses four bits), every pixel in the mask, 16-bit samples.  The unguarded kernel's error
    must stay inside the library's own rigorous bound, the guarded mode's fades must be the exact mode's."""
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, DeviceClip, Logo
    torch = gpu["torch"]
    W, H, LW, LH, X, Y0, N = 352, 240, 96, 48, 224, 18, 16
    data, alpha, alphaUV = S.make_logo(LW, LH, strength=1.5)
    assert np.abs(data[:LW * LH]).max() > 9.0                    # a = 1 / (1 - alpha) up to 10
    clip = S.make_clip_np(N, W, H, 0x5EED0031, alpha, alphaUV, X, Y0, bits=bits, period=5, fade=3, flat_every=4)
    tdt = torch.uint8 if bits <= 8 else torch.int16
    dclip = DeviceClip(*(torch.from_numpy(clip[k].view(np.uint8 if bits <= 8 else np.int16)).to(gpu["dev"]).to(tdt) for k in "YUV"), width=W, height=H, bits=bits)
    logo = Logo.from_planes(gpu["ctx"], data, LW, LH, W, H, X, Y0)
    exact = AMTAnalyzeLogo(gpu["ctx"], logo, maskratio).analyze(dclip)
    raw_an = AMTAnalyzeLogo(gpu["ctx"], logo, maskratio, mode="linear_unguarded")
    raw = raw_an.analyze(dclip)
    scale = max(1.0, float(np.abs(exact).max()))
    for k in range(3):
        err = float(np.abs(raw - exact)[:, 11 * k:11 * k + 11].max())
        assert err <= raw_an.error_bound(k, bits) * scale, (k, err, raw_an.error_bound(k, bits))
    guarded = AMTAnalyzeLogo(gpu["ctx"], logo, maskratio, mode="linear").analyze(dclip)
    er = AMTEraseLogo(gpu["ctx"], logo, "", 0, 16)
    assert er.calc_fades(guarded, N).tobytes() == er.calc_fades(exact, N).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 10])
def test_linear_mode_at_its_largest_frame_group(gpu, bits):
    """Enough frames that the linear kernel's workgroups take their maximum of 7 frames (frames x 3 logos / 2048 >= 7) with a short last
    group (5003 = 714 x 7 + 5): the guarded mode against the exact GPU kernel -- records within 1e-4 and inside the kernel's own bound,
    identical CalcFade output for every frame -- at both sample sizes (four waves per SIMD each since round 5; the 16-bit kernel parks a
    register outside its loops)."""
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Logo
    W, H, LW, LH, X, Y0, N = 352, 240, 96, 48, 224, 18, 5003
    data, alpha, alphaUV = S.make_logo(LW, LH)
    clip = S.make_clip_torch(N, W, H, 0x5EED0019, alpha, alphaUV, X, Y0, gpu["dev"], period=61, fade=7, chroma=False, bits=bits)
    Yd = clip["Y"]
    ctx = gpu["ctx"]
    logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
    exact = torch.empty((N, 33), dtype=torch.float32, device=gpu["dev"])
    lin = torch.empty_like(exact)
    AMTAnalyzeLogo(ctx, logo, 0.35).analyze_device(Yd, bits, exact)
    an = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear")
    an.analyze_device(Yd, bits, lin)
    torch.cuda.synchronize()
    e, l = exact.cpu().numpy(), lin.cpu().numpy()
    err = np.abs(e - l).reshape(N, 3, 11).max(axis=(0, 2))
    assert err.max() <= 1e-4, err
    assert all(err[k] <= an.error_bound(k, bits) for k in range(3)), (err, [an.error_bound(k, bits) for k in range(3)])
    assert an.last_refined() <= N // 20
    er = AMTEraseLogo(ctx, logo, "", 0, 16)
    assert er.calc_fades(e, N).tobytes() == er.calc_fades(l, N).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 10])
def test_linear_mode_bin_check_list(gpu, bits):
    """The linear kernel lists the (pixel, frame, fade) pairs whose interpolated mean lies within its error bound of a bin edge and
    settles them from the frame itself once a workgroup has finished (LogoScan.hpp:304 is discontinuous there).  (1) The length of the
    list is a tuning knob: records identical for every length that takes all pairs.  (2) A list that is too short makes the workgroup
    leave its frames to the exact kernel: with 16 entries many workgroups overflow; their frames are counted as refined and carry the
    exact mode's bytes, every other frame is unchanged.  (3) Without the check the tentative bins leave errors an order of magnitude above the checked
    kernel's: the check is alive (every second listed pair changes its bin)."""
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, Logo
    W, H, LW, LH, X, Y0, N = 720, 480, 160, 80, 500, 30, 2100
    data, alpha, alphaUV = S.make_logo(LW, LH, seed=0x10600071)
    clip = S.make_clip_torch(N, W, H, 0x5EED0071, alpha, alphaUV, X, Y0, gpu["dev"], period=53, fade=6, chroma=False, bits=bits)
    Yd, ctx = clip["Y"], gpu["ctx"]
    logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
    run = lambda an: (lambda o: (an.analyze_device(Yd, bits, o), torch.cuda.synchronize(), o.cpu().numpy())[2])(torch.empty((N, 33), dtype=torch.float32, device=gpu["dev"]))
    exact = run(AMTAnalyzeLogo(ctx, logo, 0.35))
    raw = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear_unguarded")
    base = run(raw)
    assert np.abs(base - exact).max() <= 2e-5
    for entries in (384, 640):
        raw.set_fixup_queue(entries)
        assert run(raw).tobytes() == base.tobytes(), entries
    guarded = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear")
    g = run(guarded)
    assert guarded.last_refined() <= N // 20 and np.abs(g - exact).max() <= 2e-5
    guarded.set_fixup_queue(16)
    g16 = run(guarded)
    r16 = guarded.last_refined()
    as_exact = (g16 == exact).all(axis=1) | np.isnan(g16).all(axis=1)
    as_before = (g16 == g).all(axis=1)
    assert r16 > N // 4 and (as_exact | as_before).all() and as_exact.sum() >= r16, (r16, int(as_exact.sum()), int(as_before.sum()))
    with pytest.raises(Exception):
        guarded.set_fixup_queue(641)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 10])
def test_linear_mode_flat_frames_on_bin_edges(gpu, bits):
    """The worst input for the linear kernel's bin check: frames without a logo whose every sample is the same multiple of 8 gray levels
    (black frames at 16, the CM boundaries of real broadcasts) -- the window mean of `s` sits exactly ON an edge of CorrelationScore's
    bins (LogoScan.hpp:304) at every mask pixel, and wherever the logo is weak the blends' means stay there.  Whatever the lists hold or
    overflow, the guarded mode must hand out the exact mode's decisions: scores within 1e-4, fades identical; a short list (16 entries) only
    moves frames to the exact kernel."""
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Logo
    W, H, LW, LH, X, Y0 = 720, 480, 160, 80, 500, 30
    sh = bits - 8
    levels = [8 * k for k in range(0, 32)] + [16, 16, 16, 235, 128, 17, 15, 255]
    N = len(levels) * 2
    data, alpha, alphaUV = S.make_logo(LW, LH, seed=0x10600072)
    clip = S.make_clip_torch(N, W, H, 0x5EED0072, alpha, alphaUV, X, Y0, gpu["dev"], period=7, fade=2, chroma=False, bits=bits)
    Yd, ctx = clip["Y"], gpu["ctx"]
    for i, lv in enumerate(levels):
        Yd[2 * i].fill_(lv << sh)                                  # (odd frames keep the synthetic picture, logo coming and going)
    logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
    run = lambda an: (lambda o: (an.analyze_device(Yd, bits, o), torch.cuda.synchronize(), o.cpu().numpy())[2])(torch.empty((N, 33), dtype=torch.float32, device=gpu["dev"]))
    exact = run(AMTAnalyzeLogo(ctx, logo, 0.35))
    er = AMTEraseLogo(ctx, logo, "", maxfade=16)
    fe = er.calc_fades(exact, N).tobytes()
    guarded = AMTAnalyzeLogo(ctx, logo, 0.35, mode="linear")
    for entries in (256, 16, 640):
        guarded.set_fixup_queue(entries)
        g = run(guarded)
        assert np.isfinite(g).all() and np.abs(g - exact).max() <= 1e-4, (entries, float(np.abs(g - exact).max()))
        assert er.calc_fades(g, N).tobytes() == fe, entries
        assert 0 <= guarded.last_refined() <= N


@pytest.mark.parametrize("bits", [8, 10])
@pytest.mark.parametrize("corner", ["top_left", "bottom_right"])
def test_logo_in_a_frame_corner_unpadded_planes(gpu, corner, bits):
    """The rectangle at the very first / the very last samples of planes WITHOUT row padding (pitch == width): the tiles' staging units, the
    [1 2 1] rows above and below, the linear kernel's exact-mean loads (aligned words around five samples) and Delogo's paired accesses
    all touch the first / last bytes of the batch.  Scan records, exact analysis records and erased planes are the oracle's bytes; the
    linear-guarded records are within 1e-4 with identical fades."""
    import ctypes as C
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, LogoFrame
    W, H, LW, LH, N = 352, 240, 96, 48, 9
    X, Y0 = (0, 0) if corner == "top_left" else (W - LW, H - LH)
    cfg = dict(W=W, H=H, LW=LW, LH=LH, IMGX=X, IMGY=Y0, N=N, period=4, fade=2, flat=3)
    cs = make_case(gpu, cfg, bits=bits, pitch_pad=0)
    ctx, orc, clip = gpu["ctx"], cs["orc"], cs["clip"]
    Y, U, V = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    assert Y.shape[2] == W
    d, t, b = oracle_eval_logos(orc, cs["lo"])
    # scan
    lf = LogoFrame(ctx, [cs["logo"]], 0.35)
    lf.scanFrames(cs["dclip"])
    want_scan = np.zeros(N * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], bits, W, H, N, _ptr(want_scan))
    assert lf.evalResults.reshape(-1).tobytes() == want_scan.tobytes()
    # analysis: exact bytes, linear-guarded within tolerance
    want = np.zeros(N * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, N, _ptr(want))
    got = AMTAnalyzeLogo(ctx, cs["logo"], 0.35).analyze(cs["dclip"])
    assert got.reshape(-1).tobytes() == want.tobytes()
    lin = AMTAnalyzeLogo(ctx, cs["logo"], 0.35, mode="linear")
    lin.set_fixup_queue(16)                                    # (short lists: more pairs go through the exact-mean loads' neighbours)
    gl = lin.analyze(cs["dclip"])
    assert (np.abs(gl.reshape(-1) - want) <= 1e-4 * np.maximum(1.0, np.abs(want))).all()
    er = AMTEraseLogo(ctx, cs["logo"], "", 0, 16)
    fades = er.calc_fades(got, N)
    assert er.calc_fades(gl, N).tobytes() == fades.tobytes()
    # erase
    er.erase(cs["dclip"], fades)
    for i in range(N):
        orc.lib.orc_erase_frame(cs["lo"], _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], bits, float(fades[i, 0]), float(fades[i, 1]))
    view = (lambda x: x.cpu().numpy().view(np.uint16)) if bits > 8 else (lambda x: x.cpu().numpy())
    assert np.array_equal(view(cs["dclip"].Y), Y) and np.array_equal(view(cs["dclip"].U), U) and np.array_equal(view(cs["dclip"].V), V)
    assert np.abs(fades).sum() > 0
