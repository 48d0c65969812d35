"""Pins the CPU oracle (oracle/amt_oracle.cpp) against the REAL reference sources compiled through the
shim (oracle/_ref/libamt_ref.so, built by oracle/build_ref.sh from /root/reference).  Bit-exact on
every output: both sides run IEEE fp32 without contraction on the same host.

Skipped where the reference build is absent and cannot be produced (no /root/reference); the committed
golden vectors (tests/golden/, produced from the same reference build) cover that case.
"""
import ctypes as C
import os

import numpy as np
import pytest

from amtlib import Oracle, Ref, _ptr, write_raw_clip
import amt_synth as S

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref/libamt_ref.so not built")

W, H = 352, 240          # small frame for CPU speed
LW, LH = 96, 48
IMGX, IMGY = 224, 18     # imgy/2 odd -> exercises the chroma parity paths


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("ref")
    orc, ref = Oracle(), Ref()
    data, alpha, alphaUV = S.make_logo(LW, LH)
    lo = orc.make_logo(data, LW, LH, W, H, IMGX, IMGY)
    path = str(tmp / "logo.lgd").encode()
    assert orc.lib.orc_logo_save(lo, path, b"synthetic", 1041) == 1
    clip = S.make_clip_np(40, W, H, 0x5EED0001, alpha, alphaUV, IMGX, IMGY, period=16, fade=6, flat_every=3)
    return dict(tmp=tmp, orc=orc, ref=ref, data=data, alpha=alpha, alphaUV=alphaUV, lo=lo, path=path, clip=clip)


def test_corr5x5_both_orders(env):
    orc, ref = env["orc"], env["ref"]
    assert ref.lib.ref_is_avx() == 1
    rng = np.random.RandomState(1)
    w = 40
    Y = (rng.rand(w * 20 + 8) * 255).astype(np.float32)
    k = (rng.randn(25 + 8)).astype(np.float32)
    for x, y in [(2, 2), (17, 9), (37, 17), (5, 11)]:
        for name in ("scalar", "avx"):
            a1, a2 = C.c_float(), C.c_float()
            r1 = getattr(orc.lib, "orc_corr5x5_" + name)(_ptr(k), _ptr(Y), x, y, w, C.byref(a1))
            r2 = getattr(ref.lib, "ref_corr5x5_" + name)(_ptr(k), _ptr(Y), x, y, w, C.byref(a2))
            assert np.float32(r1).tobytes() == np.float32(r2).tobytes()
            assert np.float32(a1.value).tobytes() == np.float32(a2.value).tobytes()
    # the author's own disabled cross-check (LogoScan.hpp:47-57): both orders agree closely
    a = C.c_float()
    s = orc.lib.orc_corr5x5_scalar(_ptr(k), _ptr(Y), 17, 9, w, C.byref(a))
    v = orc.lib.orc_corr5x5_avx(_ptr(k), _ptr(Y), 17, 9, w, C.byref(a))
    assert abs(s - v) <= 1e-4 * max(1.0, abs(s))


def test_lgd_roundtrip_bytes(env):
    """LogoData::Load + Save by the reference reproduces the oracle-written file byte for byte."""
    ref = env["ref"]
    h = ref.lib.ref_logo_load(env["path"])
    assert h, ref.lib.ref_last_error()
    out = str(env["tmp"] / "resaved.lgd").encode()
    assert ref.lib.ref_logo_save(h, out) == 1
    a = open(env["path"], "rb").read()
    b = open(out, "rb").read()
    assert len(a) == 32 + 48 + 12 * LW * LH + 540 + 4 * (LW * LH + 2 * (LW // 2) * (LH // 2)) * 2
    assert a == b
    assert np.array_equal(ref.logo_data(h), env["data"])
    ref.lib.ref_logo_free(h)


@pytest.mark.parametrize("kind", ["deint", "top", "bottom"])
@pytest.mark.parametrize("maskratio", [0.35, 0.1])
def test_create_logo_mask_tables(env, kind, maskratio):
    orc, ref = env["orc"], env["ref"]
    rh = ref.lib.ref_logo_load(env["path"])
    if kind == "deint":
        o2, r2 = orc.lib.orc_logo_deint(env["lo"]), ref.lib.ref_logo_deint(rh)
    else:
        b = 1 if kind == "bottom" else 0
        o2, r2 = orc.lib.orc_logo_field(env["lo"], b), ref.lib.ref_logo_field(rh, b)
    orc.lib.orc_logo_create_mask(o2, maskratio, 1)
    ref.lib.ref_logo_create_mask(r2, maskratio)
    data, mask, ker, sc, black, mp, cnt = orc.logo_arrays(o2)
    assert cnt > 0
    rmask, rker, rsc, rblack, rmp = ref.logo_tables(r2, cnt)
    o, r = orc.logo_info(o2), ref.logo_info(r2)
    assert list(o[:8]) == list(r[:8])
    assert np.array_equal(ref.logo_data(r2)[:2 * o[0] * o[1]], data[:2 * o[0] * o[1]])   # Y planes (deint leaves UV unset)
    assert mp == rmp and np.array_equal(mask, rmask)
    assert ker[:cnt * 25].tobytes() == rker.tobytes()
    assert sc[:cnt * 64].tobytes() == rsc.tobytes()
    assert np.float32(black).tobytes() == np.float32(rblack).tobytes()
    # EvaluateLogo on a few synthetic sources, several fades
    w, h = int(o[0]), int(o[1])
    rng = np.random.RandomState(7)
    src = (rng.rand(w * h * 2 + 8) * 255).astype(np.float32)
    work = np.zeros(w * h + 8, np.float32)
    for fade in (0.0, 0.3, 1.0, 1.9):
        for stride in (-1, 2 * w):
            if stride != -1 and kind == "deint":
                continue
            a = orc.lib.orc_evaluate_logo(o2, _ptr(src), 255.0, fade, _ptr(work), stride)
            b = ref.lib.ref_evaluate_logo(r2, _ptr(src), 255.0, fade, _ptr(work), stride)
            assert np.float32(a).tobytes() == np.float32(b).tobytes()
    ref.lib.ref_logo_free(rh)


def _logoframe_both(env, paths, evals_only=False, ncand=-1, logo_index=0):
    orc, ref, clip = env["orc"], env["ref"], env["clip"]
    Y = clip["Y"]
    n = Y.shape[0]
    nl = len(paths)
    ev_r = np.zeros(n * nl * 2, np.float32)
    best, ratio = C.c_int(), C.c_float()
    text = C.create_string_buffer(1 << 16)
    arr = (C.c_char_p * nl)(*paths)
    ok = ref.lib.ref_logoframe(arr, nl, 0.35, _ptr(Y), Y.strides[0], Y.shape[2], 8, W, H, n, 30000, 1001,
                               _ptr(ev_r), ncand, C.byref(best), C.byref(ratio), logo_index,
                               str(env["tmp"] / "logof.txt").encode(), text, len(text))
    assert ok == 1, ref.lib.ref_last_error()
    handles = []
    for p in paths:
        lo = orc.lib.orc_logo_load(p)
        if not lo:
            handles.append(None)
            continue
        d = orc.lib.orc_logo_deint(lo)
        orc.lib.orc_logo_create_mask(d, 0.35, 1)
        handles.append(d)
    harr = (C.c_void_p * nl)(*handles)
    ev_o = np.zeros(n * nl * 2, np.float32)
    orc.lib.orc_logoframe_scan(harr, nl, _ptr(Y), Y.strides[0], Y.shape[2], 8, W, H, n, _ptr(ev_o))
    obest, oratio = C.c_int(), C.c_float()
    orc.lib.orc_logoframe_select(_ptr(ev_o), n, nl, ncand, C.byref(obest), C.byref(oratio))
    otext = C.create_string_buffer(1 << 16)
    ln = orc.lib.orc_logoframe_write_result(_ptr(ev_o), n, nl, logo_index, 30000, 1001, otext, len(otext))
    assert ln >= 0
    return (ev_r, best.value, ratio.value, text.value), (ev_o, obest.value, oratio.value, otext.value)


def test_logoframe_scan_select_write(env):
    # second candidate: a weaker logo at the same place; third: a file that does not exist (ignored, :1612-1614)
    orc = env["orc"]
    d2, _, _ = S.make_logo(LW, LH, seed=0x10600002, strength=0.5)
    l2 = orc.make_logo(d2, LW, LH, W, H, IMGX, IMGY)
    p2 = str(env["tmp"] / "logo2.lgd").encode()
    assert orc.lib.orc_logo_save(l2, p2, b"other", 7) == 1
    r, o = _logoframe_both(env, [env["path"], p2], ncand=2, logo_index=0)
    assert r[0].tobytes() == o[0].tobytes()
    assert r[1] == o[1] and np.float32(r[2]).tobytes() == np.float32(o[2]).tobytes()
    assert r[3] == o[3] and len(r[3]) > 0, (r[3], o[3])
    assert r[1] == 0                                       # the true logo wins
    # frames where the logo is fully on score corr0 ~ 1 / corr1 ~ 0
    ev = r[0].reshape(-1, 2, 2)
    vis = S.logo_presence(np.arange(ev.shape[0]), 16, 6)
    on = vis >= 1.0
    assert ev[on, 0, 0].mean() > 0.5 and np.abs(ev[on, 0, 1]).mean() < 0.2


def test_analyze_logo(env):
    orc, ref, clip = env["orc"], env["ref"], env["clip"]
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    n = 19   # not a multiple of 8: last analysis frame clamps (:1133)
    out_r = np.zeros(n * 33, np.float32)
    ok = ref.lib.ref_analyze(env["path"], 0.35, _ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0],
                             Y.shape[2], U.shape[2], 8, W, H, n, _ptr(out_r))
    assert ok == 1, ref.lib.ref_last_error()
    lo = env["lo"]
    d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, 0.35, 1)
    t = orc.lib.orc_logo_field(lo, 0); orc.lib.orc_logo_create_mask(t, 0.35, 1)
    b = orc.lib.orc_logo_field(lo, 1); orc.lib.orc_logo_create_mask(b, 0.35, 1)
    out_o = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, n, _ptr(out_o))
    assert out_r.tobytes() == out_o.tobytes()


@pytest.mark.parametrize("with_logof", [False, True])
def test_erase_logo_frames_and_fades(env, with_logof):
    orc, ref, clip = env["orc"], env["ref"], env["clip"]
    n = clip["Y"].shape[0]
    Yr, Ur, Vr = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    logof = b""
    text = b""
    if with_logof:
        text = b"    14 S 0 ALL     12     17\n    20 E 0 ALL     18     23\n    30 S 0 ALL     29     33\n    39 E 0 ALL     38     39\n"
        p = env["tmp"] / "lf.txt"
        p.write_bytes(text)
        logof = str(p).encode()
    fades_r = np.zeros(n * 2, np.float32)
    ok = ref.lib.ref_erase(env["path"], logof, 16, 0.35, _ptr(Yr), _ptr(Ur), _ptr(Vr), Yr.strides[0], Ur.strides[0],
                           Yr.shape[2], Ur.shape[2], 8, W, H, n, _ptr(fades_r))
    assert ok == 1, ref.lib.ref_last_error()
    # oracle: analysis -> CalcFade -> erase
    lo = env["lo"]
    d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, 0.35, 1)
    t = orc.lib.orc_logo_field(lo, 0); orc.lib.orc_logo_create_mask(t, 0.35, 1)
    b = orc.lib.orc_logo_field(lo, 1); orc.lib.orc_logo_create_mask(b, 0.35, 1)
    Y, U, V = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    an = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], 8, n, _ptr(an))
    fr = np.zeros(n, np.int32)
    if with_logof:
        assert orc.lib.orc_read_logoframe(text, n, _ptr(fr)) == 0
    fades_o = np.zeros(n * 2, np.float32)
    for i in range(n):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(_ptr(fr), 1 if with_logof else 0, 16, _ptr(an), n, i, C.byref(ft), C.byref(fb))
        fades_o[2 * i], fades_o[2 * i + 1] = ft.value, fb.value
        orc.lib.orc_erase_frame(lo, _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], 8, ft.value, fb.value)
    assert fades_r.tobytes() == fades_o.tobytes()
    assert np.array_equal(Yr, Y) and np.array_equal(Ur, U) and np.array_equal(Vr, V)
    assert not np.array_equal(Y, clip["Y"])                 # something was erased
    if not with_logof:
        assert len(set(fades_o.tolist())) > 2              # fades vary through the transitions


def test_logoscan_accumulate_and_regress(env):
    orc, ref, clip = env["orc"], env["ref"], env["clip"]
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    so, sr = orc.lib.orc_scan_create(LW, LH, 1, 1, 12), ref.lib.ref_scan_create(LW, LH, 1, 1, 12)
    acc = []
    for i in range(Y.shape[0]):
        y = Y[i, IMGY:, IMGX:]; u = U[i, IMGY // 2:, IMGX // 2:]; v = V[i, IMGY // 2:, IMGX // 2:]
        a = orc.lib.orc_scan_add_frame_u8(so, y.ctypes.data, u.ctypes.data, v.ctypes.data, Y.shape[2], U.shape[2])
        b = ref.lib.ref_scan_add_frame_u8(sr, y.ctypes.data, u.ctypes.data, v.ctypes.data, Y.shape[2], U.shape[2])
        assert a == b
        acc.append(a)
    assert 3 <= sum(acc) < Y.shape[0]
    assert orc.lib.orc_scan_nframes(so) == ref.lib.ref_scan_nframes(sr) == sum(acc)
    npx = LW * LH + 2 * (LW // 2) * (LH // 2)
    s1, s2 = np.zeros(npx * 5), np.zeros(npx * 5)
    orc.lib.orc_scan_sums(so, _ptr(s1)); ref.lib.ref_scan_sums(sr, _ptr(s2))
    assert s1.tobytes() == s2.tobytes()
    for clean in (0, 1):
        lo = orc.lib.orc_scan_get_logo(so, 255, clean, W, H, IMGX, IMGY)
        lr = ref.lib.ref_scan_get_logo(sr, 255, clean, W, H, IMGX, IMGY)
        assert bool(lo) == bool(lr)
        if lo:
            assert orc.logo_arrays(lo)[0].tobytes() == ref.logo_data(lr).tobytes()


def test_scanlogo_full_pipeline(env):
    """The reference's exported ScanLogo() (LogoScan.hpp:1083-1098) end to end vs orc_scanlogo."""
    orc, ref = env["orc"], env["ref"]
    alpha, alphaUV = env["alpha"], env["alphaUV"]
    clip = S.make_clip_np(60, W, H, 0x5EED0004, alpha, alphaUV, IMGX, IMGY, period=20, fade=4, flat_every=2)
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    raw = str(env["tmp"] / "clip.raw").encode()
    write_raw_clip(raw, Y, U, V, W, H)
    dst = str(env["tmp"] / "scanned.lgd").encode()
    ok = ref.lib.ref_scanlogo(raw, 1041, str(env["tmp"] / "work.dat").encode(), dst, IMGX, IMGY, LW, LH, 12, 25)
    assert ok == 1, ref.lib.ref_last_error()
    nvalid = C.c_int()
    lo = orc.lib.orc_scanlogo(_ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], Y.shape[2], U.shape[2],
                              W, H, Y.shape[0], IMGX, IMGY, LW, LH, 12, 25, 1, C.byref(nvalid), None)
    assert lo and nvalid.value == 25
    out = str(env["tmp"] / "scanned_o.lgd").encode()
    assert orc.lib.orc_logo_save(lo, out, b"No Name", 1041) == 1
    assert open(dst, "rb").read() == open(out, "rb").read()
    # the frame-parallel driver (bench.py's full-stream check of BASELINE configs[3]): same bytes for any thread count, and it reports
    # how many frames the stream-order quota (LogoScan.hpp:885) let through -- 25 valid frames are reached before the clip ends
    for threads in (1, 3, 8):
        nv2, nread = C.c_int(), C.c_int()
        mf1, mf2 = np.zeros(25, np.int32), np.zeros(25, np.int32)
        lo1 = orc.lib.orc_scanlogo(_ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], Y.shape[2], U.shape[2],
                                   W, H, Y.shape[0], IMGX, IMGY, LW, LH, 12, 25, 1, None, _ptr(mf1))
        lo2 = orc.lib.orc_scanlogo_mt(_ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], Y.shape[2], U.shape[2],
                                      W, H, Y.shape[0], IMGX, IMGY, LW, LH, 12, 25, 1, C.byref(nv2), _ptr(mf2), threads, C.byref(nread))
        assert lo1 and lo2 and nv2.value == 25 and 25 <= nread.value < Y.shape[0]
        assert mf1.tobytes() == mf2.tobytes()
        out2 = str(env["tmp"] / f"scanned_mt{threads}.lgd").encode()
        assert orc.lib.orc_logo_save(lo2, out2, b"No Name", 1041) == 1
        assert open(dst, "rb").read() == open(out2, "rb").read()
    # the recovered logo resembles the true one where alpha is significant
    data = orc.logo_arrays(lo)[0]
    aY = data[:LW * LH].reshape(LH, LW)
    true_aY = env["data"][:LW * LH].reshape(LH, LW)
    m = alpha > 0.3
    assert np.abs(aY[m] - true_aY[m]).mean() < 0.35


# ------------------------------------------------------------------------------------------------
# > 8-bit clips and HD frame sizes: AMTAnalyzeLogo / AMTEraseLogo are correct for 16-bit containers in the
# reference (LogoScan.hpp:1130-1161 divides the byte pitch, :1343-1400), so the oracle is pinned there too.
# ------------------------------------------------------------------------------------------------
def _oracle_logos(orc, lo, maskratio=0.35):
    d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, maskratio, 1)
    t = orc.lib.orc_logo_field(lo, 0); orc.lib.orc_logo_create_mask(t, maskratio, 1)
    b = orc.lib.orc_logo_field(lo, 1); orc.lib.orc_logo_create_mask(b, maskratio, 1)
    return d, t, b


def _analyze_erase_both(orc, ref, lo, path, clip, w, h, bits, n, logof_text=b"", tmp=None):
    """(reference, oracle) x (analysis, fades, erased Y/U/V) on the same clip"""
    Yr, Ur, Vr = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    an_r = np.zeros(n * 33, np.float32)
    assert ref.lib.ref_analyze(path, 0.35, _ptr(Yr), _ptr(Ur), _ptr(Vr), Yr.strides[0], Ur.strides[0], Yr.shape[2], Ur.shape[2],
                               bits, w, h, n, _ptr(an_r)) == 1, ref.lib.ref_last_error()
    logof = b""
    if logof_text:
        p = tmp / f"lf_{bits}_{w}.txt"
        p.write_bytes(logof_text)
        logof = str(p).encode()
    fades_r = np.zeros(n * 2, np.float32)
    assert ref.lib.ref_erase(path, logof, 16, 0.35, _ptr(Yr), _ptr(Ur), _ptr(Vr), Yr.strides[0], Ur.strides[0], Yr.shape[2], Ur.shape[2],
                             bits, w, h, n, _ptr(fades_r)) == 1, ref.lib.ref_last_error()
    d, t, b = _oracle_logos(orc, lo)
    Y, U, V = clip["Y"].copy(), clip["U"].copy(), clip["V"].copy()
    an_o = np.zeros(n * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, n, _ptr(an_o))
    fr = np.zeros(n, np.int32)
    if logof_text:
        assert orc.lib.orc_read_logoframe(logof_text, n, _ptr(fr)) == 0
    fades_o = np.zeros(n * 2, np.float32)
    for i in range(n):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(_ptr(fr), 1 if logof_text else 0, 16, _ptr(an_o), n, i, C.byref(ft), C.byref(fb))
        fades_o[2 * i], fades_o[2 * i + 1] = ft.value, fb.value
        orc.lib.orc_erase_frame(lo, _ptr(Y[i]), _ptr(U[i]), _ptr(V[i]), Y.shape[2], U.shape[2], bits, ft.value, fb.value)
    return (an_r, fades_r, Yr, Ur, Vr), (an_o, fades_o, Y, U, V)


@pytest.mark.parametrize("bits", [10, 12])
def test_analyze_erase_16bit_containers(env, bits):
    orc, ref = env["orc"], env["ref"]
    n = 27
    clip = S.make_clip_np(n, W, H, 0x5EED0005, env["alpha"], env["alphaUV"], IMGX, IMGY, bits=bits, period=12, fade=5, flat_every=4)
    assert clip["Y"].dtype == np.uint16 and int(clip["Y"].max()) > 255
    r, o = _analyze_erase_both(orc, ref, env["lo"], env["path"], clip, W, H, bits, n)
    assert r[0].tobytes() == o[0].tobytes()                 # 33 floats per frame
    assert r[1].tobytes() == o[1].tobytes()                 # CalcFade / CalcFade2
    assert np.array_equal(r[2], o[2]) and np.array_equal(r[3], o[3]) and np.array_equal(r[4], o[4])
    assert not np.array_equal(o[2], clip["Y"]) and len(set(o[1].tolist())) > 2
    # with a logoframe file on top (ReadLogoFrameFile, :1421-1461)
    text = b"     6 S 0 ALL      4      8\n    11 E 0 ALL     10     12\n    24 S 0 ALL     23     25\n    26 E 0 ALL     26     26\n"
    r, o = _analyze_erase_both(orc, ref, env["lo"], env["path"], clip, W, H, bits, n, text, env["tmp"])
    assert r[1].tobytes() == o[1].tobytes()
    assert np.array_equal(r[2], o[2]) and np.array_equal(r[3], o[3]) and np.array_equal(r[4], o[4])


@pytest.mark.parametrize("bits", [10, 16])
def test_logoframe_scan_16bit_arithmetic_through_the_pitch_quirk(env, bits):
    """LogoFrame::ScanFrame<uint16_t> multiplies the BYTE pitch into a uint16_t pointer (LogoScan.hpp:1547,1561-1562):
    it reads rectangle row y at frame row 2*(imgy+y).  The oracle (and the product) use the element pitch; called with a
    doubled pitch the oracle addresses exactly what the reference does, which pins its 16-bit DeintY / EvaluateLogo / maxv
    arithmetic to the reference's own (the corrected addressing itself is a documented divergence)."""
    orc, ref = env["orc"], env["ref"]
    assert (W * 2) % 64 == 0 and 2 * (IMGY + LH) <= H        # shim frame pitch == 2*W bytes; doubled rows stay inside the frame
    n = 9
    clip = S.make_clip_np(n, W, H, 0x5EED0003, env["alpha"], env["alphaUV"], IMGX, IMGY, bits=bits, period=4, fade=2)
    Y = clip["Y"]
    ev_r = np.zeros(n * 2, np.float32)
    best, ratio = C.c_int(), C.c_float()
    text = C.create_string_buffer(1 << 16)
    ok = ref.lib.ref_logoframe((C.c_char_p * 1)(env["path"]), 1, 0.35, _ptr(Y), Y.strides[0], Y.shape[2], bits, W, H, n, 30000, 1001,
                               _ptr(ev_r), -1, C.byref(best), C.byref(ratio), 0, str(env["tmp"] / "lf16.txt").encode(), text, len(text))
    assert ok == 1, ref.lib.ref_last_error()
    d = orc.lib.orc_logo_deint(env["lo"]); orc.lib.orc_logo_create_mask(d, 0.35, 1)
    ev_o = np.zeros(n * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], 2 * Y.shape[2], bits, W, H, n, _ptr(ev_o))
    assert ev_r.tobytes() == ev_o.tobytes()
    assert np.isfinite(ev_o).all() and len(set(ev_o.tolist())) > n


HD_CASES = [  # (W, H, pitchY, pitchUV, bits, LW, LH, IMGX, IMGY, frames) -- BASELINE configs 2/4 (1440x1080 8-bit) and 5 (1920x1080 10-bit)
    (1440, 1080, 1472, 768, 8, 256, 128, 1120, 64, 10),
    (1920, 1080, 1920, 960, 10, 256, 128, 1600, 64, 6),
]


@pytest.mark.parametrize("case", HD_CASES, ids=["1440x1080_8bit", "1920x1080_10bit"])
def test_hd_frames_scan_analyze_erase(env, case):
    w, h, py, puv, bits, lw, lh, ix, iy, n = case
    orc, ref = env["orc"], env["ref"]
    data, alpha, alphaUV = S.make_logo(lw, lh)
    lo = orc.make_logo(data, lw, lh, w, h, ix, iy)
    path = str(env["tmp"] / f"hd_{w}.lgd").encode()
    assert orc.lib.orc_logo_save(lo, path, b"hd", 1) == 1
    clip = S.make_clip_np(n, w, h, 0x5EED0002, alpha, alphaUV, ix, iy, bits=bits, period=4, fade=3, flat_every=3, pitchY=py, pitchUV=puv)
    r, o = _analyze_erase_both(orc, ref, lo, path, clip, w, h, bits, n)
    assert r[0].tobytes() == o[0].tobytes()
    assert r[1].tobytes() == o[1].tobytes()
    assert np.array_equal(r[2], o[2]) and np.array_equal(r[3], o[3]) and np.array_equal(r[4], o[4])
    if bits == 8:                                            # the all-frames scan (8-bit: no pitch quirk)
        Y = clip["Y"]
        ev_r = np.zeros(n * 2, np.float32)
        best, ratio = C.c_int(), C.c_float()
        text = C.create_string_buffer(1 << 16)
        assert ref.lib.ref_logoframe((C.c_char_p * 1)(path), 1, 0.35, _ptr(Y), Y.strides[0], Y.shape[2], 8, w, h, n, 30000, 1001, _ptr(ev_r),
                                     -1, C.byref(best), C.byref(ratio), 0, str(env["tmp"] / "lfhd.txt").encode(), text, len(text)) == 1
        d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, 0.35, 1)
        ev_o = np.zeros(n * 2, np.float32)
        orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(Y), Y.strides[0], Y.shape[2], 8, w, h, n, _ptr(ev_o))
        assert ev_r.tobytes() == ev_o.tobytes()
