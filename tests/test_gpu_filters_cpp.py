"""The C++ filter layer (include/amt_filters.hpp: AMTAnalyzeLogo, AMTEraseLogo, LogoFrame over the C ABI) driven by a
C++ host program the way the reference's AviSynth host drives its filters, compared byte for byte with the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import amt_synth as S
from amtlib import Oracle, _ptr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "filters_host_test")
PLUGIN = os.path.join(ROOT, "tests", "cpp", "libamt_avs_plugin.so")
CFG = dict(W=352, H=240, LW=96, LH=48, IMGX=224, IMGY=18, N=43, period=16, fade=6, flat=3)


def build_exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "all"], stdout=subprocess.DEVNULL)
    return EXE


@pytest.mark.parametrize("bits", [8, 10])
def test_cpp_filters_match_oracle(tmp_path, bits):
    exe = build_exe()
    cfg = CFG
    W, H, N = cfg["W"], cfg["H"], cfg["N"]
    data, alpha, alphaUV = S.make_logo(cfg["LW"], cfg["LH"])
    data2, _, _ = S.make_logo(cfg["LW"], cfg["LH"], seed=0x10600002, strength=0.5)
    pY, pUV = W + 24, W // 2 + 12
    clip = S.make_clip_np(N, W, H, 0x5EED0007, alpha, alphaUV, cfg["IMGX"], cfg["IMGY"], bits=bits, period=cfg["period"], fade=cfg["fade"],
                          flat_every=cfg["flat"], pitchY=pY, pitchUV=pUV)
    orc = Oracle()
    lo = orc.make_logo(data, cfg["LW"], cfg["LH"], W, H, cfg["IMGX"], cfg["IMGY"])
    lo2 = orc.make_logo(data2, cfg["LW"], cfg["LH"], W, H, cfg["IMGX"], cfg["IMGY"])
    logo1, logo2 = str(tmp_path / "a.lgd"), str(tmp_path / "b.lgd")
    assert orc.lib.orc_logo_save(lo, logo1.encode(), b"A", 1) and orc.lib.orc_logo_save(lo2, logo2.encode(), b"B", 1)
    raw = tmp_path / "clip.raw"
    with open(raw, "wb") as f:
        f.write(np.array([W, H, bits, N, pY, pUV], np.int32).tobytes())
        for k in "YUV":
            f.write(clip[k].tobytes())
    out = tmp_path / "out"
    out.mkdir()
    logof_in = tmp_path / "logof_in.txt"                       # a logoframe file with two logo sections (LogoScan.hpp:1818-1819 format)
    logof_in.write_text("    14 S 0 ALL     12     17\n    20 E 0 ALL     18     23\n    30 S 0 ALL     29     33\n"
                        "    39 E 0 ALL     38     39\n")
    r = subprocess.run([exe, str(raw), logo1, logo2, str(logof_in), str(out), "0", PLUGIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout

    Y, U, V = clip["Y"], clip["U"], clip["V"]
    # ---- LogoFrame: scores, selection, logoframe text ----
    hs = []
    for l in (lo, lo2):
        d = orc.lib.orc_logo_deint(l); orc.lib.orc_logo_create_mask(d, 0.35, 1); hs.append(d)
    want2 = np.zeros(N * 2 * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 2)(*hs), 2, _ptr(Y), Y.strides[0], Y.shape[2], bits, W, H, N, _ptr(want2))
    want = np.zeros((N, 3, 2), np.float32)
    want[:, :2] = want2.reshape(N, 2, 2)
    want[:, 2] = (0, -1)                                       # unreadable logo file: LogoScan.hpp:1612-1614, 1551-1554
    got = np.fromfile(out / "eval.bin", np.float32)
    assert got.tobytes() == want.tobytes()
    best, ratio = C.c_int(), C.c_float()
    orc.lib.orc_logoframe_select(_ptr(want.reshape(-1)), N, 3, 2, C.byref(best), C.byref(ratio))
    sel = (out / "select.txt").read_text().split()
    assert int(sel[0]) == best.value
    assert np.float32(float.fromhex(sel[1])).tobytes() == np.float32(ratio.value).tobytes()
    buf = C.create_string_buffer(1 << 16)
    ln = orc.lib.orc_logoframe_write_result(_ptr(want.reshape(-1)), N, 3, best.value, 30000, 1001, buf, len(buf))
    assert (out / "logof.txt").read_bytes() == buf.raw[:ln]
    # LogoFrame::dumpResult (LogoScan.hpp:1632-1643): "<base><logo>", one "%f,%f\n" line per frame
    for i in range(3):
        assert (out / f"dump_{i}").read_text() == "".join("%f,%f\n" % (float(a), float(b)) for a, b in want[:, i])

    # ---- AMTAnalyzeLogo: BGR32 64x5 clip of ceil(N/8) frames, 8 records each, source frame numbers clamped ----
    assert (out / "analysis_vi.txt").read_text().split() == ["64", "5", str((N + 7) // 8), "100"]
    d, t, b = orc.lib.orc_logo_deint(lo), orc.lib.orc_logo_field(lo, 0), orc.lib.orc_logo_field(lo, 1)
    for h in (d, t, b):
        orc.lib.orc_logo_create_mask(h, 0.35, 1)
    an = np.zeros(N * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(Y), Y.strides[0], Y.shape[2], bits, N, _ptr(an))
    an = an.reshape(N, 33)
    idx = np.minimum(np.arange(((N + 7) // 8) * 8), N - 1)
    assert np.fromfile(out / "analysis.bin", np.float32).tobytes() == an[idx].tobytes()

    # ---- AMTEraseLogo: without and with a logoframe file ----
    text = logof_in.read_bytes()
    dt = Y.dtype
    # erased_plugin.raw: the graph AMTEraseLogo(src, AMTAnalyzeLogo(src, logo), logo) built through the factories that
    # plugin/amt_plugin.cpp registered in AvisynthPluginInit3 (names / argument specs of Amatsukaze.cpp:58-59, defaults undefined)
    for variant, name in ((0, "erased.raw"), (1, "erased_logof.raw"), (0, "erased_plugin.raw")):
        fr = np.zeros(N, np.int32)
        if variant:
            assert orc.lib.orc_read_logoframe(text, N, _ptr(fr)) == 0
        eY, eU, eV = Y.copy(), U.copy(), V.copy()
        for i in range(N):
            ft, fb = C.c_float(), C.c_float()
            orc.lib.orc_calc_fade(_ptr(fr), variant, 16, _ptr(an.reshape(-1)), N, i, C.byref(ft), C.byref(fb))
            orc.lib.orc_erase_frame(lo, _ptr(eY[i]), _ptr(eU[i]), _ptr(eV[i]), eY.shape[2], eU.shape[2], bits, ft.value, fb.value)
        gotf = np.fromfile(out / name, dt)
        per = W * H + 2 * (W // 2) * (H // 2)
        assert gotf.size == per * N
        for i in range(N):
            g = gotf[i * per:(i + 1) * per]
            assert np.array_equal(g[:W * H].reshape(H, W), eY[i, :, :W]), (name, i)
            assert np.array_equal(g[W * H:W * H + (W // 2) * (H // 2)].reshape(H // 2, W // 2), eU[i, :, :W // 2])
            assert np.array_equal(g[W * H + (W // 2) * (H // 2):].reshape(H // 2, W // 2), eV[i, :, :W // 2])

    # Prefetch threads inside one block, and two threads alternating between ADJACENT blocks (the two-entry block cache): upstream frames
    # pulled once per block either way, frames equal to the serial walk's
    assert (out / "concurrent.txt").read_text().split() == ["upstream_frames_pulled_once", "1", "frames_equal_serial", "1",
                                                            "alternating_blocks_pulled_once", "1", "alternating_frames_equal_serial", "1"]

    errs = (out / "errors.txt").read_text().splitlines()
    assert errs[0].startswith("Failed to read logo file (") and "missing.lgd" in errs[0]
    assert "mode 1" in errs[1]
    assert "mode 1" in (out / "plugin_errors.txt").read_text()
