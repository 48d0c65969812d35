"""End to end on the last BASELINE config's frame format, small N: 1920x1080 10-bit (uint16 containers), logo at
(1600, 64): field weave -> LogoFrame scan -> AMTAnalyzeLogo -> CalcFade -> AMTEraseLogo -> frame metrics -> cadence /
scene decisions, every output diffed bit-exact against the CPU oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import amt_synth as S
import frame_stats_oracle as FS
from amtlib import Oracle, _ptr

pytestmark = pytest.mark.gpu


def test_full_pipeline_1080p_10bit():
    import torch
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, Context, DeviceClip, FrameStats, Logo, LogoFrame, weave_fields
    W, H, LW, LH, X, Y0, N, bits = 1920, 1080, 256, 128, 1600, 64, 20, 10
    dev = torch.device("cuda:0")
    ctx = Context(0)
    data, alpha, alphaUV = S.make_logo(LW, LH)
    pY, pUV = 1920, 960
    clip = S.make_clip_np(N, W, H, 0x5EED0010, alpha, alphaUV, X, Y0, bits=bits, period=8, fade=3, flat_every=4, cadence="24p",
                          pitchY=pY, pitchUV=pUV)
    Y, U, V = clip["Y"], clip["U"], clip["V"]
    orc = Oracle()

    # ---- ingest: frames arrive as decoded pictures; frame i = top field of picture i, bottom field of picture min(i+1, N-1) ----
    top = list(range(N))
    bot = [min(i + 1, N - 1) for i in range(N)]
    up = lambda a: torch.from_numpy(a.view(np.int16)).to(dev)
    frames = DeviceClip(torch.zeros((N, H, pY), dtype=torch.int16, device=dev), torch.zeros((N, H // 2, pUV), dtype=torch.int16, device=dev),
                        torch.zeros((N, H // 2, pUV), dtype=torch.int16, device=dev), W, H, bits)
    weave_fields(ctx, up(Y), up(U), up(V), frames, top, bot, nv12=False)
    wY, wU, wV = np.zeros_like(Y), np.zeros_like(U), np.zeros_like(V)
    for i in range(N):
        orc.lib.orc_merge_field(_ptr(Y[top[i]]), _ptr(U[top[i]]), _ptr(V[top[i]]), _ptr(Y[bot[i]]), _ptr(U[bot[i]]), _ptr(V[bot[i]]),
                                pY, pUV, 0, bits, W, H, _ptr(wY[i]), _ptr(wU[i]), _ptr(wV[i]), pY, pUV)
    host = lambda t: t.cpu().numpy().view(np.uint16)
    assert np.array_equal(host(frames.Y), wY) and np.array_equal(host(frames.U), wU) and np.array_equal(host(frames.V), wV)

    # ---- logo scan + selection ----
    logo = Logo.from_planes(ctx, data, LW, LH, W, H, X, Y0)
    lf = LogoFrame(ctx, [logo], 0.35)
    lf.scanFrames(frames, batch=8)
    lo = orc.make_logo(data, LW, LH, W, H, X, Y0)
    d = orc.lib.orc_logo_deint(lo); orc.lib.orc_logo_create_mask(d, 0.35, 1)
    t = orc.lib.orc_logo_field(lo, 0); orc.lib.orc_logo_create_mask(t, 0.35, 1)
    b = orc.lib.orc_logo_field(lo, 1); orc.lib.orc_logo_create_mask(b, 0.35, 1)
    want_ev = np.zeros(N * 2, np.float32)
    orc.lib.orc_logoframe_scan((C.c_void_p * 1)(d), 1, _ptr(wY), wY.strides[0], wY.shape[2], bits, W, H, N, _ptr(want_ev))
    assert lf.evalResults.reshape(-1).tobytes() == want_ev.tobytes()

    # ---- analysis, fades, erase ----
    an = AMTAnalyzeLogo(ctx, logo, 0.35).analyze(frames)
    want_an = np.zeros(N * 33, np.float32)
    orc.lib.orc_analyze_frames(d, t, b, _ptr(wY), wY.strides[0], wY.shape[2], bits, N, _ptr(want_an))
    assert an.reshape(-1).tobytes() == want_an.tobytes()
    er = AMTEraseLogo(ctx, logo, "", 0, 16)
    fades = er.calc_fades(an, N)
    er.erase(frames, fades)
    eY, eU, eV = wY.copy(), wU.copy(), wV.copy()
    for i in range(N):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(None, 0, 16, _ptr(want_an), N, i, C.byref(ft), C.byref(fb))
        assert (ft.value, fb.value) == (float(fades[i, 0]), float(fades[i, 1]))
        orc.lib.orc_erase_frame(lo, _ptr(eY[i]), _ptr(eU[i]), _ptr(eV[i]), pY, pUV, bits, ft.value, fb.value)
    assert np.array_equal(host(frames.Y), eY) and np.array_equal(host(frames.U), eU) and np.array_equal(host(frames.V), eV)
    assert not np.array_equal(eY, wY)                      # the logo was there and has been touched

    # ---- whole-frame metrics and the integer decisions ----
    fs = FrameStats(ctx, W, H, bits)
    m = fs.run(frames)
    assert np.array_equal(m, FS.frame_metrics(eY[:, :, :W]))
    cad, ph = fs.cadence(m)
    ocad, oph = FS.classify_cadence(m, W, H)
    assert np.array_equal(cad, ocad) and np.array_equal(ph, oph)
    assert fs.scene_changes(m).tolist() == FS.scene_changes(m, W, H)
