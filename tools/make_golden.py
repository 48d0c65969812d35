#!/usr/bin/env python3
"""Generates tests/golden/logo_path_v1.npz from the REAL reference (oracle/_ref/libamt_ref.so, i.e. the
reference's own LogoScan.hpp / AMTLogo.hpp / ComputeKernel.cpp compiled through oracle/ref_shim).

Runs only where /root/reference exists (this container).  The fixture holds the inputs too (logo planes and
the logo-rectangle crops of every frame -- the path never reads pixels outside the rectangle), so the tests
that consume it need neither the reference nor the generator.
    python tools/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import amt_synth as S
from amtlib import Oracle, Ref, _ptr, write_raw_clip

W, H, LW, LH, X, Y0 = 352, 240, 96, 48, 224, 18
N = 40


def frames_from_crops(cy, cu, cv):
    n = cy.shape[0]
    Y = np.zeros((n, H, W), np.uint8); U = np.zeros((n, H // 2, W // 2), np.uint8); V = np.zeros_like(U)
    Y[:, Y0:Y0 + LH, X:X + LW] = cy
    U[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = cu
    V[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = cv
    return Y, U, V


def main():
    assert Ref.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    ref, orc = Ref(), Oracle()
    tmp = "/tmp/amt_golden"
    os.makedirs(tmp, exist_ok=True)
    data, alpha, alphaUV = S.make_logo(LW, LH)
    data2, _, _ = S.make_logo(LW, LH, seed=0x10600002, strength=0.5)
    clip = S.make_clip_np(N, W, H, 0x5EED0001, alpha, alphaUV, X, Y0, period=16, fade=6, flat_every=3)
    cy = clip["Y"][:, Y0:Y0 + LH, X:X + LW].copy()
    cu = clip["U"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
    cv = clip["V"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
    Y, U, V = frames_from_crops(cy, cu, cv)
    paths = []
    for i, d in enumerate((data, data2)):
        lo = orc.make_logo(d, LW, LH, W, H, X, Y0)      # only used to write the .lgd input file
        p = os.path.join(tmp, f"logo{i}.lgd").encode()
        assert orc.lib.orc_logo_save(lo, p, b"golden", 1041) == 1
        paths.append(p)
    lgd_bytes = np.frombuffer(open(paths[0], "rb").read(), np.uint8)
    # LogoFrame
    ev = np.zeros(N * 2 * 2, np.float32)
    best, ratio = C.c_int(), C.c_float()
    text = C.create_string_buffer(1 << 16)
    arr = (C.c_char_p * 2)(*paths)
    assert ref.lib.ref_logoframe(arr, 2, 0.35, _ptr(Y), Y.strides[0], W, 8, W, H, N, 30000, 1001, _ptr(ev), 2, C.byref(best),
                                 C.byref(ratio), 0, os.path.join(tmp, "lf.txt").encode(), text, len(text)) == 1
    # AMTAnalyzeLogo
    an = np.zeros(N * 33, np.float32)
    assert ref.lib.ref_analyze(paths[0], 0.35, _ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], W, W // 2, 8, W, H, N, _ptr(an)) == 1
    # AMTEraseLogo without / with a logoframe file
    out = {}
    logof_text = b"    14 S 0 ALL     12     17\n    20 E 0 ALL     18     23\n    30 S 0 ALL     29     33\n    39 E 0 ALL     38     39\n"
    open(os.path.join(tmp, "lf_in.txt"), "wb").write(logof_text)
    for tag, lf in (("nolf", b""), ("lf", os.path.join(tmp, "lf_in.txt").encode())):
        Ye, Ue, Ve = Y.copy(), U.copy(), V.copy()
        fades = np.zeros(N * 2, np.float32)
        assert ref.lib.ref_erase(paths[0], lf, 16, 0.35, _ptr(Ye), _ptr(Ue), _ptr(Ve), Ye.strides[0], Ue.strides[0], W, W // 2, 8, W, H, N, _ptr(fades)) == 1
        out[f"erase_{tag}_fades"] = fades.reshape(N, 2)
        out[f"erase_{tag}_Y"] = Ye[:, Y0:Y0 + LH, X:X + LW].copy()
        out[f"erase_{tag}_U"] = Ue[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
        out[f"erase_{tag}_V"] = Ve[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
    # LogoScan accumulate + regress
    sr = ref.lib.ref_scan_create(LW, LH, 1, 1, 12)
    valid = []
    for i in range(N):
        valid.append(ref.lib.ref_scan_add_frame_u8(sr, Y[i, Y0:, X:].ctypes.data, U[i, Y0 // 2:, X // 2:].ctypes.data, V[i, Y0 // 2:, X // 2:].ctypes.data, W, W // 2))
    npx = LW * LH + 2 * (LW // 2) * (LH // 2)
    sums = np.zeros(npx * 5)
    ref.lib.ref_scan_sums(sr, _ptr(sums))
    logos = {}
    for clean in (0, 1):
        lr = ref.lib.ref_scan_get_logo(sr, 255, clean, W, H, X, Y0)
        assert lr
        logos[clean] = ref.logo_data(lr)
    # the exported ScanLogo end to end
    clip2 = S.make_clip_np(60, W, H, 0x5EED0004, alpha, alphaUV, X, Y0, period=20, fade=4, flat_every=2)
    c2y = clip2["Y"][:, Y0:Y0 + LH, X:X + LW].copy()
    c2u = clip2["U"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
    c2v = clip2["V"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy()
    Y2, U2, V2 = frames_from_crops(c2y, c2u, c2v)
    raw = os.path.join(tmp, "clip.raw").encode()
    write_raw_clip(raw, Y2, U2, V2, W, H)
    dst = os.path.join(tmp, "scanned.lgd").encode()
    assert ref.lib.ref_scanlogo(raw, 1041, os.path.join(tmp, "work.dat").encode(), dst, X, Y0, LW, LH, 12, 25) == 1, ref.lib.ref_last_error()
    scanned = np.frombuffer(open(dst, "rb").read(), np.uint8)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "logo_path_v1.npz"),
        geom=np.array([W, H, LW, LH, X, Y0, N], np.int32), logo0=data, logo1=data2, lgd0=lgd_bytes,
        crop_y=cy, crop_u=cu, crop_v=cv,
        logoframe_evals=ev.reshape(N, 2, 2), logoframe_best=np.int32(best.value), logoframe_ratio=np.float32(ratio.value),
        logoframe_text=np.frombuffer(text.value, np.uint8), analysis=an.reshape(N, 33), logof_text=np.frombuffer(logof_text, np.uint8),
        scan_valid=np.array(valid, np.uint8), scan_sums=sums.reshape(npx, 5), scan_logo_raw=logos[0], scan_logo_clean=logos[1],
        scanlogo_crop_y=c2y, scanlogo_crop_u=c2u, scanlogo_crop_v=c2v, scanlogo_lgd=scanned, **out)
    print("wrote tests/golden/logo_path_v1.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "logo_path_v1.npz")), "bytes")


# ------------------------------------------------------------------------------------------------
# v2: > 8-bit containers and HD frame sizes (BASELINE configs 2/4: 1440x1080 8-bit; config 5: 1920x1080 10-bit)
# ------------------------------------------------------------------------------------------------
V2_CASES = [
    # name, W, H, bits, LW, LH, X, Y0, N, seed, period, fade, flat_every
    ("sd10", 352, 240, 10, 96, 48, 224, 18, 24, 0x5EED0005, 12, 5, 4),
    ("sd12", 352, 240, 12, 96, 48, 224, 18, 24, 0x5EED0006, 12, 5, 4),
    ("hd8", 1440, 1080, 8, 256, 128, 1120, 64, 6, 0x5EED0002, 3, 2, 3),
    ("fhd10", 1920, 1080, 10, 256, 128, 1600, 64, 4, 0x5EED0007, 2, 2, 3),
]
V2_LOGOF = b"     6 S 0 ALL      4      8\n    11 E 0 ALL     10     12\n    20 S 0 ALL     19     21\n    23 E 0 ALL     23     23\n"


def crops_of(clip, X, Y0, LW, LH):
    return (clip["Y"][:, Y0:Y0 + LH, X:X + LW].copy(), clip["U"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy(),
            clip["V"][:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2].copy())


def frames_of(cy, cu, cv, W, H, X, Y0):
    n, LH, LW = cy.shape
    Y = np.zeros((n, H, W), cy.dtype); U = np.zeros((n, H // 2, W // 2), cy.dtype); V = np.zeros_like(U)
    Y[:, Y0:Y0 + LH, X:X + LW] = cy
    U[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = cu
    V[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = cv
    return Y, U, V


def main_v2():
    assert Ref.available(), "build oracle/_ref first (oracle/build_ref.sh)"
    ref, orc = Ref(), Oracle()
    tmp = "/tmp/amt_golden"
    os.makedirs(tmp, exist_ok=True)
    out = {"cases": np.array([c[0] for c in V2_CASES]), "logof_text": np.frombuffer(V2_LOGOF, np.uint8)}
    for name, W, H, bits, LW, LH, X, Y0, N, seed, period, fade, flat in V2_CASES:
        data, alpha, alphaUV = S.make_logo(LW, LH)
        lo = orc.make_logo(data, LW, LH, W, H, X, Y0)
        path = os.path.join(tmp, f"{name}.lgd").encode()
        assert orc.lib.orc_logo_save(lo, path, b"golden", 1041) == 1
        clip = S.make_clip_np(N, W, H, seed, alpha, alphaUV, X, Y0, bits=bits, period=period, fade=fade, flat_every=flat)
        cy, cu, cv = crops_of(clip, X, Y0, LW, LH)
        Y, U, V = frames_of(cy, cu, cv, W, H, X, Y0)
        out[f"{name}_geom"] = np.array([W, H, bits, LW, LH, X, Y0, N], np.int32)
        out[f"{name}_logo"] = data
        out[f"{name}_crop_y"], out[f"{name}_crop_u"], out[f"{name}_crop_v"] = cy, cu, cv
        an = np.zeros(N * 33, np.float32)
        assert ref.lib.ref_analyze(path, 0.35, _ptr(Y), _ptr(U), _ptr(V), Y.strides[0], U.strides[0], W, W // 2, bits, W, H, N, _ptr(an)) == 1
        out[f"{name}_analysis"] = an.reshape(N, 33)
        tags = (("nolf", b""), ("lf", os.path.join(tmp, "lf2_in.txt").encode())) if N >= 24 else (("nolf", b""),)
        open(os.path.join(tmp, "lf2_in.txt"), "wb").write(V2_LOGOF)
        for tag, lf in tags:
            Ye, Ue, Ve = Y.copy(), U.copy(), V.copy()
            fades = np.zeros(N * 2, np.float32)
            assert ref.lib.ref_erase(path, lf, 16, 0.35, _ptr(Ye), _ptr(Ue), _ptr(Ve), Ye.strides[0], Ue.strides[0], W, W // 2, bits, W, H, N,
                                     _ptr(fades)) == 1, ref.lib.ref_last_error()
            out[f"{name}_erase_{tag}_fades"] = fades.reshape(N, 2)
            ey, eu, ev_ = crops_of({"Y": Ye, "U": Ue, "V": Ve}, X, Y0, LW, LH)
            out[f"{name}_erase_{tag}_Y"], out[f"{name}_erase_{tag}_U"], out[f"{name}_erase_{tag}_V"] = ey, eu, ev_
        best, ratio = C.c_int(), C.c_float()
        text = C.create_string_buffer(1 << 16)
        if bits == 8:
            ev = np.zeros(N * 2, np.float32)
            assert ref.lib.ref_logoframe((C.c_char_p * 1)(path), 1, 0.35, _ptr(Y), Y.strides[0], W, 8, W, H, N, 30000, 1001, _ptr(ev), -1,
                                         C.byref(best), C.byref(ratio), 0, os.path.join(tmp, "lf2.txt").encode(), text, len(text)) == 1
            out[f"{name}_logoframe_evals"] = ev.reshape(N, 1, 2)
        elif 2 * (Y0 + LH) <= H and (W * 2) % 64 == 0:
            # LogoFrame::ScanFrame<uint16_t> uses the BYTE pitch as element stride (LogoScan.hpp:1547,1561): it reads rectangle
            # row y at frame row 2*(Y0+y).  Stored with the band of rows it touches; consumers pass a doubled pitch.
            band = clip["Y"][:, 2 * Y0:2 * (Y0 + LH), X:X + LW].copy()
            Yq = np.zeros((N, H, W), np.uint16)
            Yq[:, 2 * Y0:2 * (Y0 + LH), X:X + LW] = band
            ev = np.zeros(N * 2, np.float32)
            assert ref.lib.ref_logoframe((C.c_char_p * 1)(path), 1, 0.35, _ptr(Yq), Yq.strides[0], W, bits, W, H, N, 30000, 1001, _ptr(ev), -1,
                                         C.byref(best), C.byref(ratio), 0, os.path.join(tmp, "lf2.txt").encode(), text, len(text)) == 1
            out[f"{name}_quirk_band"] = band
            out[f"{name}_logoframe_evals_bytepitch"] = ev.reshape(N, 1, 2)
    dst = os.path.join(ROOT, "tests", "golden", "logo_path_v2_hibit_hd.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    if "--v2-only" not in sys.argv:
        main()
    main_v2()
