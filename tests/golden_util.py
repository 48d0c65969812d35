"""Loads tests/golden/logo_path_v1.npz (outputs of the REAL reference, see tools/make_golden.py) and rebuilds
full frames from the stored logo-rectangle crops."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logo_path_v1.npz")


def load():
    g = dict(np.load(PATH))
    W, H, LW, LH, X, Y0, N = (int(v) for v in g["geom"])
    g.update(W=W, H=H, LW=LW, LH=LH, X=X, Y0=Y0, N=N)
    return g


def frames(g, ky="crop_y", ku="crop_u", kv="crop_v", pitch_pad=0):
    W, H, LW, LH, X, Y0 = g["W"], g["H"], g["LW"], g["LH"], g["X"], g["Y0"]
    n = g[ky].shape[0]
    Y = np.zeros((n, H, W + pitch_pad), np.uint8)
    U = np.zeros((n, H // 2, W // 2 + pitch_pad // 2), np.uint8)
    V = np.zeros_like(U)
    Y[:, Y0:Y0 + LH, X:X + LW] = g[ky]
    U[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = g[ku]
    V[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = g[kv]
    return Y, U, V


def crops(g, Y, U, V):
    LW, LH, X, Y0 = g["LW"], g["LH"], g["X"], g["Y0"]
    return (Y[:, Y0:Y0 + LH, X:X + LW], U[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2],
            V[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2])


# ---- v2: > 8-bit containers and HD frame sizes (tools/make_golden.py main_v2) ----
PATH_V2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logo_path_v2_hibit_hd.npz")


def load_v2():
    """{case name: dict(W,H,bits,LW,LH,X,Y0,N, logo, crop_*, analysis, erase_*, ...)}, logof_text"""
    g = dict(np.load(PATH_V2))
    cases = {}
    for name in (str(c) for c in g["cases"]):
        c = {k[len(name) + 1:]: v for k, v in g.items() if k.startswith(name + "_")}
        W, H, bits, LW, LH, X, Y0, N = (int(v) for v in c["geom"])
        c.update(W=W, H=H, bits=bits, LW=LW, LH=LH, X=X, Y0=Y0, N=N)
        cases[name] = c
    return cases, g["logof_text"].tobytes()


def frames_v2(c, pitch_pad=0):
    """full frames (zeros outside the logo rectangle) from a v2 case's crops"""
    W, H, LW, LH, X, Y0 = c["W"], c["H"], c["LW"], c["LH"], c["X"], c["Y0"]
    cy = c["crop_y"]
    n = cy.shape[0]
    Y = np.zeros((n, H, W + pitch_pad), cy.dtype)
    U = np.zeros((n, H // 2, W // 2 + pitch_pad // 2), cy.dtype)
    V = np.zeros_like(U)
    Y[:, Y0:Y0 + LH, X:X + LW] = cy
    U[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = c["crop_u"]
    V[:, Y0 // 2:Y0 // 2 + LH // 2, X // 2:X // 2 + LW // 2] = c["crop_v"]
    return Y, U, V


def quirk_frames_v2(c):
    """Y frames holding the rows LogoFrame::ScanFrame<uint16_t> reads through its byte-pitch stride (rows 2*(Y0+y))"""
    W, H, LW, LH, X, Y0 = c["W"], c["H"], c["LW"], c["LH"], c["X"], c["Y0"]
    band = c["quirk_band"]
    Y = np.zeros((band.shape[0], H, W), band.dtype)
    Y[:, 2 * Y0:2 * (Y0 + LH), X:X + LW] = band
    return Y
