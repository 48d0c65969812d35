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
