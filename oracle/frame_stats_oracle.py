"""CPU oracle (numpy) of the SELF-SPECIFIED whole-frame metrics and decisions.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no in-tree arithmetic for scene-change
scoring (external chapter_exe.exe, CMAnalyze.hpp:319-337) or telecine/comb analysis (external KFMDeint
plugin, Misc.cs:1300-1324) -- SURVEY.md section 0.  This file restates the specification in DESIGN.md
section 6 independently of the HIP kernels; integer arithmetic, so GPU results must match bit for bit.
"""
from __future__ import annotations

import numpy as np

WORDS = 8


def frame_metrics(Y: np.ndarray, prev_first: np.ndarray | None = None) -> np.ndarray:
    """Y: (N,H,W) integer samples.  Returns (N,8) uint64 per the AMTGPU_FS_* layout."""
    N, H, W = Y.shape
    out = np.zeros((N, WORDS), np.uint64)
    for n in range(N):
        cur = Y[n].astype(np.int64)
        prev = (Y[n - 1] if n > 0 else (prev_first if prev_first is not None else Y[0])).astype(np.int64)
        d = np.abs(cur - prev)
        out[n, 0] = d[0::2].sum()
        out[n, 1] = d[1::2].sum()
        a, b, c = cur[:-2], cur[1:-1], cur[2:]
        out[n, 2] = np.abs(a - c).sum()
        out[n, 3] = np.abs(b - ((a + c) >> 1)).sum()
        weave = cur.copy()
        weave[1::2] = prev[1::2]
        wa, wb, wc = weave[:-2], weave[1:-1], weave[2:]
        out[n, 4] = np.abs(wb - ((wa + wc) >> 1)).sum()
        out[n, 5] = cur.sum()
        out[n, 6] = np.abs(wa - wc).sum()
    return out


def scene_changes(m: np.ndarray, width: int, height: int) -> list[int]:
    out = []
    e = (m[:, 0] + m[:, 1]).astype(object)
    floor_energy = width * height * 4
    for n in range(1, len(e)):
        hist = sorted(e[max(1, n - 15):n])
        med = hist[len(hist) // 2] if hist else 0
        if e[n] >= floor_energy and e[n] > 3 * med:
            out.append(n)
    return out


def classify_cadence(m: np.ndarray, width: int, height: int):
    n = len(m)
    c0 = [int(x) for x in m[:, 3]]
    c1 = [int(x) for x in m[:, 4]]
    en = [int(a) + int(b) for a, b in zip(m[:, 0], m[:, 1])]
    code = ['C' if c0[i] * 3 < c1[i] * 2 else ('P' if c1[i] * 3 < c0[i] * 2 else 'B') for i in range(n)]
    still = width * height // 2
    cad = np.zeros(n, np.uint8)
    ph = np.zeros(n, np.uint8)
    last, last_phase = 0, 0
    for i in range(n):
        a, b = max(0, i - 4), min(n, i + 6)
        span = b - a
        nC = sum(code[k] == 'C' for k in range(a, b))
        nD = sum(code[k] != 'B' for k in range(a, b))
        motion = max(en[a:b])
        best, best_phase = -1, 0
        for p in range(5):
            hit = 0
            for k in range(a, b):
                pos = (k - a + p) % 5
                if pos <= 1:
                    hit += code[k] == 'C'
                elif pos <= 3:
                    hit += code[k] == 'P'
            if hit > best:
                best, best_phase = hit, p
        if motion < still:
            cls, p = last, ((last_phase + 1) % 5 if last == 1 else 0)
        elif nD * 2 < span:
            cls, p = 0, 0
        elif best * 10 >= span * 7:
            cls, p = 1, (i - a + best_phase) % 5
        elif nC * 10 >= span * 7:
            cls, p = 2, 0
        elif nD * 10 >= span * 7 and best * 10 >= span * 5:
            cls, p = 1, (i - a + best_phase) % 5
        else:
            cls, p = 0, 0
        cad[i], ph[i] = cls, p
        last, last_phase = cls, p
    return cad, ph


def cadence_durations(cad, ph):
    d = []
    n = len(cad)
    i = 0
    while i < n:
        if cad[i] == 1 and ph[i] == 0 and i + 5 <= n and all(cad[i + k] == 1 and ph[i + k] == k for k in range(1, 5)):
            d += [2, 3, 2, 3]
            i += 5
            continue
        d += [1, 1] if cad[i] == 0 else [2]
        i += 1
    return d
