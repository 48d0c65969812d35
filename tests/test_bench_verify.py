"""tools/bench_verify.verify_range -- the bench's whole-batch comparison with the CPU oracle -- walks a clip in chunks with an 8-frame
analysis halo, on many threads.  Here its chunk / halo / clamp logic is held against the oracle run serially over the whole clip (the
"device outputs" are the serial oracle's), for whole clips, interior ranges and ranges that end with the clip; and it must notice a
single wrong byte."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import amt_synth as S  # noqa: E402
import bench  # noqa: E402
import bench_verify as BV  # noqa: E402

W, H, X, Y0 = 352, 240, 64, 32


def _serial(bits, N):
    logos_np, alpha, alphaUV = bench.make_logos()
    ol = bench.OracleLogos(logos_np[:2], W, H, X, Y0, bits)
    clip = S.make_clip_np(N, W, H, 0x5EED0101, alpha, alphaUV, X, Y0, period=24, fade=6, bits=bits)
    Yp, Up, Vp = clip["Y"], clip["U"], clip["V"]
    ev = ol.scan(Yp, N).reshape(N, 2, 2)
    an = ol.analyze(Yp, N).reshape(N, 33)
    st = ol.metrics(Yp, N)
    fades = np.zeros((N, 2), np.float32)
    eY, eU, eV = Yp.copy(), Up.copy(), Vp.copy()
    for i in range(N):
        fades[i] = ol.fade(an.reshape(-1), N, i)
        ol.erase(eY, eU, eV, i, float(fades[i, 0]), float(fades[i, 1]))
    tt = (lambda a: torch.from_numpy(a)) if bits <= 8 else (lambda a: torch.from_numpy(a.view(np.int16)))
    return ol, tuple(map(tt, (Yp, Up, Vp))), tuple(map(tt, (eY, eU, eV))), ev, an, fades, st


@pytest.mark.parametrize("bits", [8, 10])
def test_verify_range_agrees_with_the_serial_oracle(bits):
    N = 75
    ol, P, E, ev, an, fades, st = _serial(bits, N)
    assert (fades != 0).any() and (fades == 0).any()          # the clip fades the logo in and out: the erase path is exercised
    pr = lambda lo, hi: tuple(t[lo:hi] for t in P)
    er = lambda lo, hi: tuple(t[lo:hi] for t in E)
    for (v0, v1, chunk) in ((0, N, 16), (0, N, 512), (16, 48, 8), (24, N, 16), (0, 40, 24)):
        r = BV.verify_range(torch, ol, bits, N, v0, v1, pr, er, ev[v0:], an[v0:], fades[v0:], st[v0:], base=v0, tol=0.0, chunk=chunk, threads=4)
        assert r["ok"] and r["frames"] == v1 - v0, (v0, v1, chunk, r)
    # the linear mode's comparison: records within a tolerance, everything else bytes
    r = BV.verify_range(torch, ol, bits, N, 0, N, pr, er, ev, an + np.float32(3e-6), fades, st, tol=1e-4, chunk=32, threads=3)
    assert r["ok"] and 2e-6 < r["analysis_max_abs_err"] < 1e-5


def test_verify_range_notices_one_wrong_value():
    N = 40
    ol, P, E, ev, an, fades, st = _serial(8, N)
    pr = lambda lo, hi: tuple(t[lo:hi] for t in P)
    er = lambda lo, hi: tuple(t[lo:hi] for t in E)
    run = lambda **kw: BV.verify_range(torch, ol, 8, N, 0, N, pr, kw.get("er", er), kw.get("ev", ev), kw.get("an", an), kw.get("fades", fades),
                                       kw.get("st", st), chunk=16, threads=4)
    assert run()["ok"]
    ev2 = ev.copy(); ev2[33, 1, 0] = np.nextafter(ev2[33, 1, 0], np.float32(2))
    r = run(ev=ev2); assert not r["ok"] and not r["scan"] and r["analysis"] and r["mismatches"][0]["chunk"] == [32, 40]
    an2 = an.copy(); an2[7, 20] = np.nextafter(an2[7, 20], np.float32(2))
    r = run(an=an2); assert not r["ok"] and not r["analysis"] and r["scan"]
    st2 = st.copy(); st2[0, 3] += 1
    r = run(st=st2); assert not r["metrics"] and not r["ok"]
    f2 = fades.copy(); f2[20, 1] += np.float32(0.1)
    r = run(fades=f2); assert not r["fades"] and not r["ok"]
    E2 = (E[0].clone(), E[1], E[2]); E2[0][39, H - 1, W - 1] ^= 1          # a byte OUTSIDE the logo rectangle
    r = run(er=lambda lo, hi: tuple(t[lo:hi] for t in E2)); assert not r["erase"] and r["fades"] and not r["ok"]
