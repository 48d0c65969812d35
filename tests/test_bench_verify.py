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
    # (range starts that are no multiple of 8 -- a rank's shard of a ragged split, e2e10 -- and one that ends with the clip)
    cache = {}
    for (v0, v1, chunk) in ((0, N, 16), (0, N, 512), (16, 48, 8), (24, N, 16), (0, 40, 24), (13, 50, 16), (37, N, 8), (9, 10, 8)):
        r = BV.verify_range(torch, ol, bits, N, v0, v1, pr, er, ev[v0:], an[v0:], fades[v0:], st[v0:], base=v0, tol=0.0, chunk=chunk, threads=4,
                            stage_cache=cache)
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


@pytest.mark.parametrize("bits", [8, 10])
def test_verify_scan_records_from_the_logo_rows_alone(bits):
    """the full-stream check of BASELINE configs[3] ships only the logo rectangle's rows of every frame to the host"""
    N = 41
    ol, P, E, ev, an, fades, st = _serial(bits, N)
    LH = 128
    rows = lambda lo, hi: P[0][lo:hi, Y0:Y0 + LH]
    r = BV.verify_scan_records(torch, ol, bits, 0, N, rows, ev, Y0, chunk=16, threads=3)
    assert r["records_equal_oracle"] and r["frames"] == N
    r = BV.verify_scan_records(torch, ol, bits, 5, 30, rows, ev, Y0, chunk=7, threads=5)
    assert r["records_equal_oracle"] and r["frames"] == 25
    ev2 = ev.copy(); ev2[29, 0, 1] = np.nextafter(ev2[29, 0, 1], np.float32(2))
    r = BV.verify_scan_records(torch, ol, bits, 0, N, rows, ev2, Y0, chunk=16, threads=3)
    assert not r["records_equal_oracle"] and r["mismatching_chunks"] == [[16, 32]]
