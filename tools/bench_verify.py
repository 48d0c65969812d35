"""Whole-range verification of a bench step against the CPU oracle (imported by bench.py and tools/bench_e2e.py): checker code, not
product, and never inside a timed region.

Every frame of the range [v0, v1) of a clip of NT frames is held against the oracle (oracle/libamt_oracle.so, the restatement pinned to
the real reference sources): LogoFrame scan records, frame metrics, CalcFade output and the erased Y / U / V planes as BYTES, the
AMTAnalyzeLogo records as bytes (exact mode) or within `tol` absolute (linear-guarded mode).  Matches LogoScan.hpp:1543-1568 (scan),
:1119-1161 (analysis), :1263-1341 (CalcFade / CalcFade2), :1248-1261 + :1374-1397 (Delogo).

The oracle runs on all host cores (frames dealt over threads; ctypes releases the GIL, the oracle keeps no shared state): ~2 300 frames/s
of 1440x1080 on the GPU box, so 10 000 frames take a few seconds.  Frames travel device -> pinned host buffers in chunks.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HALO = 8            # CalcFade2 reads analysis records n-8 .. n+8 (LogoScan.hpp:1265-1285)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def host_threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except Exception:
        return max(1, os.cpu_count() or 1)


class _Stage:
    """pinned landing buffers, reused for every chunk"""

    def __init__(self, torch, shapes, dtype):
        self.torch = torch
        self.gpu = torch.cuda.is_available()          # (the CPU unit test of the chunk / halo logic runs without one)
        self.bufs = [torch.empty(s, dtype=dtype).pin_memory() if self.gpu else torch.empty(s, dtype=dtype) for s in shapes]

    def fetch(self, k, t):
        b = self.bufs[k][:t.shape[0]]
        b.copy_(t, non_blocking=True)
        return b


def verify_range(torch, ol, bits, NT, v0, v1, pristine, erased, ev_g, an_g, fades_g, st_g, base=0, tol=0.0, erase=True, chunk=512, threads=None,
                 stage_cache=None):
    """pristine(lo, hi) / erased(lo, hi) -> (Y, U, V) device tensors of frames [lo, hi) before / after the step (U, V may be None when
    erase is False).  ev_g (n, nlogos, 2), an_g (n, 33), fades_g (n, 2), st_g (n, 8): the step's outputs, row i = frame base + i.
    The oracle's analysis records start at a multiple of 8 at or before v0 - 8 (CalcFade2 addresses the analysis clip in groups of eight,
    LogoScan.hpp:1271-1276, and clamps the GROUP number at the clip's end: the oracle's array must be cut where the clip's groups are),
    so pristine() is asked for up to 15 frames before v0.  stage_cache: a dict that keeps the pinned landing buffers between calls."""
    assert chunk % 8 == 0
    lib = ol.orc.lib
    nl = len(ol.deints)
    T = threads or host_threads()
    a0, a1 = max(0, v0 - HALO) // 8 * 8, min(NT, v1 + HALO)
    an_o = np.zeros((a1 - a0, 33), np.float32)               # the oracle's records of [a0, a1)
    done_an = a0
    res = {"frames": 0, "range": [v0, v1], "scan": True, "analysis": True, "fades": True, "erase": True, "metrics": True,
           "analysis_max_abs_err": 0.0, "threads": T}
    dt = np.uint8 if bits <= 8 else np.uint16
    view = (lambda t: t.numpy()) if bits <= 8 else (lambda t: t.numpy().view(np.uint16))
    stage = None
    t_start = time.perf_counter()
    bad = []
    with ThreadPoolExecutor(T) as ex:
        for c0 in range(v0, v1, chunk):
            c1 = min(v1, c0 + chunk)
            # pristine frames this chunk needs: [done_an, c1 + 8) for the analysis records CalcFade2 will read, c0 - 1 for the metrics
            hi = min(a1, c1 + HALO)
            lo = min(done_an, max(0, c0 - 1))
            pY, pU, pV = pristine(lo, hi)
            eY = eU = eV = None
            if erase:
                eY, eU, eV = erased(c0, c1)
            if stage is None:
                n_y = chunk + 3 * HALO + 2
                shp = lambda t, n: (n,) + tuple(t.shape[1:])
                shapes = [shp(pY, n_y)] + ([shp(pU, n_y), shp(pV, n_y), shp(eY, chunk), shp(eU, chunk), shp(eV, chunk)] if erase else [])
                key = (tuple(shapes), str(pY.dtype))
                stage = stage_cache.get(key) if stage_cache is not None else None
                if stage is None:
                    stage = _Stage(torch, shapes, pY.dtype)
                    if stage_cache is not None:
                        stage_cache[key] = stage
            hY = stage.fetch(0, pY)
            if erase:
                hU, hV = stage.fetch(1, pU), stage.fetch(2, pV)
                hEY, hEU, hEV = stage.fetch(3, eY), stage.fetch(4, eU), stage.fetch(5, eV)
            if stage.gpu:
                torch.cuda.synchronize()
            del pY, pU, pV, eY, eU, eV
            Y = view(hY)
            assert Y.dtype == dt
            sY, pitchY = Y.strides[0], Y.shape[2]
            n_own = c1 - c0
            ev_o = np.zeros((n_own, nl, 2), np.float32)
            st_o = np.zeros((n_own, 8), np.uint64)
            first_new, start = done_an, min(done_an, c0)

            def phase1(t):
                for i in range(start + t, hi, T):
                    y = Y[i - lo]
                    if i >= first_new:
                        lib.orc_analyze_frames(ol.deints[0], ol.top, ol.bot, _p(y), sY, pitchY, bits, 1, C.c_void_p(an_o.ctypes.data + (i - a0) * 132))
                    if c0 <= i < c1:
                        lib.orc_logoframe_scan(ol.deint_arr, nl, _p(y), sY, pitchY, bits, ol.W, ol.H, 1, C.c_void_p(ev_o.ctypes.data + (i - c0) * nl * 8))
                        prev = Y[i - 1 - lo] if i > 0 else y        # frame 0 of a clip compares with itself
                        lib.orc_frame_metrics(_p(y), sY, pitchY, bits, ol.W, ol.H, 1, _p(prev), C.c_void_p(st_o.ctypes.data + (i - c0) * 64))
            list(ex.map(phase1, range(T)))
            done_an = hi
            g = slice(c0 - base, c1 - base)
            ok_scan = ev_o.tobytes() == np.ascontiguousarray(ev_g[g], np.float32).tobytes()
            ok_met = st_o.tobytes() == np.ascontiguousarray(st_g[g]).astype(np.uint64).tobytes()
            an_c = an_o[c0 - a0:c1 - a0]
            if tol == 0.0:
                ok_an = an_c.tobytes() == np.ascontiguousarray(an_g[g], np.float32).tobytes()
            else:
                d = float(np.abs(an_c - an_g[g]).max())
                res["analysis_max_abs_err"] = max(res["analysis_max_abs_err"], d)
                ok_an = d <= tol
            res["scan"] &= ok_scan
            res["metrics"] &= ok_met
            res["analysis"] &= ok_an
            if not (ok_scan and ok_met and ok_an) and len(bad) < 8:
                bad.append({"chunk": [c0, c1], "scan": ok_scan, "metrics": ok_met, "analysis": ok_an})
            res["frames"] += n_own
            if not erase:
                continue
            # ---- CalcFade on the oracle's own records, Delogo on the pristine host copy, both against the device's.  The records the
            # fades of [c0, c1) read -- [c0 - 8, c1 + 8) clipped to the clip -- are complete.  Positions are relative to a0 (0 or a
            # multiple of 8, so CalcFade2's groups of eight line up); inside the range no clamp is reached except the clip's own ends.
            U, V, EY, EU, EV = view(hU), view(hV), view(hEY), view(hEU), view(hEV)
            fades_c = np.ascontiguousarray(fades_g[g], np.float32)
            okf = np.ones(n_own, bool)
            oke = np.ones(n_own, bool)
            n_arr = a1 - a0

            def phase2(t):
                ft, fb = C.c_float(), C.c_float()
                for i in range(c0 + t, c1, T):
                    lib.orc_calc_fade(None, 0, 16, _p(an_o), n_arr, i - a0, C.byref(ft), C.byref(fb))
                    k = i - c0
                    okf[k] = (np.float32(ft.value).tobytes() + np.float32(fb.value).tobytes()) == fades_c[k].tobytes()
                    y, u, v = Y[i - lo], U[i - lo], V[i - lo]
                    lib.orc_erase_frame(ol.hs[0], _p(y), _p(u), _p(v), pitchY, U.shape[2], bits, ft, fb)
                    oke[k] = bool(np.array_equal(y, EY[k]) and np.array_equal(u, EU[k]) and np.array_equal(v, EV[k]))
            list(ex.map(phase2, range(T)))
            res["fades"] &= bool(okf.all())
            res["erase"] &= bool(oke.all())
            if not (okf.all() and oke.all()) and len(bad) < 8:
                bad.append({"chunk": [c0, c1], "fades_bad": int((~okf).sum()), "erase_bad": int((~oke).sum()),
                            "first_bad_frame": int(c0 + np.argmax(~(okf & oke)))})
    res["seconds"] = time.perf_counter() - t_start
    res["oracle"] = "oracle/libamt_oracle.so on %d host threads, every frame of the range" % T
    if bad:
        res["mismatches"] = bad
    res["ok"] = all(res[k] for k in ("scan", "analysis", "fades", "erase", "metrics"))
    return res


def verify_metrics(torch, lib, bits, W, H, N, getY, st_g, chunk=1024, threads=None):
    """the frame metrics of EVERY frame of a Y-only clip against the C oracle (orc_frame_metrics; itself tied to the numpy statement of
    DESIGN.md section 6 by tests/test_abi_and_host.py).  getY(lo, hi) -> device tensor of frames [lo, hi).  Self-specified pass."""
    T = threads or host_threads()
    view = (lambda t: t.numpy()) if bits <= 8 else (lambda t: t.numpy().view(np.uint16))
    stage, ok, t0 = None, True, time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        for c0 in range(0, N, chunk):
            c1 = min(N, c0 + chunk)
            lo = max(0, c0 - 1)
            d = getY(lo, c1)
            if stage is None:
                stage = _Stage(torch, [(chunk + 1,) + tuple(d.shape[1:])], d.dtype)
            h = stage.fetch(0, d)
            if stage.gpu:
                torch.cuda.synchronize()
            Y = view(h)
            st_o = np.zeros((c1 - c0, 8), np.uint64)

            def work(t):
                for i in range(c0 + t, c1, T):
                    y = Y[i - lo]
                    lib.orc_frame_metrics(_p(y), Y.strides[0], Y.shape[2], bits, W, H, 1, _p(Y[i - 1 - lo]) if i > 0 else _p(y),
                                          C.c_void_p(st_o.ctypes.data + (i - c0) * 64))
            list(ex.map(work, range(T)))
            ok &= st_o.tobytes() == np.ascontiguousarray(st_g[c0:c1]).astype(np.uint64).tobytes()
    return {"frames": N, "metrics_equal_oracle": bool(ok), "seconds": time.perf_counter() - t0, "threads": T}


def verify_scan_records(torch, ol, bits, v0, v1, rows, rec, imgy, chunk=4096, threads=None):
    """EVERY LogoFrame scan record of frames [v0, v1) against the oracle (LogoScan.hpp:1543-1568; bytes).  rows(lo, hi) -> device tensor
    (hi - lo, logo rows, pitch): the Y rows [imgy, imgy + h) of those frames -- ScanFrame reads nothing else of a frame, so only the logo
    rectangle's rows travel to the host (the oracle is handed the address the full plane would start at and never looks outside the
    rows).  rec: (n, nlogos, 2) records, row i = frame i."""
    lib = ol.orc.lib
    nl = len(ol.deints)
    T = threads or host_threads()
    view = (lambda t: t.numpy()) if bits <= 8 else (lambda t: t.numpy().view(np.uint16))
    stage, ok, bad, t0 = None, True, [], time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        for c0 in range(v0, v1, chunk):
            c1 = min(v1, c0 + chunk)
            d = rows(c0, c1)
            if stage is None:
                stage = _Stage(torch, [(chunk,) + tuple(d.shape[1:])], d.dtype)
            h = stage.fetch(0, d)
            if stage.gpu:
                torch.cuda.synchronize()
            del d
            Y = view(h)
            sY, pitch = Y.strides[0], Y.shape[2]
            fake = Y.ctypes.data - imgy * pitch * Y.itemsize       # where row 0 of frame c0 would be
            ev_o = np.zeros((c1 - c0, nl, 2), np.float32)
            per = -(-(c1 - c0) // T)

            def work(t):
                a, b = t * per, min(c1 - c0, (t + 1) * per)
                if a < b:
                    lib.orc_logoframe_scan(ol.deint_arr, nl, C.c_void_p(fake + a * sY), sY, pitch, bits, ol.W, ol.H, b - a,
                                           C.c_void_p(ev_o.ctypes.data + a * nl * 8))
            list(ex.map(work, range(T)))
            same = ev_o.tobytes() == np.ascontiguousarray(rec[c0:c1], np.float32).tobytes()
            ok &= same
            if not same and len(bad) < 8:
                bad.append([c0, c1])
    out = {"frames": v1 - v0, "records_equal_oracle": bool(ok), "seconds": time.perf_counter() - t0, "threads": T}
    if bad:
        out["mismatching_chunks"] = bad
    return out
