"""CPU-side checks of the product library: it loads, exports every symbol include/amt_gpu.h declares, and
its host-only entry points (logo model, .lgd format, evaluation tables, decisions) agree with the oracle
bit for bit.  No GPU compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import amt_synth as S
from amtlib import ROOT, Oracle, _ptr

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frame_stats_oracle as FS


@pytest.fixture(scope="module")
def lib():
    from amatsukaze_amd import build as b
    b.build()
    from amatsukaze_amd import binding
    return binding.load()


def test_exports_every_declared_symbol(lib):
    from amatsukaze_amd import binding
    hdr = open(os.path.join(ROOT, "include", "amt_gpu.h")).read()
    declared = set(re.findall(r"\b(amtgpu_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(binding.SIGNATURES), declared ^ set(binding.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.amtgpu_abi_version() == 5


def test_hip_runtime_count_is_reported(lib):
    """amtgpu_hip_runtimes_loaded: the copies of libamdhip64 mapped into the process, by path -- one here (the library's own)"""
    buf = C.create_string_buffer(1024)
    n = lib.amtgpu_hip_runtimes_loaded(buf, len(buf))
    paths = buf.value.decode().split()
    assert n == len(paths) >= 1 and all("libamdhip64.so" in p for p in paths)
    assert lib.amtgpu_hip_runtimes_loaded(None, 0) == n
    tiny = C.create_string_buffer(8)
    assert lib.amtgpu_hip_runtimes_loaded(tiny, len(tiny)) == n and len(tiny.value) <= 7


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not lib.amtgpu_context_create(0)
    from amatsukaze_amd import AmtError, Context
    with pytest.raises((AmtError, Exception)):
        Context(0)


def test_product_never_touches_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "amatsukaze_amd")):
        if os.path.basename(dp) == "build":
            continue
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "amt_oracle" not in src and "libamt_ref" not in src and "frame_stats_oracle" not in src, f


def test_lgd_and_mask_tables_match_oracle(lib, tmp_path):
    LW, LH, W, H, X, Y0 = 96, 48, 352, 240, 224, 18
    data, _, _ = S.make_logo(LW, LH)
    orc = Oracle()
    lo = orc.make_logo(data, LW, LH, W, H, X, Y0)
    p1, p2 = str(tmp_path / "o.lgd").encode(), str(tmp_path / "g.lgd").encode()
    assert orc.lib.orc_logo_save(lo, p1, b"synthetic", 1041) == 1
    lg = lib.amtgpu_logo_load(None, p1)
    assert lg
    assert lib.amtgpu_logo_save(None, lg, p2, b"synthetic", 1041) == 1
    assert open(p1, "rb").read() == open(p2, "rb").read()
    info = np.zeros(8, np.int32)
    lib.amtgpu_logo_get_info(lg, _ptr(info))
    assert info.tolist() == [LW, LH, 1, 1, W, H, X, Y0]
    for kind, maskratio in [(0, 0.35), (1, 0.35), (2, 0.35), (0, 0.1), (0, 0.95)]:
        o2 = orc.lib.orc_logo_deint(lo) if kind == 0 else orc.lib.orc_logo_field(lo, kind - 1)
        orc.lib.orc_logo_create_mask(o2, maskratio, 1)
        _, mask, ker, sc, black, mp, cnt = orc.logo_arrays(o2)
        gmp, gcnt, gblack = C.c_int(), C.c_int(), C.c_float()
        h = LH if kind == 0 else LH // 2
        gmask = np.zeros(LW * h, np.uint8)
        assert lib.amtgpu_logo_mask_tables(None, lg, kind, maskratio, C.byref(gmp), C.byref(gcnt), C.byref(gblack), _ptr(gmask), None, None) == 1
        assert (gmp.value, gcnt.value) == (mp, cnt)
        gker, gsc = np.zeros(cnt * 25, np.float32), np.zeros(cnt * 64, np.float32)
        assert lib.amtgpu_logo_mask_tables(None, lg, kind, maskratio, None, None, None, None, _ptr(gker), _ptr(gsc)) == 1
        assert np.array_equal(gmask, mask)
        assert gker.tobytes() == ker[:cnt * 25].tobytes()
        assert gsc.tobytes() == sc[:cnt * 64].tobytes()
        assert np.float32(gblack.value).tobytes() == np.float32(black).tobytes()
    lib.amtgpu_logo_destroy(lg)
    assert not lib.amtgpu_logo_load(None, b"/nonexistent.lgd")


def test_logo_header_access_and_utf16_paths(lib, tmp_path):
    """LogoFile_GetName / GetServiceId / SetName / SetServiceId (LogoGUISupport.hpp:254-275) through amtgpu_logo_get_header /
    _set_header, and the *W entry points that take the reference's own string type -- NUL-terminated UTF-16 (const tchar* =
    wchar_t* on Windows, LogoScan.hpp:1083-1086; C# CharSet.Unicode, AmatsukazeNatives.cs:391-393) -- on a Japanese file name."""
    LW, LH, W, H, X, Y0 = 96, 48, 352, 240, 224, 18
    data, _, _ = S.make_logo(LW, LH)
    orc = Oracle()
    lo = orc.make_logo(data, LW, LH, W, H, X, Y0)
    p1 = str(tmp_path / "o.lgd").encode()
    assert orc.lib.orc_logo_save(lo, p1, "NHK総合".encode("utf-8"), 1024) == 1
    lg = lib.amtgpu_logo_load(None, p1)
    name = C.create_string_buffer(256)
    sid = C.c_int()
    assert lib.amtgpu_logo_get_header(lg, name, 256, C.byref(sid)) == 1
    assert name.value.decode("utf-8") == "NHK総合" and sid.value == 1024
    assert lib.amtgpu_logo_get_header(lg, name, 4, None) == 0                   # buffer too small: refused, nothing written past it
    assert lib.amtgpu_logo_set_header(lg, "テレビ東京".encode("utf-8"), 1072) == 1
    assert lib.amtgpu_logo_get_header(lg, name, 256, C.byref(sid)) == 1 and name.value.decode("utf-8") == "テレビ東京" and sid.value == 1072

    def w16(s):
        return C.create_string_buffer(s.encode("utf-16-le") + b"\0\0")
    jp = tmp_path / "ロゴ_テスト.lgd"
    assert lib.amtgpu_logo_saveW(None, lg, w16(str(jp)), "テレビ東京".encode("utf-8"), 1072) == 1
    assert jp.exists()
    want = str(tmp_path / "w.lgd").encode()
    assert lib.amtgpu_logo_save(None, lg, want, "テレビ東京".encode("utf-8"), 1072) == 1
    assert jp.read_bytes() == open(want, "rb").read()
    lg2 = lib.amtgpu_logo_loadW(None, w16(str(jp)))
    assert lg2
    assert lib.amtgpu_logo_get_header(lg2, name, 256, C.byref(sid)) == 1 and name.value.decode("utf-8") == "テレビ東京" and sid.value == 1072
    # a lone surrogate in a path is replaced (U+FFFD), never emitted as invalid UTF-8: the call fails cleanly on the missing file
    assert not lib.amtgpu_logo_loadW(None, C.create_string_buffer(b"\x00\xd8x\x00\x00\x00"))
    lib.amtgpu_logo_destroy(lg)
    lib.amtgpu_logo_destroy(lg2)


def test_stats_decisions_match_oracle(lib):
    rng = np.random.RandomState(5)
    W, H, n = 352, 240, 150
    m = np.zeros((n, 8), np.uint64)
    # synthetic metric streams: 24p (C C P P B), 30p (all C), interlaced (all B), still
    for i in range(n):
        seg = i // 50
        motion = 400000 + int(rng.randint(0, 50000))
        c_lo, c_hi = 100000, 600000
        if seg == 0:
            pos = i % 5
            comb, combp = (c_lo, c_hi) if pos <= 1 else ((c_hi, c_lo) if pos <= 3 else (c_lo, c_lo))
        elif seg == 1:
            comb, combp = c_lo, c_hi
        else:
            comb, combp = c_hi, c_hi - 1000
        if 120 <= i < 135:
            motion = 1000
        m[i] = [motion // 2, motion // 2, 300000, comb + int(rng.randint(0, 2000)), combp, 10000000, 300000, 0]
    m[77, 0] = m[77, 1] = W * H * 20
    cad, ph = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    assert lib.amtgpu_kfm_cadence(_ptr(m), n, W, H, _ptr(cad), _ptr(ph)) == 1
    ocad, oph = FS.classify_cadence(m, W, H)
    assert np.array_equal(cad, ocad) and np.array_equal(ph, oph)
    assert (cad[5:40] == 1).all() and (cad[60:90] == 2).all() and (cad[105:118] == 0).all()
    assert (cad[120:135] == cad[119]).all()           # still frames inherit
    sc = np.zeros(n, np.int32)
    k = C.c_int()
    assert lib.amtgpu_cm_scene_changes(_ptr(m), n, W, H, _ptr(sc), n, C.byref(k)) == 1
    assert sc[:k.value].tolist() == FS.scene_changes(m, W, H) and 77 in sc[:k.value].tolist()


def _metric_stream(rng, n, W, H, kind):
    m = np.zeros((n, 8), np.uint64)
    if kind == 0:                                            # noise: every branch of the classifiers, ties included
        m[:, 0] = rng.randint(0, W * H * 6, n); m[:, 1] = rng.randint(0, W * H * 6, n)
        m[:, 3] = rng.randint(1, W * H * 10, n) // 1000 * 1000; m[:, 4] = rng.randint(1, W * H * 10, n) // 1000 * 1000
        return m
    for i in range(n):                                       # segments of 24p / 30p / 30i / noise with stills and cuts
        seg = (i // 60) % 4
        mot = 0 if (i // 25) % 7 == 3 else int(rng.randint(W * H, W * H * 3))
        if i % 97 == 0:
            mot *= 20
        base = int(rng.randint(W * H, 2 * W * H))
        if seg == 0:
            c = (base, base * 3) if i % 5 < 2 else ((base * 3, base) if i % 5 < 4 else (base, base))
        elif seg == 1:
            c = (base, base * 4)
        elif seg == 2:
            c = (base, base + int(rng.randint(0, W * H // 4)))
        else:
            c = (base * int(rng.randint(1, 4)), base * int(rng.randint(1, 4)))
        m[i, 0], m[i, 1], m[i, 3], m[i, 4] = mot, mot // 2, c[0], c[1]
    return m


@pytest.mark.parametrize("n", [0, 1, 2, 4, 5, 6, 9, 10, 11, 15, 16, 17, 31, 40, 333, 1500])
def test_sliding_window_decisions_match_the_oracle_at_every_length(lib, n):
    """the replicated decisions run in O(1) per frame (sorted sliding windows, per-offset 3:2 hit counts): same outputs as the
    oracle's per-frame window scans at every clip length around the window sizes (10 and 15 frames), on noise and on structure"""
    W, H = 352, 240
    for kind in (0, 1):
        m = _metric_stream(np.random.RandomState(1000 * kind + n), n, W, H, kind)
        cad, ph, sc, k = np.zeros(max(1, n), np.uint8), np.zeros(max(1, n), np.uint8), np.zeros(max(1, n), np.int32), C.c_int()
        assert lib.amtgpu_kfm_cadence(_ptr(m), n, W, H, _ptr(cad), _ptr(ph)) == 1
        assert lib.amtgpu_cm_scene_changes(_ptr(m), n, W, H, _ptr(sc), max(1, n), C.byref(k)) == 1
        ocad, oph = FS.classify_cadence(m, W, H)
        assert np.array_equal(cad[:n], ocad) and np.array_equal(ph[:n], oph)
        assert sc[:k.value].tolist() == FS.scene_changes(m, W, H)


@pytest.mark.parametrize("fps", [(30000, 1001), (24000, 1001), (60000, 1001), (25, 1), (5, 1), (1, 1)])
def test_logoframe_decisions_on_the_host_match_the_oracle(lib, fps):
    """amtgpu_logoframe_decide_host = LogoFrame::selectLogo + the text of writeResult (LogoScan.hpp:1647-1827) from scan records alone:
    bytes of the oracle's text (itself pinned to the real reference, tests/test_oracle_vs_ref.py) at clip lengths around the
    window sizes, for on/off runs, noise, zeros, infinities and NaNs (a logo whose blackScore is 0 scores 0/0)"""
    O = Oracle()
    rng = np.random.RandomState(fps[0] + fps[1])
    for t, n in enumerate([1, 2, 3, 7, 14, 15, 16, 29, 30, 31, 32, 61, 200, 1000, 5000, 20000]):
        nl = 1 + t % 3
        ev = np.zeros((n, nl, 2), np.float32)
        mode = t % 4
        for l in range(nl):
            on = ((np.arange(n) // (40 + 17 * l + t)) % 2).astype(np.float32)
            if mode == 0:
                ev[:, l, 0] = rng.uniform(-1, 1, n); ev[:, l, 1] = rng.uniform(-1, 1, n)
            elif mode == 1:
                ev[:, l, 0] = on * 0.9 - 0.1 + rng.uniform(-0.3, 0.3, n)
                ev[:, l, 1] = np.where(on > 0, rng.uniform(-0.05, 0.05, n), -0.7 + rng.uniform(-0.2, 0.2, n))
            elif mode == 2:
                ev[:, l, 0] = np.where(rng.randint(0, 3, n) == 0, 0, on); ev[:, l, 1] = np.where(rng.randint(0, 4, n) == 0, 0, -0.5 * on)
            else:
                ev[:, l, 0] = on * 0.9 - 0.1 + rng.uniform(-0.6, 0.6, n); ev[:, l, 1] = -0.3 + rng.uniform(-0.6, 0.6, n)
                ev[rng.randint(0, 50, n) == 0, l, 0] = np.inf
                ev[rng.randint(0, 60, n) == 0, l, 1] = np.nan
                # NaN evidence: corr0 = +inf with corr1 = -inf (max(0, inf) + min(0, -inf)); and the harmless mirror image
                k = rng.randint(0, 45, n) == 0
                ev[k, l, 0], ev[k, l, 1] = np.inf, -np.inf
                k = rng.randint(0, 70, n) == 0
                ev[k, l, 0], ev[k, l, 1] = -np.inf, np.inf
        ev = np.ascontiguousarray(ev)
        ob, orat = C.c_int(), C.c_float()
        O.lib.orc_logoframe_select(_ptr(ev), n, nl, -1, C.byref(ob), C.byref(orat))
        for li in (-1, nl - 1):
            want = C.create_string_buffer(1 << 20)
            wl = O.lib.orc_logoframe_write_result(_ptr(ev), n, nl, ob.value if li < 0 else li, fps[0], fps[1], want, len(want))
            assert wl >= 0
            best, ratio, tl = C.c_int(-2), C.c_float(), C.c_int()
            assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, -1, li, fps[0], fps[1], C.byref(best), C.byref(ratio), None, 0, C.byref(tl)) == 1
            got = C.create_string_buffer(max(1, tl.value))
            assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, -1, li, fps[0], fps[1], C.byref(best), C.byref(ratio), got, tl.value, C.byref(tl)) == 1
            assert best.value == ob.value and np.float32(ratio.value).tobytes() == np.float32(orat.value).tobytes()
            assert got.raw[:tl.value] == want.raw[:wl], (n, nl, li, mode)
    assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, -1, nl, fps[0], fps[1], None, None, None, 0, None) == 0       # logo index outside
    assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, nl + 1, -1, fps[0], fps[1], None, None, None, 0, None) == 0   # more candidates than logos
    small = C.create_string_buffer(4)
    tl = C.c_int()
    assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, -1, -1, fps[0], fps[1], None, None, small, 4, C.byref(tl)) == (0 if tl.value > 4 else 1)


def test_logoframe_text_survives_nan_evidence(lib):
    """one frame with (corr0, corr1) = (+inf, -inf) makes the evidence NaN; the sliding median used to search for it with `!=`
    and ran off its window.  Every position of the NaN frame in a short clip, and runs of them."""
    O = Oracle()
    for n in (1, 5, 17, 40, 200):
        for pos in sorted({0, 1, n // 2, n - 1}):
            for run in (1, 3, 20):
                ev = np.zeros((n, 1, 2), np.float32)
                ev[:, 0, 0] = 0.8
                ev[pos:pos + run, 0, 0], ev[pos:pos + run, 0, 1] = np.inf, -np.inf
                want = C.create_string_buffer(1 << 16)
                wl = O.lib.orc_logoframe_write_result(_ptr(ev), n, 1, 0, 30000, 1001, want, len(want))
                got = C.create_string_buffer(1 << 16)
                tl = C.c_int()
                assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, 1, -1, 0, 30000, 1001, None, None, got, len(got), C.byref(tl)) == 1
                assert got.raw[:tl.value] == want.raw[:wl], (n, pos, run)


def test_c_and_numpy_stat_oracles_agree():
    """the two restatements of the self-specified metrics (C for the CPU baseline, numpy for readability)"""
    orc = Oracle()
    for bits in (8, 10):
        clip = S.make_clip_np(5, 96, 38, 0x5EED0003, bits=bits, pitchY=104)
        Y = clip["Y"]
        out = np.zeros((5, 8), np.uint64)
        orc.lib.orc_frame_metrics(_ptr(Y), Y.strides[0], Y.shape[2], bits, 96, 38, 5, None, _ptr(out))
        assert np.array_equal(out, FS.frame_metrics(Y[:, :, :96]))


def test_cpp_filter_layer_builds(lib):
    """include/amt_filters.hpp (AMTAnalyzeLogo / AMTEraseLogo / LogoFrame over the C ABI) and its host test program compile
    with plain g++ against the library; running it needs a GPU (tests/test_gpu_filters_cpp.py)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(root, "tests", "cpp", "filters_host_test"))


def test_kfm_and_cm_output_file_contracts(tmp_path, lib):
    """The files the reference READS back from the external detectors, written by this build and parsed here exactly as the
    reference parses them: chapter_exe output (CMAnalyze.hpp:411-439), KFM timecode (FilteredSource.hpp:163-212: integer ms
    per frame, '# total:', base fps = the one of 60/120/240 x 1000/1001 with the smallest rounding error), durations."""
    import ctypes as C
    import re
    # cadence: 40 frames of 3:2 film (phase advancing), 20 of 30p, 20 of 60i
    n = 80
    cad = np.array([1] * 40 + [2] * 20 + [0] * 20, np.uint8)
    ph = np.array([i % 5 for i in range(40)] + [0] * 40, np.uint8)
    dur_path, tc_path, ch_path = (str(tmp_path / f) for f in ("k.duration.txt", "k.timecode.txt", "chapter_exe.txt"))
    nout, nout2 = C.c_int(), C.c_int()
    assert lib.amtgpu_kfm_write_durations(cad.ctypes.data, ph.ctypes.data, n, dur_path.encode(), C.byref(nout)) == 1
    assert lib.amtgpu_kfm_write_timecode(cad.ctypes.data, ph.ctypes.data, n, 30000, 1001, tc_path.encode(), C.byref(nout2)) == 1
    durs = [int(x) for x in open(dur_path).read().split()]
    assert sum(durs) == 2 * n and len(durs) == nout.value == nout2.value
    # --- readTimecodeFile ---
    tcs, total = [], None
    for line in open(tc_path).read().splitlines():
        if not line:
            continue
        m = re.search(r"#\s*total:\s*([+-]?([0-9]*[.])?[0-9]+).*", line)
        if m:
            total = float(m.group(1)) * 1000
            break
        if line[0] != "#":
            tcs.append(int(line))
    assert total is not None and len(tcs) == len(durs)
    tick = 1001.0 / 60.0
    want, t = [], 0
    for d in durs:
        want.append(int(t * tick + 0.5)); t += d
    assert tcs == want
    assert abs(total - n * 1001.0 / 30.0) < 1e-3                 # same duration as the source: within 0.1 s (FilteredSource.hpp:592-594)
    # --- readTimecode: base fps inference ---
    codes = tcs + [total]
    best, mind = None, codes[-1]
    for fps in (60, 120, 240):
        mult = fps / 1001.0
        diff = sum(abs(round(ts * mult) / mult - ts) for ts in codes)
        if diff < mind - len(codes) * 10e-10:
            best, mind = fps, diff
    assert best == 60
    # --- chapter_exe: readSceneChanges ---
    sc = np.array([97, 194, 900], np.int32)
    assert lib.amtgpu_cm_write_chapter_exe(sc.ctypes.data, len(sc), 1000, ch_path.encode()) == 1
    lines = open(ch_path).read().splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("----"))
    got = []
    for l in lines[k + 1:]:
        if re.search(r"mute\s*(\d+):\s*(\d+)\s*-\s*(\d+).*", l):
            continue
        m = re.search(r"\s*SCPos:\s*(\d+).*", l)
        if m:
            got.append(int(m.group(1)))
    assert got == sc.tolist()


def test_release_library_reads_no_experiment_knobs(lib):
    from amatsukaze_amd import binding
    """AMTGPU_DBG / AMTGPU_LDSPAD / AMTGPU_FPI / AMTGPU_G exist only in instrumented builds (-DAMT_EXPERIMENT,
    amatsukaze_amd/build.py build_variant): a stray environment variable must not be able to change the release
    library's results, so the strings are not even in it."""
    blob = open(binding.LIB_PATH, "rb").read()
    for knob in (b"AMTGPU_DBG", b"AMTGPU_LDSPAD", b"AMTGPU_FPI", b"AMTGPU_G\0", b"AMTGPU_VERBOSE"):
        assert knob not in blob, knob


def test_avisynth_plugin_registration(lib):
    """plugin/amt_plugin.cpp exports AvisynthPluginInit3 and registers the logo filters under the reference's names and
    argument specifications (Amatsukaze.cpp:58-59); registration itself needs no GPU."""
    import subprocess
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp, "all"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(cpp, "filters_host_test"), "--registration", os.path.join(cpp, "libamt_avs_plugin.so")],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[:2] == ["AMTAnalyzeLogo\tcs[maskratio]i", "AMTEraseLogo\tccs[logof]s[mode]i[maxfade]i"]
    assert lines[2] == "AMTAnalyzeLogoFast\tcs[maskratio]i"   # the opt-in linear-guarded analysis (not a reference name)
    assert len(lines) == 4 and lines[3]                       # the description string AviSynth shows
    sym = subprocess.run(["nm", "-D", "--defined-only", os.path.join(cpp, "libamt_avs_plugin.so")], capture_output=True, text=True).stdout
    assert " T AvisynthPluginInit3" in sym


def test_amts_layout_is_the_reference_structs(lib, tmp_path):
    """The byte layout amts_file.cpp parses is pinned to the reference's OWN definitions, not to this repo's reading of them:
    tests/golden/amts_ref_layout.json holds sizeof/offsetof of VideoFormat / AudioFormat / FilterSourceFrame / FilterAudioFrame /
    DecoderSetting as compiled from StreamUtils.hpp:526-536,633-641,778-781 and StreamReform.hpp:145-160 under -fshort-wchar
    (oracle/ref_shim/layout_probe.cpp), and amts_ref_sample.dat is a file those structs were fwrite()n into the way SaveAMTSource does
    (AMTSource.hpp:835-852), padding bytes poisoned with 0xEE.  Where oracle/_ref/layout_probe exists (this container) it is re-run and
    must reproduce the committed fixtures; everywhere, amtgpu_amts_load must read every field of the sample back."""
    import json
    import struct
    import amts_util as A
    gold = os.path.join(os.path.dirname(__file__), "golden")
    layout = json.load(open(os.path.join(gold, "amts_ref_layout.json")))
    sample = open(os.path.join(gold, "amts_ref_sample.dat"), "rb").read()
    probe = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "layout_probe")
    if os.path.exists(probe):
        live = json.loads(subprocess.run([probe, "layout"], capture_output=True, text=True, check=True).stdout)
        assert live == layout
        subprocess.run([probe, "sample", str(tmp_path / "s.dat")], check=True)
        assert (tmp_path / "s.dat").read_bytes() == sample
    # the test-side writer every other amts test uses packs the same layout
    assert layout["sizeof(wchar_t)"] == 2
    assert struct.calcsize("<9i3B2?3x") == layout["sizeof(VideoFormat)"] == 44
    assert [layout["VideoFormat." + k] for k in ("format", "width", "frameRateDenom", "colorPrimaries", "colorSpace", "progressive",
                                                  "fixedFrameRate")] == [0, 4, 32, 36, 38, 39, 40]
    assert struct.calcsize("<2i") == layout["sizeof(AudioFormat)"] and layout["AudioFormat.sampleRate"] == 4
    assert struct.calcsize("<?3xiddqqii") == layout["sizeof(FilterSourceFrame)"] == 48
    assert [layout["FilterSourceFrame." + k] for k in ("halfDelay", "frameIndex", "pts", "frameDuration", "framePTS", "fileOffset",
                                                        "keyFrame", "cmType")] == [0, 4, 8, 16, 24, 32, 40, 44]
    assert struct.calcsize("<i4xqi4x") == layout["sizeof(FilterAudioFrame)"] == 24
    assert [layout["FilterAudioFrame." + k] for k in ("frameIndex", "waveOffset", "waveLength")] == [0, 8, 16]
    assert struct.calcsize("<3i") == layout["sizeof(DecoderSetting)"] and layout["DecoderSetting.hevc"] == 8
    # the library reads the reference-written bytes
    path = tmp_path / "amts_ref.dat"
    path.write_bytes(sample)
    h = lib.amtgpu_amts_load(None, str(path).encode())
    assert h
    info = np.zeros(19, np.int32)
    nf, na = C.c_int(), C.c_int()
    assert lib.amtgpu_amts_get_info(h, _ptr(info), C.byref(nf), C.byref(na))
    assert info.tolist() == [layout["VS_H264"], 1440, 1080, 1440, 1080, 4, 3, 30000, 1001, 1, 6, 9, 0, 1, layout["AUDIO_32_LFE"], 48000,
                             0, layout["DECODER_CUVID"], 1]
    assert nf.value == 7 and na.value == 3
    b1, b2 = C.create_string_buffer(256), C.create_string_buffer(256)
    assert lib.amtgpu_amts_get_paths(h, b1, 256, b2, 256)
    assert b1.value.decode("utf-8") == "D:\\rec\\\u756a\u7d44 #12.ts" and b2.value.decode() == "C:\\tmp\\amt0\\a0-0.wav"
    gp, go = np.zeros(7, np.int64), np.zeros(7, np.int64)
    gk, gc, gh = np.zeros(7, np.int32), np.zeros(7, np.int32), np.zeros(7, np.uint8)
    assert lib.amtgpu_amts_get_frames(h, _ptr(gp), _ptr(go), _ptr(gk), _ptr(gh), _ptr(gc))
    want_pts = [(1 << 33) - 6006 + 3003 * (i - (1 if i >= 4 else 0)) for i in range(7)]
    assert gp.tolist() == want_pts
    assert go.tolist() == [5000000000 + 188 * i for i in range(7)]
    assert gk.tolist() == [i if i % 3 == 0 else -1 for i in range(7)]
    assert gh.tolist() == [0, 0, 0, 1, 1, 0, 0]
    assert gc.tolist() == [layout["CMTYPE_CM"] if i & 1 else 1 for i in range(7)]
    pics = [p & ((1 << 33) - 1) for p in sorted(set(want_pts))]
    top, bot = np.zeros(7, np.int32), np.zeros(7, np.int32)
    assert lib.amtgpu_amts_weave_plan(h, _ptr(np.array(pics, np.int64)), len(pics), _ptr(top), _ptr(bot))
    wt, wb = A.reference_plan(want_pts, [bool(x) for x in gh], pics)
    assert top.tolist() == wt and bot.tolist() == wb
    lib.amtgpu_amts_destroy(h)


def test_amts_stream_index_reader_and_weave_plan(lib, tmp_path):
    """amtgpu_amts_load parses a file laid out like SaveAMTSource's (AMTSource.hpp:835-852) and amtgpu_amts_weave_plan reproduces
    AMTSource::OnFrameOutput's picture -> frame matching (:482-566): half-delayed frames take the previous picture's top field."""
    import amts_util as A
    step = 3003                                                       # 90 kHz ticks of a 29.97 fps frame
    base = (1 << 33) - 4 * step                                       # the stream wraps its 33-bit PTS after four frames
    fpts, half = [], []
    for i in range(12):
        fpts.append(base + i * step); half.append(False)
    # a 3:2 pulldown stretch (:524-552): picture k yields a half-delayed frame AND a plain one with the same PTS
    for j in range(4):
        p = base + (12 + j) * step
        fpts += [p, p]; half += [True, False]
    for i in range(16, 22):
        fpts.append(base + i * step); half.append(i % 5 == 0)
    frames = [dict(halfDelay=h, frameIndex=i, pts=float(p), frameDuration=float(step), framePTS=p, fileOffset=188 * 1000 * i,
                   keyFrame=(i // 15) * 15, cmType=i % 3) for i, (p, h) in enumerate(zip(fpts, half))]
    path = tmp_path / "amts0.dat"
    vf = (1, 1440, 1080, 1440, 1080, 4, 3, 30000, 1001, 1, 1, 1, False, True)
    A.write_amts(path, "D:\\\\録画\\\\番組.ts", "D:\\\\tmp\\\\audio0.wav", vf, (2, 48000), frames, [(0, 0, 4096), (1, 4096, 4100)], (0, 2, 1))
    h = lib.amtgpu_amts_load(None, str(path).encode())
    assert h
    info = np.zeros(19, np.int32)
    nf, na = C.c_int(), C.c_int()
    assert lib.amtgpu_amts_get_info(h, _ptr(info), C.byref(nf), C.byref(na))
    assert info.tolist() == [1, 1440, 1080, 1440, 1080, 4, 3, 30000, 1001, 1, 1, 1, 0, 1, 2, 48000, 0, 2, 1]
    assert nf.value == len(frames) and na.value == 2
    b1, b2 = C.create_string_buffer(256), C.create_string_buffer(256)
    assert lib.amtgpu_amts_get_paths(h, b1, 256, b2, 256)
    assert b1.value.decode("utf-8") == "D:\\\\録画\\\\番組.ts" and b2.value.decode() == "D:\\\\tmp\\\\audio0.wav"
    assert not lib.amtgpu_amts_get_paths(h, b1, 4, b2, 256)            # buffer too small
    gp, go = np.zeros(nf.value, np.int64), np.zeros(nf.value, np.int64)
    gk, gc, gh = np.zeros(nf.value, np.int32), np.zeros(nf.value, np.int32), np.zeros(nf.value, np.uint8)
    assert lib.amtgpu_amts_get_frames(h, _ptr(gp), _ptr(go), _ptr(gk), _ptr(gh), _ptr(gc))
    assert gp.tolist() == fpts and gh.tolist() == [int(x) for x in half]
    assert go.tolist() == [f["fileOffset"] for f in frames] and gk.tolist() == [f["keyFrame"] for f in frames]
    assert gc.tolist() == [f["cmType"] for f in frames]
    # decoded pictures in output order: PTS truncated to 33 bits, one picture dropped (discontinuity), one unknown PTS
    pics = [base + i * step for i in range(22)]
    del pics[9]
    pics.insert(5, base + 5 * step - 7)
    pics = [p & ((1 << 33) - 1) for p in pics]
    top, bot = np.zeros(nf.value, np.int32), np.zeros(nf.value, np.int32)
    assert lib.amtgpu_amts_weave_plan(h, _ptr(np.array(pics, np.int64)), len(pics), _ptr(top), _ptr(bot))
    wt, wb = A.reference_plan(fpts, half, pics)
    assert top.tolist() == wt and bot.tolist() == wb
    assert (top < 0).sum() >= 1 and any(t != b for t, b in zip(wt, wb) if t >= 0)     # a lost frame and real two-picture weaves
    lib.amtgpu_amts_destroy(h)
    # truncated / corrupt files fail loudly, not with garbage
    data = path.read_bytes()
    for cut in (3, 40, len(data) - 5):
        bad = tmp_path / f"cut{cut}.dat"
        bad.write_bytes(data[:cut])
        assert not lib.amtgpu_amts_load(None, str(bad).encode())
    (tmp_path / "neg.dat").write_bytes(b"\\xff" * 8 + data[8:])
    assert not lib.amtgpu_amts_load(None, str(tmp_path / "neg.dat").encode())


@pytest.mark.parametrize("threads,grain", [(2, 1), (3, 7), (5, 50), (8, 1000), (32, 4096)])
def test_decisions_do_not_depend_on_how_the_clip_is_cut_over_threads(lib, threads, grain):
    """amtgpu_host_set_parallelism: the replicated decisions cut the clip into frame ranges (window filters are local; the state machines
    stay sequential).  Ranges shorter than every window, ranges of a few frames, many ranges: the oracle's outputs, byte for byte."""
    O = Oracle()
    W, H = 352, 240
    try:
        lib.amtgpu_host_set_parallelism(threads, grain)
        for n in (1, 2, 9, 31, 64, 333, 2500, 20011):
            rng = np.random.RandomState(n + threads)
            # ---- logo selection + logoframe text ----
            nl = 3
            ev = np.zeros((n, nl, 2), np.float32)
            for l in range(nl):
                on = ((np.arange(n) // (37 + 11 * l)) % 2).astype(np.float32)
                ev[:, l, 0] = on * 0.9 - 0.1 + rng.uniform(-0.5, 0.5, n)
                ev[:, l, 1] = np.where(on > 0, rng.uniform(-0.1, 0.1, n), -0.6 + rng.uniform(-0.3, 0.3, n))
                k = rng.randint(0, 90, n) == 0
                ev[k, l, 0], ev[k, l, 1] = np.inf, -np.inf
            ob, orat = C.c_int(), C.c_float()
            O.lib.orc_logoframe_select(_ptr(ev), n, nl, -1, C.byref(ob), C.byref(orat))
            want = C.create_string_buffer(1 << 20)
            wl = O.lib.orc_logoframe_write_result(_ptr(ev), n, nl, ob.value, 30000, 1001, want, len(want))
            got = C.create_string_buffer(1 << 20)
            best, ratio, tl = C.c_int(-2), C.c_float(), C.c_int()
            assert lib.amtgpu_logoframe_decide_host(_ptr(ev), n, nl, -1, -1, 30000, 1001, C.byref(best), C.byref(ratio), got, len(got), C.byref(tl)) == 1
            assert best.value == ob.value and np.float32(ratio.value).tobytes() == np.float32(orat.value).tobytes()
            assert got.raw[:tl.value] == want.raw[:wl], (n, threads, grain)
            # ---- scene changes + cadence ----
            if n > 2500:
                continue                                   # (the numpy oracle walks every window in Python)
            for kind in (0, 1):
                m = _metric_stream(np.random.RandomState(1000 * kind + n), n, W, H, kind)
                cad, ph, sc, k = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.int32), C.c_int()
                assert lib.amtgpu_kfm_cadence(_ptr(m), n, W, H, _ptr(cad), _ptr(ph)) == 1
                assert lib.amtgpu_cm_scene_changes(_ptr(m), n, W, H, _ptr(sc), n, C.byref(k)) == 1
                ocad, oph = FS.classify_cadence(m, W, H)
                assert np.array_equal(cad, ocad) and np.array_equal(ph, oph), (n, threads, grain, kind)
                assert sc[:k.value].tolist() == FS.scene_changes(m, W, H)
    finally:
        lib.amtgpu_host_set_parallelism(0, 0)


def test_release_build_defines_no_knob_and_knobs_are_fenced(tmp_path):
    """The kernels carry ablation macros that compute WRONG results by design, phase timers and shape parameters.  The release library
    is built with none of them (build.FLAGS / EXTRA_FLAGS carry no -DAMT_), and csrc/build_knobs.h -- included first by every source --
    turns any of them on the command line into a compile error unless the build says it is an instrumented variant."""
    import re
    import subprocess
    from amatsukaze_amd import build as B
    flags = list(B.FLAGS) + [f for v in B.EXTRA_FLAGS.values() for f in v]
    assert not [f for f in flags if f.startswith("-DAMT_")], flags
    csrc = os.path.join(ROOT, "amatsukaze_amd", "csrc")
    fence = open(os.path.join(csrc, "build_knobs.h")).read()
    fenced = set(re.findall(r"defined\((AMT_[A-Z0-9_]+)\)", fence)) - {"AMT_INSTRUMENTED_BUILD"}
    used = set()
    for f in os.listdir(csrc):
        text = open(os.path.join(csrc, f)).read()
        if f != "build_knobs.h":
            used |= set(re.findall(r"#\s*(?:ifn?def|if|elif)\b[^\n]*?\b(AMT_[A-Z0-9_]+)", text))
            used |= set(re.findall(r"defined\((AMT_[A-Z0-9_]+)\)", text))
        if f in B.SOURCES:
            first = re.search(r'^#include\s+[<"]([^>"]+)[>"]', text, re.M)
            assert first and first.group(1) == "build_knobs.h", f"{f}: build_knobs.h must be its first include"
    # internal helper macros (defined by the sources themselves, never on a command line) are not knobs
    internal = {"AMT_GPU_H", "AMT_HD", "AMT_TILE_HD", "AMT_HIP", "AMT_TRACE_SCOPE", "AMT_TRACE_CAT", "AMT_TRACE_CAT2", "AMT_TRACE_BLOCK_BEGIN",
                "AMT_LTICK", "AMT_PTICK", "AMT_PDUMP", "AMT_TICK", "AMT_STATS_LAUNCH", "AMT_LAUNCH", "AMT_PAIR_OCC_ATTR", "AMT_INSTRUMENTED_BUILD"}
    assert used - internal <= fenced, f"knobs the fence does not know: {sorted(used - internal - fenced)}"
    # ... and the fence bites: an ablation on the command line of a release-style compile fails, the instrumented spelling passes
    src = tmp_path / "t.cpp"
    src.write_text('#include "build_knobs.h"\nint main() { return 0; }\n')
    cc = ["g++", "-fsyntax-only", "-I", csrc, str(src)]
    assert subprocess.run(cc, capture_output=True).returncode == 0
    for knob in ("AMT_LIN_NO_FIXUP", "AMT_PAIR_NO_SUM", "AMT_LIN_G=3", "AMT_FUSED_TIMING", "AMT_STATS_ROWS8=8", "AMT_EXPERIMENT"):
        r = subprocess.run(cc + ["-D" + knob], capture_output=True, text=True)
        assert r.returncode != 0 and "build_variant" in r.stderr, knob
        assert subprocess.run(cc + ["-D" + knob, "-DAMT_INSTRUMENTED_BUILD"], capture_output=True).returncode == 0, knob
