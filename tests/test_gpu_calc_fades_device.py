"""CalcFade / CalcFade2 on the device (amtgpu_erase_calc_fades_device, LogoScan.hpp:1263-1341) against the CPU oracle's
orc_calc_fade: bytes, on crafted and random analysis records -- clip ends (the n < 8 negative-index quirk and the last
analyze frame), ties, NaNs, the abrupt branch, logoframe files, shard windows with an 8-frame halo -- and the whole
analyse -> decide -> erase chain without a host round trip against the host-decided one."""
import ctypes as C

import numpy as np
import pytest

from amtlib import _ptr
from test_gpu_parity import SMALL, gpu, make_case      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

LOGOF = ("    14 S 0 ALL     12     17\n    20 E 0 ALL     18     23\n    30 S 0 ALL     29     33\n"
         "    39 E 0 ALL     38     39\n")


def oracle_fades(orc, an, n, text="", maxfade=16):
    fr = np.zeros(max(1, n), np.int32)
    if text:
        assert orc.lib.orc_read_logoframe(text.encode(), n, _ptr(fr)) == 0
    want = np.zeros((n, 2), np.float32)
    an = np.ascontiguousarray(an, np.float32)
    for i in range(n):
        ft, fb = C.c_float(), C.c_float()
        orc.lib.orc_calc_fade(_ptr(fr) if text else None, 1 if text else 0, maxfade, _ptr(an), n, i, C.byref(ft), C.byref(fb))
        want[i] = (ft.value, fb.value)
    return want


def records(rng, n, kind):
    an = rng.rand(n, 33).astype(np.float32) + 0.5
    if kind == "ties":            # quantised scores: equal minima, the first one must win
        an = np.round(an * 4) / 4
    elif kind == "nan":
        an[rng.rand(n, 33) < 0.1] = np.nan
        an[rng.rand(n) < 0.05] = np.nan          # whole records of NaN: argmin stays at 0
    elif kind == "switch":        # the logo comes and goes abruptly every 13 frames: the per-field branch fires
        for i in range(n):
            b = 10 if (i // 13) % 2 else 0
            an[i, b] = 0.01
            an[i, 11 + (3 if i % 13 == 0 else b)] = 0.001
            an[i, 22 + (8 if i % 13 == 0 else b)] = 0.002
    return an.astype(np.float32)


@pytest.mark.parametrize("n", [1, 5, 8, 9, 17, 45, 1000])
@pytest.mark.parametrize("kind", ["random", "ties", "nan", "switch"])
def test_device_fades_equal_oracle(gpu, n, kind):
    from amatsukaze_amd import AMTEraseLogo
    torch = gpu["torch"]
    cs = make_case(gpu, dict(SMALL, N=1))
    an = records(np.random.RandomState(n * 7 + len(kind)), n, kind)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    d_an = torch.from_numpy(an).to(gpu["dev"])
    got = er.calc_fades_device(d_an, n).cpu().numpy()
    want = oracle_fades(cs["orc"], an, n)
    assert got.tobytes() == want.tobytes()
    assert er.calc_fades(an, n).tobytes() == want.tobytes()          # and the host routine agrees
    if kind == "switch" and n >= 45:
        assert any(a != b for a, b in want.tolist())                 # the abrupt branch fired


@pytest.mark.parametrize("maxfade", [16, 4, 0])
def test_device_fades_with_logoframe_file(gpu, maxfade):
    from amatsukaze_amd import AMTEraseLogo
    torch = gpu["torch"]
    cs = make_case(gpu, dict(SMALL, N=1))
    n = 40
    an = records(np.random.RandomState(3), n, "switch")
    er = AMTEraseLogo(gpu["ctx"], cs["logo"], LOGOF, 0, maxfade)
    got = er.calc_fades_device(torch.from_numpy(an).to(gpu["dev"]), n).cpu().numpy()
    assert got.tobytes() == oracle_fades(cs["orc"], an, n, LOGOF, maxfade).tobytes()


def test_device_fades_of_a_shard_window(gpu):
    """a shard hands over the records of its own frames plus 8 either side (CalcFade2's window) and gets the whole clip's answers"""
    from amatsukaze_amd import AMTEraseLogo, AmtError
    torch = gpu["torch"]
    cs = make_case(gpu, dict(SMALL, N=1))
    n = 203
    an = records(np.random.RandomState(5), n, "switch")
    want = oracle_fades(cs["orc"], an, n)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"])
    for first, cnt in ((0, 50), (50, 75), (125, 78), (195, 8), (0, 203), (7, 1)):
        a0, a1 = max(0, first - 8), min(n, first + cnt + 8)
        d = torch.from_numpy(an[a0:a1]).to(gpu["dev"])
        got = er.calc_fades_device(d, n, first, cnt, analysis_first=a0).cpu().numpy()
        assert got.tobytes() == want[first:first + cnt].tobytes(), (first, cnt)
    with pytest.raises(AmtError, match="do not cover"):              # a window without its halo is refused, not read out of bounds
        er.calc_fades_device(torch.from_numpy(an[50:100]).to(gpu["dev"]), n, 50, 50, analysis_first=50)
    with pytest.raises(AmtError, match="outside the clip"):
        er.calc_fades_device(torch.from_numpy(an).to(gpu["dev"]), n, 200, 10)


@pytest.mark.parametrize("bits", [8, 10])
def test_stream_ordered_analyse_decide_erase(gpu, bits):
    """analysis records, fades and the erase stay on the device (no synchronise in between): erased frames == the host-decided path's"""
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, DeviceClip
    torch = gpu["torch"]
    cs = make_case(gpu, SMALL, bits=bits, pitch_pad=32)
    dc = cs["dclip"]
    n = dc.num_frames
    an = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"], "", 0, 16)
    host_clip = DeviceClip(dc.Y.clone(), dc.U.clone(), dc.V.clone(), dc.width, dc.height, bits)
    rec = an.analyze(host_clip)
    fades = er.calc_fades(rec, n)
    er.erase(host_clip, fades)
    d_rec = torch.empty((n, 33), dtype=torch.float32, device=gpu["dev"])
    an.analyze_device(dc.Y, bits, d_rec)
    d_f = er.calc_fades_device(d_rec, n)
    er.erase_device_fades(dc, d_f)
    gpu["ctx"].synchronize()
    assert d_f.cpu().numpy().tobytes() == fades.tobytes()
    assert torch.equal(dc.Y, host_clip.Y) and torch.equal(dc.U, host_clip.U) and torch.equal(dc.V, host_clip.V)
    assert np.abs(fades).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 10, 16])
def test_erase_into_a_writable_copy(gpu, bits):
    """amtgpu_erase_batch_dfades_to: Delogo reads the SOURCE batch and writes the rectangle into a destination that already holds a copy
    of the frames (AMTEraseLogo::GetFrameT's env->MakeWritable, LogoScan.hpp:1346-1347).  The destination equals the in-place result
    byte for byte, the source is untouched, and a second call on the same pair changes nothing (the bench's steps rely on that instead of
    restoring the rectangles)."""
    from amatsukaze_amd import AMTAnalyzeLogo, AMTEraseLogo, DeviceClip
    torch = gpu["torch"]
    cs = make_case(gpu, SMALL, bits=bits, pitch_pad=32)
    src = cs["dclip"]
    n = src.num_frames
    an = AMTAnalyzeLogo(gpu["ctx"], cs["logo"], 0.35)
    er = AMTEraseLogo(gpu["ctx"], cs["logo"], "", 0, 16)
    d_rec = torch.empty((n, 33), dtype=torch.float32, device=gpu["dev"])
    an.analyze_device(src.Y, bits, d_rec)
    d_f = er.calc_fades_device(d_rec, n)
    inplace = DeviceClip(src.Y.clone(), src.U.clone(), src.V.clone(), src.width, src.height, bits)
    er.erase_device_fades(inplace, d_f)
    keep = [src.Y.clone(), src.U.clone(), src.V.clone()]
    dst = DeviceClip(src.Y.clone(), src.U.clone(), src.V.clone(), src.width, src.height, bits)
    for _ in range(2):
        er.erase_device_fades(src, d_f, dst=dst)
        gpu["ctx"].synchronize()
        assert torch.equal(dst.Y, inplace.Y) and torch.equal(dst.U, inplace.U) and torch.equal(dst.V, inplace.V)
        assert torch.equal(src.Y, keep[0]) and torch.equal(src.U, keep[1]) and torch.equal(src.V, keep[2])
    assert not torch.equal(dst.Y, src.Y) and float(d_f.abs().sum()) > 0
    with pytest.raises(ValueError):
        er.erase_device_fades(src, d_f, dst=DeviceClip(src.Y[:-1].clone(), src.U[:-1].clone(), src.V[:-1].clone(), src.width, src.height, bits))
