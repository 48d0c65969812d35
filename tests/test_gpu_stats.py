"""GPU whole-frame metrics vs the numpy oracle of the SELF-SPECIFIED passes (parity unpinned vs the
reference: it has no in-tree arithmetic for them).  Integer metrics -> bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import amt_synth as S
import frame_stats_oracle as FS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from amatsukaze_amd import Context
    return dict(torch=torch, ctx=Context(0), dev=torch.device("cuda:0"))


@pytest.mark.parametrize("W,H,bits,pad", [(352, 240, 8, 0), (360, 242, 8, 24), (1440, 1080, 8, 32), (352, 240, 10, 0), (2304, 64, 8, 0),
                                          (1920, 1080, 10, 0), (720, 480, 8, 16), (48, 34, 8, 0), (16, 18, 8, 0)])
def test_frame_metrics_bit_exact(gpu, W, H, bits, pad):
    from amatsukaze_amd import DeviceClip, FrameStats
    torch = gpu["torch"]
    N = 70 if W < 400 else (37 if W < 1000 else 5)      # 37 / 70: more than one frame run, ragged tail
    clip = S.make_clip_np(N, W, H, 0x5EED0003, bits=bits, pitchY=W + pad, pitchUV=W // 2 + pad // 2)
    Y = clip["Y"]
    t = torch.from_numpy(Y.view(np.uint8 if bits <= 8 else np.int16)).to(gpu["dev"])
    dclip = DeviceClip(t, t[:, :H // 2, :W // 2], t[:, :H // 2, :W // 2], W, H, bits)
    fs = FrameStats(gpu["ctx"], W, H, bits)
    got = fs.run(dclip)
    want = FS.frame_metrics(Y[:, :, :W])
    assert np.array_equal(got, want)
    # with an explicit previous frame for frame 0 (sharded clips: the halo frame)
    out = torch.zeros((N - 3, 8), dtype=torch.int64, device=gpu["dev"])
    fs.run_device(t[3:], out, prevY=t[2])
    gpu["ctx"].synchronize()
    assert np.array_equal(out.cpu().numpy().astype(np.uint64), want[3:])


def test_cadence_and_scene_decisions(gpu, tmp_path):
    """mixed 24p/30i/30p clip: GPU metrics -> host decisions == oracle decisions; 3:2 segments are found."""
    import ctypes as C
    from amatsukaze_amd import DeviceClip, FrameStats
    torch = gpu["torch"]
    W, H = 352, 240
    segs = [("24p", 60), ("30i", 40), ("30p", 40), ("24p", 35)]
    parts, start = [], 0
    for cad, n in segs:
        parts.append(S.make_clip_np(n, W, H, 0x5EED0003, cadence=cad, start=start)["Y"])
        start += n
    Y = np.concatenate(parts)
    t = torch.from_numpy(Y).to(gpu["dev"])
    fs = FrameStats(gpu["ctx"], W, H, 8)
    m = fs.run(DeviceClip(t, t, t, W, H))
    assert np.array_equal(m, FS.frame_metrics(Y))
    cad, ph = fs.cadence(m)
    ocad, oph = FS.classify_cadence(m, W, H)
    assert np.array_equal(cad, ocad) and np.array_equal(ph, oph)
    sc = fs.scene_changes(m)
    assert sc.tolist() == FS.scene_changes(m, W, H)
    # the synthetic clip cuts scenes every 97 frames
    assert 97 in sc.tolist()
    # segment interiors are classified as generated (edges may lag by the 10-frame window)
    assert (cad[10:50] == 1).mean() > 0.9
    assert (cad[70:95] == 0).mean() > 0.9
    assert (cad[110:135] == 2).mean() > 0.9
    # duration file contract: integers, sum == 2*N (60p ticks)
    n_out = C.c_int()
    path = tmp_path / "kfm.duration.txt"
    assert gpu["ctx"].lib.amtgpu_kfm_write_durations(cad.ctypes.data, ph.ctypes.data, len(cad), str(path).encode(), C.byref(n_out)) == 1
    d = [int(x) for x in path.read_text().split()]
    assert d == FS.cadence_durations(cad, ph) and sum(d) == 2 * len(cad) and len(d) == n_out.value
    assert d.count(3) > 10


def test_config3_frame_size_cadence_segments(gpu):
    """BASELINE configs[2]'s shape (1920x1080 8-bit, 24p / 30i / 30p segments, band-limited grain, picture-replacing scene cuts)
    generated on the device, 600 frames per segment: metrics of probe blocks equal the numpy oracle's bytes at the full frame size,
    the host decisions equal the oracle's on the whole clip's metrics, AND the detectors find what the generator put there --
    the cadence of every segment and the scene cuts every 97 frames (bench.py reports the same figures at 18 000 frames)."""
    import amt_synth as S
    from amatsukaze_amd import DeviceClip, FrameStats
    torch = gpu["torch"]
    W, H, SEG = 1920, 1080, 600
    parts, start = [], 0
    for cad in ("24p", "30i", "30p"):
        parts.append(S.make_clip_torch(SEG, W, H, 0x5EED0003, None, None, 0, 0, gpu["dev"], cadence=cad, start=start, chroma=False,
                                       noise="soft")["Y"])
        start += SEG
    Y = torch.cat(parts)
    del parts
    N = Y.shape[0]
    fs = FrameStats(gpu["ctx"], W, H, 8)
    out = torch.zeros((N, 8), dtype=torch.int64, device=gpu["dev"])
    fs.run_device(Y, out)
    gpu["ctx"].synchronize()
    m = out.cpu().numpy().astype(np.uint64)
    for b0 in (0, SEG - 12, 2 * SEG - 12, N - 24):            # clip start, both cadence changes, clip end
        blk = Y[max(0, b0 - 1):b0 + 24].cpu().numpy()
        want = FS.frame_metrics(blk)
        assert np.array_equal(m[b0:b0 + 24], want[(1 if b0 > 0 else 0):]), b0
    cad, ph = fs.cadence(m)
    ocad, oph = FS.classify_cadence(m, W, H)
    assert np.array_equal(cad, ocad) and np.array_equal(ph, oph)
    sc = fs.scene_changes(m).tolist()
    assert sc == FS.scene_changes(m, W, H)
    # ground truth: 24p -> 1, 30i -> 0, 30p -> 2 (10 frames after a cadence change belong to the classifier's window)
    for k, code in enumerate((1, 0, 2)):
        assert (cad[k * SEG + 10:(k + 1) * SEG] == code).mean() > 0.98, (k, code)
    cuts = set(range(97, N, 97))
    assert len(cuts & set(sc)) >= 0.95 * len(cuts) and len(set(sc) - cuts) <= 2
