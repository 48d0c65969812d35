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
                                          (1920, 1080, 10, 0), (720, 480, 8, 16), (48, 34, 8, 0), (16, 18, 8, 0),
                                          # a ragged row in an unpadded pitch: the last 16-byte column of the bottom row would straddle the
                                          # end of the frame's buffer -- these take the plain-load form of the kernel
                                          (362, 242, 8, 0), (362, 50, 8, 2), (366, 26, 8, 0), (354, 26, 8, 0), (46, 30, 10, 0), (360, 242, 8, 0)])
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


def test_contexts_on_cu_range_streams_run_side_by_side():
    """amtgpu_stream_create_cu_range: two contexts on complementary parts of the device -- the frame metrics on the first 64 compute
    units, the analysis on the others -- give the results of the whole-device run (a partition changes where a kernel runs, nothing
    else); ranges outside the device are refused."""
    import torch
    import amt_synth as S
    from amatsukaze_amd import AMTAnalyzeLogo, Context, FrameStats, Logo
    dev = torch.device("cuda:0")
    W, H, N = 352, 240, 300
    data, alpha, alphaUV = S.make_logo(96, 48)
    Y = S.make_clip_torch(N, W, H, 0x5EED0044, alpha, alphaUV, 224, 18, dev, period=40, fade=6, chroma=False)["Y"]
    whole = Context(0)
    ncu = whole.cu_count()
    assert ncu >= 128
    logo = Logo.from_planes(whole, data, 96, 48, W, H, 224, 18)
    want_m = torch.zeros((N, 8), dtype=torch.int64, device=dev)
    FrameStats(whole, W, H, 8).run_device(Y, want_m)
    want_a = torch.zeros((N, 33), dtype=torch.float32, device=dev)
    AMTAnalyzeLogo(whole, logo, 0.35).analyze_device(Y, 8, want_a)
    torch.cuda.synchronize()
    cm, ca = Context(0), Context(0)
    sm, sa = cm.use_cu_range(0, 64), ca.use_cu_range(64, ncu - 64)
    cur = torch.cuda.current_stream()
    sm.wait_stream(cur); sa.wait_stream(cur)
    got_m = torch.zeros((N, 8), dtype=torch.int64, device=dev)
    got_a = torch.zeros((N, 33), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    fs, an = FrameStats(cm, W, H, 8), AMTAnalyzeLogo(ca, logo, 0.35)
    for _ in range(3):
        fs.run_device(Y, got_m)
        an.analyze_device(Y, 8, got_a)
    torch.cuda.synchronize()
    assert torch.equal(got_m, want_m) and got_a.cpu().numpy().tobytes() == want_a.cpu().numpy().tobytes()
    assert not cm.lib.amtgpu_stream_create_cu_range(cm.h, ncu - 8, 16)
    assert b"outside the device" in cm.lib.amtgpu_last_error(cm.h)
