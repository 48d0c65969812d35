"""Frame assembly (AMTSource::MergeField, AMTSource.hpp:291-355): the oracle restatement against its definition written
out in numpy (CPU), and the HIP kernel against the oracle (GPU, through the C ABI)."""
import numpy as np
import pytest

from amtlib import Oracle, _ptr


def make_pictures(rng, P, W, H, bits, pitchY, pitchUV, nv12):
    dt = np.uint8 if bits <= 8 else np.uint16
    hi = (1 << bits)
    Y = rng.integers(0, hi, (P, H, pitchY)).astype(dt)
    if nv12:
        UV = rng.integers(0, hi, (P, H // 2, pitchUV)).astype(dt)        # U0 V0 U1 V1 ...
        return Y, UV, None
    return Y, rng.integers(0, hi, (P, H // 2, pitchUV)).astype(dt), rng.integers(0, hi, (P, H // 2, pitchUV)).astype(dt)


def expect_weave(Y, U, V, top, bot, W, H, nv12):
    """definition: even rows from the top picture, odd rows from the bottom picture, per plane"""
    dY = np.where((np.arange(H) % 2 == 0)[:, None], Y[top, :, :W], Y[bot, :, :W])
    rows = (np.arange(H // 2) % 2 == 0)[:, None]
    if nv12:
        dU = np.where(rows, U[top, :, 0:W:2], U[bot, :, 0:W:2])
        dV = np.where(rows, U[top, :, 1:W:2], U[bot, :, 1:W:2])
    else:
        dU = np.where(rows, U[top, :, :W // 2], U[bot, :, :W // 2])
        dV = np.where(rows, V[top, :, :W // 2], V[bot, :, :W // 2])
    return dY, dU, dV


# heights are multiples of 4: Copy1 walks row pairs of the CHROMA planes too (AMTSource.hpp:294, 345-346)
CASES = [(64, 32, 8, 0, False), (70, 36, 8, 6, False), (64, 32, 8, 0, True), (66, 28, 10, 2, True), (352, 240, 12, 32, False)]


@pytest.mark.parametrize("W,H,bits,pad,nv12", CASES)
def test_oracle_merge_field_known_answer(W, H, bits, pad, nv12):
    rng = np.random.default_rng(W * 131 + H + bits)
    spY, spUV = W + pad, (W if nv12 else W // 2) + pad
    Y, U, V = make_pictures(rng, 3, W, H, bits, spY, spUV, nv12)
    orc = Oracle()
    dt = Y.dtype
    pY, pUV = W + 16, W // 2 + 8
    for top, bot in ((0, 0), (1, 2), (2, 0)):
        oY = np.full((H, pY), 7, dt); oU = np.full((H // 2, pUV), 7, dt); oV = np.full((H // 2, pUV), 7, dt)
        orc.lib.orc_merge_field(_ptr(Y[top]), _ptr(U[top]), _ptr(V[top]) if V is not None else None, _ptr(Y[bot]), _ptr(U[bot]),
                                _ptr(V[bot]) if V is not None else None, spY, spUV, int(nv12), bits, W, H, _ptr(oY), _ptr(oU), _ptr(oV),
                                pY, pUV)
        eY, eU, eV = expect_weave(Y, U, V, top, bot, W, H, nv12)
        assert np.array_equal(oY[:, :W], eY) and np.array_equal(oU[:, :W // 2], eU) and np.array_equal(oV[:, :W // 2], eV)
        assert np.all(oY[:, W:] == 7) and np.all(oU[:, W // 2:] == 7) and np.all(oV[:, W // 2:] == 7)   # padding untouched


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,bits,pad,nv12", CASES + [(1440, 1080, 8, 32, False), (1920, 1080, 10, 0, True)])
def test_gpu_weave_matches_oracle(W, H, bits, pad, nv12):
    import torch
    from amatsukaze_amd import Context, DeviceClip, weave_fields
    rng = np.random.default_rng(W * 7 + H + bits)
    P, N = 5, 6
    spY, spUV = W + pad, (W if nv12 else W // 2) + pad
    Y, U, V = make_pictures(rng, P, W, H, bits, spY, spUV, nv12)
    top = [0, 1, 2, 3, 4, 2]
    bot = [0, 2, 2, 4, 0, 1]
    dev = torch.device("cuda:0")
    tdt = torch.uint8 if bits <= 8 else torch.int16
    view = (lambda a: a) if bits <= 8 else (lambda a: a.view(np.int16))
    dYs, dUs = torch.from_numpy(view(Y)).to(dev), torch.from_numpy(view(U)).to(dev)
    dVs = torch.from_numpy(view(V)).to(dev) if V is not None else None
    pY, pUV = W + 16, W // 2 + 8
    out = DeviceClip(torch.full((N, H, pY), 7, dtype=tdt, device=dev), torch.full((N, H // 2, pUV), 7, dtype=tdt, device=dev),
                     torch.full((N, H // 2, pUV), 7, dtype=tdt, device=dev), W, H, bits)
    ctx = Context(0)
    weave_fields(ctx, dYs, dUs, dVs, out, top, bot, nv12)
    torch.cuda.synchronize()
    orc = Oracle()
    dt = Y.dtype
    for i in range(N):
        oY = np.full((H, pY), 7, dt); oU = np.full((H // 2, pUV), 7, dt); oV = np.full((H // 2, pUV), 7, dt)
        t, b = top[i], bot[i]
        orc.lib.orc_merge_field(_ptr(Y[t]), _ptr(U[t]), _ptr(V[t]) if V is not None else None, _ptr(Y[b]), _ptr(U[b]),
                                _ptr(V[b]) if V is not None else None, spY, spUV, int(nv12), bits, W, H, _ptr(oY), _ptr(oU), _ptr(oV),
                                pY, pUV)
        gY = out.Y[i].cpu().numpy().view(dt); gU = out.U[i].cpu().numpy().view(dt); gV = out.V[i].cpu().numpy().view(dt)
        assert gY.tobytes() == oY.tobytes() and gU.tobytes() == oU.tobytes() and gV.tobytes() == oV.tobytes()
    # identity indices (frame-coded stream): NULL index arrays
    out2 = DeviceClip(torch.zeros((P, H, pY), dtype=tdt, device=dev), torch.zeros((P, H // 2, pUV), dtype=tdt, device=dev),
                      torch.zeros((P, H // 2, pUV), dtype=tdt, device=dev), W, H, bits)
    weave_fields(ctx, dYs, dUs, dVs, out2, None, None, nv12)
    torch.cuda.synchronize()
    eY, eU, eV = expect_weave(Y, U, V, 3, 3, W, H, nv12)
    assert np.array_equal(out2.Y[3].cpu().numpy().view(dt)[:, :W], eY)
    assert np.array_equal(out2.U[3].cpu().numpy().view(dt)[:, :W // 2], eU)
    assert np.array_equal(out2.V[3].cpu().numpy().view(dt)[:, :W // 2], eV)


@pytest.mark.gpu
def test_gpu_frames_assembled_from_an_amts_index(tmp_path):
    """decoded pictures + an amts%d.dat stream index -> frames: amtgpu_amts_weave_plan feeds amtgpu_weave_fields_batch's index
    arrays (AMTSource::OnFrameOutput + MakeFrame, AMTSource.hpp:482-566, 357-366); half-delayed frames take their top field
    from the previous picture."""
    import torch
    import amts_util as A
    from amatsukaze_amd import AmtsFile, Context, DeviceClip, weave_fields
    W, H, bits, P = 352, 240, 8, 9
    rng = np.random.default_rng(99)
    Y, U, V = make_pictures(rng, P, W, H, bits, W, W // 2, False)
    step, base = 3003, 900000
    fpts, half = [], []
    for i in range(P):
        if i in (3, 6):                                   # picture i yields a half-delayed frame and a plain one (same PTS)
            fpts += [base + i * step] * 2; half += [True, False]
        else:
            fpts.append(base + i * step); half.append(i == 0)          # frame 0 half-delayed with nothing before it: cannot be made
    frames = [dict(halfDelay=h, frameIndex=i, pts=float(p), frameDuration=float(step), framePTS=p, fileOffset=188 * i, keyFrame=0, cmType=0)
              for i, (p, h) in enumerate(zip(fpts, half))]
    path = tmp_path / "amts0.dat"
    A.write_amts(path, "in.ts", "in.wav", (1, W, H, W, H, 1, 1, 30000, 1001, 1, 1, 1, False, True), (2, 48000), frames, [])
    af = AmtsFile(path)
    assert af.num_frames == len(frames) and af.info["width"] == W and af.srcpath == "in.ts"
    top, bot = af.weave_plan([base + i * step for i in range(P)])
    wt, wb = A.reference_plan(fpts, half, [base + i * step for i in range(P)])
    assert top.tolist() == wt and bot.tolist() == wb and top[0] == -1 and (top[1:] >= 0).all()
    made = np.nonzero(top >= 0)[0]
    dev = torch.device("cuda:0")
    out = DeviceClip(torch.zeros((len(made), H, W), dtype=torch.uint8, device=dev), torch.zeros((len(made), H // 2, W // 2), dtype=torch.uint8, device=dev),
                     torch.zeros((len(made), H // 2, W // 2), dtype=torch.uint8, device=dev), W, H, bits)
    ctx = Context(0)
    weave_fields(ctx, torch.from_numpy(Y).to(dev), torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev), out, top[made], bot[made], False)
    torch.cuda.synchronize()
    orc = Oracle()
    for j, i in enumerate(made):
        oY = np.zeros((H, W), np.uint8); oU = np.zeros((H // 2, W // 2), np.uint8); oV = np.zeros((H // 2, W // 2), np.uint8)
        t, b = int(top[i]), int(bot[i])
        orc.lib.orc_merge_field(_ptr(Y[t]), _ptr(U[t]), _ptr(V[t]), _ptr(Y[b]), _ptr(U[b]), _ptr(V[b]), W, W // 2, 0, bits, W, H,
                                _ptr(oY), _ptr(oU), _ptr(oV), W, W // 2)
        assert out.Y[j].cpu().numpy().tobytes() == oY.tobytes() and out.U[j].cpu().numpy().tobytes() == oU.tobytes()
        assert out.V[j].cpu().numpy().tobytes() == oV.tobytes()
    assert any(top[i] != bot[i] for i in made)
