"""The host <-> HBM plumbing of the C ABI (amt_gpu_upload.hip): the pinned ring with its staging worker threads, registered host
ranges, gather / strided uploads, the scatter download, owned markers and the keep-alive -- what arrives must be what was sent,
whatever the thread count, size or alignment."""
import ctypes as C

import numpy as np
import pytest

from amatsukaze_amd import binding

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from amatsukaze_amd import Context
    ctx = Context(0)
    return dict(torch=torch, ctx=ctx, lib=ctx.lib, dev=torch.device("cuda:0"))


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_staged_upload_round_trips(env, threads):
    torch, ctx, lib = env["torch"], env["ctx"], env["lib"]
    ctx.check(lib.amtgpu_context_set_upload_threads(ctx.h, threads))
    rng = np.random.default_rng(threads)
    for n in (1, 4097, (1 << 20) + 3, (16 << 20) + 5, (50 << 20) + 1):           # below / above the parallel threshold, across ring slots
        host = rng.integers(0, 256, n, dtype=np.uint8)
        d = torch.zeros(n, dtype=torch.uint8, device=env["dev"])
        torch.cuda.synchronize()          # uploads run on the context's side stream, which is not ordered behind torch's fill
        for rep in range(3):                                                       # back-to-back jobs: workers still spinning from the last one
            ctx.check(lib.amtgpu_frames_upload(ctx.h, _p(d), C.c_void_p(host.ctypes.data), n))
        ctx.check(lib.amtgpu_frames_upload_wait(ctx.h))
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy(), host), n
    ctx.check(lib.amtgpu_context_set_upload_threads(ctx.h, 8))
    assert not lib.amtgpu_context_set_upload_threads(ctx.h, 0)


@pytest.mark.parametrize("threads", [1, 8])
def test_strided_and_gather_uploads(env, threads):
    torch, ctx, lib = env["torch"], env["ctx"], env["lib"]
    ctx.check(lib.amtgpu_context_set_upload_threads(ctx.h, threads))
    rng = np.random.default_rng(7)
    # strided: 3000 rows of 700 bytes out of a pitch-1024 host image into a pitch-768 device image (2.1 MB: parallel path at 8 threads)
    rows, w, hp, dp = 3000, 700, 1024, 768
    host = rng.integers(0, 256, (rows, hp), dtype=np.uint8)
    d = torch.zeros((rows, dp), dtype=torch.uint8, device=env["dev"])
    g = torch.zeros((40 * 80, 320), dtype=torch.uint8, device=env["dev"])
    torch.cuda.synchronize()              # (the side stream is not ordered behind torch's fills)
    ctx.check(lib.amtgpu_frames_upload_strided(ctx.h, _p(d), dp, C.c_void_p(host.ctypes.data), hp, w, rows))
    # gather: 40 separately allocated frames, rows [10, 90) x 300 bytes of each, destinations continuing one another
    frames = [rng.integers(0, 256, (100, 512), dtype=np.uint8) for _ in range(40)]
    ptrs = (C.c_void_p * 40)(*[f.ctypes.data + 10 * 512 + 16 for f in frames])
    ctx.check(lib.amtgpu_frames_upload_gather(ctx.h, _p(g), 320, ptrs, 512, 300, 80, 40))
    ctx.check(lib.amtgpu_frames_upload_wait(ctx.h))
    ctx.synchronize()
    got = d.cpu().numpy()
    assert np.array_equal(got[:, :w], host[:, :w]) and not got[:, w:].any()
    gg = g.cpu().numpy().reshape(40, 80, 320)
    for i, f in enumerate(frames):
        assert np.array_equal(gg[i, :, :300], f[10:90, 16:316]) and not gg[i, :, 300:].any()
    ctx.check(lib.amtgpu_context_set_upload_threads(ctx.h, 8))


def test_registered_host_ranges_skip_the_ring(env):
    torch, ctx, lib = env["torch"], env["ctx"], env["lib"]
    rng = np.random.default_rng(9)
    pool = rng.integers(0, 256, 24 << 20, dtype=np.uint8)                          # "the decoder's frame pool"
    ctx.check(lib.amtgpu_frames_register(ctx.h, C.c_void_p(pool.ctypes.data), pool.size))
    try:
        d = torch.zeros(8 << 20, dtype=torch.uint8, device=env["dev"])
        s = torch.zeros((2000, 512), dtype=torch.uint8, device=env["dev"])
        t = torch.zeros(2000, dtype=torch.uint8, device=env["dev"])
        torch.cuda.synchronize()
        off = (5 << 20) + 3
        ctx.check(lib.amtgpu_frames_upload(ctx.h, _p(d), C.c_void_p(pool.ctypes.data + off), d.numel()))
        rows, w, hp = 2000, 500, 2048
        ctx.check(lib.amtgpu_frames_upload_strided(ctx.h, _p(s), 512, C.c_void_p(pool.ctypes.data + 1000), hp, w, rows))
        # a source that only PARTLY lies inside the registered range goes through the ring like any pageable memory
        tail = np.concatenate([pool[-1000:], rng.integers(0, 256, 1000, dtype=np.uint8)])
        ctx.check(lib.amtgpu_frames_upload(ctx.h, _p(t), C.c_void_p(tail.ctypes.data), 2000))
        ctx.check(lib.amtgpu_frames_upload_wait(ctx.h))
        ctx.synchronize()
        assert np.array_equal(d.cpu().numpy(), pool[off:off + d.numel()])
        assert np.array_equal(s.cpu().numpy()[:, :w], np.lib.stride_tricks.as_strided(pool[1000:], (rows, w), (hp, 1)))
        assert np.array_equal(t.cpu().numpy(), tail)
    finally:
        ctx.check(lib.amtgpu_frames_unregister(ctx.h, C.c_void_p(pool.ctypes.data)))
    assert not lib.amtgpu_frames_unregister(ctx.h, C.c_void_p(pool.ctypes.data))       # not registered any more
    assert b"not registered" in lib.amtgpu_last_error(ctx.h)


def test_download_scatter(env):
    torch, ctx, lib = env["torch"], env["ctx"], env["lib"]
    rng = np.random.default_rng(11)
    src = rng.integers(0, 256, 100000, dtype=np.uint8)
    d = torch.from_numpy(src).to(env["dev"])
    a = np.zeros((50, 128), np.uint8)
    b = np.zeros((30, 64), np.uint8)
    pieces = (binding.Scatter * 2)(binding.Scatter(a.ctypes.data + 8, 128, 1000, 100, 50), binding.Scatter(b.ctypes.data, 64, 70000, 64, 30))
    ctx.check(lib.amtgpu_download_scatter(ctx.h, _p(d), src.size, pieces, 2))
    assert np.array_equal(a[:, 8:108], src[1000:6000].reshape(50, 100)) and not a[:, :8].any() and not a[:, 108:].any()
    assert np.array_equal(b, src[70000:70000 + 30 * 64].reshape(30, 64))
    bad = (binding.Scatter * 1)(binding.Scatter(b.ctypes.data, 64, 99990, 64, 1))       # reaches past the downloaded bytes
    assert not lib.amtgpu_download_scatter(ctx.h, _p(d), src.size, bad, 1)
    assert b"outside" in lib.amtgpu_last_error(ctx.h)


def test_owned_markers(env):
    torch, ctx, lib = env["torch"], env["ctx"], env["lib"]
    m1, m2 = lib.amtgpu_marker_create(ctx.h), lib.amtgpu_marker_create(ctx.h)
    assert m1 and m2 and m1 != m2
    ctx.check(lib.amtgpu_marker_wait_on(ctx.h, m1))                                    # never recorded: returns at once
    x = torch.ones(1 << 20, device=env["dev"])
    ctx.check(lib.amtgpu_marker_record_on(ctx.h, m1))
    ctx.check(lib.amtgpu_marker_wait_on(ctx.h, m1))
    lib.amtgpu_marker_destroy(ctx.h, m1)
    lib.amtgpu_marker_destroy(ctx.h, m2)
