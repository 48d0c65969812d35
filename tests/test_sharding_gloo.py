"""world_size-2 `gloo` runs of the multi-GPU path's exchange steps (amatsukaze_amd/sharding.py) on CPU.
Per-shard numbers come from the oracle (checker side); the product code under test is the sharding logic."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as G
from amatsukaze_amd import sharding as SH
from amtlib import ROOT, Oracle, _ptr


def test_shard_range_partitions():
    for n in (0, 1, 7, 40, 10001):
        for w in (1, 2, 3, 8):
            r = [SH.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert SH.halo_range(10, 20, 25) == (2, 25) and SH.halo_range(0, 5, 100) == (0, 13)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = G.load()
        orc = Oracle()
        W, H, LW, LH, X, Y0, N = (g[k] for k in ("W", "H", "LW", "LH", "X", "Y0", "N"))
        N = N - 1                                   # 39 frames: ragged shards (20 + 19)
        Y, U, V = G.frames(g)
        first, last = SH.shard_range(N, rank, world)
        # --- sharded all-frames logo scan: local records, all_gather, decisions on the whole clip ---
        los = [orc.make_logo(g[k], LW, LH, W, H, X, Y0) for k in ("logo0", "logo1")]
        ds = []
        for l in los:
            d = orc.lib.orc_logo_deint(l); orc.lib.orc_logo_create_mask(d, 0.35, 1); ds.append(d)
        n_loc = last - first
        ev = np.zeros(n_loc * 2 * 2, np.float32)
        Ys = np.ascontiguousarray(Y[first:last])
        orc.lib.orc_logoframe_scan((C.c_void_p * 2)(*ds), 2, _ptr(Ys), Ys.strides[0], Ys.shape[2], 8, W, H, n_loc, _ptr(ev))
        full = SH.gather_frame_records(torch.from_numpy(ev.reshape(n_loc, 2, 2)), N).numpy()
        ok_scan = full.tobytes() == g["logoframe_evals"][:N].tobytes()
        # --- sharded logo generation: quota in stream order + exact all-reduce of the accumulators ---
        max_valid = 9
        so = orc.lib.orc_scan_create(LW, LH, 1, 1, 12)
        local_valid = [i for i in range(first, last) if g["scan_valid"][i]]
        quota = SH.stream_order_quota(len(local_valid), max_valid)
        for i in local_valid[:quota]:
            assert orc.lib.orc_scan_add_frame_u8(so, Y[i, Y0:, X:].ctypes.data, U[i, Y0 // 2:, X // 2:].ctypes.data,
                                                 V[i, Y0 // 2:, X // 2:].ctypes.data, Y.shape[2], U.shape[2]) == 1
        npx = LW * LH + 2 * (LW // 2) * (LH // 2)
        s5 = np.zeros(npx * 5)
        orc.lib.orc_scan_sums(so, _ptr(s5))
        s5 = s5.reshape(npx, 5)
        sums = torch.from_numpy(np.ascontiguousarray(s5[:, [0, 2, 4]]).astype(np.int64).reshape(-1))
        ysz, csz = LW * LH, (LW // 2) * (LH // 2)
        plane = torch.tensor([s5[0, 1], s5[0, 3], s5[ysz, 1], s5[ysz, 3], s5[ysz + csz, 1], s5[ysz + csz, 3]]).to(torch.int64)
        sums, plane, nfr = SH.allreduce_scan_sums(sums, plane, quota)
        # single-process answer: first max_valid valid frames of the first N
        so1 = orc.lib.orc_scan_create(LW, LH, 1, 1, 12)
        taken = 0
        for i in range(N):
            if taken >= max_valid:
                break
            taken += orc.lib.orc_scan_add_frame_u8(so1, Y[i, Y0:, X:].ctypes.data, U[i, Y0 // 2:, X // 2:].ctypes.data,
                                                   V[i, Y0 // 2:, X // 2:].ctypes.data, Y.shape[2], U.shape[2])
        w5 = np.zeros(npx * 5)
        orc.lib.orc_scan_sums(so1, _ptr(w5))
        w5 = w5.reshape(npx, 5)
        ok_sum = (nfr == taken == max_valid and np.array_equal(sums.numpy().reshape(npx, 3), w5[:, [0, 2, 4]].astype(np.int64))
                  and plane.tolist() == [int(w5[0, 1]), int(w5[0, 3]), int(w5[ysz, 1]), int(w5[ysz, 3]), int(w5[ysz + csz, 1]), int(w5[ysz + csz, 3])])
        q.put((rank, ok_scan, ok_sum, quota))
    finally:
        dist.destroy_process_group()


def test_sharded_scan_and_logo_generation_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok1 and ok2 for _, ok1, ok2, _ in res), res
    assert sum(r[3] for r in res) == 9


def _coll_worker(rank, world, port, q):
    """the host-memory collectives handed to the C ABI's sharded drivers (AmtGpuCollectives), called the way C calls them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = SH.TorchCollectives()
        send = np.array([rank * 10 + 1, rank * 10 + 2], np.int64)
        recv = np.zeros(2 * world, np.int64)
        ok = c.struct.allgather(None, send.ctypes.data, recv.ctypes.data, 16)
        buf = np.array([1, 2 ** 40 + rank, -5], np.int64)             # sums that do not fit a double's 24-bit float cousin
        ok2 = c.struct.allreduce_sum_i64(None, buf.ctypes.data, 3)
        q.put((rank, ok, recv.tolist(), ok2, buf.tolist(), c.struct.rank, c.struct.world, c.error is None))
    finally:
        dist.destroy_process_group()


def test_torch_collectives_callbacks_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_coll_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, recv, ok2, buf, srank, sworld, noerr in res:
        assert ok == 1 and ok2 == 1 and noerr and srank == rank and sworld == 2
        assert recv == [1, 2, 11, 12]
        assert buf == [2, 2 ** 41 + 1, -10]


def _fs_worker(rank, world, port, q):
    """per-rank metrics of a contiguous range with the frame before it as halo (numpy oracle = checker side),
    amtgpu_framestats_allgather through the C ABI with torch.distributed collectives, replicated decisions"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import frame_stats_oracle as FS
        import amt_synth as S
        from amatsukaze_amd import binding
        lib = binding.load()
        W, H, N = 64, 36, 47                                 # ragged: 24 + 23
        Y = np.concatenate([S.make_clip_np(24, W, H, 0x5EED0099, cadence="24p")["Y"], S.make_clip_np(23, W, H, 0x5EED0098, cadence="30i", start=24)["Y"]])
        whole = FS.frame_metrics(Y)
        first, last = SH.shard_range(N, rank, world)
        local = FS.frame_metrics(Y[first:last], prev_first=Y[first - 1] if first > 0 else None)
        coll = SH.TorchCollectives()
        fs = lib.amtgpu_framestats_create(None, W, H, 8)     # the exchange and the decisions are host code: no context needed
        out = np.zeros((N, 8), np.uint64)
        ok = lib.amtgpu_framestats_allgather(fs, coll.ref(), local.ctypes.data_as(C.c_void_p), first, last - first, N, out.ctypes.data_as(C.c_void_p))
        same = bool(ok) and np.array_equal(out, whole)
        cad, ph = np.zeros(N, np.uint8), np.zeros(N, np.uint8)
        lib.amtgpu_kfm_cadence(out.ctypes.data_as(C.c_void_p), N, W, H, cad.ctypes.data_as(C.c_void_p), ph.ctypes.data_as(C.c_void_p))
        ocad, oph = FS.classify_cadence(whole, W, H)
        sc, nsc = np.zeros(N, np.int32), C.c_int()
        lib.amtgpu_cm_scene_changes(out.ctypes.data_as(C.c_void_p), N, W, H, sc.ctypes.data_as(C.c_void_p), N, C.byref(nsc))
        dec = bool(np.array_equal(cad, ocad) and np.array_equal(ph, oph) and sc[:nsc.value].tolist() == FS.scene_changes(whole, W, H))
        # ranges that do not tile the clip (rank 1 claims one frame too few) and a rank-local failure: every rank returns 0, nobody hangs
        bad1 = lib.amtgpu_framestats_allgather(fs, coll.ref(), local.ctypes.data_as(C.c_void_p), first, last - first - (1 if rank == 1 else 0), N,
                                               out.ctypes.data_as(C.c_void_p))
        bad2 = lib.amtgpu_framestats_allgather(fs, coll.ref(), local.ctypes.data_as(C.c_void_p), first if rank == 0 else N, last - first, N,
                                               out.ctypes.data_as(C.c_void_p))
        lib.amtgpu_framestats_destroy(fs)
        q.put((rank, same, dec, int(bad1), int(bad2), coll.error is None))
    finally:
        dist.destroy_process_group()


def test_sharded_frame_metrics_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, dec, bad1, bad2, noerr in res:
        assert same, "gathered frame metrics differ from the single-process ones"
        assert dec, "replicated cadence / scene-change decisions differ"
        assert bad1 == 0 and bad2 == 0 and noerr
