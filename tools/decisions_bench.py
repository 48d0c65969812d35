"""Host-side decision routines over a whole 4-hour clip's records (431 568 frames, BASELINE configs[4]): the part of the sharded
end-to-end pass that runs REPLICATED on every rank after the exchanges, i.e. its Amdahl term.  Needs no GPU:
amtgpu_logoframe_decide_host / amtgpu_cm_scene_changes / amtgpu_kfm_cadence work on records in host memory.

    python tools/decisions_bench.py [frames] [reps]
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from amatsukaze_amd import binding  # noqa: E402


def synth(n, nl=3, seed=7):
    rng = np.random.RandomState(seed)
    on = ((np.arange(n) // 2700) % 2).astype(np.float32)
    ev = np.zeros((n, nl, 2), np.float32)
    for l in range(nl):
        ev[:, l, 0] = on * (0.9 - 0.2 * l) - 0.05 + rng.uniform(-0.25, 0.25, n)
        ev[:, l, 1] = np.where(on > 0, rng.uniform(-0.05, 0.05, n), -0.6 + rng.uniform(-0.2, 0.2, n))
    W, H = 1920, 1080
    m = np.zeros((n, 8), np.uint64)
    base = rng.randint(W * H, 3 * W * H, n).astype(np.uint64)
    cut = (np.arange(n) % 97) == 0
    m[:, 0] = base + cut * np.uint64(8 * W * H); m[:, 1] = base + cut * np.uint64(8 * W * H)
    seg = (np.arange(n) // 1800) % 3                  # 24p / 30i / 30p
    cyc = np.arange(n) % 5
    comb = rng.randint(W * H, 2 * W * H, n).astype(np.uint64)
    m[:, 3] = comb; m[:, 4] = comb
    p24 = seg == 0
    m[p24 & (cyc < 2), 4] *= np.uint64(3); m[p24 & (cyc >= 2) & (cyc < 4), 3] *= np.uint64(3)
    m[seg == 2, 4] *= np.uint64(3)
    still = (np.arange(n) // 500) % 7 == 3
    m[still, 0] = 1000; m[still, 1] = 1000
    return np.ascontiguousarray(ev), m, W, H


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 431568
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = binding.load()
    ev, m, W, H = synth(n)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    best, ratio, tl = C.c_int(), C.c_float(), C.c_int()
    text = C.create_string_buffer(1 << 22)
    cad, ph = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    sc, k = np.zeros(n, np.int32), C.c_int()
    t = {"logoframe_select_and_text": [], "cadence": [], "scene_changes": []}
    for _ in range(reps):
        t0 = time.perf_counter()
        assert lib.amtgpu_logoframe_decide_host(p(ev), n, 3, -1, -1, 30000, 1001, C.byref(best), C.byref(ratio), text, len(text), C.byref(tl)) == 1
        t1 = time.perf_counter()
        assert lib.amtgpu_kfm_cadence(p(m), n, W, H, p(cad), p(ph)) == 1
        t2 = time.perf_counter()
        assert lib.amtgpu_cm_scene_changes(p(m), n, W, H, p(sc), n, C.byref(k)) == 1
        t3 = time.perf_counter()
        for key, v in zip(t, (t1 - t0, t2 - t1, t3 - t2)):
            t[key].append(v * 1e3)
    h = hashlib.sha256(text.raw[:tl.value] + cad.tobytes() + ph.tobytes() + sc[:k.value].tobytes() + bytes([best.value & 255])).hexdigest()
    out = {"frames": n, "reps": reps, "host_cpus": os.cpu_count(), "ms_min": {key: round(min(v), 3) for key, v in t.items()},
           "ms_total_min": round(sum(min(v) for v in t.values()), 3), "decisions_sha256": h, "scene_changes": k.value, "text_bytes": tl.value,
           "cadence_histogram": np.bincount(cad, minlength=3).tolist()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
