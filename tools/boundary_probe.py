"""filters_host_test --bench under a list of environment variants (the frame source of round 3 that allocated a fresh frame per faded
picture, runtime knobs):
per variant the filter layer's rates and the latency histogram of the block-launching GetFrame calls.  Run on the GPU box from the repo root:
    python tools/boundary_probe.py [frames] > gpurun_out/boundary_probe.json"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import amt_synth as S
import numpy as np
from amatsukaze_amd import binding
# The .lgd files are written WITHOUT touching the GPU (amtgpu_logo_from_planes / _save are host code and take a NULL context): this
# process must not hold a HIP context while filters_host_test runs -- two processes with queues on one GPU are time-sliced by the
# driver's scheduler in 10 ms quanta, which is what round 3 read as an "idle-queue pick-up tick" (profiles/r04_notes.md).
lib = binding.load()
tmp = tempfile.mkdtemp()
paths = []
for i in range(3):
    data = S.make_logo(256, 128, seed=0x10600002 + i, strength=0.5 + 0.1 * i)[0] if i else S.make_logo(256, 128)[0]
    data = np.ascontiguousarray(data, np.float32)
    h = lib.amtgpu_logo_from_planes(None, 256, 128, 1, 1, 1440, 1080, 1120, 64, data.ctypes.data)
    p = os.path.join(tmp, f"logo{i}.lgd")
    assert h and lib.amtgpu_logo_save(None, h, p.encode(), f"b{i}".encode(), 1)
    lib.amtgpu_logo_destroy(h)
    paths.append(p)
subprocess.check_call(["make", "-C", "tests/cpp", "filters_host_test"], stdout=subprocess.DEVNULL)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
only = set(sys.argv[2:])
VARIANTS = [("baseline", {}), ("baseline_again", {}), ("swap_order", {"AMT_BENCH_SWAP": "1"}), ("fresh_fades_r03_source", {"AMT_BENCH_FRESH_FADES": "1"}),
            ("no_sdma", {"HSA_ENABLE_SDMA": "0"}), ("one_hw_queue", {"GPU_MAX_HW_QUEUES": "1"}),
            ("no_interrupt", {"HSA_ENABLE_INTERRUPT": "0"})]
out = {}
for name, env in VARIANTS:
    if only and name not in only:
        continue
    e = dict(os.environ); e.update(env)
    r = subprocess.run(["tests/cpp/filters_host_test", "--bench", "1440", "1080", str(n)] + paths + ["0"], capture_output=True, text=True, env=e, timeout=300)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out[name] = {k: v for k, v in d.items() if k.endswith("_fps") or k.endswith("_hist")}
    except Exception as ex:
        out[name] = {"error": (r.stderr or r.stdout)[-300:] + repr(ex)}
    print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
print(json.dumps(out))
