"""filters_host_test --bench under a list of environment variants (the keep-alive settings of amtgpu_context_set_keepalive, runtime knobs):
per variant the filter layer's rates and the latency histogram of the block-launching GetFrame calls.  Run on the GPU box from the repo root:
    python tools/boundary_probe.py [frames] > gpurun_out/boundary_probe.json"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import amt_synth as S
from amatsukaze_amd import Context, Logo
ctx = Context(0)
tmp = tempfile.mkdtemp()
paths = []
for i in range(3):
    data = S.make_logo(256, 128, seed=0x10600002 + i, strength=0.5 + 0.1 * i)[0] if i else S.make_logo(256, 128)[0]
    l = Logo.from_planes(ctx, data, 256, 128, 1440, 1080, 1120, 64)
    p = os.path.join(tmp, f"logo{i}.lgd"); l.save(p, f"b{i}", 1); paths.append(p)
del ctx
subprocess.check_call(["make", "-C", "tests/cpp", "filters_host_test"], stdout=subprocess.DEVNULL)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
only = set(sys.argv[2:])
VARIANTS = [("baseline", {}), ("baseline_again", {}), ("swap_order", {"AMT_BENCH_SWAP": "1"}),
            ("keepalive_1000_0", {"AMT_KEEPALIVE": "1000,0"}), ("keepalive_200_0", {"AMT_KEEPALIVE": "200,0"}),
            ("keepalive_1000_1000", {"AMT_KEEPALIVE": "1000,1000"}), ("keepalive_5000_5000", {"AMT_KEEPALIVE": "5000,5000"}),
            ("keepalive_200_200", {"AMT_KEEPALIVE": "200,200"}),
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
        out[name] = {k: v for k, v in d.items() if k.endswith("_fps") or k.endswith("_hist") or k == "keepalive_us"}
    except Exception as ex:
        out[name] = {"error": (r.stderr or r.stdout)[-300:] + repr(ex)}
    print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
print(json.dumps(out))
