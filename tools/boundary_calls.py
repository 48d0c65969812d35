"""Where a block of the filter layer spends its wall time: filters_host_test --bench against the call-tracing build of the library
(-DAMT_TRACE_CALLS: host-side begin / duration of every C ABI call and of the waits inside them).  Build the variant where hipcc is
(`python tools/boundary_calls.py --build`, it travels with the tree), run on the GPU box from the repo root:
    python tools/boundary_calls.py [frames] [ENV=VALUE ...] > gpurun_out/boundary_calls.txt"""
import collections, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
VDIR = os.path.join(ROOT, "amatsukaze_amd", "build", "trace_lib")
if "--build" in sys.argv:
    from amatsukaze_amd import build as B
    so = B.build_variant("trace", ["AMT_TRACE_CALLS"])
    os.makedirs(VDIR, exist_ok=True)
    shutil.copy(so, os.path.join(VDIR, "libamt_gpu.so"))
    print("built", os.path.join(VDIR, "libamt_gpu.so"))
    sys.exit(0)
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
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "filters_host_test"], stdout=subprocess.DEVNULL)
args = [a for a in sys.argv[1:] if "=" not in a]
n = int(args[0]) if args else 2048
env = dict(os.environ, LD_LIBRARY_PATH=VDIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
env.update(dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a))
r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "filters_host_test"), "--bench", "1440", "1080", str(n)] + paths + ["0"], capture_output=True, text=True, env=env,
                   timeout=600)
print(r.stdout.strip().splitlines()[-1][:600])
recs = [l.split() for l in r.stderr.splitlines() if l.startswith("amt_trace ")]
recs = [(x[1], float(x[2]), float(x[3])) for x in recs]
print(len(recs), "calls traced")
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, t0, d in recs:
    e = tot[name]; e[0] += 1; e[1] += d; e[2] = max(e[2], d)
for name, (c, s, m) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    slow = sum(1 for nm, _, d in recs if nm == name and d > 5000)
    print(f"{name:45s} calls {c:6d}  total {s / 1e3:9.1f} ms  max {m / 1e3:7.2f} ms  calls > 5 ms: {slow}")
# the blocks of the analysis passes on one clock: microseconds after the host began the block
blk = collections.defaultdict(list)
for name, t0, d in recs:
    if name.startswith("blk."):
        blk[name].append(d)
for name in sorted(blk):
    v = blk[name]
    print(f"{name:40s} per block (us): " + " ".join(f"{x:8.0f}" for x in v[2:18]))
# every traced call of two consecutive steady-state blocks of the first analysis pass (records are appended when a call ENDS)
ends = [i for i, r in enumerate(recs) if r[0] == "amtgpu_analyze_batch_host"]
if len(ends) > 6:
    print("--- all calls of blocks 4 and 5 (t = begin, us since the first call) ---")
    for nm, t, dd in recs[ends[3] + 1:ends[5] + 1]:
        if not nm.startswith(("blk.", "gpu.")):
            print(f"   t={t / 1e3:10.3f} ms  {dd / 1e3:8.3f} ms  {nm}")
# the call sequence around the first few slow calls (> 5 ms)
shown = 0
for i, (name, t0, d) in enumerate(recs):
    if d > 5000 and "scan" not in name and shown < 4:
        shown += 1
        print(f"--- around slow call #{i} ({name}, {d / 1e3:.2f} ms) ---")
        for nm, t, dd in recs[max(0, i - 14):i + 6]:
            print(f"   t={t / 1e3:10.3f} ms  {dd / 1e3:8.3f} ms  {nm}")
