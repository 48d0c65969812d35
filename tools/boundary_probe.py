"""filters_host_test --bench at two clip lengths (fixed vs per-frame cost of the filter layer); run on the GPU box from the repo root"""
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
subprocess.check_call(["make", "-C", "tests/cpp", "filters_host_test"], stdout=subprocess.DEVNULL)
for n in [int(x) for x in sys.argv[1:]] or [2048, 8192]:
    r = subprocess.run(["tests/cpp/filters_host_test", "--bench", "1440", "1080", str(n)] + paths + ["0"], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(n, {k: v for k, v in d.items() if k.endswith("_fps")})
print(" ".join(paths))
