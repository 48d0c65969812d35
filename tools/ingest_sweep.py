"""amtgpu_frames_upload with 1..16 staging threads, and from a registered pool (tools/bench_ingest.py); run on the GPU box from the repo root"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import amt_synth as S
import bench_ingest
from amatsukaze_amd import Context, Logo
ctx = Context(0)
dev = torch.device("cuda:0")
data, alpha, alphaUV = S.make_logo(256, 128)
logos = [Logo.from_planes(ctx, data, 256, 128, 1440, 1080, 1120, 64)]
K = dict(W=1440, H=1080, PITCH_Y=1472, PITCH_UV=768, IMGX=1120, IMGY=64, LH=128, MASKRATIO=0.35, ROOT=ROOT)
out = {}
for t in (1, 2, 4, 8, 16):
    r = bench_ingest.measure(ctx, logos, alpha, alphaUV, dev, K, threads=t, with_link=False)
    out[f"threads_{t}"] = {k: r[k] for k in ("y_plane", "logo_rectangle_rows", "y_plane_registered", "logo_rectangle_rows_registered", "register_GBs")}
    print(t, json.dumps(out[f"threads_{t}"]["y_plane"]), file=sys.stderr, flush=True)
print(json.dumps(out))
