#!/bin/bash
# round 4: where the e2e10 chunk time goes (fenced diagnostic), e2e10 at HEAD, configs[2] interior cadence misses
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
AMT_E2E_PHASES=1 timeout 600 python bench.py --workload e2e10 --no-verify > gpurun_out/g_e2e_phases.json 2> gpurun_out/g_e2e_phases.err; echo "phases rc=$?"
timeout 600 python bench.py --workload e2e10 > gpurun_out/g_e2e.json 2> gpurun_out/g_e2e.err; echo "e2e rc=$?"
timeout 600 python - > gpurun_out/g_kfm.json 2> gpurun_out/g_kfm.err <<'PY'
import sys, json, types
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench
from amatsukaze_amd import Context
ctx = Context(0)
logos_np, alpha, alphaUV = bench.make_logos()
r = bench.config_kfm(ctx, torch.device("cuda:0"), logos_np, alpha, alphaUV, types.SimpleNamespace())
print(json.dumps(r))
PY
echo "kfm rc=$?"; tail -c 300 gpurun_out/g_kfm.err
