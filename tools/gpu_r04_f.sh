#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/boundary_calls.py 2048 2>&1 | grep -E "^\{|^blk\.|^gpu\." | cut -c1-330
python tools/boundary_probe.py 6144 baseline keepalive_1000_0 2>&1 | tail -1 | cut -c1-1400
( timeout 900 python -m pytest tests/test_gpu_upload.py tests/test_gpu_parity.py tests/test_gpu_filters_cpp.py tests/test_gpu_golden.py tests/test_ingest.py -m gpu -x -q 2>&1 | tail -3 )
timeout 300 python tools/ingest_sweep.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, {n: (round(x['ingest_GBs'],1), round(x['pipelined_fps'])) for n,x in v.items() if isinstance(x,dict)})"
