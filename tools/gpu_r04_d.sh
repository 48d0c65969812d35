#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/boundary_calls.py 2048 > gpurun_out/d_calls_baseline.txt 2>&1; echo "calls rc=$?"
grep -E "^gpu\.|analyze_batch_host|upload_wait|upload_gather " gpurun_out/d_calls_baseline.txt
timeout 300 python tools/boundary_calls.py 2048 AMT_KEEPALIVE=1000,0 > gpurun_out/d_calls_keepalive.txt 2>&1; echo "calls rc=$?"
grep -E "^\{|^gpu\.|analyze_batch_host|upload_wait|upload_gather " gpurun_out/d_calls_keepalive.txt | cut -c1-400
( timeout 300 python -m pytest tests/test_gpu_upload.py -m gpu -x -q 2>&1 | tail -2 )
