#!/bin/bash
# round 4: kernel timeline of three e2e10 chunks (where the time between the kernels goes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/h_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/h_trace -o run -- python bench.py --workload e2e10 --no-verify --e2e-frames 12288 > gpurun_out/h_e2e.json 2> gpurun_out/h_e2e.err
echo "rc=$?"; ls -la gpurun_out/h_trace/* | head
