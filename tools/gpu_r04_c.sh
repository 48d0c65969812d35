#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_upload.py tests/test_gpu_stats.py tests/test_gpu_parity.py tests/test_gpu_filters_cpp.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c_pytest.log )
tail -4 gpurun_out/c_pytest.log
timeout 900 python tools/stats_bench.py > gpurun_out/c_stats_bench.json 2> gpurun_out/c_stats_bench.err; echo "stats rc=$?"; cat gpurun_out/c_stats_bench.err | tail -12
timeout 300 python tools/boundary_calls.py 2048 > gpurun_out/c_calls_baseline.txt 2>&1; echo "calls rc=$?"
head -40 gpurun_out/c_calls_baseline.txt
timeout 600 python tools/boundary_probe.py 6144 baseline keepalive_1000_0 keepalive_200_200 > gpurun_out/c_boundary.json 2> gpurun_out/c_boundary.err; echo "boundary rc=$?"; cat gpurun_out/c_boundary.err | cut -c1-700
tools/ubench/pcie_ceiling > gpurun_out/c_pcie.json 2>&1; cat gpurun_out/c_pcie.json
timeout 600 python tools/ingest_sweep.py > gpurun_out/c_ingest_sweep.json 2> gpurun_out/c_ingest_sweep.err; echo "ingest rc=$?"; tail -5 gpurun_out/c_ingest_sweep.err | cut -c1-300
