#!/bin/bash
# round 4, GPU call B: new upload / wide-logo tests, frame_stats variants, call trace of the filter layer, ingest sweep with the spinning pool
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_upload.py tests/test_gpu_eval_shapes.py tests/test_gpu_stats.py tests/test_gpu_filters_cpp.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest.log )
tail -4 gpurun_out/b_pytest.log
timeout 900 python tools/stats_bench.py > gpurun_out/b_stats_bench.json 2> gpurun_out/b_stats_bench.err; echo "stats rc=$?"; cat gpurun_out/b_stats_bench.err | tail -12
timeout 300 python tools/boundary_calls.py 2048 > gpurun_out/b_calls_baseline.txt 2>&1; echo "calls rc=$?"
timeout 300 python tools/boundary_calls.py 2048 AMT_KEEPALIVE=1000,0 > gpurun_out/b_calls_keepalive.txt 2>&1; echo "calls rc=$?"
head -60 gpurun_out/b_calls_baseline.txt
timeout 600 python tools/ingest_sweep.py > gpurun_out/b_ingest_sweep.json 2> gpurun_out/b_ingest_sweep.err; echo "ingest rc=$?"; tail -5 gpurun_out/b_ingest_sweep.err
