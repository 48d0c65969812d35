#!/bin/bash
# round 6: the bench line again (defaults, and as the driver runs it) after a change that touches bench.py only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; echo "bench rc=$?"; tail -n 1 gpurun_out/r6_bench.json | head -c 600; echo
cp bench_detail.json gpurun_out/r6_bench_detail.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver.json 2> gpurun_out/r6_bench_driver.err; echo "driver-style bench rc=$?"; wc -c gpurun_out/r6_bench_driver.json
