#!/bin/bash
# round 5, second GPU call: GPU tests again (sharded bench tests read the detail now), frame_stats variants, rocprofv3 + PMC of the 16-bit kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b_pytest.log )
tail -5 gpurun_out/r5b_pytest.log
timeout 900 python tools/stats_bench.py deal1 deal1_no_nt deal1_run64 deal1_rows16 run64 no_nt > gpurun_out/r5b_stats.json 2> gpurun_out/r5b_stats.err; cut -c1-330 gpurun_out/r5b_stats.err
timeout 900 bash tools/gpu_prof_bench16.sh r05 > gpurun_out/r5b_prof16.log 2>&1; echo "prof16 rc=$?"; tail -60 gpurun_out/r5b_prof16.log | cut -c1-250
