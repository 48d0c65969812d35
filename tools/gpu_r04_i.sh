#!/bin/bash
# round 4: e2e10 with the untimed warm-up and the faster replicated decisions; the tests that run it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python bench.py --workload e2e10 > gpurun_out/i_e2e.json 2> gpurun_out/i_e2e.err; echo "e2e rc=$?"
AMT_E2E_PHASES=1 timeout 600 python bench.py --workload e2e10 --no-verify > gpurun_out/i_e2e_phases.json 2> gpurun_out/i_e2e_phases.err; echo "phases rc=$?"
timeout 900 python -m pytest tests/test_gpu_e2e_1080p10.py tests/test_gpu_sharded.py tests/test_gpu_stats.py -x -q > gpurun_out/i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/i_pytest.log
