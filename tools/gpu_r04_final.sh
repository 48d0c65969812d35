#!/bin/bash
# round 4: full GPU test suite, the bench line, rocprofv3 kernel stats + PMC traffic of the bench's own launches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f_pytest.log )
tail -4 gpurun_out/f_pytest.log
timeout 900 python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; echo "bench rc=$?"; tail -c 400 gpurun_out/f_bench.err
timeout 900 bash tools/gpu_prof_bench.sh r04 > gpurun_out/f_prof.log 2>&1; echo "prof rc=$?"; tail -5 gpurun_out/f_prof.log | cut -c1-300
