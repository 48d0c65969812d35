#!/bin/bash
# round 5, first GPU call: the GPU test suite, the bench line (must parse: compact, last on stdout), the decisions' Amdahl term on the box's cores
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a_pytest.log )
tail -5 gpurun_out/r5a_pytest.log
timeout 1200 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r5a_bench.err | head -c 600; echo
wc -c gpurun_out/r5a_bench.json; tail -n 1 gpurun_out/r5a_bench.json | head -c 3000; echo
python tools/decisions_bench.py 431568 7 > gpurun_out/r5a_decisions.json 2>&1; cat gpurun_out/r5a_decisions.json
