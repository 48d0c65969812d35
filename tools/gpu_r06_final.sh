#!/bin/bash
# round 6: full GPU test suite, the bench line as the driver runs it and with the defaults, rocprofv3 kernel stats + PMC traffic of the bench's own launches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_pytest.log )
tail -4 gpurun_out/r6_pytest.log
timeout 900 bash tools/gpu_prof_bench.sh r06 > gpurun_out/r6_prof.log 2>&1; echo "prof rc=$?"; tail -5 gpurun_out/r6_prof.log | cut -c1-300
cp gpurun_out/profb_r06/r06_pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null      # the line below quotes this round's own traffic figures
timeout 900 bash tools/gpu_prof_bench16.sh r06 > gpurun_out/r6_prof16.log 2>&1; echo "prof16 rc=$?"
cp gpurun_out/profb16_r06/r06_pmc_traffic16.json profiles/r06_pmc_traffic16.json 2>/dev/null
timeout 1500 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; echo "bench rc=$?"; tail -n 1 gpurun_out/r6_bench.json | head -c 1500; echo
cp bench_detail.json gpurun_out/r6_bench_detail.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver.json 2> gpurun_out/r6_bench_driver.err; echo "driver-style bench rc=$?"; wc -c gpurun_out/r6_bench_driver.json
timeout 600 python tools/stress_linear.py > gpurun_out/r6_stress_linear.txt 2>&1; tail -2 gpurun_out/r6_stress_linear.txt
bash tools/gpu_r06_pmc.sh > /dev/null 2>&1
