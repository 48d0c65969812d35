#!/bin/bash
# round 4, GPU call A: tests, bench line, boundary tick variants, PCIe ceiling + ingest sweep, e2e10 hash invariance
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log ) 
tail -3 gpurun_out/a_pytest.log
timeout 600 python bench.py --steps 40 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/a_bench.err
timeout 600 python tools/boundary_probe.py 6144 > gpurun_out/a_boundary.json 2> gpurun_out/a_boundary.err; echo "boundary rc=$?"
tools/ubench/pcie_ceiling > gpurun_out/a_pcie.json 2>&1; echo "pcie rc=$?"; cat gpurun_out/a_pcie.json
timeout 600 python tools/ingest_sweep.py > gpurun_out/a_ingest_sweep.json 2> gpurun_out/a_ingest_sweep.err; echo "ingest rc=$?"
# e2e10: hash must not depend on the chunk size or the number of ranks (shared-GPU dry run at 2 ranks)
timeout 600 python bench.py --workload e2e10 --e2e-frames 8192 --e2e-chunk 1024 > gpurun_out/a_e2e_c1024.json 2> gpurun_out/a_e2e.err; echo "e2e rc=$?"
timeout 600 python bench.py --workload e2e10 --e2e-frames 8192 --e2e-chunk 640 > gpurun_out/a_e2e_c640.json 2>> gpurun_out/a_e2e.err; echo "e2e rc=$?"
AMT_BENCH_SHARED_GPU=1 timeout 900 python bench.py --workload e2e10 --gpus 2 --e2e-frames 8192 > gpurun_out/a_e2e_n2.json 2>> gpurun_out/a_e2e.err; echo "e2e n2 rc=$?"
grep -ho '"decisions_sha256": "[0-9a-f]*"' gpurun_out/a_e2e_*.json
tail -c 1500 gpurun_out/a_e2e.err
