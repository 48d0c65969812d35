#!/bin/bash
# round 6: the XCD-aware workgroup map of the tile kernels -- parity, A/B against the plain map, HBM traffic (FETCH_SIZE only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; R=$PWD
python tools/lin_check.py > gpurun_out/xcd_lincheck.log 2>&1
timeout 900 python -m pytest tests/test_gpu_eval_shapes.py tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x > gpurun_out/xcd_pytest.log 2>&1
bash tools/ab_lin.sh lin wg_plain_map > gpurun_out/xcd_ab_lin.log 2>&1
bash tools/ab_lin.sh scan wg_plain_map > gpurun_out/xcd_ab_scan.log 2>&1
cd /tmp && export TMPDIR=/tmp
for v in default wg_plain_map; do
  if [ "$v" != default ]; then export AMTGPU_LIB=$R/amatsukaze_amd/libamt_gpu_flags_$v.so; else unset AMTGPU_LIB; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/xcd_pmc_lin_$v -- python $R/tools/prof_run.py --what analyze --frames 4096 --iters 1 --mode linear > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/xcd_pmc_scan_$v -- python $R/tools/prof_run.py --what scan --frames 4096 --iters 1 --logos 3 > /dev/null 2>&1
done
find $R/gpurun_out -name "*.db" -delete
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/xcd_pmc_*")):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:50]; agg[k] += float(r["Counter_Value"]); n[k] += 1
    print(d.split("/")[-1])
    for k in agg: print("   ", k, n[k], "FETCH_SIZE", agg[k])
PY
tail -3 $R/gpurun_out/xcd_lincheck.log; tail -2 $R/gpurun_out/xcd_pytest.log; cat $R/gpurun_out/xcd_ab_lin.log $R/gpurun_out/xcd_ab_scan.log
