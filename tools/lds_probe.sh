#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "AMTGPU_LPPAD=8" "AMTGPU_DBG_WOFF=1"; do
  rm -rf /tmp/lp; env $cfg AMTGPU_PXT=2 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/lp -- python $REPO/tools/prof_run.py --what analyze --frames 512 --iters 1 > /tmp/lp.log 2>&1
  python - "$cfg" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/lp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "logo_corr" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[1], {k: int(v) for k, v in agg.items()})
PY
  grep logo_corr /tmp/lp.log; tail -2 /tmp/lp.log
done
rocprofv3 -L 2>/dev/null | grep -i "LDS" | head -20
