#!/bin/bash
# rocprofv3 recipe used for profiles/: kernel trace + stats, then PMC counters in separate passes (never
# combined with tracing).  Run on the GPU box from the repo root:  bash tools/gpu_prof.sh <tag> <what> <frames>
set -u
TAG=${1:-r01}; WHAT=${2:-analyze}; FRAMES=${3:-1024}; EXTRA=${4:-}   # EXTRA: more prof_run.py arguments, e.g. "--mode linear" or "--logos 3"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $REPO/tools/prof_run.py --what $WHAT --frames $FRAMES --iters 3 $EXTRA > $OUT/kt.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python $REPO/tools/prof_run.py --what $WHAT --frames $FRAMES --iters 1 $EXTRA > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.db" -delete
# condense: keep the stats and per-kernel counter sums only
python - <<PY
import csv, glob, os, collections
out = "$OUT"
rows = []
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rows.append(("== " + os.path.basename(f), open(f).read()))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?").split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for name, txt in rows:
        fo.write(name + "\n" + txt + "\n")
    fo.write("== PMC sums over all dispatches of the run (1 iteration)\n")
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"    {c} = {v:.0f}\n")
print(open(out + "/summary.txt").read()[:6000])
PY
find $OUT -name "*.csv" -size +2M -delete
