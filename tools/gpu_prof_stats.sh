#!/bin/bash
# PMC counters of frame_stats_kernel, whole device and on a 40-CU partition: bash tools/gpu_prof_stats.sh <tag>
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in full mask40; do
  [ $mode = mask40 ] && export AMT_STATS_MASK_CUS=40 || unset AMT_STATS_MASK_CUS
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" \
             "FETCH_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/${mode}_pmc$i -- python $REPO/tools/prof_run.py --what stats --frames 4096 --iters 1 > $OUT/${mode}_pmc$i.log 2>&1
  done
done
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections
out = "$OUT"
for mode in ("full", "mask40"):
    agg = collections.defaultdict(float)
    for f in glob.glob(out + f"/{mode}_pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "frame_stats_kernel" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", mode)
    for c, v in sorted(agg.items()):
        print(f"    {c} = {v:.0f}")
PY
find $OUT -name "*.csv" -size +1M -delete
