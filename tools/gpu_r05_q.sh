#!/bin/bash
# round 5: do the line-aligned dealings of frame_stats_kernel actually fetch less?  (TCC_EA0_RDREQ of default / deal1 / deal2, 1440x1080 8-bit)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export AMT_STATS_FRAMES=2048 AMT_STATS_PITCHES=1
for v in default deal1 deal2; do
  if [ $v != default ]; then export AMTGPU_LIB=$R/amatsukaze_amd/libamt_gpu_stats_$v.so; else unset AMTGPU_LIB; fi
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_READ_sum --output-format csv -d $R/gpurun_out/r5q_pmc/$v -- python $R/tools/stats_bench.py --child > $R/gpurun_out/r5q_$v.log 2>&1
done
find $R/gpurun_out/r5q_pmc -name "*.db" -delete
python - <<PY
import csv, glob, collections
for v in ("default", "deal1", "deal2"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"$R/gpurun_out/r5q_pmc/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "frame_stats" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, rows in sorted(agg.items()):
        rows.sort()
        per = [sum(x for _, x in rows[9 * i:9 * i + 9]) / max(1, len(rows[9 * i:9 * i + 9])) for i in range(4)]
        lines = 2048 * 1440 * 1080 / 128
        print(v, c, "pitches 1472/1536/1440/1600:", [f"{p / lines:.4f}" for p in per])
PY
