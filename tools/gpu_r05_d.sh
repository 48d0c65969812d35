#!/bin/bash
# round 5, fourth GPU call: L2-side request counters of frame_stats_kernel (is the 1.2x "traffic" real 128-byte line fetches?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export AMT_STATS_FRAMES=2048
for v in default no_nt rows8bit_16; do
  for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    tag=$(echo $ctrs | tr ' ' '_')
    if [ $v != default ]; then export AMTGPU_LIB=$R/amatsukaze_amd/libamt_gpu_stats_$v.so; else unset AMTGPU_LIB; fi
    rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/r5d_pmc/$v/$tag -- python $R/tools/stats_bench.py --child > $R/gpurun_out/r5d_${v}_$tag.log 2>&1
  done
done
find $R/gpurun_out/r5d_pmc -name "*.db" -delete
python - <<PY
import csv, glob, collections
for v in ("default", "no_nt", "rows8bit_16"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"$R/gpurun_out/r5d_pmc/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "frame_stats" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, rows in sorted(agg.items()):
        rows.sort()
        # 9 launches per shape: 1440x1080 8-bit (pitch 1472), 1920x1080 8-bit, 1920x1080 10-bit; 2048 frames each?  (the child's default shapes use their own N)
        per = [sum(x for _, x in rows[9 * i:9 * i + 9]) / max(1, len(rows[9 * i:9 * i + 9])) for i in range(3)]
        print(v, c, [f"{p:.4g}" for p in per], len(rows))
PY
