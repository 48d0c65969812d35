#!/bin/bash
# round 5, third GPU call: why frame_stats sits at 0.70 of HBM peak at 1440 wide (pitch 1472) and 0.75+ at 1920: the same frames at other pitches,
# with the HBM traffic of each (rocprofv3 --pmc FETCH_SIZE, its own pass)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
R=$(pwd)
export AMT_STATS_PITCHES=1
python tools/stats_bench.py --child > gpurun_out/r5c_pitches.json 2> gpurun_out/r5c_pitches.err; cat gpurun_out/r5c_pitches.json
cd /tmp && export TMPDIR=/tmp
AMT_STATS_FRAMES=2048 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r5c_pmc -- python $R/tools/stats_bench.py --child > $R/gpurun_out/r5c_pmc.log 2>&1
find $R/gpurun_out/r5c_pmc -name "*.db" -delete
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$R/gpurun_out/r5c_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frame_stats" in r.get("Kernel_Name", ""):
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
# 9 launches per pitch (1 + 8), pitches in order 1472, 1536, 1440, 1600; 2048 frames each
for i, p in enumerate((1472, 1536, 1440, 1600)):
    v = [x for _, x in rows[9 * i:9 * i + 9]]
    if v:
        b = 2 * sum(v) / len(v) * 1024 / 2048
        print(f"pitch {p}: fetch {b:.0f} B per frame = {b / (1440 * 1080):.4f} x algorithmic ({len(v)} launches)")
PY
