#!/bin/bash
# FETCH_SIZE calibration for the logo kernels' request shapes (tools/ubench/fetch_calib.hip): runs the microbenchmark under
# rocprofv3 --pmc FETCH_SIZE and divides the known byte counts by what the counter reports.
#   bash tools/gpu_fetch_calib.sh r03      -> gpurun_out/fetch_calib_r03/r03_fetch_calibration.json  (copy into profiles/)
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fetch_calib_$TAG
mkdir -p $OUT
[ -x $REPO/tools/ubench/fetch_calib ] || hipcc --offload-arch=gfx950 -O2 $REPO/tools/ubench/fetch_calib.hip -o $REPO/tools/ubench/fetch_calib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- $REPO/tools/ubench/fetch_calib > $OUT/known.json 2> $OUT/pmc.log
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, json
out = "$OUT"; tag = "$TAG"
known = json.load(open(out + "/known.json"))
rows = []
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
rect = [r for r in rows if "rect_kernel" in r["Kernel_Name"]]
res = {"note": "known bytes / (FETCH_SIZE KiB x 1024) per request shape, tools/ubench/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE; "
               "factor_requested prices the bytes the lanes asked for, factor_lines64/128 the distinct 64 B / 128 B lines they touch"}
for name, k in known.items():
    if name.startswith("stream_kernel"):
        m = [r for r in rows if name in r["Kernel_Name"]]
    else:
        m = [rect[k["launch_index"]]] if len(rect) > k["launch_index"] else []
    if not m: continue
    fetched = float(m[0]["Counter_Value"]) * 1024
    res[name] = dict(k, fetch_size_bytes=fetched, factor_requested=k["bytes_requested"] / fetched,
                     factor_lines64=k["bytes_lines64"] / fetched, factor_lines128=k["bytes_lines128"] / fetched)
json.dump(res, open(out + f"/{tag}_fetch_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
