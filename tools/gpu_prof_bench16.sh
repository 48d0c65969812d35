#!/bin/bash
# rocprofv3 over the 16-bit kernels (BASELINE configs[4]'s format: 1920x1080 YUV420P10 in 16-bit containers), through the bench's own e2e10
# workload on 8 192 frames (two 4 096-frame chunks + the 32-frame warm-up pass):
#   1. --kernel-trace --stats            -> profiles/<tag>_bench16_rocprofv3_kernel_stats.csv
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, never combined with tracing) -> profiles/<tag>_pmc_traffic16.json
# Run on the GPU box from the repo root:  bash tools/gpu_prof_bench16.sh r05
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profb16_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload e2e10 --e2e-frames 8192 --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- $CMD > $OUT/pmc_$ctr.log 2>&1
done
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, json, collections
out = "$OUT"; tag = "$TAG"
rows = []
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rd = list(csv.reader(open(f)))
    rows = [rd[0]] + [r for r in rd[1:] if "amt::" in r[0]]
with open(out + f"/{tag}_bench16_rocprofv3_kernel_stats.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["# bench.py --workload e2e10 --e2e-frames 8192 (1920x1080 10-bit, two 4096-frame chunks + a 32-frame warm-up pass), analysis-mode linear"])
    for r in rows: w.writerow(r)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "?")
            if "amt::" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
# frames every kernel saw over the run: chunks of 4096 (+ 8-frame analysis halos inside the clip) and the 32-frame warm-up (+ 8 after it)
FR_OWN, FR_AN = 8192 + 32, (4096 + 8) * 2 + 32 + 8
W, H = 1920, 1080
alg = {"frame_stats_kernel": W * H * 2, "logo_eval_linear_kernel16": 2 * 256 * 128 + 132, "logo_eval_pair_kernel": 3 * 2 * 256 * 128 + 24,
       "delogo_kernel": 2 * 2 * (256 * 128 + 2 * 128 * 64)}
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --workload e2e10 --e2e-frames 8192; bytes = (2 x FETCH_SIZE + "
               "WRITE_SIZE) KiB per MI355X_MICROARCH.md, summed over the run's launches and divided by the frames the kernel processed "
               f"({FR_OWN}; the analysis {FR_AN}: chunk + halo); algorithmic_bytes_per_frame: the rectangle (or the Y plane) once, 2 bytes per sample "
               "(delogo: read + write, every frame counted although frames with fade 0 are skipped)"}
for k, d in agg.items():
    nm = next((s for s in ("logo_eval_linear_kernel16", "logo_eval_pair_kernel", "logo_eval_fused_kernel", "frame_stats_kernel", "delogo_kernel",
                           "analysis_mark_kernel", "calc_fades_kernel", "rect_range_flag_kernel") if s in k), k.split("(")[0])
    fr = FR_AN if nm in ("logo_eval_linear_kernel16", "analysis_mark_kernel", "rect_range_flag_kernel", "logo_eval_fused_kernel") else FR_OWN
    fetch = 2 * d.get("FETCH_SIZE", 0) * 1024 / fr
    write = d.get("WRITE_SIZE", 0) * 1024 / fr
    e = {"hbm_bytes_per_frame": fetch + write, "fetch_bytes_per_frame": fetch, "write_bytes_per_frame": write,
         "launches_profiled": n[k].get("FETCH_SIZE", 0), "kernel": k.split("(")[0][:90]}
    if nm in alg:
        e["algorithmic_bytes_per_frame"] = alg[nm]; e["traffic_over_algorithmic"] = (fetch + write) / alg[nm]
    res[nm if nm not in res else nm + "#2"] = e
json.dump(res, open(out + f"/{tag}_pmc_traffic16.json", "w"), indent=1)
print(open(out + f"/{tag}_bench16_rocprofv3_kernel_stats.csv").read()[:3000])
print(json.dumps(res, indent=1)[:3500])
PY
find $OUT -name "*.csv" -size +2M -delete
