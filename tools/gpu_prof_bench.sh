#!/bin/bash
# rocprofv3 over the bench's own launches (10 000-frame batch, same launch geometry as the timed run):
#   1. --kernel-trace --stats            -> profiles/<tag>_bench_rocprofv3_kernel_stats.csv (rows of this repo's kernels)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, never combined with tracing) for both analysis modes
#      -> profiles/<tag>_pmc_traffic.json: HBM bytes per frame per kernel (FETCH_SIZE doubled per MI355X_MICROARCH.md)
# Run on the GPU box from the repo root:  bash tools/gpu_prof_bench.sh r03
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-strong --no-ingest --cpu-frames 0 --no-verify --no-alt-mode --no-configs --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $REPO/bench.py --steps 20 --warmup 2 $COMMON > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_exact -- python $REPO/bench.py --steps 10 --warmup 2 --analysis-mode exact $COMMON > $OUT/kt_exact.log 2>&1
for mode in linear exact; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${mode}_$ctr -- python -X faulthandler $REPO/bench.py --steps 2 --warmup 0 --analysis-mode $mode $COMMON > $OUT/pmc_${mode}_$ctr.log 2>&1
  done
done
find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, json, os, collections
out = "$OUT"; tag = "$TAG"
ours = ("amt::",)
def stats(d):
    rows = []
    for f in glob.glob(out + "/" + d + "/**/*kernel_stats.csv", recursive=True):
        rd = list(csv.reader(open(f)))
        rows = [rd[0]] + [r for r in rd[1:] if any(o in r[0] for o in ours)]
    return rows
with open(out + f"/{tag}_bench_rocprofv3_kernel_stats.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    for d, label in (("kt", "analysis-mode linear (headline)"), ("kt_exact", "analysis-mode exact")):
        w.writerow(["# " + label])
        for r in stats(d): w.writerow(r)
def pmc(mode):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(out + f"/pmc_{mode}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "?")
                if "amt::" not in k: continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return agg, n
N = 10000
def name_of(k, mode):
    if "logo_eval_linear_kernel" in k: return "logo_eval_linear_kernel.analysis"
    if "logo_eval_pair_kernel" in k: return "logo_eval_pair_kernel.scan"
    if "logo_eval_fused_kernel" in k:
        if ", 2>" in k: return "logo_eval_fused_kernel.scan"
        return "logo_eval_fused_kernel.analysis" if mode == "exact" else "logo_eval_fused_kernel.analysis_refine"
    for s in ("frame_stats_kernel", "delogo_kernel", "analysis_mark_kernel"):
        if s in k: return s
    return k.split("(")[0]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 (10 000-frame launches, the bench's own "
               "launch geometry); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced "
               "reads on gfx950), divided by launches and by 10 000 frames", "frames_per_launch": N}
for mode in ("linear", "exact"):
    agg, n = pmc(mode)
    for k, d in agg.items():
        nm = name_of(k, mode)
        if nm in res and mode == "exact" and nm != "logo_eval_fused_kernel.analysis": continue
        launches = max(1, n[k].get("FETCH_SIZE", 1))
        fetch = 2 * d.get("FETCH_SIZE", 0) * 1024 / launches / N
        write = d.get("WRITE_SIZE", 0) * 1024 / max(1, n[k].get("WRITE_SIZE", 1)) / N
        res[nm] = {"hbm_bytes_per_frame": fetch + write, "fetch_bytes_per_frame": fetch, "write_bytes_per_frame": write, "launches_profiled": launches,
                   "kernel": k.split("(")[0][:80]}
json.dump(res, open(out + f"/{tag}_pmc_traffic.json", "w"), indent=1)
print(open(out + f"/{tag}_bench_rocprofv3_kernel_stats.csv").read()[:3000])
print(json.dumps(res, indent=1)[:3000])
PY
find $OUT -name "*.csv" -size +2M -delete
