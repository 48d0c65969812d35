#!/bin/bash
# kernel + memcpy trace of the filter-layer bench (tests/cpp/filters_host_test --bench): which launches the layer makes and how long they take
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R && L=$(python tools/boundary_probe.py 256 | tail -1)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/btrace -- $R/tests/cpp/filters_host_test --bench 1440 1080 ${1:-4096} $L 0 > $R/gpurun_out/btrace.log 2>&1
find $R/gpurun_out/btrace -name "*.db" -delete
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/btrace/**/*kernel_stats.csv", recursive=True) + glob.glob("$R/gpurun_out/btrace/**/*memory_copy_stats.csv", recursive=True):
    print("==", f.split("/")[-1])
    for r in list(csv.reader(open(f)))[:12]: print([c[:60] for c in r[:6]])
PY
tail -1 $R/gpurun_out/btrace.log | cut -c1-400
find $R/gpurun_out/btrace -name "*.csv" -size +1M -delete
