#!/bin/bash
# round 4: timeline of scan || slim frame metrics (do the two kernels overlap at all?)  usage: gpu_r04_k.sh <variant> ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf gpurun_out/k_trace_$v
  AMTGPU_LIB=$PWD/amatsukaze_amd/libamt_gpu_cosched_$v.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k_trace_$v -o run -- python tools/coschedule_probe.py --child > gpurun_out/k_child_$v.json 2> gpurun_out/k_child_$v.err
  echo "$v rc=$?"
done
