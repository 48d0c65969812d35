#!/bin/bash
# per-pass timings for a few engine settings: bash tools/gpu_try2.sh <tag> "<env cfg>" ...
set -u
TAG=${1:-try}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
shift
for cfg in "$@"; do
  for what in analyze scan; do
    echo "== $cfg  [$what]" >> $OUT/sweep.txt
    env $cfg timeout 300 python tools/prof_run.py --what $what --frames 2048 --iters 3 2>&1 | grep -v amdgpu.ids >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
