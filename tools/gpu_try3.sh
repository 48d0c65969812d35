#!/bin/bash
# scan pass with 3 candidate logos (the bench's shape), a few engine settings: bash tools/gpu_try3.sh <tag> "<env cfg>" ...
set -u
TAG=${1:-try}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
shift
for cfg in "$@"; do
  echo "== $cfg  [scan x3 logos, 4096 frames]" >> $OUT/sweep.txt
  env $cfg timeout 300 python tools/prof_run.py --what scan --logos 3 --frames 4096 --iters 3 2>&1 | grep -v amdgpu.ids >> $OUT/sweep.txt
done
cat $OUT/sweep.txt
