#!/bin/bash
# Variant sweep of the correlation kernel on the GPU box: bash tools/gpu_sweep.sh <tag>
set -u
TAG=${1:-sweep}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for v in "1 256" "1 512" "1 1024" "2 256" "2 512" "4 256"; do
  set -- $v
  echo "== PXT=$1 NT=$2" >> $OUT/sweep.txt
  AMTGPU_PXT=$1 AMTGPU_NT=$2 timeout 300 python tools/prof_run.py --what all --frames 2048 --iters 3 >> $OUT/sweep.txt 2>&1
done
cat $OUT/sweep.txt
