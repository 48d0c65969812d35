#!/bin/bash
# quick A/B on the GPU box: parity tests, then per-kernel timings for a few engine settings
set -u
TAG=${1:-try}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
shift
for cfg in "$@"; do
  echo "== $cfg" >> $OUT/sweep.txt
  env $cfg timeout 300 python tools/prof_run.py --what all --frames 2048 --iters 3 2>&1 | grep -v amdgpu.ids >> $OUT/sweep.txt
done
cat $OUT/sweep.txt
