#!/bin/bash
# quick timing of single components on the GPU box: bash tools/gpu_quick.sh <frames> ; prints per-kernel avg ms
FR=${1:-4096}
for what in "analyze --mode linear" "analyze --mode exact" "scan --logos 3" "stats"; do
  echo "== $what"; python tools/prof_run.py --what $what --frames $FR --iters 5 2>&1 | grep -v amdgpu.ids
done
