#!/bin/bash
# A/B of the pair kernel's build variants on the GPU box (run from the repo root): scan of 3 logos, HIP-event kernel times.
#   usage: tools/exp_pair.sh FRAMES "lib:G lib:G ..."     (lib = suffix of amatsukaze_amd/libamt_gpu_<lib>.so, built with build_variant;
#   "release" = the shipped library, G ignored)
F=${1:-4096}
for v in ${2:-release}; do
  lib=${v%%:*}; g=${v##*:}
  if [ "$lib" = release ]; then echo "== release"; python tools/prof_run.py --what scan --frames $F --iters 3 --logos 3 2>&1 | grep logo_eval
  else echo "== $lib G=$g"; AMTGPU_LIB=amatsukaze_amd/libamt_gpu_$lib.so AMTGPU_G=$g python tools/prof_run.py --what scan --frames $F --iters 3 --logos 3 2>&1 | grep -E "logo_eval|Error|error" | head -3
  fi
done
