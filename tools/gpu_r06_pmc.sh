#!/bin/bash
# round 6: PMC counter sums of the evaluation kernels at HEAD (tools/gpu_prof.sh, 4096 frames each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
R=$PWD
( cd $R && bash tools/gpu_prof.sh r06lin analyze 4096 "--mode linear" > gpurun_out/pmc6_lin.log 2>&1 )
( cd $R && bash tools/gpu_prof.sh r06scan scan 4096 "--logos 3" > gpurun_out/pmc6_scan.log 2>&1 )
ls gpurun_out/prof_r06*/summary.txt
