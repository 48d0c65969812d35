#!/bin/bash
# round 5: PMC counter sums of the evaluation kernels and the frame metrics at HEAD (tools/gpu_prof.sh, 4096 frames each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
R=$PWD
( cd $R && bash tools/gpu_prof.sh r05lin analyze 4096 "--mode linear" > gpurun_out/pmc5_lin.log 2>&1 )
( cd $R && bash tools/gpu_prof.sh r05scan scan 4096 "--logos 3" > gpurun_out/pmc5_scan.log 2>&1 )
( cd $R && bash tools/gpu_prof.sh r05stats stats 4096 > gpurun_out/pmc5_stats.log 2>&1 )
ls gpurun_out/prof_r05*/summary.txt
