# ablations of the exact fused kernel (instrumented build -DAMT_EXPERIMENT: AMTGPU_DBG bit 0 = no ordered sum, 1 = no staging, 2 = no fade loop)
for d in 0 1 2 4 6 7; do echo "== AMTGPU_DBG=$d"; AMTGPU_LIB=amatsukaze_amd/libamt_gpu_exp.so AMTGPU_DBG=$d python tools/prof_run.py --what analyze --mode exact --frames ${1:-10000} --iters 3 2>&1 | grep logo_eval; done
