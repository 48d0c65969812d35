for v in ${1:-release}; do
  if [ $v = release ]; then echo "== release (4 waves)"; python tools/prof_run.py --what analyze --mode linear --frames 10000 --iters 3 2>&1 | grep logo_eval
  else echo "== $v"; AMTGPU_LIB=amatsukaze_amd/libamt_gpu_$v.so python tools/prof_run.py --what analyze --mode linear --frames 10000 --iters 3 2>&1 | grep logo_eval; fi
done
