# A/B of measurement builds on one box: tools/ab_variants.sh base v1 v2 ...   (run through gpurun; "base" = the product library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=build/variants/libplayrender_$v.so; fi
  PR_PERF_LIB=$lib python tools/perf/perf_train_leg.py 20 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernel_ms_per_step']
print('$v', d['ms_per_step'], k)" | tee -a gpurun_out/ab/ab.log
done
