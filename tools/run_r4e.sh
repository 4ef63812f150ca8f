R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; cd $R
for lib in "" build/variants/libplayrender_twoslab.so "" build/variants/libplayrender_twoslab.so; do echo "== train f16x3 lib=$lib"; PR_PERF_LIB=$lib PR_PERF_PRECISION=f16x3 python tools/perf/perf_train_leg.py 20 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'])"; done
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "backward_matches_oracle_autograd or reference_gradient_fixtures" 2>&1 | tail -3
