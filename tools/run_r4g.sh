R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; cd $R
timeout 1500 python -m pytest tests/test_gpu.py -x -q -m gpu -k "backward_matches_oracle_autograd or backward_full_size or reference_gradient_fixtures or train_mode_batchnorm" 2>&1 | tail -8
for prec in fp32 f16x3 f16x3; do echo "== train precision=$prec"; PR_PERF_PRECISION=$prec python tools/perf/perf_train_leg.py 20 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'])"; done
