R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; cd $R
PR_PERF_LIB=build/variants/libplayrender_trace.so python tools/perf/mlp_trace.py tennis 2>&1 | grep -v amdgpu | tail -4
for w in tennis minecraft; do for lib in "" build/variants/libplayrender_early.so; do echo "== $w lib=$lib"; PR_PERF_LIB=$lib python tools/perf/perf_native_frame.py $w fp32 2>&1 | grep -E "wall|mlp"; PR_PERF_LIB=$lib python tools/perf/perf_native_frame.py $w f16x3 2>&1 | grep -E "wall|mlp"; done; done
for lib in "" build/variants/libplayrender_early.so "" build/variants/libplayrender_early.so; do echo "== train lib=$lib"; PR_PERF_LIB=$lib python tools/perf/perf_train_leg.py 20 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['kernel_ms_per_step'])"; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-distinct-frames --no-minecraft --no-reference-graph --no-shard-balance --no-native-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], 'split', d['split_precision']['value'], d['split_precision']['mlp_ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu.py -x -q -m gpu -k "composer_matches or backward_matches or gated or train_mode or golden or native_evaluation or frame_graph or full_size or divergence" 2>&1 | tail -5
