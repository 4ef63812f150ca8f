#!/bin/bash
# split-precision training with the fp16-pair forward: gradient tests, step time, kernel stats
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "backward or train or gradient" 2>&1 | tail -4 > gpurun_out/r4/f16fwd_tests.log
for i in 1 2; do PR_PERF_PRECISION=f16x3 timeout 300 python tools/perf/perf_train_leg.py 20 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('f16x3 step', d['ms_per_step'], d['ms_per_step_median'])"; done | tee gpurun_out/r4/f16fwd_step.txt
cd /tmp; rm -rf /tmp/tr3
PR_PERF_PRECISION=f16x3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr3 -- python $ROOT/tools/perf/perf_train_leg.py 6 3 > /dev/null 2>&1
cd $ROOT
python tools/summarise_train_trace.py /tmp/tr3/*/*_kernel_trace.csv > gpurun_out/r4/f16fwd_trace_summary.json
head -30 gpurun_out/r4/f16fwd_trace_summary.json
cat gpurun_out/r4/f16fwd_tests.log
