R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4
cd $R
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "native_evaluation_frame or frame_graph or arena or recorded_training" > gpurun_out/r4/t1.log 2>&1
tail -15 gpurun_out/r4/t1.log
timeout 600 python bench.py --only native_eval_frame > gpurun_out/r4/native_leg.json 2> gpurun_out/r4/native_leg.err
tail -3 gpurun_out/r4/native_leg.err
