R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; cd $R
timeout 1500 python -m pytest tests/test_gpu.py -x -q -m gpu -k "fused_scene_setup or native_evaluation or frame_graph or environment_model or two_cameras or render_sharded or observation" 2>&1 | tail -8
for a in "tennis fp32" "tennis f16x3" "minecraft fp32" "minecraft f16x3"; do python tools/perf/perf_native_frame.py $a 2>&1 | grep -E "wall|one frame"; done
