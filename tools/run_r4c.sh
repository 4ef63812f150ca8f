R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4; cd $R
timeout 1500 python -m pytest tests/test_gpu.py -x -q -m gpu -k "ranks_on_one_gpu or bench_self or bench_collectives or rccl_world1 or ray_chunks or sweep_slice or arena or recorded" 2>&1 | tail -15
