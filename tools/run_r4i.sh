#!/bin/bash
# round 4: the single-product fp16 tier - parity-at-tier test, headline workload timing beside f16x3, native frames
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "half_precision or composer_matches_oracle" 2>&1 | tail -5 > gpurun_out/r4/f16_tests.log
for P in f16x3 f16; do
  timeout 600 python bench.py --steps 5 --warmup 2 --precision $P --no-cpu-baseline --no-train-step --no-distinct-frames --no-reference-graph \
      --no-minecraft --no-shard-balance --no-native-frame > gpurun_out/r4/bench_$P.json 2> gpurun_out/r4/bench_$P.err
done
for W in tennis minecraft; do
  timeout 300 python tools/perf/perf_native_frame.py $W f16 > gpurun_out/r4/native_${W}_f16.txt 2>&1
done
cat gpurun_out/r4/f16_tests.log
python - <<'PY'
import json
for p in ("f16x3", "f16"):
    try:
        d = json.load(open(f"gpurun_out/r4/bench_{p}.json"))
        print(p, d["value"], d["ms_per_step"], d["roofline"]["mlp_ms_per_step"])
    except Exception as e:
        print(p, "failed", e)
PY
tail -4 gpurun_out/r4/native_tennis_f16.txt gpurun_out/r4/native_minecraft_f16.txt
