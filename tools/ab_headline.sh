#!/bin/bash
# A/B of measurement builds of the split / f16 kernels on the headline workload: tools/ab_headline.sh <variant> [<variant> ...]
mkdir -p gpurun_out/r4
{
for rep in 1 2; do
  for P in f16x3 f16; do
    python tools/perf/perf_headline.py $P 10
    for V in "$@"; do PR_PERF_LIB=build/variants/libplayrender_$V.so python tools/perf/perf_headline.py $P 10; done
  done
done
} 2>&1 | grep "ms/frame" | tee gpurun_out/r4/split_ab.txt
