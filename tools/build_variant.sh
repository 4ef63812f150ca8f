#!/bin/bash
# Measurement build of the library: tools/build_variant.sh <name> "<extra compiler flags>"  ->  build/variants/libplayrender_<name>.so
# (work-skipping switches exist at compile time only; run with PR_PERF_LIB=build/variants/libplayrender_<name>.so python tools/perf/perf_train_leg.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/variants"
make -s -C "$ROOT/playableenvironments_amd/csrc" EXTRA="$2" OBJDIR="$ROOT/build/obj_$1" OUT="$ROOT/build/variants/libplayrender_$1.so"
