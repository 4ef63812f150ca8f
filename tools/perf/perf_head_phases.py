"""Phase timing of k_head_bwd_group (GPU box, measurement build -DPR_HEAD_TIMING only):
    tools/build_variant.sh headtime -DPR_HEAD_TIMING && PR_PERF_LIB=build/variants/libplayrender_headtime.so python tools/perf/perf_head_phases.py"""
import ctypes as C
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib  # noqa: E402

_lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
lib = _lib.load()
args = types.SimpleNamespace(steps=10, warmup=3)
dev = torch.device("cuda", 0)
out = (C.c_ulonglong * 16)()
raw = C.CDLL(_lib.library_path())
bench.train_step_leg(args, dev, 1, 0, None, lib, precision=os.environ.get("PR_PERF_PRECISION", "fp32"))
raw.pr_debug_head_phases(out, 1)
bench.train_step_leg(args, dev, 1, 0, None, lib, precision=os.environ.get("PR_PERF_PRECISION", "fp32"))
raw.pr_debug_head_phases(out, 0)
names = ["records + barrier", "operand load + barrier", "prefetch A issue", "product", "claim + prefetch B + barrier", "epilogue",
         "barrier", "row write-out + barrier"]
for phase in (0, 1):
    total = sum(out[phase * 8 + i] for i in range(8))
    print(f"head backward phase {phase + 1}: {total / 1e6:.1f} Mticks (thread 0 of every workgroup)")
    for i, n in enumerate(names):
        print(f"  {n:32s} {out[phase * 8 + i] / 1e6:10.1f} Mticks {100.0 * out[phase * 8 + i] / max(total, 1):5.1f} %")
