"""Phase timing of k_chain_bwd_group (GPU box, measurement build -DPR_CHAIN_TIMING only):
    tools/build_variant.sh chaintime -DPR_CHAIN_TIMING && PR_PERF_LIB=build/variants/libplayrender_chaintime.so python tools/perf/perf_chain_phases.py"""
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
raw.pr_debug_chain_phases(out, 1)
bench.train_step_leg(args, dev, 1, 0, None, lib, precision=os.environ.get("PR_PERF_PRECISION", "fp32"))
raw.pr_debug_chain_phases(out, 0)
names = ["entry load", "K loops", "wait after K loop", "masked store", "wait after store", "gradient write-out", "mask fetch",
         "input products + global store", "tile end barrier"]
total = sum(out[i] for i in range(9))
for i, n in enumerate(names):
    print(f"{n:32s} {out[i] / 1e6:10.1f} Mticks {100.0 * out[i] / total:5.1f} %")
