"""Timing harness (GPU box): bench.py's `train_step` leg alone (BASELINE.json configs[4] renderer part: minecraft, 3 frames x
2880 rays, perturb, train-mode BatchNorm, forward + backward + Adam), so that a kernel trace holds nothing else:

    python tools/perf/perf_train_leg.py [steps] [warmup]
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/perf/perf_train_leg.py 5
    python tools/trace_timeline.py <dir>/*/*_kernel_trace.csv --all"""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib  # noqa: E402


def main():
    # PR_PERF_LIB: a measurement build of the library (make EXTRA=-D... OUT=...), for A/B timings on one box
    if os.environ.get("PR_PERF_LIB"):
        _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
    args = types.SimpleNamespace(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20,
                                 warmup=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = bench.train_step_leg(args, dev, 1, 0, None, _lib.load(), precision=os.environ.get("PR_PERF_PRECISION", "fp32"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
