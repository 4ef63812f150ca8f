"""Debugging (GPU box): worst gradient errors of a split-precision training call against the oracle's autograd over a few scenes, for
A/B runs of the weight scale of the fp16-pair forward (-DPR_TRAIN_SPLIT_SCALE=k):
    [PR_PERF_LIB=build/variants/libplayrender_scale5.so] python tools/perf/dbg_split_scale.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402

if os.environ.get("PR_PERF_LIB"):
    _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
from tests.test_gpu import SMALL_NETS, _gradients  # noqa: E402

cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
for seed in (1234, 1, 2, 3, 4):
    for precision in ("fp32", "f16x3"):
        grads = _gradients(cfg, synthetic.minecraft_scene(seed=seed), 16, 3.0, False, precision=precision, min_divergence=0.0)
        worst = sorted(((float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30), k) for k, (a, b) in grads.items()), reverse=True)
        print(f"scene seed {seed} {precision}: worst relative gradient error {worst[0][0]:.2e} ({worst[0][1]}), tensors over 1e-4: "
              f"{sum(1 for v, _ in worst if v > 1e-4)}", flush=True)
