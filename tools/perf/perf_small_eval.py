"""Timing probe (GPU box): the EVALUATION kernel on the training step's workload (minecraft, 3 frames x 2880 rays) - separates
the cost of the small sample count from the cost of the training-only code of the forward pass."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.to(dev)
    size = (288, 512)
    sc = bench.to_device(synthetic.minecraft_scene(batch=3, seed=77, image_size=size), dev)
    comp = model.object_composer
    comp.gate_feature_head = False
    for mode in ("eval", "train_nograd"):
        model.eval() if mode == "eval" else model.train()

        def step():
            with torch.no_grad():
                return model(*bench.scene_args(sc, size), 2880, mode != "eval", 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.pr_profile_enable(1)
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        lib.pr_profile_enable(0)
        ms, cnt = bench.profile_arrays()
        lib.pr_profile_collect(ms, cnt)
        print(mode, f"step {dt * 1e3:.3f} ms, mlp {ms[0] / n:.3f} ms in {cnt[0] / n:.1f} launches, composite {ms[1] / n:.3f} ms")


if __name__ == "__main__":
    main()
