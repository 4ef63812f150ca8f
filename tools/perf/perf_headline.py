"""Timing harness (GPU box): the headline workload (bench.py: tennis renderer, 256 x 256 frame, 64 + 128 hierarchical samples) at one
precision - wall ms per frame and the MLP launches' HIP-event time - for A/B runs of measurement builds:

    [PR_PERF_LIB=build/variants/libplayrender_<name>.so] python tools/perf/perf_headline.py [fp32|f16x3|f16] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def main():
    if os.environ.get("PR_PERF_LIB"):
        _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
    precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    model.eval().to(dev)
    model.object_composer.precision = precision
    size = (256, 256)
    scene = bench.to_device(synthetic.tennis_scene(seed=1234, image_size=size), dev)
    lib = _lib.load()

    def step():
        with torch.no_grad():
            return model(*bench.scene_args(scene, size), 0, False, mode="scene_encodings")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib.pr_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(frames):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    lib.pr_profile_enable(0)
    ms, launches = bench.profile_arrays()
    lib.pr_profile_collect(ms, launches)
    print(f"{os.environ.get('PR_PERF_LIB', 'shipped library')} {precision}: {dt * 1e3:.2f} ms/frame, mlp {ms[0] / frames:.2f} ms, "
          f"composite {ms[1] / frames:.2f} ms")


if __name__ == "__main__":
    main()
