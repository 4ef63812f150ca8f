"""Timing probe (GPU box): the evaluation kernel's rate against the number of 64-sample tiles per resident workgroup (512 on an
MI355X: 256 CUs x 2) - one object, every sample in the box, so tiles = rays / 2.  Shows what the last, partly filled round of
tiles costs on workloads of a few rounds (the training step's: 5.1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import ObjectComposer, _lib, configs, synthetic  # noqa: E402
from tests.helpers import composer_inputs  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    cfg = configs.tennis_single_player_config()
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    comp.eval().to(dev)
    comp.gate_feature_head = False
    full = [v.to(dev) for v in composer_inputs(cfg, synthetic.single_player_scene(image_size=(128, 128)))]
    flop = bench.flops_per_sample(cfg["model"]["object_models"][0])
    # "thrash": a 512 MB elementwise pass between the launches evicts the packed weights (2.9 MB) from every XCD's L2 - what the
    # other objects' weights and feature rows do to an object's weights in a multi-object frame
    thrash = torch.zeros(128 << 20, dtype=torch.float32, device=dev) if "thrash" in sys.argv else None
    for rounds in (1.0, 1.1, 2.0, 2.1, 4.0, 4.5, 5.0, 5.1, 5.5, 6.0, 8.0, 8.1, 16.0, 16.1):
        tiles = int(round(rounds * 512))
        rays = tiles * 2
        inputs = list(full)
        inputs[1] = full[1][..., :rays, :].contiguous()
        with torch.no_grad():
            for _ in range(3):
                comp(*inputs, False)
            torch.cuda.synchronize()
            lib.pr_profile_enable(1)
            n = 10
            for _ in range(n):
                if thrash is not None:
                    thrash.add_(1.0)
                comp(*inputs, False)
            torch.cuda.synchronize()
            lib.pr_profile_enable(0)
        ms, cnt = bench.profile_arrays()
        lib.pr_profile_collect(ms, cnt)
        t = ms[0] / n
        print(f"rounds {rounds:5.1f} tiles {tiles:5d} mlp {t:7.3f} ms  {rays * 32 * flop / (t * 1e-3) / 1e12:6.1f} TFLOP/s  {t / rounds * 1e3:6.1f} us per round")


if __name__ == "__main__":
    main()
