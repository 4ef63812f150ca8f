"""Manual measurement (not collected by pytest): the reference's PyTorch op graph - as restated by the
oracle: materialised per-sample tensors, boolean-mask compaction, sort + gather compose, 1000-ray
chunks - executed by PyTorch-ROCm on the MI355X, on the benchmark workload of bench.py.  It answers
"how fast is the reference's own PyTorch path on this GPU", the yardstick of north_star's >= 10x target.

    python tools/perf/perf_reference_graph_on_gpu.py [rays_side]
"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import render_oracle as ro  # noqa: E402
from playableenvironments_amd import ObjectComposer, configs, synthetic  # noqa: E402
from tests.helpers import composer_inputs, grid_pixels  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    scene = synthetic.tennis_scene(seed=1234, image_size=(256, 256))
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, side))]
    sd = {k: v.detach().cuda() for k, v in comp.state_dict().items()}
    for chunk in (1000, 4000):
        with torch.no_grad():
            ro.batchified_composer_call(cfg, sd, *[v[..., :2000, :] if v.dim() == 5 and v.size(-2) > 2000 else v for v in inputs],
                                        False, chunk=chunk)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ro.batchified_composer_call(cfg, sd, *inputs, False, chunk=chunk)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"PyTorch-ROCm op graph, {side}x{side} = {side * side} rays, chunk {chunk}: {dt:.3f} s -> "
              f"{side * side / dt / 1e6:.4f} Mrays/s ({side * side / dt / 65536:.3f} frames/s at 256x256), "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


def train_step():
    """The C5-shaped training step of tools/perf/perf_train_step.py through torch.autograd on the GPU."""
    from tests.perf_train_step import patch_pixels
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=20000, alpha_bias=1.0, bender_scale=1e4)
    scene = synthetic.minecraft_scene(batch=3, seed=5)
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=patch_pixels())]
    o, d, n, w2o, sty, dfm, ins = inputs
    sd = {k: v.detach().cuda().clone() for k, v in comp.state_dict().items()}
    names = [k for k, _ in comp.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    for t in (w2o, sty, dfm):
        t.requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in names], lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        out = ro.composer_forward(cfg, sd, o, d, n, w2o, sty, dfm, ins, True, training=True)
        out["coarse"]["global"]["integrated_features"].square().mean().backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rays = d.numel() // 3
    print(f"PyTorch-ROCm autograd training step (minecraft, {rays} rays): {dt * 1e3:.1f} ms -> {rays / dt / 1e6:.4f} Mrays/s "
          f"trained, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        train_step()
    else:
        main()
