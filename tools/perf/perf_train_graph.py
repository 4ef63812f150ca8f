"""Timing harness (GPU box): bench.py's training-step leg as eager launches and as ONE recorded HIP graph
(frame_graph.GraphedStep): ms per step of both, and that replays draw fresh noise / fresh patches.

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/perf/perf_train_graph.py      (correct replays, no faster than eager)
    PR_ALLOW_UNSAFE_GRAPH=1 python tools/perf/perf_train_graph.py               (the runtime's default path: 8.4 ms per step, but the
                                                                           replays go wrong after a host synchronisation)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402
from playableenvironments_amd.frame_graph import GraphedStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.train().to(dev)
    comp = model.object_composer
    comp.batchnorm_check = "deferred"
    size = (288, 512)
    sc = bench.to_device(synthetic.minecraft_scene(batch=3, seed=77, image_size=size), dev)
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)
    params = list(comp.parameters())
    opt = torch.optim.Adam(params, lr=1e-5, fused=True, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        for attempt in range(20):
            try:       # (a patch that misses an object: BatchNorm without samples raises, like torch - the patch is re-drawn)
                out = model(*bench.scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
                break
            except ValueError:
                if attempt == 19:
                    raise
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    print(f"eager   : {eager * 1e3:.3f} ms / step; normalised samples of the last step {comp.last_normalised_samples['coarse'].tolist()}")

    if os.environ.get("PR_ALLOW_UNSAFE_GRAPH"):
        from playableenvironments_amd import frame_graph
        frame_graph.graph_runtime_is_safe = lambda: True
    graphed = GraphedStep(step, warmup=3)
    losses = []
    seeds = []
    for _ in range(3):
        losses.append(float(graphed.replay().detach()))
        seeds.append(int(comp.last_noise_seed.item()))
        print("normalised samples after a replay:", comp.last_normalised_samples["coarse"].tolist())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed.replay()
    torch.cuda.synchronize()
    g = (time.perf_counter() - t0) / steps
    print(f"graphed : {g * 1e3:.3f} ms / step   losses of three replays {losses}   seed words {seeds}")
    print("normalised samples of the last replay:", comp.last_normalised_samples["coarse"].tolist())


if __name__ == "__main__":
    main()
