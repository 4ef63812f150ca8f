"""Device-time split (GPU box) of bench.py's training iteration, eager launches: HIP events after the forward pass, the backward
pass and the optimiser step - the time the DEVICE spends between them (its idle gaps included), next to the host's enqueue times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.train().to(dev)
    comp = model.object_composer
    comp.batchnorm_check = os.environ.get("PR_BN_CHECK", "deferred")
    size = (288, 512)
    sc = bench.to_device(synthetic.minecraft_scene(batch=3, seed=77, image_size=size), dev)
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)
    params = list(comp.parameters())
    opt = torch.optim.Adam(params, lr=1e-5, fused=True)
    n = 40
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    host = [0.0, 0.0, 0.0]

    def step(i):
        e = ev[i] if i >= 0 else [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        e[0].record()
        opt.zero_grad(set_to_none=True)
        try:
            out = model(*bench.scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
        except ValueError:
            out = model(*bench.scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        e[1].record()
        t1 = time.perf_counter()
        loss.backward()
        e[2].record()
        t2 = time.perf_counter()
        opt.step()
        e[3].record()
        t3 = time.perf_counter()
        if i >= 0:
            host[0] += t1 - t0
            host[1] += t2 - t1
            host[2] += t3 - t2

    for _ in range(10):
        step(-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    seg = [0.0, 0.0, 0.0, 0.0]
    for i in range(n):
        for j in range(3):
            seg[j] += ev[i][j].elapsed_time(ev[i][j + 1])
        if i + 1 < n:
            seg[3] += ev[i][3].elapsed_time(ev[i + 1][0])
    print(f"step {total:.2f} ms; device: forward+loss {seg[0] / n:.2f}, backward {seg[1] / n:.2f}, optimiser {seg[2] / n:.2f}, "
          f"between steps {seg[3] / (n - 1):.2f} ms; host enqueue: forward {host[0] / n * 1e3:.2f}, backward {host[1] / n * 1e3:.2f}, "
          f"optimiser {host[2] / n * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
