"""Manual measurement (not collected by pytest): how much of a C5-shaped training step is host time.

    python tools/perf/perf_train_host.py            enqueue time per step (no synchronisation in the loop) vs synchronised time
    python tools/perf/perf_train_host.py isolate    host time of the forward call, the backward call and the optimiser step with
                                               the device idle at the start of each (pure enqueue cost), and their device time
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from playableenvironments_amd import configs, synthetic  # noqa: E402
from playableenvironments_amd.object_composer import ObjectComposer  # noqa: E402
from tests.helpers import composer_inputs  # noqa: E402
from tests.perf_train_step import patch_pixels  # noqa: E402


def main():
    isolate = len(sys.argv) > 1 and sys.argv[1] == "isolate"
    cfg = configs.minecraft_config()
    scene = synthetic.minecraft_scene(batch=3, seed=5)
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=20000, alpha_bias=1.0, bender_scale=1e4)
    comp = comp.cuda().train()
    comp.batchnorm_check = "deferred"
    o, d, n, w2o, sty, dfm, ins = [v.cuda() for v in composer_inputs(cfg, scene, pixels=patch_pixels())]
    for t in (w2o, sty, dfm):
        t.requires_grad_(True)
    opt = torch.optim.Adam(comp.parameters(), lr=1e-5, fused=True)
    host = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0}
    device = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host[name] += t1 - t0
        device[name] += t2 - t0
        return out

    def step():
        opt.zero_grad(set_to_none=True)
        if not isolate:
            out = comp(o, d, n, w2o, sty, dfm, ins, True)
            out["coarse"]["global"]["integrated_features"].square().mean().backward()
            opt.step()
            return
        out = timed("forward", lambda: comp(o, d, n, w2o, sty, dfm, ins, True))
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        timed("backward", loss.backward)
        timed("optimizer", opt.step)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for k in host:
        host[k] = device[k] = 0.0
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    enqueue = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / steps
    if isolate:
        print("host enqueue ms / call-to-completion ms: " +
              ", ".join(f"{k} {host[k] / steps * 1e3:.2f} / {device[k] / steps * 1e3:.2f}" for k in host))
    else:
        print(f"enqueue {enqueue * 1e3:.2f} ms/step, synchronised {total * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
