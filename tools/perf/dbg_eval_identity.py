"""Identity probe (GPU box): SHA-256 of every result field of evaluation renders at the split-precision tiers - the hierarchical tennis
frame (128 x 128, 64 + 128 samples) and the shipped minecraft frame (96 x 128) at precision f16x3 and f16 - so that two builds of the
library can be compared bit for bit:
    python tools/perf/dbg_eval_identity.py > new.txt;  PR_PERF_LIB=build/variants/libplayrender_<name>.so python tools/perf/dbg_eval_identity.py > old.txt;  diff new.txt old.txt"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def flat(d, prefix=""):
    for k in sorted(d):
        v = d[k]
        if isinstance(v, dict):
            yield from flat(v, prefix + k + ".")
        elif torch.is_tensor(v):
            yield prefix + k, v
        elif isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if torch.is_tensor(t):
                    yield f"{prefix}{k}[{i}]", t


def main():
    if os.environ.get("PR_PERF_LIB"):
        _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
    dev = torch.device("cuda", 0)
    for world, cfg, scene_fn, size in (("tennis 64+128", configs.tennis_config(hierarchical=(64, 128)), synthetic.tennis_scene, (128, 128)),
                                       ("minecraft", configs.minecraft_config(), synthetic.minecraft_scene, (96, 128))):
        torch.manual_seed(0)
        model = EnvironmentModel(cfg)
        model.frame_replay = None
        synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=0.5, bender_scale=1e4)
        model.eval().to(dev)
        scene = bench.to_device(scene_fn(seed=1234, image_size=size), dev)
        for precision in ("f16x3", "f16"):
            model.object_composer.precision = precision
            for gate in (True, False):
                model.object_composer.gate_feature_head = gate
                with torch.no_grad():
                    out = model(*bench.scene_args(scene, size), 0, False, mode="scene_encodings")
                torch.cuda.synchronize()
                for name, t in flat(out):
                    digest = hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
                    print(f"{world:14s} {precision:6s} gate={int(gate)} {name:60s} {digest}")


if __name__ == "__main__":
    main()
