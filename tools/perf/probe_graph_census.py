"""Node census (pr_graph_node_census) of the recordings EnvironmentModel.frame_replay makes: the renderer-only frame, the
observation-driven frame with the package's own encoders, and a torch reduction that is known to record a memset node.
    python tools/perf/probe_graph_census.py          (prints one line per recording)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from playableenvironments_amd import configs, synthetic                      # noqa: E402
from playableenvironments_amd import environment_model as em                 # noqa: E402
from playableenvironments_amd import frame_graph                             # noqa: E402
from playableenvironments_amd.frame_graph import OBSERVATION_KEYS, SCENE_KEYS  # noqa: E402


def census_of(fn, tensors):
    try:
        call = frame_graph.CapturedCall(fn, tensors, warmup=1)
        return call.census
    except frame_graph.UnsafeRecording as error:
        return f"refused: {error}"


def main():
    print("runtime switch set:", frame_graph.graph_runtime_is_safe())
    x = torch.randn(1 << 22, device="cuda")
    print("elementwise      ", census_of(lambda t: t * 2 + 1, [x]))
    print("sum of 4M floats ", census_of(lambda t: t.sum(), [x]))
    print("mean over dim    ", census_of(lambda t: t.view(4096, 1024).mean(dim=0), [x]))
    print("zeros + add      ", census_of(lambda t: torch.zeros_like(t) + t, [x]))
    for world in ("tennis", "minecraft"):
        cfg = getattr(configs, world + "_config")(encoders=True)
        model = em.EnvironmentModel(cfg).cuda().eval()
        size = (288, 512)
        scene_fn = getattr(synthetic, world + "_scene")
        sc = {k: v.cuda() for k, v in scene_fn(seed=5, image_size=size).items() if torch.is_tensor(v)}
        batch = {k: v.cuda() for k, v in synthetic.observation_batch(scene_fn(batch=1, seed=3, image_size=size), boxes_seed=3).items()}
        model._in_replay = True
        with torch.no_grad():
            print(world, "scene_encodings ", census_of(
                lambda *ts: model.forward_from_scene_encoding(*ts[:3], size, *ts[3:], 0, False, patch_stride=[4, 8]),
                [sc[k] for k in SCENE_KEYS]))
            print(world, "observations    ", census_of(
                lambda *ts: model.forward_from_observations(*ts, 0, False, patch_stride=[4, 8]),
                [batch[k] for k in OBSERVATION_KEYS]))


if __name__ == "__main__":
    main()
