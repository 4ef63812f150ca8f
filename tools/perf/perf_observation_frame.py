"""Manual measurement (GPU box): host time of the evaluators' observation-driven frame (forward_from_observations, 288 x 512,
strides [4, 8], this package's encoders) - cProfile by cumulative and own time.
    python tools/perf/perf_observation_frame.py [tennis|minecraft]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402
from playableenvironments_amd.frame_graph import OBSERVATION_KEYS  # noqa: E402

world = sys.argv[1] if len(sys.argv) > 1 else "tennis"
dev = torch.device("cuda", 0)
cfg = (configs.tennis_config if world == "tennis" else configs.minecraft_config)(encoders=True)
torch.manual_seed(0)
model = EnvironmentModel(cfg)
model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
model.eval().to(dev)
size = (288, 512)
make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
batch = bench.to_device(synthetic.observation_batch(make(seed=1234, image_size=size)), dev)


def step():
    with torch.no_grad():
        return model.forward_from_observations(*[batch[k] for k in OBSERVATION_KEYS], 0, False, 1200, patch_stride=[4, 8])


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{world}: host issue {1e3 * (t1 - t0) / n:.3f} ms/frame, wall {1e3 * (t2 - t0) / n:.3f} ms/frame")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(30)
