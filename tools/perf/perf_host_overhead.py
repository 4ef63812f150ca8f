"""Manual measurement (GPU box): where does a minecraft eval frame spend its time - host (python + ctypes) or device?"""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from playableenvironments_amd import configs, synthetic
from playableenvironments_amd.environment_model import EnvironmentModel

cfg = configs.minecraft_config()
torch.manual_seed(0)
model = EnvironmentModel(cfg)
model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
model.eval().cuda()
size = (256, 256)
scene = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synthetic.minecraft_scene(seed=1234, image_size=size).items()}

def step():
    with torch.no_grad():
        return model(scene["camera_rotations"], scene["camera_translations"], scene["focals"], size, scene["object_rotation_parameters"],
                     scene["object_translation_parameters"], scene["object_style"], scene["object_deformation"], scene["object_in_scene"],
                     0, False, mode="scene_encodings")
for _ in range(3):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / n:.2f} ms/frame, total {1e3 * (t2 - t0) / n:.2f} ms/frame")
torch.cuda.set_sync_debug_mode("warn")      # any op that blocks the host on the device prints a warning
step()
torch.cuda.set_sync_debug_mode("default")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
