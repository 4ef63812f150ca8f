"""Manual measurement (GPU box): a minecraft frame captured into a HIP graph (torch.cuda.CUDAGraph) and replayed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from playableenvironments_amd import configs, synthetic
from playableenvironments_amd.environment_model import EnvironmentModel

for precision in ("fp32", "f16x3"):
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.eval().cuda()
    model.object_composer.precision = precision
    size = (256, 256)
    keys = ("camera_rotations", "camera_translations", "focals", "object_rotation_parameters", "object_translation_parameters",
            "object_style", "object_deformation", "object_in_scene")
    scene = {k: v.cuda() for k, v in synthetic.minecraft_scene(seed=1234, image_size=size).items() if torch.is_tensor(v)}
    other = {k: v.cuda() for k, v in synthetic.minecraft_scene(seed=99, image_size=size).items() if torch.is_tensor(v)}
    static = {k: scene[k].clone() for k in keys}

    def render(s):
        with torch.no_grad():
            return model(s["camera_rotations"], s["camera_translations"], s["focals"], size, s["object_rotation_parameters"],
                         s["object_translation_parameters"], s["object_style"], s["object_deformation"], s["object_in_scene"],
                         0, False, mode="scene_encodings")["coarse"]["global"]["integrated_features"]

    for _ in range(3):
        render(static)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        render(static)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = render(static)
    for s in (scene, other):
        for k in keys:
            static[k].copy_(s[k])
        graph.replay()
        torch.cuda.synchronize()
        want = render(s)
        torch.cuda.synchronize()
        print(f"minecraft graph replay, {precision}: identical to the eager render: {torch.equal(out, want)}")
    n = 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"minecraft graph replay, {precision}: host {1e3 * (t1 - t0) / n:.3f} ms/frame, total {1e3 * (t2 - t0) / n:.2f} ms/frame "
          f"({n / (t2 - t0):.1f} frames/s)")
