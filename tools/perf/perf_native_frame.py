"""Manual measurement (GPU box): the frame the reference's evaluators and play loop render - 288 x 512, strided grids [4, 8]
(72 x 128 + 36 x 64 = 11 520 rays; environment_model_backpropagated_autoencoder.py:173-236) - through the plain drop-in call
``EnvironmentModel.forward_from_scene_encoding(..., 0, False, patch_stride=[4, 8])``: host issue time, wall time of back-to-back
frames, device time between the frames' first launches, and where the Python time goes (cProfile, by own time).

    python tools/perf/perf_native_frame.py [tennis|minecraft] [fp32|f16x3] [profile]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib, configs, synthetic  # noqa: E402
from playableenvironments_amd.environment_model import EnvironmentModel  # noqa: E402


def build(world: str, dev, batch: int = 1):
    cfg = configs.tennis_config() if world == "tennis" else configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    model.frame_replay = None      # eager launches: what this script measures / what counter passes can instrument
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.eval().to(dev)
    size = (288, 512)
    make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
    scene = bench.to_device(make(batch=batch, seed=1234, image_size=size), dev)
    return cfg, model, scene, size


def main():
    if os.environ.get("PR_PERF_LIB"):      # a measurement build of the library (tools/build_variant.sh), for A/B timings on one box
        _lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
    world = sys.argv[1] if len(sys.argv) > 1 else "tennis"
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    dev = torch.device("cuda", 0)
    batch = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("batch=")), 1)
    cfg, model, scene, size = build(world, dev, batch)
    model.object_composer.precision = precision
    if "nogate" in sys.argv:
        model.object_composer.gate_feature_head = False

    def step():
        with torch.no_grad():
            return model.forward_from_scene_encoding(*bench.scene_args(scene, size), 0, False, 1200, patch_stride=[4, 8])

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 200
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    for i in range(n):
        marks[i].record()
        step()
    marks[n].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gaps = bench.event_gaps_ms(marks)
    print(f"{world} {precision}: host issue {1e3 * (t1 - t0) / n:.3f} ms/frame, wall {1e3 * (t2 - t0) / n:.3f} ms/frame, "
          f"device gap median {bench.median(gaps):.3f} ms")
    # device-only time: one frame at a time, the host waits in between (no queueing behind the previous frame)
    single = []
    for _ in range(20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        single.append(e0.elapsed_time(e1))
    print(f"  one frame, first launch to last (host-paced): median {bench.median(single):.3f} ms")
    lib = bench._lib_handle()
    lib.pr_profile_enable(1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    lib.pr_profile_enable(0)
    ms, launches = bench.profile_arrays()
    lib.pr_profile_collect(ms, launches)
    with torch.no_grad():
        inputs = bench.composer_call_inputs_strided(model, cfg, scene, size, [4, 8])
    roof = bench.leg_roofline(model.object_composer, cfg, inputs, ms[0] / 10)
    print(f"  mlp {ms[0] / 10:.3f} ms, composite {ms[1] / 10:.3f} ms, executed {roof['flop_executed'] / 1e9:.1f} GFLOP (algorithmic "
          f"{roof['flop_algorithmic'] / 1e9:.1f}), {roof['achieved']} TFLOP/s = {roof['frac']} of peak; evaluated {roof['evaluated_samples']}")
    if "profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
