#!/usr/bin/env python
"""Headline benchmark: Mrays/s of the HIP renderer on BASELINE.json configs[1].

Workload (per GPU): the tennis renderer (4 objects) with the hierarchical override - 64 coarse +
128 resampled positions per object and ray, coarse and fine networks - on one 256x256 frame
(65 536 rays) of the seeded synthetic tennis scene, eval mode, fp32.  A "step" is one full render
from the scene encoding (camera, object poses, style, deformation - resident in HBM) to the result
tensors of ``EnvironmentModel.forward(mode="scene_encodings")``.  With N GPUs every rank renders
its own frame (weak scaling) and the rendered ``fine.global.integrated_features`` maps are exchanged
with one RCCL all_gather inside the timed region.

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel (the fused fp32-MFMA MLP,
``k_mlp_mfma``): algorithmic FLOPs per launch (in-box samples actually evaluated x
FLOP/sample of the object's networks) / average launch duration measured with HIP events on the
launch stream.  ``cpu_baseline`` times the CPU oracle (a restatement of the reference's PyTorch op
graph, 1000-ray chunks like the reference's full-frame path) on a bounded ray subset of the same
frame on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz


def flops_per_sample(model_cfg: dict) -> float:
    """Matmul FLOPs (2 per MAC) of one evaluated sample, SURVEY.md section 8d."""
    n = model_cfg["nerf_model"]
    din = 6 if n["architecture"].endswith("skybox_adain_style_nerf_model_v3") else 3
    enc = din * (1 + 2 * n["position_encoder"]["octaves"])
    w, layers, f = n["layers_width"], n["backbone_layers_count"], n["output_features"]
    mac = enc * w + (layers - 2) * w * w + (w + enc) * w + w * w + w * (w // 2) + (w // 2) * f
    if din == 3:
        mac += w  # sigma head
    b = model_cfg["ray_bender_model"]
    if b["architecture"].endswith("positional_ray_bender_model"):
        benc = 3 * (1 + 2 * b["position_encoder"]["octaves"]) + model_cfg["deformation_features"]
        bw, bl = b["layers_width"], b["layers_count"]
        mac += benc * bw + (bl - 2) * bw * bw + (bw + benc) * bw + bw * 3
    return 2.0 * mac


def train_step_leg(args, dev, world, rank, dist, lib):
    """Secondary figure (never the headline): one data-parallel training step of the renderer in the shape of
    BASELINE.json configs[4] / SURVEY.md C5 - minecraft, 3 frames per GPU, one 48x48 patch at strides [4, 8] per frame
    (2880 rays), perturb=True, train-mode BatchNorm, forward + backward (pr_render_backward) + gradient all-reduce
    over RCCL + Adam on the composer parameters.  The loss reads global.integrated_features only, which is where the
    shipped configurations send gradients (every other renderer loss weight is 0)."""
    from playableenvironments_amd import configs, synthetic, parallel
    from playableenvironments_amd.environment_model import EnvironmentModel
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.train().to(dev)
    # the BatchNorm sample-count check is read back asynchronously (raised at the next call) so that the host can
    # enqueue the backward pass and the next step while the device works
    model.object_composer.batchnorm_check = "deferred"
    size = (288, 512)
    scene = synthetic.minecraft_scene(batch=3, seed=77, image_size=size)   # same frames on every rank: weak scaling
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)          # produced by trainable encoders in the reference
    params = list(model.object_composer.parameters())
    opt = torch.optim.Adam(params, lr=1e-5, fused=True)
    steps, warmup = max(1, args.steps), max(2, args.warmup)

    def step():
        opt.zero_grad(set_to_none=True)
        for attempt in range(20):
            try:
                out = model(sc["camera_rotations"], sc["camera_translations"], sc["focals"], size,
                            sc["object_rotation_parameters"], sc["object_translation_parameters"], sc["object_style"],
                            sc["object_deformation"], sc["object_in_scene"], 2880, True, 0, patch_size=48,
                            patch_stride=[4, 8], mode="scene_encodings")
                break
            except ValueError:
                # a random patch that misses an object leaves its BatchNorm without samples: torch (and the reference)
                # raise; a trainer would skip the batch - here the patch is re-drawn.  With the deferred check the error
                # concerns the PREVIOUS step (whose update was harmless: no samples, no gradient) and this call simply
                # proceeds
                if attempt == 19:
                    raise
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        loss.backward()
        parallel.allreduce_gradients(params)
        opt.step()
        return out

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rays = int(out["coarse"]["global"]["opacity"].numel())
    return {
        "value": round(rays * world * steps / dt / 1e6, 4),
        "unit": "Mrays/s trained (forward + backward + optimiser step)",
        "ms_per_step": round(dt / steps * 1e3, 3),
        "rays_per_gpu_per_step": rays,
        "workload": "minecraft shipped config, 3 frames/GPU x (48x48 patch @ strides [4, 8] = 2880 rays), perturb, train-mode "
                    "BatchNorm - BASELINE.json configs[4] renderer part",
        "parallelism": f"data parallel x{world}" + (", one flat RCCL all_reduce of the parameter gradients" if world > 1 else ""),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--image", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "f16x3"], default="fp32",
                    help="fp32 = exact fp32 MFMA (default, the headline); f16x3 = fp32 emulated with three fp16 MFMAs")
    ap.add_argument("--no-split-precision", action="store_true", help="skip the secondary f16x3 measurement")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--cpu-rays", type=int, default=64, help="the CPU baseline renders a cpu_rays x cpu_rays pixel grid")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads of the CPU baseline (all 256 host cores are >50x SLOWER on these small ops)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the renderer)")
    # PR_BENCH_DEVICE / PR_BENCH_BACKEND: test knobs (several ranks on one GPU over gloo exercise the multi-rank path
    # where only one device exists); the defaults are one rank per GPU over RCCL
    device_index = int(os.environ.get("PR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from playableenvironments_amd import configs, synthetic, _lib
    from playableenvironments_amd.environment_model import EnvironmentModel

    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    model.eval().to(dev)
    model.object_composer.precision = args.precision
    size = (args.image, args.image)
    # the same frame on every rank: weak scaling with exactly the same work per GPU (a different frame per rank would
    # make the max-over-ranks time follow the heaviest frame instead of the system)
    scene = synthetic.tennis_scene(seed=1234, image_size=size)
    scene_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}

    def step():
        with torch.no_grad():
            out = model(scene_dev["camera_rotations"], scene_dev["camera_translations"], scene_dev["focals"], size,
                        scene_dev["object_rotation_parameters"], scene_dev["object_translation_parameters"],
                        scene_dev["object_style"], scene_dev["object_deformation"], scene_dev["object_in_scene"],
                        0, False, mode="scene_encodings")
        feats = out["fine"]["global"]["integrated_features"]
        if world > 1:
            # one RCCL collective for the rendered feature maps (50 MB per frame); every rank receives the stack,
            # rank 0 is the consumer (decoder / writer) in the reference's evaluation flow.  It runs on RCCL's own
            # stream behind this step's kernels and overlaps the NEXT step's rendering; the step after that (or the end
            # of the timed region) waits for it.
            gather.submit(feats)
            gather.drain_done()
        return out

    from playableenvironments_amd.parallel import AsyncFeatureGather

    class _Gather(AsyncFeatureGather):
        def drain_done(self):          # the benchmark has no consumer: drop finished stacks, keep the one in flight
            self._done.clear()

    gather = _Gather(depth=1)

    def drain():
        gather.drain(keep=False)

    lib = _lib.load()

    def timed(steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between barrier + synchronize; max over ranks."""
        for _ in range(warmup):
            step()
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        lib.pr_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()                      # every gather of the timed steps completes inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        lib.pr_profile_enable(0)
        kernel_ms = (C.c_double * 2)()
        kernel_launches = (C.c_int32 * 2)()
        _lib.check(lib.pr_profile_collect(kernel_ms, kernel_launches), "pr_profile_collect")
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kernel_ms, kernel_launches

    elapsed, ms, launches = timed(args.steps, args.warmup)
    split = None
    if args.precision == "fp32" and not args.no_split_precision:
        # secondary measurement, never the headline: the same step with the MLP on the split-precision
        # kernel (fp32 emulated with three fp16 MFMAs; same parity tolerance in tests/test_gpu.py)
        model.object_composer.precision = "f16x3"
        split_s, split_ms, _ = timed(args.steps, max(1, args.warmup))
        model.object_composer.precision = "fp32"
        split = (split_s, split_ms[0] / max(1, args.steps))

    # algorithmic FLOPs of the MLP launches of one step: evaluated samples x FLOP/sample
    comp = model.object_composer
    comp_inputs = None
    with torch.no_grad():
        from playableenvironments_amd.environment_model import camera_rays, euler_to_matrix
        rows = torch.arange(size[0] * size[1], dtype=torch.int32) // size[1]
        cols = torch.arange(size[0] * size[1], dtype=torch.int32) % size[1]
        c2w = euler_to_matrix(scene_dev["camera_rotations"], scene_dev["camera_translations"])
        o, d, n = camera_rays(c2w, scene_dev["focals"] * cfg["data"]["focal_length_multiplier"], size[0], size[1], rows, cols)
        w2o, _ = model.compute_transformation_matrix_w2o_o2w(scene_dev["object_rotation_parameters"],
                                                             scene_dev["object_translation_parameters"])
        ex = comp(o, d, n, w2o, scene_dev["object_style"].unsqueeze(-3), scene_dev["object_deformation"].unsqueeze(-3),
                  scene_dev["object_in_scene"].unsqueeze(-2), False, _export=True)
    torch.cuda.synchronize()
    helper = comp.object_id_helper
    flops = 0.0
    evaluated = {}
    for ty in ("coarse", "fine"):
        ev = sum(p["evaluated"].cpu() for p in ex[ty]["_samples"])
        evaluated[ty] = [int(v) for v in ev]
        for k in range(helper.objects_count):
            flops += float(ev[k]) * flops_per_sample(cfg["model"]["object_models"][helper.model_idx_by_object_idx(k)])

    # HBM traffic of the dominant kernel from the committed PMC pass (collected separately: counters
    # cannot ride along with the timed run), and the fp32 MFMA rate this box sustains
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(pmc_path) and size == (256, 256):
        with open(pmc_path) as f:
            traffic = json.load(f)["k_mlp_mfma"]["hbm_bytes_per_launch_avg"]
    probe = {}
    for name, rnd in (("constant_operands", 0), ("random_operands", 1)):
        tf, pms = C.c_double(), C.c_double()
        _lib.check(lib.pr_probe_mfma_f32(100000, rnd, C.byref(tf), C.byref(pms), None), "pr_probe_mfma_f32")
        probe[name] = round(tf.value, 1)

    rays_per_gpu = size[0] * size[1]
    total_rays = rays_per_gpu * world * args.steps
    value = total_rays / elapsed / 1e6
    mlp_ms = ms[0] / max(1, args.steps)
    achieved = flops / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    result = {
        "metric": "Mrays/s",
        "value": round(value, 4),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"tennis renderer, {size[0]}x{size[1]} frame per GPU, 4 objects, 64+128 hierarchical samples/ray "
                        "(coarse+fine networks), eval - BASELINE.json configs[1]",
            "rays_per_gpu": rays_per_gpu,
            "frames_per_gpu": 1,
            "parallelism": f"frame shard x{world}" + (" + RCCL all_gather of feature maps" if world > 1 else ""),
        },
        "frames_per_s_256x256": round(value * 1e6 / 65536.0, 3),
        "roofline": {
            "bound": "mfma",
            "kernel": "k_mlp_mfma (fused fp32 MFMA MLP, all launches of one step)",
            "achieved": round(achieved, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (average over the launches of a step), rocprofv3 PMC passes of tools/collect_pmc.sh, profiles/r01_pmc_summary.json",
            "peak_measured": probe,
            "flop_per_step": flops,
            "mlp_ms_per_step": round(mlp_ms, 3),
            "mlp_launches_per_step": int(launches[0] / max(1, args.steps)),
            "composite_ms_per_step": round(ms[1] / max(1, args.steps), 3),
            "evaluated_samples": evaluated,
        },
    }

    if args.precision == "f16x3":
        result["dtype"] = "f16x3 (fp32 emulated with three fp16 MFMAs, fp32 accumulate)"
        result["roofline"]["kernel"] = "k_mlp_split (fused split-precision MFMA MLP); achieved/peak are in fp32-equivalent algorithmic FLOPs"
    if split is not None:
        result["split_precision"] = {
            "value": round(rays_per_gpu * world * args.steps / split[0] / 1e6, 4),
            "unit": "Mrays/s",
            "ms_per_step": round(split[0] / args.steps * 1e3, 3),
            "mlp_ms_per_step": round(split[1], 3),
            "algorithmic_tflops": round(flops / (split[1] * 1e-3) / 1e12, 2) if split[1] > 0 else None,
            "note": "same workload with ObjectComposer.precision='f16x3' (k_mlp_split): every fp32 product as three fp16 "
                    "MFMAs, ~22-bit operands, fp32 accumulation; passes the same oracle/golden parity tolerance; "
                    "reported beside the exact-fp32 headline, not as it",
        }
    if not args.no_train_step:
        result["train_step"] = train_step_leg(args, dev, world, rank, dist, lib)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import render_oracle as ro
        from tests.helpers import composer_inputs, grid_pixels
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        torch.set_num_threads(threads)
        sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
        n_side = args.cpu_rays
        inputs = composer_inputs(cfg, scene, pixels=grid_pixels(size[0], size[1], n_side))
        with torch.no_grad():
            t0 = time.perf_counter()
            ro.batchified_composer_call(cfg, sd, *inputs, False, chunk=1000)
            cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(n_side * n_side / cpu_s / 1e6, 6),
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{n_side}x{n_side} pixel grid ({n_side * n_side} rays) of the same frame and weights, "
                      f"oracle/render_oracle.py in 1000-ray chunks, {cpu_s:.1f} s wall, {threads} torch threads "
                      f"of {os.cpu_count()} host cores",
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
