#!/usr/bin/env python
"""Headline benchmark: Mrays/s of the HIP renderer on BASELINE.json configs[1].

Workload (per GPU): the tennis renderer (4 objects) with the hierarchical override - 64 coarse +
128 resampled positions per object and ray, coarse and fine networks - on one 256x256 frame
(65 536 rays) of the seeded synthetic tennis scene, eval mode, fp32.  A "step" is one full render
from the scene encoding (camera, object poses, style, deformation - resident in HBM) to the result
tensors of ``EnvironmentModel.forward(mode="scene_encodings")``.  With N GPUs every rank renders
its own copy of the frame (weak scaling with identical work per GPU) and the rendered
``fine.global.integrated_features`` maps are exchanged with one RCCL all_gather inside the timed region.

``python bench.py --gpus N`` launches itself under ``torch.distributed.run`` (one process per GPU) when it
is not already running under it, so both ``python bench.py --gpus 8`` and
``python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`` work.

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel (the fused fp32-MFMA MLP,
``k_mlp_mfma_group`` - the tile loop of ``k_mlp_mfma``, one launch per model type for its four objects): FLOPs per launch /
average launch duration measured with HIP events on the launch stream.  Two FLOP counts are given: ``algorithmic`` (in-box samples x FLOP/sample of the
object's networks - what the reference evaluates) and ``executed`` (minus the feature-head FLOPs of the
samples whose density is <= 0, which the sigma-gated head skips exactly); ``achieved`` uses the EXECUTED
count.  Secondary legs, none of which is the headline: the split-precision kernel, the same-GPU PyTorch
op graph of the reference (``reference_graph_on_gpu``), PSNR against the CPU oracle, 8 distinct frames
sharded over the ranks (BASELINE.json configs[3]), a data-parallel training step (configs[4], with its own
roofline), configs[0] at full size, configs[2] (the shipped minecraft renderer), and ``cpu_baseline``: the CPU oracle (a restatement of the reference's
PyTorch op graph, 1000-ray chunks like the reference's full-frame path) on bounded ray subsets of the same
frame on this box's host cores with 1 / 16 / all threads.
"""
import argparse
import ctypes as C
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
SCENE_KEYS = ("camera_rotations", "camera_translations", "focals", "object_rotation_parameters",
              "object_translation_parameters", "object_style", "object_deformation", "object_in_scene")


LINE_BUDGET = 4096      # bytes of the ONE stdout line (r04's 22.6 KB line was not parsed by the driver; everything else -> bench_full.json)


def _dig(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_result(result: dict, full_path=None) -> dict:
    """The driver's line: the contract keys, `roofline` and `cpu_baseline` of the headline, and a flat handful of scalar summaries
    of the secondary legs.  Every note, per-run list and per-shard table stays in the full record (`full`)."""
    r = result
    roof = dict(r.get("roofline") or {})
    line = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median",
                                  "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if "frames_per_s_256x256" in r:
        line["frames_per_s_256x256"] = r["frames_per_s_256x256"]       # (BASELINE.json: "Mrays/s (and frames/s at 256x256)")
    cfgw = dict(r.get("config") or {})
    line["config"] = cfgw
    line["roofline"] = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_from_this_library",
                                                   "peak_measured", "flop_per_step", "flop_per_step_algorithmic", "algorithmic_tflops",
                                                   "mlp_ms_per_step", "mlp_launches_per_step", "composite_ms_per_step", "clock") if k in roof}
    if isinstance(line["roofline"].get("kernel"), str):
        line["roofline"]["kernel"] = line["roofline"]["kernel"].split(" ")[0]
    cpu = r.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                "sample": f"{_dig(cpu, 'runs', default=[{}])[0].get('rays')} rays of the same frame, oracle in 1000-ray chunks, "
                                          f"2 warm-ups + median of 5 ({_dig(cpu, 'runs', default=[{}])[0].get('seconds')} s each)",
                                "cpu_model": cpu.get("cpu_model"), "host_cores": cpu.get("host_cores"), "usable_cores": cpu.get("usable_cores"),
                                "full_size_value": _dig(cpu, "full_size", "value"), "gpu_over_cpu": cpu.get("gpu_over_cpu")}
    dist = r.get("distributed") or {}
    line["distributed"] = {k: dist[k] for k in ("world_size", "backend", "nccl_version", "rank_devices") if k in dist}
    fg = r.get("feature_gather") or {}
    if "all_gather" in fg:
        line["feature_gather"] = {"bytes_per_rank": fg.get("bytes_per_rank"),
                                  "all_gather_ms": _dig(fg, "all_gather", "ms"),
                                  "all_gather_GB_per_s": _dig(fg, "all_gather", "receive_GB_per_s_per_receiving_rank"),
                                  "gather_dst0_ms": _dig(fg, "gather_dst0", "ms"),
                                  "gather_dst0_GB_per_s": _dig(fg, "gather_dst0", "receive_GB_per_s_per_receiving_rank")}
    summary = {
        "identical_frames_mrays": _dig(r, "identical_frames", "value"),
        "split_precision_mrays": _dig(r, "split_precision", "value"),
        "split_precision_sclk_mhz": _dig(r, "split_precision", "clock", "sclk_mhz"),
        "half_precision_mrays": _dig(r, "half_precision", "value"),
        "train_step_ms": _dig(r, "train_step", "ms_per_step"),
        "train_step_ms_median": _dig(r, "train_step", "ms_per_step_median"),
        "train_step_frac": _dig(r, "train_step", "roofline", "frac"),
        "train_step_f16x3_ms": _dig(r, "train_step", "f16x3", "ms_per_step"),
        "train_step_f16x3_ms_median": _dig(r, "train_step", "f16x3", "ms_per_step_median"),
        "train_step_with_decoder_ms": _dig(r, "train_step_with_decoder", "maps_route", "ms_per_step"),
        "distinct_frames_shipped_mrays": _dig(r, "distinct_frames", "shipped_p72", "value"),
        "distinct_frames_hierarchical_mrays": _dig(r, "distinct_frames", "hierarchical_64_128", "value"),
        "native_frame_tennis_ms": _dig(r, "native_eval_frame", "tennis", "fp32", "scene_encoding_default", "ms_per_frame"),
        "native_frame_tennis_f16x3_ms": _dig(r, "native_eval_frame", "tennis", "f16x3", "scene_encoding_default", "ms_per_frame"),
        "native_frame_tennis_frac": _dig(r, "native_eval_frame", "tennis", "fp32", "roofline", "frac"),
        "native_frame_minecraft_ms": _dig(r, "native_eval_frame", "minecraft", "fp32", "scene_encoding_default", "ms_per_frame"),
        "native_frame_minecraft_frac": _dig(r, "native_eval_frame", "minecraft", "fp32", "roofline", "frac"),
        "evaluator_default_call_tennis_ms": _dig(r, "native_eval_frame", "tennis", "fp32", "observations_default", "ms_per_frame"),
        "evaluator_default_call_minecraft_ms": _dig(r, "native_eval_frame", "minecraft", "fp32", "observations_default", "ms_per_frame"),
        "config0_frac": _dig(r, "config0_single_player_128", "roofline", "frac"),
        "config2_minecraft_ms": _dig(r, "config2_minecraft_256", "fp32", "ms_per_frame"),
        "config2_minecraft_frac": _dig(r, "config2_minecraft_256", "roofline", "frac"),
        "reference_graph_fastest_mrays": _dig(r, "reference_graph_on_gpu", "value"),
        "hip_over_reference_graph": _dig(r, "reference_graph_on_gpu", "hip_over_reference_graph"),
        "f16x3_over_reference_graph": _dig(r, "reference_graph_on_gpu", "tiers", "f16x3", "over_reference_graph"),
        "f16_over_reference_graph": _dig(r, "reference_graph_on_gpu", "tiers", "f16", "over_reference_graph"),
        "psnr_db": _dig(r, "psnr_db", "fp32"),
        "psnr_db_f16x3": _dig(r, "psnr_db", "f16x3"),
        "psnr_db_f16": _dig(r, "psnr_db", "f16"),
    }
    line["summary"] = {k: v for k, v in summary.items() if v is not None}
    line["library_sha256"] = (r.get("library_sha256") or "")[:16]
    line["full"] = full_path
    return line


def compact_line(result: dict, full_path=None, budget: int = LINE_BUDGET) -> str:
    """json of compact_result, guaranteed < `budget` bytes: optional blocks are dropped (last first) rather than ever exceeding it."""
    line = compact_result(result, full_path)
    text = json.dumps(line, separators=(",", ":"))
    for optional in ("summary", "feature_gather", "distributed"):
        if len(text.encode()) < budget:
            break
        if optional == "distributed" and isinstance(line.get("distributed"), dict):
            line["distributed"].pop("rank_devices", None)
        else:
            line.pop(optional, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) < budget and "\n" not in text, len(text)
    return text


def emit(result: dict, full_path: str) -> None:
    """Full record -> `full_path` (and gpurun_out/ when it exists, so that a gpurun call brings it home); compact line -> stdout, LAST."""
    paths = [full_path]
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch) and os.path.dirname(os.path.abspath(full_path)) != scratch:
        paths.append(os.path.join(scratch, os.path.basename(full_path)))
    written = None
    for path in paths:
        try:
            with open(path, "w") as f:
                json.dump(result, f, indent=1)
            written = written or path
        except OSError as e:
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(result, os.path.relpath(written, ROOT) if written else None), flush=True)


def flops_per_sample(model_cfg: dict, head_only: bool = False) -> float:
    """Matmul FLOPs (2 per MAC) of one evaluated sample, SURVEY.md section 8d.  ``head_only``: the three feature-head
    products alone (what the sigma gate skips for a sample that cannot contribute)."""
    n = model_cfg["nerf_model"]
    din = 6 if n["architecture"].endswith("skybox_adain_style_nerf_model_v3") else 3
    enc = din * (1 + 2 * n["position_encoder"]["octaves"])
    w, layers, f = n["layers_width"], n["backbone_layers_count"], n["output_features"]
    head = w * w + w * (w // 2) + (w // 2) * f
    if head_only:
        return 2.0 * head
    mac = enc * w + (layers - 2) * w * w + (w + enc) * w + head
    if din == 3:
        mac += w  # sigma head
    b = model_cfg["ray_bender_model"]
    if b["architecture"].endswith("positional_ray_bender_model"):
        benc = 3 * (1 + 2 * b["position_encoder"]["octaves"]) + model_cfg["deformation_features"]
        bw, bl = b["layers_width"], b["layers_count"]
        mac += benc * bw + (bl - 2) * bw * bw + (bw + benc) * bw + bw * 3
    return 2.0 * mac


def scene_args(sc, size):
    return [sc["camera_rotations"], sc["camera_translations"], sc["focals"], size, sc["object_rotation_parameters"],
            sc["object_translation_parameters"], sc["object_style"], sc["object_deformation"], sc["object_in_scene"]]


def to_device(scene, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}


def _lib_handle():
    from playableenvironments_amd import _lib
    return _lib.load()


def profile_arrays():
    from playableenvironments_amd import _lib
    return (C.c_double * _lib.PR_PROFILE_CATEGORIES)(), (C.c_int32 * _lib.PR_PROFILE_CATEGORIES)()


def library_sha256() -> str:
    """sha256 of the libplayrender.so this process loads: stamped on the bench line and on profiles/*_pmc_summary.json, so that a
    ``traffic`` figure collected from an older build is detectable."""
    import hashlib
    from playableenvironments_amd import _lib
    with open(_lib.library_path(), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def composer_call_inputs(model, cfg, scene_dev, size):
    """What EnvironmentModel.forward(mode="scene_encodings") hands to ObjectComposer.forward for every pixel of the frame(s)."""
    from playableenvironments_amd.environment_model import camera_rays, euler_to_matrix
    with torch.no_grad():
        rows = torch.arange(size[0] * size[1], dtype=torch.int32) // size[1]
        cols = torch.arange(size[0] * size[1], dtype=torch.int32) % size[1]
        c2w = euler_to_matrix(scene_dev["camera_rotations"], scene_dev["camera_translations"])
        o, d, n = camera_rays(c2w, scene_dev["focals"] * cfg["data"]["focal_length_multiplier"], size[0], size[1], rows, cols)
        w2o, _ = model.compute_transformation_matrix_w2o_o2w(scene_dev["object_rotation_parameters"],
                                                             scene_dev["object_translation_parameters"])
    return [o, d, n, w2o, scene_dev["object_style"].unsqueeze(-3), scene_dev["object_deformation"].unsqueeze(-3),
            scene_dev["object_in_scene"].unsqueeze(-2)]


def flop_counts(comp, cfg, inputs):
    """Matmul FLOPs of the MLP launches of one composer call: evaluated samples x FLOP/sample (algorithmic, what the reference
    computes), minus the feature-head FLOPs of the samples the sigma gate skipped (executed).  One extra call with the sample
    counters exported."""
    with torch.no_grad():
        ex = comp(*inputs, False, _export=True)
    torch.cuda.synchronize()
    helper = comp.object_id_helper
    flops = executed = 0.0
    evaluated, head_samples = {}, {}
    for ty in ("coarse", "fine"):
        if ty not in ex:
            continue
        ev = sum(p["evaluated"].cpu() for p in ex[ty]["_samples"])
        hd = sum(p["head_evaluated"].cpu() for p in ex[ty]["_samples"])
        evaluated[ty] = [int(v) for v in ev]
        head_samples[ty] = [int(v) for v in hd]
        for k in range(helper.objects_count):
            mcfg = cfg["model"]["object_models"][helper.model_idx_by_object_idx(k)]
            flops += float(ev[k]) * flops_per_sample(mcfg)
            executed += float(ev[k]) * flops_per_sample(mcfg) - float(ev[k] - hd[k]) * flops_per_sample(mcfg, head_only=True)
    return flops, executed, evaluated, head_samples


def leg_roofline(comp, cfg, inputs, mlp_ms):
    """Per-leg roofline of the MLP launches: executed FLOPs of the leg's own frame(s) / HIP-event time of its MLP launches."""
    flops, executed, evaluated, _ = flop_counts(comp, cfg, inputs)
    achieved = executed / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "mlp_ms": round(mlp_ms, 3), "flop_executed": executed,
            "flop_algorithmic": flops, "evaluated_samples": evaluated}


def median(values):
    v = sorted(values)
    return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2]) if v else None


def event_gaps_ms(events):
    """Device time between consecutive events of a list recorded at the start of every timed step (+ one at the end)."""
    return [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]


def shard_balance_leg(model, cfg, scene_dev, size, shards=8, reps=3):
    """What an N-rank ray-sharded render of ONE frame would hand to each rank, measured at N = 1: the ``shards`` virtual shards of
    the frame are rendered one after the other on this GPU (composer call on the shard's rays), for contiguous ranges of the
    pixel list (parallel.shard_range) and for 8 x 8-pixel tiles dealt round robin (parallel.tile_shard_lists).  A sharded frame
    takes as long as its slowest shard: max / mean of the evaluated samples and of the measured time."""
    from playableenvironments_amd import parallel
    comp = model.object_composer
    inputs = composer_call_inputs(model, cfg, scene_dev, size)
    total = size[0] * size[1]
    dev = inputs[1].device
    schemes = {
        "contiguous_ranges": [torch.arange(*parallel.shard_range(total, r, shards), device=dev) for r in range(shards)],
        "tiles_8x8_round_robin": [l.to(dev) for l in parallel.tile_shard_lists([size], shards)],
    }
    out = {}
    for name, lists in schemes.items():
        samples, times = [], []
        for idx in lists:
            sub = list(inputs)
            sub[1] = inputs[1].index_select(-2, idx)
            with torch.no_grad():
                ex = comp(*sub, False, _export=True)           # (warm-up + the sample counts)
                samples.append(sum(int(p["evaluated"].sum()) for ty in ("coarse", "fine") if ty in ex for p in ex[ty]["_samples"]))
                gaps = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    comp(*sub, False)
                    e1.record()
                    e1.synchronize()
                    gaps.append(e0.elapsed_time(e1))
            times.append(median(gaps))
        mean_s, mean_t = sum(samples) / shards, sum(times) / shards
        out[name] = {"evaluated_samples_max": max(samples), "evaluated_samples_mean": round(mean_s, 1),
                     "samples_max_over_mean": round(max(samples) / mean_s, 3) if mean_s else None,
                     "ms_max": round(max(times), 3), "ms_mean": round(mean_t, 3),
                     "ms_max_over_mean": round(max(times) / mean_t, 3) if mean_t else None,
                     "per_shard_samples": samples, "per_shard_ms": [round(t, 3) for t in times]}
    return out


def feature_gather_leg(feats, dist, world, rank, dev, reps=10):
    """The exchange step of a sharded render on its own: every rank contributes its rendered feature map; once as all_gather
    (every rank receives the stack) and once as gather(dst=0) (rank 0, the decoder / writer of the evaluation flow, alone)."""
    if dist is None:
        return {"world_size": 1, "note": "one rank: no exchange (gather_ray_shards returns the local tensor)"}
    src = feats.contiguous()
    nbytes = src.numel() * src.element_size()
    out = {"world_size": world, "bytes_per_rank": nbytes, "tensor": list(src.shape)}
    stack = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
    parts = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None

    def all_gather():
        dist.all_gather_into_tensor(stack, src)

    def gather_dst0():
        dist.gather(src, parts, dst=0)

    for name, fn in (("all_gather", all_gather), ("gather_dst0", gather_dst0)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        dt = max_over_ranks(time.perf_counter() - t0, dist, dev) / reps
        received = nbytes * (world - 1)
        out[name] = {"ms": round(dt * 1e3, 3), "bytes_received_per_receiving_rank": received,
                     "receive_GB_per_s_per_receiving_rank": round(received / dt / 1e9, 2),
                     "GB_per_s_per_link_if_direct": round(nbytes / dt / 1e9, 2),
                     "receiving_ranks": world if name == "all_gather" else 1}
    out["link_note"] = ("xGMI is point to point: with direct transfers each of the (world - 1) links into a receiving rank carries one "
                        "rank's map (bytes_per_rank) per collective - GB_per_s_per_link_if_direct, against ~153 GB/s per link per direction")
    return out


def max_over_ranks(value, dist, dev):
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """File descriptor 1 -> stderr for the duration of the block (C-level writers included), C stdio flushed on the way out."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def train_traffic():
    """HBM bytes per training step from the committed PMC passes of tools/collect_pmc_train.sh (all library kernels of a step, and
    the three largest), with the library stamp they were collected from."""
    for name in ("r06_pmc_train_summary.json", "r05_pmc_train_summary.json", "r04_pmc_train_summary.json", "r03_pmc_train_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return {"traffic": None}
    with open(path) as f:
        pmc = json.load(f)
    big = {k.split("::")[1]: int(v["hbm_MB"] * 1e6) for k, v in list(pmc["per_step"].items())[:3]}
    return {"traffic": int(pmc["hbm_MB_per_step_all_library_kernels"] * 1e6),
            "traffic_unit": "HBM bytes per training step, all library kernels (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes of "
                            f"tools/collect_pmc_train.sh, profiles/{name}: the fp32 step): ~1.5 TB/s over the step - the step is bound "
                            "by the matrix pipes, not by HBM",
            "traffic_largest_kernels": big,
            "traffic_from_this_library": pmc.get("library_sha256") == library_sha256()}


def train_step_leg(args, dev, world, rank, dist, lib, arena=True, precision="fp32"):
    """Secondary figure (never the headline): one data-parallel training step of the renderer in the shape of
    BASELINE.json configs[4] / SURVEY.md C5 - minecraft, 3 frames per GPU, one 48x48 patch at strides [4, 8] per frame
    (2880 rays), perturb=True, train-mode BatchNorm, forward + backward (pr_render_backward) + gradient all-reduce
    over RCCL + Adam on the composer parameters.  The loss reads global.integrated_features only, which is where the
    shipped configurations send gradients (every other renderer loss weight is 0).  Its roofline: 3 x the forward
    matmul FLOPs of the samples that were evaluated (dX + dW + forward) / step time, against the fp32
    matrix peak, with the HIP-event time of the forward MLP launches and of the backward dX / dW products."""
    from playableenvironments_amd import configs, synthetic, parallel, _lib
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
    scene = synthetic.minecraft_scene(batch=3, seed=77 + rank, image_size=size)   # every rank trains on its own frames (data parallel)
    sc = to_device(scene, dev)
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)          # produced by trainable encoders in the reference
    comp = model.object_composer
    comp.precision = precision      # "f16x3": split-precision products (fp16 pairs of scaled operands) in phase 1 of the forward and in the backward
    # the optimiser works on the composer's parameter ARENA (parallel.flatten_parameters: every parameter a view of one flat
    # tensor - names, state_dict and values unchanged; Adam is element-wise, so the update is bit for bit the per-tensor one):
    # one fused launch instead of a multi-tensor sweep over 170 tensors.  arena=False: torch's Adam on the separate tensors.
    flat = parallel.flatten_parameters(comp) if arena else None
    params = list(comp.parameters())
    # arena: parallel.ArenaAdam - torch.optim.Adam's update as ONE full-grid launch of pr_adam_step over the arena (torch's fused
    # kernel gives a single tensor one block per 65 536 elements: 32 blocks, 0.10 ms of a step); separate tensors: torch's fused Adam
    opt = parallel.ArenaAdam([flat], lr=1e-5) if arena else torch.optim.Adam(params, lr=1e-5, fused=True)
    steps, warmup = max(1, args.steps), max(2, args.warmup)
    K = comp.object_id_helper.objects_count
    evaluated = torch.zeros((K,), dtype=torch.int64, device=dev)
    retries = [0]
    # the all-reduce of the renderer's flat gradient buffer starts inside backward() (behind pr_render_backward's launches, in front
    # of whatever autograd still has to walk: the pose / style producers) and is waited for in front of the optimiser step
    overlap = parallel.OverlappedGradientAllReduce(comp) if dist is not None else None

    def iteration():
        out = model(*scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
        loss = out["coarse"]["global"]["integrated_features"].square().mean()
        loss.backward()
        if overlap is None or overlap.finish() == 0:
            parallel.allreduce_gradients(params)
        if arena:
            parallel.flat_gradient(flat, comp)
        opt.step()
        seen = comp.last_normalised_samples["coarse"]
        evaluated.add_(seen)
        return out

    def step():
        """eager: one launch per kernel.  A random patch that misses an object leaves its BatchNorm with an EMPTY batch, which torch
        (and therefore the reference) accepts: statistics untouched, no gradient.  Only a batch of exactly one sample raises; the call
        would be repeated with a new patch and counted (never observed: a ray that meets a box brings all its samples)."""
        opt.zero_grad(set_to_none=True)       # (arena: flat_gradient has moved the views' gradients to the arena)
        for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
            sc[k].grad = None                 # (stand-ins for encoder outputs: their gradients are consumed, not accumulated over steps)
        for attempt in range(20):
            try:
                return iteration()
            except ValueError:
                retries[0] += 1
                if attempt == 19:
                    raise

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        evaluated.zero_()
        retries[0] = 0
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            out = fn()
        marks[steps].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return max_over_ranks(time.perf_counter() - t0, dist, dev), out, event_gaps_ms(marks)

    dt, out, gaps = timed(step)
    redrawn = retries[0]
    counts = [int(v) / steps for v in evaluated.cpu()]
    # per-kernel HIP-event times from a second, untimed pass (an event pair around each of the ~100 launches of a step
    # would slow the timed steps down)
    lib.pr_profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    lib.pr_profile_enable(0)
    ms, launches = profile_arrays()
    _lib.check(lib.pr_profile_collect(ms, launches), "pr_profile_collect")
    if overlap is not None:
        overlap.remove()
    rays = int(out["coarse"]["global"]["opacity"].numel())
    helper = comp.object_id_helper
    fwd_flops = sum(counts[k] * flops_per_sample(cfg["model"]["object_models"][helper.model_idx_by_object_idx(k)])
                    for k in range(K))
    step_ms = dt / steps * 1e3
    achieved = 3.0 * fwd_flops / (step_ms * 1e-3) / 1e12
    per = lambda i: round(ms[i] / steps, 3)
    gemm_ms = (ms[2] + ms[3]) / steps
    return {
        "value": round(rays * world * steps / dt / 1e6, 4),
        "unit": "Mrays/s trained (forward + backward + optimiser step)",
        "ms_per_step": round(step_ms, 3),
        "ms_per_step_median": round(median(gaps), 3),
        "ms_per_step_note": f"ms_per_step = wall time of the {steps} timed steps / {steps} (mean); median = of the device time between the "
                            "steps' first launches (events on the launch stream, rank 0)",
        "redrawn_patches": redrawn,
        "starved_note": "renderer calls repeated inside the timed region because an object's train-mode BatchNorm saw exactly one sample "
                        "(torch / the reference raise for that; a patch that misses an object altogether is an empty batch, which passes)",
        "rays_per_gpu_per_step": rays,
        "workload": "minecraft shipped config, 3 frames/GPU x (48x48 patch @ strides [4, 8] = 2880 rays), perturb, train-mode "
                    "BatchNorm - BASELINE.json configs[4] renderer part",
        "parallelism": f"data parallel x{world}" + (", one flat RCCL all_reduce of the parameter gradients started inside backward() "
                                                     "(parallel.OverlappedGradientAllReduce)" if world > 1 else ""),
        "optimizer": ("parallel.ArenaAdam on the composer's parameter arena (parallel.flatten_parameters + pr_adam_step: torch.optim.Adam's "
                      "update, one full-grid launch)") if arena else "torch.optim.Adam(fused=True) on the 170 separate parameter tensors",
        "roofline": {
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "flop_per_step": 3.0 * fwd_flops,
            "definition": "3 x forward matmul FLOPs of the evaluated samples (forward + dX + dW) / whole step time "
                          "(incl. compositing, BatchNorm passes, optimiser)",
            **train_traffic(),
            "evaluated_samples_per_step": [round(c, 1) for c in counts],
            "kernel_ms_per_step": {"forward_mlp": per(0), "forward_composite": per(1), "backward_dx_gemm": per(2),
                                   "backward_dw_gemm": per(3), "backward_composite": per(4)},
            "kernel_ms_note": "HIP-event time of every launch of an EAGER pass, summed per category: backward_dx_gemm = the fused head / chain "
                              "tile kernels (every input-gradient product), backward_dw_gemm = the one weight-gradient launch + its reduction",
            "backward_gemm_tflops": round(2.0 * fwd_flops / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms > 0 else None,
            "forward_mlp_tflops": round(fwd_flops / (ms[0] / steps * 1e-3) / 1e12, 2) if ms[0] > 0 else None,
        },
    }


class _ResidualBlock(torch.nn.Module):
    """3x3 conv - BatchNorm - ReLU - 3x3 conv - BatchNorm (reflection padding) around a shortcut; a 1x1 conv + BatchNorm
    shortcut where the channel count changes."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        nn = torch.nn
        self.body = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(cin, cout, 3, bias=False), nn.BatchNorm2d(cout), nn.ReLU(True),
                                  nn.ReflectionPad2d(1), nn.Conv2d(cout, cout, 3, bias=False), nn.BatchNorm2d(cout))
        self.shortcut = None if cin == cout else nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        return (x if self.shortcut is None else self.shortcut(x)) + self.body(x)


class StandInDecoder(torch.nn.Module):
    """Seeded stand-in for the image decoder behind the renderer in BASELINE.json configs[4] (the reference's DecoderV6 with the
    shipped minecraft settings, model/autoencoder_models/decoder_v6.py:10-96: bottleneck 128 channels at 1/8 resolution, three
    residual blocks per level, bilinear x2 + 3x3 conv + BatchNorm + ReLU per upsampling, the 1/4-resolution map concatenated as
    a skip, a 7x7 conv + sigmoid head).  Own code on stock torch.nn (MIOpen convolutions): it exists to put the decoder's
    launches and its gradient w.r.t. the renderer's maps into the measured step, not to reproduce trained images."""

    def __init__(self, image_channels: int = 3, bottleneck: int = 128, blocks: int = 3, levels=(2, 1)):
        super().__init__()
        nn = torch.nn
        width = bottleneck
        stages = []
        for idx, upsamplings in enumerate(reversed(list(levels))):
            layers = [_ResidualBlock(width * (2 if (b == 0 and idx > 0) else 1), width) for b in range(blocks)]
            for _ in range(upsamplings):
                layers += [nn.UpsamplingBilinear2d(scale_factor=2),
                           nn.Conv2d(width, width // 2, 3, padding=1, padding_mode="reflect", bias=False),
                           nn.BatchNorm2d(width // 2), nn.ReLU(True)]
                width //= 2
            stages.append(nn.Sequential(*layers))
        self.stages = nn.ModuleList(stages)
        self.head = nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(width, image_channels, 7), nn.Sigmoid())

    def forward(self, maps):
        """maps: [(N, 64, h/4, w/4), (N, 128, h/8, w/8)] - finest first, as forward_decoder takes them."""
        x = maps[-1]
        for i, stage in enumerate(self.stages):
            x = stage(x)
            if i != len(self.stages) - 1:
                x = torch.cat([x, maps[-i - 2]], dim=-3)
        return self.head(x)


def train_step_with_decoder_leg(args, dev, world, rank, dist, renderer_only_ms):
    """BASELINE.json configs[4] end to end in shape: the training step of ``train_step_leg`` with an image decoder behind the
    renderer - the renderer's per-stride maps -> decoder CNN -> image loss -> backward through the decoder into the maps ->
    pr_render_backward -> gradient all-reduce -> Adam on both parameter sets.  Two routes for the renderer -> decoder hand-over
    (SURVEY.md section 8 f-1): "maps" = ``decoder_features`` written channels-first by the compositing kernel, "ray_major" = the
    reference's route (ray-major integrated_features folded / split / permuted by wire_format, torch ops).  The decoder runs on
    a side stream (its forward, and therefore its autograd nodes); the dependency chain renderer forward -> decoder forward ->
    decoder backward -> renderer backward leaves only the decoder's optimiser update to overlap with the renderer's backward."""
    from playableenvironments_amd import configs, synthetic, parallel
    from playableenvironments_amd import wire_format as wf
    from playableenvironments_amd.environment_model import EnvironmentModel
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.train().to(dev)
    model.object_composer.batchnorm_check = "deferred"
    torch.manual_seed(1)
    decoder = StandInDecoder().train().to(dev)
    size = (288, 512)
    sc = to_device(synthetic.minecraft_scene(batch=3, seed=77 + rank, image_size=size), dev)
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        sc[k].requires_grad_(True)
    # both parameter sets as arenas (parallel.flatten_parameters), as in train_step_leg
    arena_render = parallel.flatten_parameters(model.object_composer)
    arena_decoder = parallel.flatten_parameters(decoder)
    render_params = list(model.object_composer.parameters())
    decoder_params = list(decoder.parameters())
    opt_render = parallel.ArenaAdam([arena_render], lr=1e-5)
    opt_decoder = parallel.ArenaAdam([arena_decoder], lr=1e-5)
    g = torch.Generator().manual_seed(5)
    target = torch.rand((3, 3, 192, 192), generator=g).to(dev)
    side = torch.cuda.Stream(dev)
    counts = [64, 128]
    steps, warmup = max(1, args.steps), max(2, args.warmup)

    def make_step(route):
        def step():
            opt_render.zero_grad(set_to_none=True)
            opt_decoder.zero_grad(set_to_none=True)
            out = model(*scene_args(sc, size), 2880, True, 0, patch_size=48, patch_stride=[4, 8], mode="scene_encodings",
                        **({"_decoder_features": counts} if route == "maps" else {}))
            if route == "maps":
                maps = out["coarse"]["global"]["decoder_features"]
            else:
                _, maps = wf.decoder_patches(out["coarse"]["global"]["integrated_features"], 48, [4, 8], counts)
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                image = decoder([m.reshape([-1] + list(m.shape[-3:])) for m in maps])
                loss = (image - target).square().mean()
            main.wait_stream(side)
            loss.backward()
            parallel.allreduce_gradients(render_params)         # (the renderer's gradients are one flat buffer: one collective)
            parallel.allreduce_gradients(decoder_params)
            parallel.flat_gradient(arena_render, model.object_composer)
            parallel.flat_gradient(arena_decoder, decoder)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                opt_decoder.step()
            opt_render.step()
            main.wait_stream(side)
            return loss
        return step

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            fn()
        marks[steps].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = max_over_ranks(time.perf_counter() - t0, dist, dev)
        return dt / steps * 1e3, median(event_gaps_ms(marks))

    results = {}
    for route in ("maps", "ray_major"):
        mean_ms, median_ms = timed(make_step(route))
        results[route] = {"ms_per_step": round(mean_ms, 3), "ms_per_step_median": round(median_ms, 3)}

    # the decoder alone on fixed maps (forward + backward to the maps + Adam), for the shares
    fixed = [torch.randn((3, 64, 48, 48), device=dev, requires_grad=True), torch.randn((3, 128, 24, 24), device=dev, requires_grad=True)]

    def decoder_only():
        opt_decoder.zero_grad(set_to_none=True)
        (decoder(fixed) - target).square().mean().backward()
        parallel.flat_gradient(arena_decoder, decoder)
        opt_decoder.step()
    decoder_ms, _ = timed(decoder_only)
    combined = results["maps"]["ms_per_step"]
    return {
        "workload": "train_step (minecraft, 3 frames x 2880 rays, perturb, train-mode BatchNorm) + a stand-in decoder with DecoderV6's "
                    "shipped layer shapes (1.3 M parameters, 64 @ 48x48 + 128 @ 24x24 -> 3 x 192x192 per frame), MSE against seeded "
                    "images, Adam on both parameter sets - BASELINE.json configs[4] in shape (the reference's decoder weights and "
                    "perceptual losses are not part of the path)",
        "value": round(2880 * 3 * world / (combined * 1e-3) / 1e6, 4), "unit": "Mrays/s trained, decoder included",
        "maps_route": results["maps"], "ray_major_route": results["ray_major"],
        "route_note": "maps = decoder_features written channels-first per stride by the compositing kernel (and their gradient read by "
                      "its backward); ray_major = integrated_features + wire_format's fold / split / permute as in the reference's glue",
        "renderer_only_ms": round(renderer_only_ms, 3), "decoder_only_ms": round(decoder_ms, 3),
        "renderer_share": round(renderer_only_ms / combined, 3),
        "overlap_ms": round(renderer_only_ms + decoder_ms - combined, 3),
        "overlap_note": "renderer_only + decoder_only - combined (positive = hidden time): the decoder runs on a side stream, but the step "
                        "is one dependency chain (renderer forward -> decoder forward -> decoder backward -> renderer backward), so only "
                        "the decoder's optimiser update and host enqueue time can hide behind the renderer's backward",
        "decoder_parameters": sum(p.numel() for p in decoder_params),
    }


def minecraft_leg(dev, lib, frames=20, balance=False):
    """BASELINE.json configs[2]: the shipped minecraft renderer (background P=16, skybox P=1, two players P=32 that share one
    model, static / dynamic overlap fix), one 256x256 frame, evaluation - both precisions, with the MLP / compositing share."""
    from playableenvironments_amd import configs, synthetic
    from playableenvironments_amd.environment_model import EnvironmentModel
    cfg = configs.minecraft_config()
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.eval().to(dev)
    model.frame_replay = None          # (the library's per-launch timers below need the launches issued by the call)
    size = (256, 256)
    scene = to_device(synthetic.minecraft_scene(seed=1234, image_size=size), dev)
    out = {"workload": "shipped minecraft renderer, 256x256 frame, 4 objects, 16 + 1 + 32 + 32 samples/ray, overlap fix, eval - "
                       "BASELINE.json configs[2]"}
    for precision in ("fp32", "f16x3", "f16"):      # ("f16": the throughput tier, not a parity configuration)
        model.object_composer.precision = precision

        def step():
            with torch.no_grad():
                return model(*scene_args(scene, size), 0, False, mode="scene_encodings")
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / frames
        lib.pr_profile_enable(1)
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        lib.pr_profile_enable(0)
        ms, _ = profile_arrays()
        lib.pr_profile_collect(ms, _)
        out[precision] = {"ms_per_frame": round(dt * 1e3, 3), "frames_per_s": round(1.0 / dt, 1),
                          "mrays_per_s": round(size[0] * size[1] / dt / 1e6, 3), "mlp_ms": round(ms[0] / 5, 3),
                          "composite_ms": round(ms[1] / 5, 3)}
        if precision == "fp32":
            out["roofline"] = leg_roofline(model.object_composer, cfg, composer_call_inputs(model, cfg, scene, size), ms[0] / 5)
    model.object_composer.precision = "fp32"
    if balance:
        out["shard_balance"] = shard_balance_leg(model, cfg, scene, size)
    return out


def native_eval_frame_leg(dev, lib, frames=200):
    """The frame the reference's evaluators and play loop actually render: 288 x 512, the strided grids [4, 8] of
    the autoencoder subclasses (72 x 128 + 36 x 64 = 11 520 rays; environment_model_backpropagated_autoencoder.py:173-236,
    evaluation/reconstructed_dataset_creator.py:121, model/playable_environment_model.py:281-285), shipped tennis and minecraft
    renderers, through the PLAIN drop-in calls - ``forward_from_scene_encoding(..., 0, False, 1200, patch_stride=[4, 8])`` and
    ``forward_from_observations`` (this package's encoders in front) - eager and as a replayed ``FrameGraph``, fp32 and f16x3,
    with and without a DecoderV6-shaped stand-in behind ``decoder_features``.  Per entry: wall ms per frame of back-to-back frames
    (what a dataset evaluator sees), the host's issue time per frame, the device time of ONE frame with an idle queue (what the
    play loop sees: render, show, wait for input), the HIP-event time of the MLP / compositing launches and the MLP roofline."""
    from playableenvironments_amd import configs, synthetic
    from playableenvironments_amd.environment_model import EnvironmentModel
    from playableenvironments_amd.frame_graph import FrameGraph, OBSERVATION_KEYS
    size = (288, 512)
    strides = [4, 8]
    out = {"workload": "288x512 frame, strided grids [4, 8] = 11 520 rays, shipped renderers, eval - the frame of the reference's "
                       "evaluators / play loop (SURVEY.md C3)",
           "rays": sum((size[0] // s) * (size[1] // s) for s in strides), "frames_timed": frames}

    def timed(fn, n=frames, warmup=5):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        single = []
        for _ in range(15):
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize(dev)
            single.append((e0.elapsed_time(e1), (time.perf_counter() - h0) * 1e3))
        return {"ms_per_frame": round((t2 - t0) / n * 1e3, 3), "host_issue_ms": round((t1 - t0) / n * 1e3, 3),
                "one_frame_device_ms": round(median([a for a, _ in single]), 3),
                "one_frame_latency_ms": round(median([b for _, b in single]), 3)}

    for world in ("tennis", "minecraft"):
        cfg = (configs.tennis_config if world == "tennis" else configs.minecraft_config)(encoders=True)
        torch.manual_seed(0)
        model = EnvironmentModel(cfg)
        synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
        model.eval().to(dev)
        comp = model.object_composer
        default_replay = model.frame_replay       # what a caller that changes nothing gets ("clone": recorded on the 2nd call, replayed)
        make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
        scene = to_device(make(seed=1234, image_size=size), dev)
        batch = to_device(synthetic.observation_batch(make(seed=1234, image_size=size)), dev)
        decoder = StandInDecoder().to(dev).eval()
        entry = {"frame_replay_default": default_replay}
        for precision in ("fp32", "f16x3", "f16"):
            comp.precision = precision
            model.frame_replay = None             # the *_eager entries: every call issues its launches

            def eager():
                with torch.no_grad():
                    return model.forward_from_scene_encoding(*scene_args(scene, size), 0, False, 1200, patch_stride=strides)

            def eager_decoder():
                with torch.no_grad():
                    res = model.forward_from_scene_encoding(*scene_args(scene, size), 0, False, 1200, patch_stride=strides,
                                                            _decoder_features=[64, 128])
                    maps = res["coarse"]["global"]["decoder_features"]
                    return decoder([m.reshape([-1] + list(m.shape[-3:])) for m in maps])

            def eager_observations():
                with torch.no_grad():
                    return model.forward_from_observations(*[batch[k] for k in OBSERVATION_KEYS], 0, False, 1200, patch_stride=strides)

            cur = {"scene_encoding_eager": timed(eager)}
            lib.pr_profile_enable(1)
            for _ in range(10):
                eager()
            torch.cuda.synchronize(dev)
            lib.pr_profile_enable(0)
            ms, launches = profile_arrays()
            lib.pr_profile_collect(ms, launches)
            cur["scene_encoding_eager"].update(mlp_ms=round(ms[0] / 10, 4), composite_ms=round(ms[1] / 10, 4))
            if precision == "fp32":
                with torch.no_grad():
                    inputs = composer_call_inputs_strided(model, cfg, scene, size, strides)
                cur["roofline"] = leg_roofline(comp, cfg, inputs, ms[0] / 10)
            graph = FrameGraph(model, scene, size, patch_stride=strides)
            cur["scene_encoding_frame_graph"] = timed(lambda: graph.render(scene))
            del graph
            if precision == "f16":       # the throughput tier (not a parity configuration): the renderer's own entries only
                entry[precision] = cur
                continue
            # the SAME plain calls with the model as constructed (EnvironmentModel.frame_replay's default): what the unchanged
            # evaluator / play loop pays per frame
            model.frame_replay = default_replay
            cur["scene_encoding_default"] = timed(eager)
            cur["observations_default"] = timed(eager_observations, n=max(20, frames // 4))
            model.frame_replay = "alias"       # ... and without the copy-out (static result tensors)
            cur["scene_encoding_auto_replay"] = timed(eager)
            cur["observations_auto_replay"] = timed(eager_observations, n=max(20, frames // 4))
            model.frame_replay = None
            model._replays.clear()
            cur["scene_encoding_eager_with_decoder"] = timed(eager_decoder)
            cur["observations_eager"] = timed(eager_observations, n=max(20, frames // 4))
            graph = FrameGraph(model, batch, mode="observations", patch_stride=strides)
            cur["observations_frame_graph"] = timed(lambda: graph.render(batch), n=max(20, frames // 4))
            del graph
            entry[precision] = cur
        comp.precision = "fp32"
        out[world] = entry
        del model, decoder
    out["note"] = ("ms_per_frame = wall time of back-to-back frames / frames (no synchronisation in the loop: the evaluator's throughput); "
                   "host_issue_ms = the Python + launch time of a call; one_frame_device_ms / one_frame_latency_ms = first launch to last "
                   "launch / call to completion of ONE frame on an idle queue (the play loop's latency: host-paced for eager calls); "
                   "*_frame_graph = the same frame replayed from a captured HIP graph (FrameGraph, bit-identical results); "
                   "*_default = the unchanged plain call on the model as constructed (frame_replay = 'clone': recorded on the second call "
                   "of a shape, replayed, results copied out); *_auto_replay = the same with frame_replay = 'alias' (no copy-out); "
                   "*_eager = frame_replay = None; "
                   "with_decoder = + a DecoderV6-shaped stand-in (bench.StandInDecoder) consuming decoder_features; observations_* = "
                   "forward_from_observations with this package's CNN encoders in front (PyTorch-ROCm / MIOpen + pr_roi_pool)")
    return out


def composer_call_inputs_strided(model, cfg, scene_dev, size, strides):
    """composer_call_inputs for the strided grids of a frame."""
    from playableenvironments_amd.environment_model import camera_rays, euler_to_matrix, strided_grid_pixels
    rows, cols = strided_grid_pixels(size[0], size[1], strides)
    c2w = euler_to_matrix(scene_dev["camera_rotations"], scene_dev["camera_translations"])
    o, d, n = camera_rays(c2w, scene_dev["focals"] * cfg["data"]["focal_length_multiplier"], size[0], size[1], rows, cols)
    w2o, _ = model.compute_transformation_matrix_w2o_o2w(scene_dev["object_rotation_parameters"],
                                                         scene_dev["object_translation_parameters"])
    return [o, d, n, w2o, scene_dev["object_style"].unsqueeze(-3), scene_dev["object_deformation"].unsqueeze(-3),
            scene_dev["object_in_scene"].unsqueeze(-2)]


def distinct_frames_leg(model, cfg, label, size, dev, world, rank, dist, steps, frames=8, lib=None):
    """BASELINE.json configs[3]: a batch of 8 DISTINCT seeded frames sharded over the ranks with
    EnvironmentModel.render_sharded (parallel.shard_frames), the rendered feature maps gathered with one collective.
    Unlike the headline's identical frames, distinct frames carry different amounts of work (in-box samples): the batch
    time follows the heaviest shard.  Reports the whole-batch rate, every rank's render time and the max over ranks."""
    from playableenvironments_amd import synthetic
    scenes = [synthetic.tennis_scene(seed=4000 + i, image_size=size) for i in range(frames)]
    batch = {k: torch.cat([s[k] for s in scenes], dim=0).to(dev) for k in SCENE_KEYS}
    ty = "fine" if "hierarchical" in label else "coarse"

    def run():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        # the evaluation flow: rank 0 is the consumer of the rendered maps (decoder / writer) - one gather to it, not an all-gather
        out = model.render_sharded(*scene_args(batch, size), False, shard="frames" if frames >= world else "tiles",
                                   fields=("integrated_features",), dst=0)
        e1.record()
        return out, e0, e1

    run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    local_ms = 0.0
    for _ in range(steps):
        out, e0, e1 = run()
        torch.cuda.synchronize()
        local_ms += e0.elapsed_time(e1)
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist, dev)
    per_rank = [local_ms / steps]
    if dist is not None:
        t = torch.tensor(per_rank, dtype=torch.float64, device=dev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank = [float(p.item()) for p in parts]
    feats = out[ty]["global"]["integrated_features"] if out is not None else None      # (rank 0 holds the assembled maps)
    rays = frames * size[0] * size[1]
    roofline = None
    if world == 1 and lib is not None:
        lib.pr_profile_enable(1)
        run()
        torch.cuda.synchronize()
        lib.pr_profile_enable(0)
        kernel_ms, _ = profile_arrays()
        lib.pr_profile_collect(kernel_ms, _)
        roofline = leg_roofline(model.object_composer, cfg, composer_call_inputs(model, cfg, batch, size), kernel_ms[0])
    return {
        "roofline": roofline,
        "workload": f"{frames} distinct tennis frames ({label}), {size[0]}x{size[1]}, sharded by frame over {world} rank(s), "
                    "feature maps gathered on rank 0 (gather(dst=0): the evaluator's consumer)",
        "value": round(rays * steps / dt / 1e6, 4),
        "unit": "Mrays/s (whole batch, strong scaling over a fixed batch)",
        "frames_per_s": round(frames * steps / dt, 3),
        "ms_per_batch": round(dt / steps * 1e3, 3),
        "per_rank_ms": [round(v, 3) for v in per_rank],
        "max_rank_ms": round(max(per_rank), 3),
        "gathered_shape": list(feats.shape) if feats is not None else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--image", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "f16x3", "f16"], default="fp32",
                    help="fp32 = exact fp32 MFMA (default, the headline); f16x3 = fp32 emulated with three fp16 MFMAs; "
                         "f16 = plain fp16 operands (throughput tier, not a parity configuration)")
    ap.add_argument("--no-split-precision", action="store_true", help="skip the secondary f16x3 measurement")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--no-minecraft", action="store_true", help="skip the secondary configs[2] (minecraft) measurement")
    ap.add_argument("--no-native-frame", action="store_true", help="skip the native evaluation frame legs (288x512, strides [4, 8])")
    ap.add_argument("--only", default=None, help="run ONE secondary leg alone and print its JSON (native_eval_frame | train_step)")
    ap.add_argument("--no-distinct-frames", action="store_true", help="skip the 8-distinct-frames legs (configs[3])")
    ap.add_argument("--no-reference-graph", action="store_true", help="skip the same-GPU PyTorch op graph measurement")
    ap.add_argument("--no-gate", action="store_true", help="disable the sigma-gated feature head (measurement)")
    ap.add_argument("--cpu-rays", type=int, default=32, help="the 16-thread CPU baseline renders a cpu_rays x cpu_rays pixel grid "
                                                             "(2 warm-ups + median of 5)")
    ap.add_argument("--cpu-full-size", action="store_true",
                    help="also run the CPU oracle ONCE on every ray of the frame (~190 s of host time; recorded at 0.000348 Mrays/s in "
                         "profiles/r04_bench.json).  Off by default: the cpu_baseline sample is bounded so that the default run takes ~1.5 min")
    ap.add_argument("--no-cpu-full-size", action="store_true", help=argparse.SUPPRESS)     # (accepted: the old default-on switch)
    ap.add_argument("--no-telemetry", action="store_true", help="do not sample shader clock / package power during the timed regions")
    ap.add_argument("--no-shard-balance", action="store_true", help="skip the virtual-shard balance measurement (N = 1 only)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads of the main CPU baseline run (all 256 host cores are >50x SLOWER on these small ops)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the FULL record (every leg, note and table) is written; stdout carries one compact line < 4 KB")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # self-launch: one process per GPU under torch.distributed.run, rendezvous on the loopback interface
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the renderer)")
    # PR_BENCH_DEVICE / PR_BENCH_BACKEND: test knobs (several ranks on one GPU over gloo exercise the multi-rank path
    # where only one device exists); the defaults are one rank per GPU over RCCL
    if "PR_BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    device_index = int(os.environ.get("PR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    backend = None
    # PR_BENCH_FORCE_DIST: initialise the process group (and run every collective of the multi-rank legs) with ONE rank too -
    # how the RCCL code path of `--gpus N` is exercised on a one-GPU box
    multi = world > 1 or bool(os.environ.get("PR_BENCH_FORCE_DIST"))
    if multi:
        import torch.distributed as dist
        # (NCCL_DEBUG=VERSION makes RCCL print a version banner on STDOUT, which has to stay the one JSON line: the version is
        # reported through torch.cuda.nccl.version() below; with NCCL_DEBUG_FILE set by the caller its lines are copied as well)
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PR_BENCH_BACKEND", "nccl")
        # the communication libraries announce themselves on the process's STDOUT from C (RCCL: a version / host banner when the
        # first communicator is created; gloo: one "connected to N peer ranks" line per rank) - through C stdio, i.e. flushed when
        # the process exits, behind the JSON line.  Stdout has to stay that one line: the file descriptor points at stderr while the
        # group and its first communicator are created, and C's buffers are flushed before it is restored.
        with _c_stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            probe = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev)

    from playableenvironments_amd import configs, synthetic, _lib
    from playableenvironments_amd.environment_model import EnvironmentModel

    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    model = EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    model.eval().to(dev)
    # every launch of the timed steps is issued by the call itself (the library's HIP-event timers behind `roofline` sit on the
    # launches; a replayed recording issues none) - at 206 ms of device work per step the host's 3 ms are hidden anyway.  The
    # default ("clone": record + replay) is measured where it matters, on the small native frames (native_eval_frame *_default).
    model.frame_replay = None
    comp = model.object_composer
    comp.precision = args.precision
    comp.gate_feature_head = not args.no_gate
    size = (args.image, args.image)
    # every rank renders ITS OWN frame (seed 1234 + rank; rank 0's is the N = 1 frame): weak scaling over distinct frames, whose
    # in-box sample counts differ - the step time follows the heaviest.  The identical-frame run (every rank the seed-1234
    # frame: exactly the same work per GPU) is the labelled secondary "identical_frames".
    scene = synthetic.tennis_scene(seed=1234 + rank, image_size=size)
    scene_dev = to_device(scene, dev)
    active = {"scene": scene_dev}

    def step():
        with torch.no_grad():
            out = model(*scene_args(active["scene"], size), 0, False, mode="scene_encodings")
        feats = out["fine"]["global"]["integrated_features"]
        if multi:
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
    if args.only:
        if args.only == "native_eval_frame":
            print(json.dumps({"native_eval_frame": native_eval_frame_leg(dev, lib)}))
        elif args.only == "train_step":
            print(json.dumps({"train_step": train_step_leg(args, dev, world, rank, dist, lib)}))
        else:
            raise SystemExit(f"--only {args.only}: unknown leg")
        return

    def timed(steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between barrier + synchronize; max over ranks."""
        for _ in range(warmup):
            step()
        drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        lib.pr_profile_enable(1)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            step()
        marks[steps].record()
        drain()                      # every gather of the timed steps completes inside the timed region
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
        lib.pr_profile_enable(0)
        kernel_ms, kernel_launches = profile_arrays()
        _lib.check(lib.pr_profile_collect(kernel_ms, kernel_launches), "pr_profile_collect")
        dt = max_over_ranks(dt, dist, dev)
        timed.gaps = event_gaps_ms(marks)
        return dt, kernel_ms, kernel_launches

    # shader clock / package power during the timed regions (amdgpu sysfs, sampled by a thread of this process; one GPU only - the node
    # is found by the power rise of a matrix probe): says whether a measured region ran clock limited.  Never fails the run.
    telemetry = None
    if world == 1 and not args.no_telemetry:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gpu_telemetry

            def _load():
                tf, pms = C.c_double(), C.c_double()
                lib.pr_probe_mfma_f32(20000, 1, C.byref(tf), C.byref(pms), None)
            props = torch.cuda.get_device_properties(dev)
            card = gpu_telemetry.find_card(_load, pci=(props.pci_domain_id, props.pci_bus_id, props.pci_device_id))
            if card is not None:
                telemetry = gpu_telemetry.Telemetry(card)
                telemetry.start()
        except Exception as error:      # (no sysfs in a container, ...: the line simply carries no clock block)
            print(f"bench.py: no clock / power telemetry ({type(error).__name__}: {error})", file=sys.stderr)
            telemetry = None

    def watched(label, fn):
        if telemetry is None:
            return fn()
        telemetry.label = label
        try:
            return fn()
        finally:
            telemetry.label = None

    elapsed, ms, launches = watched("headline", lambda: timed(args.steps, args.warmup))
    step_gaps = timed.gaps
    identical = None
    if multi:
        active["scene"] = to_device(synthetic.tennis_scene(seed=1234, image_size=size), dev)
        same_s, _, _ = timed(max(1, min(args.steps, 5)), 1)
        active["scene"] = scene_dev
        identical = same_s / max(1, min(args.steps, 5))
    split = half = None
    if args.precision == "fp32" and not args.no_split_precision:
        # secondary measurement, never the headline: the same step with the MLP on the split-precision
        # kernel (fp32 emulated with three fp16 MFMAs; same parity tolerance in tests/test_gpu.py)
        comp.precision = "f16x3"
        split_s, split_ms, _ = watched("f16x3", lambda: timed(args.steps, max(1, args.warmup)))
        # ... and on the single-product fp16 tier (throughput configuration, ~1e-3 relative error: not a parity configuration)
        comp.precision = "f16"
        half_s, half_ms, _ = watched("f16", lambda: timed(args.steps, max(1, args.warmup)))
        comp.precision = "fp32"
        split = (split_s, split_ms[0] / max(1, args.steps))
        half = (half_s, half_ms[0] / max(1, args.steps))

    # FLOPs of the MLP launches of one step (this rank's frame)
    call_inputs = composer_call_inputs(model, cfg, scene_dev, size)
    flops, executed, evaluated, head_samples = flop_counts(comp, cfg, call_inputs)

    # HBM traffic of the dominant kernel from the committed PMC pass (collected separately: counters
    # cannot ride along with the timed run), and the fp32 MFMA rate this box sustains
    traffic = None
    traffic_source = traffic_stamp = None
    sha = library_sha256()
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc_path) and size == (256, 256):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if "k_mlp_mfma" not in pmc:        # (a summary of failed counter passes: keep looking)
                continue
            traffic = pmc["k_mlp_mfma"]["hbm_bytes_per_launch_avg"]
            traffic_stamp = pmc.get("library_sha256")
            traffic_source = "profiles/" + name
            break
    probe = {}
    for name, rnd in (("constant_operands", 0), ("random_operands", 1)):
        tf, pms = C.c_double(), C.c_double()
        _lib.check(lib.pr_probe_mfma_f32(100000, rnd, C.byref(tf), C.byref(pms), None), "pr_probe_mfma_f32")
        probe[name] = round(tf.value, 1)

    distributed = {"world_size": (dist.get_world_size() if multi else 1), "backend": backend,
                   "launched_by": "torch.distributed.run" if world > 1 else ("single process, process group forced" if multi else "single process")}
    if multi:
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
        if backend == "nccl":
            try:
                distributed["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())     # (= RCCL on ROCm)
            except Exception as e:     # (a build without the query: the backend string above still says what ran)
                distributed["nccl_version"] = f"unavailable ({type(e).__name__})"
            distributed["devices"] = torch.cuda.device_count()
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cuda:{device_index} {torch.cuda.get_device_properties(dev).name}")
        distributed["rank_devices"] = names
        debug_file = os.environ.get("NCCL_DEBUG_FILE")
        if debug_file and os.path.exists(debug_file.replace("%h", socket.gethostname()).replace("%p", str(os.getpid()))):
            with open(debug_file.replace("%h", socket.gethostname()).replace("%p", str(os.getpid()))) as f:
                distributed["nccl_debug_version"] = [line.strip() for line in f if "version" in line.lower()][:4]
    rays_per_gpu = size[0] * size[1]
    total_rays = rays_per_gpu * world * args.steps
    value = total_rays / elapsed / 1e6
    mlp_ms = ms[0] / max(1, args.steps)
    achieved = executed / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    result = {
        "metric": "Mrays/s",
        "value": round(value, 4),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "ms_per_step_median": round(median(step_gaps), 3),
        "ms_per_step_note": f"ms_per_step (and value) = wall time of the {args.steps} timed steps between barrier + synchronize, max over "
                            f"ranks, / {args.steps}; median = of the device time between the steps' first launches (rank 0)",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"tennis renderer, {size[0]}x{size[1]} frame per GPU (rank r renders the frame of seed 1234 + r), 4 objects, "
                        "64+128 hierarchical samples/ray (coarse+fine networks), eval - BASELINE.json configs[1]",
            "rays_per_gpu": rays_per_gpu,
            "frames_per_gpu": 1,
            "parallelism": f"frame shard x{world}" + (" + RCCL all_gather of feature maps" if world > 1 else ""),
        },
        "frames_per_s_256x256": round(value * 1e6 / 65536.0, 3),
        "library_sha256": sha,
        "distributed": distributed,
        "roofline": {
            "bound": "mfma",
            "kernel": "k_mlp_mfma_group (fused fp32 MFMA MLP; one launch per model type evaluates its four objects; all launches of one step)",
            "achieved": round(achieved, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "traffic_unit": f"HBM bytes per launch (average over the launches of a step), rocprofv3 PMC passes of tools/collect_pmc.sh, {traffic_source}",
            "traffic_library_sha256": traffic_stamp,
            "traffic_from_this_library": bool(traffic_stamp == sha) if traffic is not None else None,
            "peak_measured": probe,
            "flop_per_step": executed,
            "flop_per_step_algorithmic": flops,
            "algorithmic_tflops": round(flops / (mlp_ms * 1e-3) / 1e12, 2) if mlp_ms > 0 else None,
            "flop_note": "achieved = EXECUTED matmul FLOPs / HIP-event time of the MLP launches; executed = algorithmic (every in-box "
                         "sample through every layer, what the reference computes) minus the three feature-head products of the "
                         "samples with density <= 0, whose compositing weight is exactly 0 (sigma-gated head, bit-identical results)",
            "mlp_ms_per_step": round(mlp_ms, 3),
            "mlp_launches_per_step": int(launches[0] / max(1, args.steps)),
            "composite_ms_per_step": round(ms[1] / max(1, args.steps), 3),
            "evaluated_samples": evaluated,
            "feature_head_samples": head_samples,
            "sigma_gate": bool(comp.gate_feature_head),
        },
    }

    clocks = {}
    if telemetry is not None:
        telemetry.finish()
        clocks = {label: telemetry.summary(label) for label in ("headline", "f16x3", "f16")}
        if clocks["headline"].get("samples"):
            result["roofline"]["clock"] = {k: clocks["headline"].get(k) for k in ("sclk_mhz", "power_w", "power_cap_w") if k in clocks["headline"]}
            result["roofline"]["clock_note"] = ("mean shader clock / package power over the headline's warm-up + timed steps (amdgpu sysfs, 20 ms samples, "
                                                "tools/gpu_telemetry.py); nominal 2 400 MHz: the fp32 kernel is not clock limited, the fp16 tiers are "
                                                "(DESIGN.md 11.3)")
    if args.precision == "f16x3":
        result["dtype"] = "f16x3 (fp32 emulated with three fp16 MFMAs, fp32 accumulate)"
        result["roofline"]["kernel"] = "k_mlp_split (fused split-precision MFMA MLP); achieved/peak are in fp32-equivalent FLOPs"
    if args.precision == "f16":
        result["dtype"] = "f16 (fp16 operands, fp32 accumulate: throughput tier, ~1e-3 relative error)"
        result["roofline"]["kernel"] = "k_mlp_f16 (fused fp16 MFMA MLP); achieved is in executed FLOPs, peak is the fp32 pipe's"
    if split is not None:
        result["split_precision"] = {
            "value": round(rays_per_gpu * world * args.steps / split[0] / 1e6, 4),
            "unit": "Mrays/s",
            "ms_per_step": round(split[0] / args.steps * 1e3, 3),
            "mlp_ms_per_step": round(split[1], 3),
            "executed_tflops": round(executed / (split[1] * 1e-3) / 1e12, 2) if split[1] > 0 else None,
            "clock": {k: clocks["f16x3"].get(k) for k in ("sclk_mhz", "power_w") if k in clocks.get("f16x3", {})} or None,
            "note": "same workload with ObjectComposer.precision='f16x3' (k_mlp_split): every fp32 product as three fp16 "
                    "MFMAs, ~22-bit operands, fp32 accumulation; passes the same oracle/golden parity tolerance; "
                    "reported beside the exact-fp32 headline, not as it",
        }
    if half is not None:
        result["half_precision"] = {
            "value": round(rays_per_gpu * world * args.steps / half[0] / 1e6, 4),
            "unit": "Mrays/s",
            "ms_per_step": round(half[0] / args.steps * 1e3, 3),
            "mlp_ms_per_step": round(half[1], 3),
            "executed_tflops": round(executed / (half[1] * 1e-3) / 1e12, 2) if half[1] > 0 else None,
            "note": "same workload with ObjectComposer.precision='f16' (k_mlp_f16: the split kernel's hi x hi product only - plain fp16 "
                    "operands, fp32 accumulation, one MFMA per step): the throughput tier for interactive play; ~1e-3 relative error on "
                    "the rendered features (tests/test_gpu.py: >= 40 dB PSNR against the oracle), NOT a parity configuration and never "
                    "the headline",
        }
    if identical is not None:
        result["identical_frames"] = {
            "value": round(rays_per_gpu * world / identical / 1e6, 4), "unit": "Mrays/s", "ms_per_step": round(identical * 1e3, 3),
            "note": "secondary: every rank renders the SAME frame (seed 1234) - exactly the same work per GPU; the headline's ranks "
                    "render distinct frames"}
    final_feats = None
    if multi:
        final_feats = step()["fine"]["global"]["integrated_features"]
        drain()
    result["feature_gather"] = feature_gather_leg(final_feats, dist, world, rank, dev)
    del final_feats
    if world == 1 and not args.no_shard_balance:
        result["shard_balance"] = {
            "what": "the 8 virtual shards of ONE 256x256 frame rendered one after the other on this GPU (composer calls), contiguous "
                    "ranges of the pixel list vs 8x8 tiles dealt round robin (render_sharded shard='rays' / 'tiles'); a sharded frame "
                    "takes as long as its slowest shard",
            "tennis_hierarchical_64_128": shard_balance_leg(model, cfg, scene_dev, size),
        }
    if not args.no_distinct_frames:
        steps_d = max(1, min(args.steps, 3))
        result["distinct_frames"] = {
            "hierarchical_64_128": distinct_frames_leg(model, cfg, "hierarchical 64+128, the headline networks", size, dev, world,
                                                       rank, dist, steps_d, lib=lib),
        }
        shipped_cfg = configs.tennis_config()
        torch.manual_seed(0)
        shipped = EnvironmentModel(shipped_cfg)
        synthetic.randomize_module_state(shipped.object_composer, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
        shipped.eval().to(dev)
        shipped.frame_replay = None
        result["distinct_frames"]["shipped_p72"] = distinct_frames_leg(shipped, shipped_cfg, "shipped 4+4+32+32 positions - BASELINE.json "
                                                                       "configs[3]", size, dev, world, rank, dist, max(steps_d, 3), lib=lib)
        if "shard_balance" in result:
            result["shard_balance"]["tennis_shipped"] = shard_balance_leg(
                shipped, shipped_cfg, to_device(synthetic.tennis_scene(seed=1234, image_size=size), dev), size)
        del shipped
    if not args.no_train_step:
        result["train_step"] = train_step_leg(args, dev, world, rank, dist, lib)
        # the same step with precision="f16x3" (opt-in split precision: fp16-pair products in the training forward's phase 1, the
        # backward chains and every weight gradient; fp32 accumulation, gradients at fp32 round-off) - beside the fp32 figure, never as it
        split_train = train_step_leg(args, dev, world, rank, dist, lib, precision="f16x3")
        result["train_step"]["f16x3"] = {
            "ms_per_step": split_train["ms_per_step"], "ms_per_step_median": split_train["ms_per_step_median"], "value": split_train["value"],
            "unit": split_train["unit"], "kernel_ms_per_step": split_train["roofline"]["kernel_ms_per_step"],
            "note": "ObjectComposer.precision='f16x3' on a training call (PR_FLAG_SPLIT_BACKWARD): phase 1 of the forward and the backward "
                    "chains on fp16 pairs (x = hi + lo, three v_mfma_f32_32x32x16_f16 per product, weights packed as w x 2^8, activation and gradient tiles "
                    "scaled by a power of two per tile: k_mlp_mfma_train_group_split, k_chain_bwd_group_f16), the weight gradients on fp16 "
                    "pairs of 16-row half slabs scaled by powers of two (k_gemm_tn_all_f16), fp32 accumulation everywhere; the head "
                    "phases stay fp32.  Same gradient tests as fp32 "
                    "(reference fixtures at 1e-4, oracle autograd, float64 arbitration at shipped sizes)"}
        if world == 1:      # (a comparison leg: not repeated on every GPU count of a scaling run)
            separate = train_step_leg(args, dev, world, rank, dist, lib, arena=False)
            result["train_step"]["separate_parameter_tensors"] = {
                "ms_per_step": separate["ms_per_step"], "ms_per_step_median": separate["ms_per_step_median"],
                "note": "the same step with torch's fused Adam on the separate parameter tensors (a multi-tensor sweep: ~0.25 ms of 38-workgroup launches)"}
        result["train_step_with_decoder"] = train_step_with_decoder_leg(args, dev, world, rank, dist, result["train_step"]["ms_per_step"])
    if rank == 0 and world == 1 and not args.no_native_frame:
        result["native_eval_frame"] = native_eval_frame_leg(dev, lib)
    if rank == 0 and world == 1 and not args.no_minecraft:
        result["config2_minecraft_256"] = minecraft_leg(dev, lib, balance=not args.no_shard_balance)
        if "shard_balance" in result and "shard_balance" in result["config2_minecraft_256"]:
            result["shard_balance"]["minecraft_shipped"] = result["config2_minecraft_256"].pop("shard_balance")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result.update(baseline_legs(args, cfg, comp, scene, size, dev, value))
    graph_rate = _dig(result, "reference_graph_on_gpu", "value")
    if graph_rate:
        # north_star: ">= 10 x the reference PyTorch renderer's Mrays/s on one MI355X at matched PSNR" - every tier against the FASTEST chunk
        # size of the reference's op graph on this GPU, with the tier's PSNR against the oracle beside it (80 dB = the formula's ceiling)
        tiers = {}
        for tier, rate, key in (("fp32", result["value"], "fp32"), ("f16x3", _dig(result, "split_precision", "value"), "f16x3"),
                                ("f16", _dig(result, "half_precision", "value"), "f16")):
            if rate:
                tiers[tier] = {"over_reference_graph": round(rate / graph_rate, 2), "psnr_db": _dig(result, "psnr_db", key)}
        result["reference_graph_on_gpu"]["tiers"] = tiers
    if rank == 0:
        emit(result, args.full_json)
    if multi:
        dist.destroy_process_group()


def usable_cores() -> int:
    """CPUs this process can actually run on: os.cpu_count() capped by the scheduler affinity and the cgroup CPU quota (a
    container on a 256-thread host may own far fewer; oversubscribing them with 256 spinning OpenMP threads is pathological)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    n = min(n, max(1, int(int(fields[0]) / int(fields[1]))))
            else:
                quota = int(fields[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(quota / int(f.read().split()[0]))))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def baseline_legs(args, cfg, comp, scene, size, dev, gpu_mrays):
    """Everything that runs the ORACLE (test infrastructure) as a yardstick, on rank 0 at N=1 only: the CPU baseline on
    the host cores, the same PyTorch op graph executed by PyTorch-ROCm on this GPU (what north_star's ">= 10x the
    reference PyTorch renderer on one MI355X" compares against), and the PSNR of the HIP result against it."""
    from oracle import render_oracle as ro
    from playableenvironments_amd import configs, synthetic, ObjectComposer, _lib
    from tests.helpers import composer_inputs, grid_pixels
    out = {}
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    n_side = args.cpu_rays
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(size[0], size[1], n_side))

    # ---- CPU baseline: main run = `cpu_threads` threads on the n_side^2 subset, 2 warm-ups + median of 5; 1 thread and all usable
    # cores once on a smaller subset; one FULL-SIZE run (every ray of the frame) with the main run's threads
    host_cores = os.cpu_count() or 1
    usable = usable_cores()
    main_threads = max(1, min(args.cpu_threads, usable))

    def cpu_run(threads, sub):
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            res = ro.batchified_composer_call(cfg, sd, *sub, False, chunk=1000)
            return time.perf_counter() - t0, res

    for _ in range(2):
        cpu_run(main_threads, inputs)
    timings = []
    for _ in range(5):
        cpu_s, want = cpu_run(main_threads, inputs)
        timings.append(cpu_s)
    cpu_s = median(timings)
    runs = [{"threads": main_threads, "rays": n_side * n_side, "seconds": round(cpu_s, 3), "seconds_all": [round(t, 3) for t in timings],
             "seconds_mean": round(sum(timings) / len(timings), 3), "protocol": "2 warm-ups, median of 5",
             "value": round(n_side * n_side / cpu_s / 1e6, 7), "unit": "Mrays/s"}]
    small = max(8, n_side // 2)
    small_inputs = composer_inputs(cfg, scene, pixels=grid_pixels(size[0], size[1], small))
    for threads in (1, usable):
        cpu_run(threads, small_inputs)
        t, _ = cpu_run(threads, small_inputs)
        runs.append({"threads": threads, "rays": small * small, "seconds": round(t, 3), "protocol": "1 warm-up, 1 run",
                     "value": round(small * small / t / 1e6, 7), "unit": "Mrays/s"})
    full_size = None
    if args.cpu_full_size and not args.no_cpu_full_size:
        every = composer_inputs(cfg, scene, pixels=grid_pixels(size[0], size[1], size[0]))
        assert every[1].size(-2) == size[0] * size[1]
        t, _ = cpu_run(main_threads, every)
        full_size = {"rays": size[0] * size[1], "threads": main_threads, "seconds": round(t, 2),
                     "value": round(size[0] * size[1] / t / 1e6, 7), "unit": "Mrays/s",
                     "note": "ONE run of the whole frame of the headline workload (every ray, 1000-ray chunks), no extrapolation"}
        del every
    torch.set_num_threads(max(1, min(args.cpu_threads, host_cores)))
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    main_run = runs[0]
    out["cpu_baseline"] = {
        "value": main_run["value"],
        "unit": "Mrays/s",
        "cores": main_run["threads"],
        "kind": "port",
        "sample": f"{n_side}x{n_side} pixel grid ({n_side * n_side} rays) of the same frame and weights, "
                  f"oracle/render_oracle.py in 1000-ray chunks, 2 warm-ups then the median of 5 runs ({main_run['seconds']:.2f} s), "
                  f"{main_run['threads']} torch threads of {host_cores} host cores; `full_size` = one run of all "
                  f"{size[0] * size[1]} rays",
        "runs": runs,
        "full_size": full_size,
        "extrapolation": "linear in the number of rays (rays are independent; every run renders a uniform pixel grid of the same frame)",
        "cpu_model": cpu_model,
        "host_cores": host_cores,
        "usable_cores": usable,
        "usable_cores_note": "min(os.cpu_count(), scheduler affinity, cgroup cpu quota): the 'all cores' run uses this many threads",
        "torch": torch.__version__,
        "gpu_over_cpu": round(gpu_mrays / main_run["value"], 1) if main_run["value"] > 0 else None,
    }

    # ---- PSNR of the HIP renderer against the oracle on the same rays (evaluation/metrics/psnr.py:10-34, features
    # rescaled to [0, 1] with the oracle's min / max - SURVEY.md section 8d)
    a = want["fine"]["global"]["integrated_features"]
    lo, hi = float(a.min()), float(a.max())
    psnr = {}
    before = comp.precision
    for precision in ("fp32", "f16x3", "f16"):     # (f16: the throughput tier - NOT a parity configuration; its PSNR says what it costs)
        comp.precision = precision
        with torch.no_grad():
            got = comp(*[v.to(dev) for v in inputs], False)
        b = got["fine"]["global"]["integrated_features"].cpu()
        psnr[precision] = round(ro.psnr((a - lo) / (hi - lo), (b - lo) / (hi - lo)), 2)
        psnr[precision + "_max_abs_diff"] = float((a - b).abs().max())
    comp.precision = before
    out["psnr_db"] = {**psnr, "against": "CPU oracle (pinned bitwise to the reference), fine.global.integrated_features rescaled "
                      f"to [0, 1] on the {n_side * n_side}-ray subset; 80 dB is the formula's ceiling (its 1e-8 floor)"}

    # ---- the reference's PyTorch op graph on this GPU (PyTorch-ROCm): EVERY ray of the frame, at the reference's own chunk size
    # (1000 rays: an RTX 8000 memory limit, model/environment_model.py:584) and at the chunk sizes a 288 GB part allows - the honest
    # same-GPU baseline is the FASTEST of them
    if not args.no_reference_graph:
        ref_inputs = composer_inputs(cfg, scene)
        gin = [v.to(dev) for v in ref_inputs]
        gsd = {k: v.to(dev) for k, v in sd.items()}
        rays_total = size[0] * size[1]
        graph, spread, skipped = {}, {}, {}
        with torch.no_grad():       # warm-up: kernels of every shape class compiled / loaded
            ro.batchified_composer_call(cfg, gsd, *[v[..., :1000, :] if v.dim() == 5 and v.size(-2) > 1000 else v for v in gin], False, chunk=1000)
        for chunk in (1000, 4000, 16384, 65536):
            if chunk > rays_total:
                continue
            # materialised tensors of the op graph: ~12 live (chunk, 192 positions, 256) fp32 tensors per object pass + the merged
            # (chunk, 768, 192) feature gathers
            need = chunk * (12 * 192 * 256 * 4 + 3 * 768 * 192 * 4)
            free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            if need > 0.8 * free:
                skipped[f"chunk_{chunk}"] = f"estimated {need / 2**30:.0f} GiB of materialised tensors, {free / 2**30:.0f} GiB available"
                continue
            rates = []
            try:
                with torch.no_grad():
                    for i in range(6):                 # first run = warm-up of this chunk size's allocations
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        ro.batchified_composer_call(cfg, gsd, *gin, False, chunk=chunk)
                        torch.cuda.synchronize()
                        if i:
                            rates.append(rays_total / (time.perf_counter() - t0) / 1e6)
            except torch.OutOfMemoryError as e:
                skipped[f"chunk_{chunk}"] = "out of memory: " + str(e)[:120]
                torch.cuda.empty_cache()
                continue
            graph[f"chunk_{chunk}"] = round(median(rates), 5)
            spread[f"chunk_{chunk}"] = {"runs": [round(r, 5) for r in rates], "min": round(min(rates), 5), "max": round(max(rates), 5)}
            torch.cuda.empty_cache()
        del gin, gsd
        torch.cuda.empty_cache()
        fastest = max(graph, key=graph.get)
        out["reference_graph_on_gpu"] = {
            "value": graph[fastest], "unit": "Mrays/s", "fastest": fastest,
            "by_chunk": graph, "skipped": skipped,
            "protocol": "all rays of the frame; per chunk size 1 warm-up run, then the median of 5 runs", "spread": spread,
            "sample": f"the oracle's restatement of the reference's op graph (materialised per-sample tensors, boolean compaction, "
                      f"sort + gather compose) run by PyTorch-ROCm on this GPU on all {rays_total} rays of the same frame; chunk_1000 is "
                      "what render_full_frame_* uses (an RTX 8000 memory limit), the larger chunks are what this GPU's memory allows",
            "hip_over_reference_graph": round(gpu_mrays / graph[fastest], 2),
            "hip_over_reference_graph_note": "exact-fp32 headline / the FASTEST chunk size of the reference's graph on this GPU",
            "hip_over_reference_graph_by_chunk": {k: round(gpu_mrays / v, 2) for k, v in graph.items()},
        }

    # ---- BASELINE.json configs[0] at full size: one 128x128 frame, one player object, 32 samples per ray, every
    # ray inside the box - HIP renderer and CPU oracle on all 16 384 rays
    c1_cfg = configs.tennis_single_player_config()
    torch.manual_seed(0)
    c1 = ObjectComposer(c1_cfg)
    synthetic.randomize_module_state(c1, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    c1.eval()
    c1_inputs = composer_inputs(c1_cfg, synthetic.single_player_scene(image_size=(128, 128)))
    c1_sd = {k: v.detach().clone() for k, v in c1.state_dict().items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        c1_want = ro.batchified_composer_call(c1_cfg, c1_sd, *c1_inputs, False, chunk=1000)
        c1_cpu = time.perf_counter() - t0
        c1 = c1.to(dev)
        gin = [v.to(dev) for v in c1_inputs]
        c1(*gin, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            c1_got = c1(*gin, False)
        torch.cuda.synchronize()
        c1_gpu = (time.perf_counter() - t0) / reps
        lib = _lib.load()
        lib.pr_profile_enable(1)
        for _ in range(5):
            c1(*gin, False)
        torch.cuda.synchronize()
        lib.pr_profile_enable(0)
        kernel_ms, _counts = profile_arrays()
        lib.pr_profile_collect(kernel_ms, _counts)
        c1_roofline = leg_roofline(c1, c1_cfg, gin, kernel_ms[0] / 5)
    diff = float((c1_want["coarse"]["global"]["integrated_features"] - c1_got["coarse"]["global"]["integrated_features"].cpu()).abs().max())
    out["config0_single_player_128"] = {
        "workload": "BASELINE.json configs[0]: 128x128 frame, 1 object (player: NeRF + ray bender), 32 samples/ray, all in the box",
        "hip_mrays_per_s": round(16384 / c1_gpu / 1e6, 4), "hip_ms": round(c1_gpu * 1e3, 3),
        "cpu_oracle_mrays_per_s": round(16384 / c1_cpu / 1e6, 6), "cpu_seconds": round(c1_cpu, 2),
        "cpu_threads": torch.get_num_threads(), "max_abs_diff_features": diff,
        "roofline": c1_roofline,
    }
    return out


if __name__ == "__main__":
    main()
