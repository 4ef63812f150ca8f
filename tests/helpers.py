"""Shared helpers of the test-suite (tests may use the oracle; the product never does)."""
import contextlib

import torch

from oracle import render_oracle as ro


def grid_pixels(h, w, n):
    r = torch.linspace(0, h - 1, n).long()
    c = torch.linspace(0, w - 1, n).long()
    rr, cc = torch.meshgrid(r, c, indexing="ij")
    return rr.reshape(-1), cc.reshape(-1)


def composer_inputs(config, scene, strides=None, pixels=None):
    """Scene encoding -> the seven tensors ObjectComposer.forward takes (CPU, via the oracle's ray set-up)."""
    rows = cols = None
    if strides:
        rows, cols = ro.strided_grid_pixels(scene["image_size"][0], scene["image_size"][1], strides)
    if pixels is not None:
        rows, cols = pixels
    o, d, n = ro.world_rays_from_cameras(config, scene["camera_rotations"], scene["camera_translations"],
                                         scene["focals"], scene["image_size"], rows, cols)
    w2o, _ = ro.object_matrices(scene["object_rotation_parameters"], scene["object_translation_parameters"])
    return (o, d, n, w2o, scene["object_style"].unsqueeze(-3), scene["object_deformation"].unsqueeze(-3),
            scene["object_in_scene"].unsqueeze(-2))


def poison_device_memory():
    """Leaves NaN bit patterns in the blocks torch's caching allocator hands out next (the renderer's workspaces are
    ``torch.empty``): a kernel that reads scratch it never wrote - padded columns, rows beyond the compacted count - then
    produces NaNs instead of passing by luck on fresh (zero) pages."""
    blocks = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]
    small = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22) for _ in range(4)]
    del blocks, small


@contextlib.contextmanager
def oracle_in_float64():
    """The oracle's op graph in float64 (every tensor it creates takes torch's default dtype): the arbiter where a tolerance
    is wider than fp32 round-off.  Pass weights / inputs through ``to_double``."""
    before = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.set_default_dtype(before)


def to_double(x):
    if torch.is_tensor(x):
        return x.detach().cpu().double() if x.is_floating_point() else x.detach().cpu()
    if isinstance(x, dict):
        return {k: to_double(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_double(v) for v in x)
    return x


def arbitrate(exact, oracle32, hip, factor=4.0, floor=1e-6, path="", out=None):
    """Per field of a composer result: (max |HIP - fp64|, max |fp32 oracle - fp64|, ok) with
    ok = |HIP - fp64| <= factor x |fp32 oracle - fp64| + floor x max |fp64| - the HIP path may be as far from the exact
    result as the fp32 restatement of the reference is (times a small factor), not farther."""
    out = out if out is not None else {}
    for k in exact:
        if k in ("pytorch_hook", "extra_outputs") or k.startswith("_"):
            continue
        if isinstance(exact[k], dict):
            arbitrate(exact[k], oracle32[k], hip[k], factor, floor, path + k + ".", out)
            continue
        e, a, b = (t.detach().cpu().double() for t in (exact[k], oracle32[k], hip[k]))
        if k == "weights":
            e, a, b = torch.sort(e, -1)[0], torch.sort(a, -1)[0], torch.sort(b, -1)[0]
        clean = lambda t: torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0)
        err_hip = float(clean(b - e).abs().max()) if e.numel() else 0.0
        err_ref = float(clean(a - e).abs().max()) if e.numel() else 0.0
        scale = float(clean(e).abs().max()) if e.numel() else 0.0
        out[path + k] = (err_hip, err_ref, err_hip <= factor * err_ref + floor * scale)
    return out


def bender_kink_margin(run_oracle):
    """Smallest relative distance of a sample to a KINK of the ray benders' Jacobian while ``run_oracle()`` (a callable that runs
    the oracle) executes: a raw displacement at its clamp bound (``minimum(maximum(delta, lo - x), hi - x)``,
    model/nerf_models/positional_ray_bender_model.py:81-163 - the Jacobian switches between the network's and -1) or a hidden
    unit's pre-activation at 0.  The Hutchinson divergence estimate is a function of that Jacobian: it is DISCONTINUOUS there, and a
    sample within fp32 rounding of a kink legitimately lands on either side (randomized backward sweep, seed 7 case 0: one sample
    6e-9 of the box size from its clamp bound moved object_2's integrated_divergence by 3 % while every other field agreed to 4e-9)."""
    import torch.nn.functional as F
    from oracle import render_oracle as ro
    original = ro.bender_forward
    margins = []

    def traced(sd, prefix, cfg, bbox, x, deformation):
        with torch.no_grad():
            if x.numel():
                pe_cfg = cfg["position_encoder"]
                size = bbox[:, 1] - bbox[:, 0]
                w = ro.annealing_weights(sd[prefix + "positional_encoder.current_step"], pe_cfg["octaves"], pe_cfg["num_steps"])
                enc = ro.positional_encoding(x / size, pe_cfg["octaves"], pe_cfg["append_original"], w)
                h = torch.cat([enc, deformation], dim=-1)
                for i in range(cfg["layers_count"]):
                    if i == cfg["skip_layer_idx"]:
                        h = torch.cat([h, enc, deformation], dim=-1)
                    pre = F.linear(h, sd[prefix + f"backbone_layers.{i}.weight"], sd[prefix + f"backbone_layers.{i}.bias"])
                    margins.append(float((pre.abs() / pre.abs().max().clamp_min(1e-30)).min()))
                    h = F.relu(pre)
                delta = F.linear(h, sd[prefix + "output_head.weight"]) * size
                lo, hi = bbox[:, 0].unsqueeze(0) - x, bbox[:, 1].unsqueeze(0) - x
                margins.append(float(torch.minimum((delta - lo).abs(), (delta - hi).abs()).min() / size.max()))
        return original(sd, prefix, cfg, bbox, x, deformation)

    ro.bender_forward = traced
    try:
        run_oracle()
    finally:
        ro.bender_forward = original
    return min(margins) if margins else 1.0


def compare_results(want, got, rtol, atol, path="", out=None):
    """NaN-aware comparison of two composer result dicts; ``weights`` are compared after sorting
    (tie order inside equal-t groups is unspecified in the reference).  Returns {field: (maxdiff, ok)}."""
    out = out if out is not None else {}
    for k in want:
        if k in ("pytorch_hook", "extra_outputs") or k.startswith("_"):
            continue
        if isinstance(want[k], dict):
            compare_results(want[k], got[k], rtol, atol, path + k + ".", out)
            continue
        a, b = want[k].detach().cpu().float(), got[k].detach().cpu().float()
        assert a.shape == b.shape, f"{path + k}: shape {tuple(b.shape)} != {tuple(a.shape)}"
        if k == "weights":
            a, _ = torch.sort(a, dim=-1)
            b, _ = torch.sort(b, dim=-1)
        nan_ok = torch.equal(torch.isnan(a), torch.isnan(b))
        diff = torch.nan_to_num(a - b, nan=0.0, posinf=0.0, neginf=0.0).abs().max().item()
        ok = nan_ok and torch.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
        out[path + k] = (diff, ok)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Stand-in encoders for the observation-driven modes (EnvironmentModel.forward_from_observations needs one object encoder
# and one object-parameters encoder per object model; the reference's are CNNs with roi_pool crops, out of scope).  Small
# deterministic, differentiable torch modules with the call contracts of the reference's modules, so that the reference
# itself (build container) and this package (GPU box) can run the same scene.
class StandInObjectEncoder(torch.nn.Module):
    """Contract of ObjectEncoderV4.forward (model/object_encoder_v4.py:80-178): (observations (..., O, C, 3, H, W),
    bounding_box (..., O, C, 4), camera_rotations, camera_translations, global_frame_indexes, video_frame_indexes,
    video_indexes) -> (style (..., O, S), deformation (..., O, D), attention, crops)."""

    def __init__(self, style_features: int, deformation_features: int, seed: int):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.to_style = torch.nn.Parameter(torch.randn(7, style_features, generator=g))
        self.to_deformation = torch.nn.Parameter(torch.randn(7, deformation_features, generator=g))

    def forward(self, observations, bounding_box, camera_rotations, camera_translations, global_frame_indexes,
                video_frame_indexes, video_indexes):
        colour = observations[..., 0, :, :, :].mean(dim=(-1, -2))                      # first camera, (..., O, 3)
        code = torch.cat([colour, bounding_box[..., 0, :]], dim=-1)                    # (..., O, 7)
        style = torch.tanh(code @ self.to_style)
        deformation = torch.tanh(code @ self.to_deformation)
        attention = torch.zeros(list(code.shape[:-1]) + [1, 1, 2, 2], device=code.device)
        crops = observations[..., 0:1, :, :4, :4]
        return style, deformation, attention, crops


class StandInStaticParameters(torch.nn.Module):
    """Static object models sit at the world origin (model/static_object_parameters_encoder.py): (observations) ->
    rotations, translations (..., O, 3, objects)."""

    def __init__(self, objects_count: int):
        super().__init__()
        self.objects_count = objects_count

    def forward(self, observations):
        shape = list(observations.shape[:-4]) + [3, self.objects_count]
        zeros = torch.zeros(shape, device=observations.device)
        return zeros, zeros.clone()


class StandInDynamicParameters(torch.nn.Module):
    """Contract of ClassicObjectParametersEncoder.forward (model/classic_object_parameters_encoder.py:129-237):
    (observations, transformation_matrix_w2c, camera_rotations, focals, bounding_boxes (..., O, C, 4, n),
    bounding_boxes_validity (..., O, C, n)) -> rotations, translations (..., O, 3, n).  The pose is an affine function of
    the first camera's box centre: ``translation = origin + u * (cx - .5) + v * (cy - .5)``, rotation about ``up``."""

    def __init__(self, origin, u, v, up_axis: int, gain: float = 1.0):
        super().__init__()
        self.register_buffer("origin", torch.as_tensor(origin, dtype=torch.float32))
        self.register_buffer("u", torch.as_tensor(u, dtype=torch.float32))
        self.register_buffer("v", torch.as_tensor(v, dtype=torch.float32))
        self.up_axis = up_axis
        self.gain = torch.nn.Parameter(torch.tensor(float(gain)))

    def forward(self, observations, transformation_matrix_w2c, camera_rotations, focals, bounding_boxes,
                bounding_boxes_validity):
        box = bounding_boxes[..., 0, :, :]                                             # first camera, (..., O, 4, n)
        cx = (box[..., 0, :] + box[..., 2, :]) / 2 - 0.5                               # (..., O, n)
        cy = (box[..., 1, :] + box[..., 3, :]) / 2 - 0.5
        translation = (self.origin.unsqueeze(-1) + self.u.unsqueeze(-1) * cx.unsqueeze(-2) * self.gain
                       + self.v.unsqueeze(-1) * cy.unsqueeze(-2) * self.gain)          # (..., O, 3, n)
        rotation = torch.zeros_like(translation)
        rotation[..., self.up_axis, :] = cx * 0.5
        return rotation, translation


def stand_in_encoders(config, world: str, seed: int = 5):
    """(object_encoders, object_parameters_encoders) for a tennis or minecraft configuration: one module per object model."""
    static = config["model"]["static_object_models"]
    enc, par = [], []
    for m, mcfg in enumerate(config["model"]["object_models"]):
        enc.append(StandInObjectEncoder(mcfg["style_features"], mcfg["deformation_features"], seed + m))
        count = int(config["model"]["object_parameters_encoder"][m]["objects_count"])
        if m < static:
            par.append(StandInStaticParameters(count))
        elif world == "tennis":   # z up, court in the xy plane
            par.append(StandInDynamicParameters(origin=(0.0, 1.0, 0.01), u=(8.0, 0.0, 0.0), v=(0.0, -30.0, 0.0), up_axis=2))
        else:                     # minecraft: y up
            par.append(StandInDynamicParameters(origin=(0.0, 0.0, 0.0), u=(0.0, 0.0, 8.0), v=(8.0, 0.0, 0.0), up_axis=1))
    return enc, par


from playableenvironments_amd.synthetic import observation_batch  # noqa: E402,F401  (shared with bench.py)


# ---------------------------------------------------------------------------------------------------------------------
# forward_pose_consistency / forward_keypoint_consistency against the reference's recorded outputs (tests/golden/consistency,
# oracle/make_golden.py consistency): shared by the CPU suite (oracle composer behind the product's host logic) and the GPU suite
CONSISTENCY_KEYS = ("camera_rotations", "camera_translations", "focals", "bounding_boxes", "bounding_boxes_validity",
                    "global_frame_indexes", "video_frame_indexes", "video_indexes")


def run_consistency_fixture(z, model, device, monkeypatch):
    """Replays the fixture's random draws through ``model`` and returns {label: (reference tensor, product tensor)}."""
    from playableenvironments_amd import ray_sampling
    t = lambda name: torch.from_numpy(z[name]).to(device)
    queues = {"object": [], "keypoints": []}
    for kind in queues:
        i = 0
        while f"draw/{kind}/{i}/0" in z.files:
            queues[kind].append(tuple(t(f"draw/{kind}/{i}/{j}") for j in range(3)))
            i += 1
    assert len(queues["object"]) == 2 and len(queues["keypoints"]) == 2
    monkeypatch.setattr(ray_sampling, "sample_rays_at_object", lambda *a, **k: queues["object"].pop(0))
    monkeypatch.setattr(ray_sampling, "sample_rays_at_keypoints", lambda *a, **k: queues["keypoints"].pop(0))
    common = [t("in/" + k) for k in CONSISTENCY_KEYS] + [t("se/" + k) for k in ("object_style", "object_deformation",
                                                                                "object_rotation_parameters",
                                                                                "object_translation_parameters")]
    with torch.no_grad():
        pose = model(t("in/optical_flow"), *common, 30, False, mode="pose_consistency")
        kp = model(t("in/observations"), *common, t("in/keypoints"), t("in/bounding_boxes_validity"), 20, False,
                   mode="keypoint_consistency")
    assert not queues["object"] and not queues["keypoints"]
    pairs = {}
    for name, (previous, following) in pose["coarse"].items():
        for tag, (positions, opacity) in (("previous", previous), ("following", following)):
            pairs[f"pose/{name}/{tag}/positions"] = positions
            pairs[f"pose/{name}/{tag}/opacity"] = opacity
    for name, (positions, confidence, opacity, sampled) in kp["coarse"].items():
        for tag, value in (("positions", positions), ("confidence", confidence), ("opacity", opacity), ("sampled", sampled)):
            pairs[f"keypoint/{name}/{tag}"] = value
    recorded = [k for k in z.files if k.startswith(("pose/", "keypoint/"))]
    assert sorted(recorded) == sorted(pairs), (sorted(recorded), sorted(pairs))
    return {k: (torch.from_numpy(z[k]), v.detach().cpu()) for k, v in pairs.items()}
