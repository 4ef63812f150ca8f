"""CPU oracle for the volumetric-renderer hot path of willi-menapace/PlayableEnvironments.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
(``playableenvironments_amd``) never does and has no CPU fallback.

It is an own-code, functional restatement (plain PyTorch, fp32; tensors are created on the device of
the inputs, so tests can also time this op graph with PyTorch-ROCm on the GPU) of the reference's
algorithm for:  camera rays -> ray/object transforms -> slab test -> stratified / hierarchical
sample placement -> AABB cull + compaction -> ray-bender MLP -> AdaIN style NeRF MLP (or skybox)
-> per-object alpha compositing -> cross-object sort/compose (+ static/dynamic overlap fix)
-> integrate.  It deliberately keeps the reference's *materialising* op graph (per-sample
broadcasts, boolean-mask compaction, ``sort`` + ``gather`` compose, optional ray chunking) so that
it (a) matches the reference's outputs to fp32 round-off and (b) is a fair "reference CPU path"
when timed.  Every function cites the reference file:line it follows (paths relative to the
reference root).

Parity pin: ``oracle/check_against_reference.py`` runs the imported reference and this module on
identical inputs/weights in the build container (max |diff| reported in DESIGN.md), and
``oracle/make_golden.py`` writes reference-generated fixtures to ``tests/golden``; the CPU test
suite re-checks this module against those fixtures wherever it runs.

Arbitration mode: every tensor this module creates takes torch's DEFAULT dtype, so the same op graph runs in float64 when the
caller sets ``torch.set_default_dtype(torch.float64)`` and passes double weights / inputs (``tests/helpers.oracle_in_float64``):
where a test's tolerance is wider than fp32 round-off, |HIP - fp64| is judged against |this module in fp32 - fp64|.

Weights are addressed by the reference's own ``state_dict`` key names
(``object_models_coarse.<m>.nerf_model.backbone_layers.<i>.weight`` ...), so a reference
checkpoint (or the product module's ``state_dict()``) can be passed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# Config helpers
# --------------------------------------------------------------------------------------------

class ObjectLayout:
    """Object-instance <-> model bookkeeping (model/utils/object_ids_helper.py:4-153).

    Static models come first; model ``m`` owns ``object_parameters_encoder[m].objects_count``
    consecutive object instances."""

    def __init__(self, config: dict):
        model_cfg = config["model"]
        self.static_models = int(model_cfg["static_object_models"])
        self.model_count = len(model_cfg["object_models"])
        counts = [int(e["objects_count"]) for e in model_cfg["object_parameters_encoder"]]
        self.model_of_object: List[int] = []
        for m in range(self.model_count):
            self.model_of_object += [m] * counts[m]
        self.objects_count = len(self.model_of_object)
        self.static_objects = sum(counts[: self.static_models])
        self.dynamic_objects = self.objects_count - self.static_objects


def _bbox_tensor(model_cfg: dict, device=None) -> Tensor:
    return torch.as_tensor(model_cfg["bounding_box"], dtype=torch.get_default_dtype(), device=device)  # (3, 2)


# --------------------------------------------------------------------------------------------
# Geometry (utils/lib_3d)
# --------------------------------------------------------------------------------------------

def euler_to_matrix(rotations: Tensor, translations: Tensor) -> Tensor:
    """(..., 3) x/y/z radians + (..., 3) translation -> (..., 4, 4), R = Ry @ (Rx @ Rz).

    utils/lib_3d/transformations_3d.py:10-96."""
    cx, sx = torch.cos(rotations[..., 0]), torch.sin(rotations[..., 0])
    cy, sy = torch.cos(rotations[..., 1]), torch.sin(rotations[..., 1])
    cz, sz = torch.cos(rotations[..., 2]), torch.sin(rotations[..., 2])
    lead = list(rotations.shape[:-1])
    zeros = lambda: torch.zeros(lead + [3, 3], dtype=torch.get_default_dtype(), device=rotations.device)
    rx, ry, rz = zeros(), zeros(), zeros()
    rx[..., 0, 0] += 1.0
    rx[..., 1, 1] += cx
    rx[..., 1, 2] += -sx
    rx[..., 2, 1] += sx
    rx[..., 2, 2] += cx
    ry[..., 1, 1] += 1.0
    ry[..., 0, 0] += cy
    ry[..., 2, 0] += -sy
    ry[..., 0, 2] += sy
    ry[..., 2, 2] += cy
    rz[..., 2, 2] += 1.0
    rz[..., 0, 0] += cz
    rz[..., 0, 1] += -sz
    rz[..., 1, 0] += sz
    rz[..., 1, 1] += cz
    rot = torch.matmul(ry, torch.matmul(rx, rz))
    out = torch.zeros(lead + [4, 4], dtype=torch.get_default_dtype(), device=rotations.device)
    out[..., :3, :3] = rot
    out[..., :3, 3] = translations
    out[..., 3, 3] = 1.0
    return out


def create_camera_rays(lead: List[int], height: int, width: int, focal: Tensor):
    """Pinhole rays in the camera frame, d = ((c - W/2)/f, -(r - H/2)/f, -1).

    utils/lib_3d/ray_helper.py:15-52.  ``focal`` is a (*lead) tensor."""
    focal = focal.unsqueeze(-1).unsqueeze(-1)
    dev = focal.device
    rows, cols = torch.meshgrid(torch.arange(0, height, device=dev), torch.arange(0, width, device=dev), indexing="ij")
    dx = (cols - width / 2) / focal
    dy = -(rows - height / 2) / focal
    dz = -torch.ones_like(dx)
    directions = torch.stack([dx, dy, dz], -1)
    normals = torch.zeros(lead + [3], device=dev)
    normals[..., 2] = -1
    origins = torch.zeros_like(normals)
    return directions, origins, normals


def strided_grid_pixels(height: int, width: int, strides) -> Tuple[Tensor, Tensor]:
    """Pixel (row, col) lists of the strided full-frame sampler: pixel ``i*s + s//2`` on a
    ``H/s x W/s`` grid for every stride, strides concatenated smallest first.

    utils/lib_3d/ray_helper.py:433-482 and :533-582.  Returns int64 (R,) rows and cols."""
    if not isinstance(strides, (list, tuple)):
        strides = [strides]
    all_r, all_c = [], []
    for s in strides:
        if height % s or width % s:
            raise Exception("image size not divisible by the stride")
        r = torch.arange(height // s) * s + s // 2
        c = torch.arange(width // s) * s + s // 2
        rr, cc = torch.meshgrid(r, c, indexing="ij")
        all_r.append(rr.reshape(-1))
        all_c.append(cc.reshape(-1))
    return torch.cat(all_r), torch.cat(all_c)


def transform_points(points: Tensor, matrix: Tensor, rotation=True, translation=True) -> Tensor:
    """p' = R p (+ t) as broadcast-multiply + sum.  utils/lib_3d/ray_helper.py:1180-1201."""
    out = points
    if rotation:
        out = torch.sum(out.unsqueeze(-2) * matrix[..., :3, :3], -1)
    if translation:
        out = out + matrix[..., :3, -1]
    return out


def transform_rays(origins: Tensor, directions: Tensor, normals: Tensor, matrix: Tensor):
    """Origins rotated+translated, directions/normals rotated.  ray_helper.py:1203-1227."""
    o = transform_points(origins, matrix)
    n = transform_points(normals, matrix, translation=False)
    d = transform_points(directions, matrix.unsqueeze(-3), translation=False)
    return o, d, n


def raywise_z_bounds(origins: Tensor, directions: Tensor, bbox: Tensor, valid: Tensor):
    """Per-ray AABB slab test in the object frame; misses / absent objects -> (0, 0).

    model/object_composer.py:104-151.  ``bbox`` is (3, 2) [lo, hi]; eps is added to the
    direction *before* the division (sign-asymmetric quirk kept)."""
    eps = 1e-6
    corners = torch.stack([bbox[:, 0], bbox[:, 1]], 0)            # (2, 3)
    corners = corners - origins.unsqueeze(-2)                      # (..., 2, 3)
    corners = corners.unsqueeze(-3)                                # (..., 1, 2, 3)
    z = corners / (directions.unsqueeze(-2) + eps)                 # (..., R, 2, 3)
    z_near, _ = z.min(dim=-2)
    z_far, _ = z.max(dim=-2)
    z_near, _ = z_near.max(dim=-1)
    z_far, _ = z_far.min(dim=-1)
    valid = valid.unsqueeze(-1)
    _, valid = torch.broadcast_tensors(z_far, valid)
    mask = torch.logical_or(z_far <= z_near, valid == False)  # noqa: E712
    z_far = z_far.clone()
    z_near = z_near.clone()
    z_far[mask] = 0.0
    z_near[mask] = 0.0
    return z_near, z_far


def stratified_positions(origins: Tensor, directions: Tensor, z_near: Tensor, z_far: Tensor,
                         count: int, perturb: bool, rand: Optional[Tensor] = None):
    """t_i = near (1 - s_i) + far s_i with s = linspace(0, 1, P); optional stratified jitter;
    x = o + d t.  utils/lib_3d/ray_helper.py:1229-1282.

    ``rand``: explicit U[0,1) tensor of the shape of ``t`` (otherwise drawn from torch's global
    CPU generator exactly where the reference draws it).  Returns (x, t, rand_used)."""
    s = torch.linspace(0.0, 1.0, count, device=z_near.device)
    t = z_near.unsqueeze(-1) * (1.0 - s) + z_far.unsqueeze(-1) * s
    used = None
    if perturb:
        mid = (t[..., 1:] + t[..., :-1]) / 2
        upper = torch.cat([mid, t[..., -1:]], dim=-1)
        lower = torch.cat([t[..., :1], mid], dim=-1)
        used = torch.rand(t.size(), device=t.device) if rand is None else rand
        t = lower + (upper - lower) * used
    x = origins.unsqueeze(-2).unsqueeze(-2) + directions.unsqueeze(-2) * t.unsqueeze(-1)
    return x, t, used


def sample_pdf(bins: Tensor, weights: Tensor, count: int, perturb: bool,
               rand: Optional[Tensor] = None):
    """Inverse-CDF sampling of ``count`` values from the piecewise-constant pdf ``weights`` over
    ``bins``.  utils/lib_3d/ray_helper.py:1348-1403 (the reference adds 1e-5 to its argument in
    place; here the addition is out of place, same values)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    used = None
    if not perturb:
        u = torch.linspace(0.0, 1.0, count, device=cdf.device)
        u = u.expand(list(cdf.shape[:-1]) + [count]).contiguous()
    else:
        used = torch.rand(list(cdf.shape[:-1]) + [count], device=cdf.device) if rand is None else rand
        u = used.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(idx - 1, min=0)
    above = torch.clamp(idx, max=cdf.size(-1) - 1)
    cdf_lo = torch.gather(cdf, -1, below)
    cdf_hi = torch.gather(cdf, -1, above)
    bin_lo = torch.gather(bins, -1, below)
    bin_hi = torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    frac = (u - cdf_lo) / denom
    return bin_lo + frac * (bin_hi - bin_lo), used


def hierarchical_positions(origins: Tensor, directions: Tensor, count: int, ref_t: Tensor,
                           weights: Tensor, perturb: bool, rand: Optional[Tensor] = None):
    """Fine sample placement: resample ``count`` t from the coarse weights, merge with the coarse
    t and sort.  utils/lib_3d/ray_helper.py:1320-1346."""
    mids = (ref_t[..., 1:] + ref_t[..., :-1]) / 2
    new_t, used = sample_pdf(mids, weights[..., 1:-1], count, perturb, rand)
    new_t = new_t.detach()
    merged, _ = torch.sort(torch.cat([ref_t, new_t], dim=-1), dim=-1)
    x = origins.unsqueeze(-2).unsqueeze(-2) + directions.unsqueeze(-2) * merged.unsqueeze(-1)
    return x, merged, used


# --------------------------------------------------------------------------------------------
# Networks (model/nerf_models, model/positional_encoder.py, model/layers/adain.py)
# --------------------------------------------------------------------------------------------

def positional_encoding(x: Tensor, octaves: int, append_original: bool,
                        octave_weights: Optional[Tensor] = None) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (each block as wide as x).

    model/positional_encoder.py:41-65; with ``octave_weights`` the annealed variant of
    model/annealable_positional_encoder.py:46-76."""
    freqs = 2.0 ** torch.linspace(0.0, octaves - 1, octaves, device=x.device)
    parts = [x] if append_original else []
    for k in range(octaves):
        for fn in (torch.sin, torch.cos):
            e = fn(freqs[k] * x)
            if octave_weights is not None:
                e = e * octave_weights[k]
            parts.append(e)
    return torch.cat(parts, dim=-1)


def annealing_weights(step: Tensor, octaves: int, num_steps: int) -> Tensor:
    """(1 - cos(pi clamp(step * octaves / num_steps - k, 0, 1))) / 2 for k = 0..octaves-1.

    model/annealable_positional_encoder.py:59-63 (``step`` is the int32 ``current_step`` buffer)."""
    alpha = step * octaves / num_steps
    k = torch.arange(octaves, dtype=torch.get_default_dtype(), device=step.device)
    return (1 - torch.cos(math.pi * torch.clamp(alpha - k, min=0.0, max=1.0))) / 2


def _in_box(x: Tensor, bbox: Tensor) -> Tensor:
    """Closed-interval AABB test.  model/nerf_models/ray_bending_style_nerf_model.py:62-85."""
    ok = None
    for a in range(3):
        cur = torch.logical_and(x[..., a] >= bbox[a, 0], x[..., a] <= bbox[a, 1])
        ok = cur if ok is None else torch.logical_and(ok, cur)
    return ok


def bender_forward(sd: Dict[str, Tensor], prefix: str, cfg: dict, bbox: Tensor, x: Tensor,
                   deformation: Tensor) -> Tensor:
    """PositionalRayBender: x/size -> annealed PE -> cat deformation -> MLP with skip ->
    Linear(no bias) -> * size -> clamp so x + delta stays in the box.

    model/nerf_models/positional_ray_bender_model.py:81-163."""
    pe_cfg = cfg["position_encoder"]
    size = bbox[:, 1] - bbox[:, 0]
    w = annealing_weights(sd[prefix + "positional_encoder.current_step"], pe_cfg["octaves"],
                          pe_cfg["num_steps"])
    enc = positional_encoding(x / size, pe_cfg["octaves"], pe_cfg["append_original"], w)
    h = torch.cat([enc, deformation], dim=-1)
    for i in range(cfg["layers_count"]):
        if i == cfg["skip_layer_idx"]:
            h = torch.cat([h, enc, deformation], dim=-1)
        h = F.relu(F.linear(h, sd[prefix + f"backbone_layers.{i}.weight"],
                            sd[prefix + f"backbone_layers.{i}.bias"]))
    delta = F.linear(h, sd[prefix + "output_head.weight"]) * size
    lo = bbox[:, 0].unsqueeze(0) - x
    hi = bbox[:, 1].unsqueeze(0) - x
    return torch.minimum(torch.maximum(delta, lo), hi)


def _adain(sd: Dict[str, Tensor], prefix: str, h: Tensor, style: Tensor, training: bool,
           update_stats: bool) -> Tensor:
    """AffineTransformAdaIn: [scale | bias] = Linear(style); BatchNorm1d(affine=False)(h) * scale
    + bias.  model/layers/adain.py:5-61."""
    enc = F.linear(style, sd[prefix + "affine_transform.weight"], sd[prefix + "affine_transform.bias"])
    scale, bias = enc.chunk(2, 1)
    rm = sd[prefix + "ada_in.normalization.running_mean"]
    rv = sd[prefix + "ada_in.normalization.running_var"]
    if training:
        # nn.BatchNorm1d counts the batch first, then torch.nn.functional.batch_norm raises for EXACTLY one value per channel
        # (_verify_batch_size); an empty batch passes: empty output, running statistics untouched, the counter incremented
        if update_stats:
            sd[prefix + "ada_in.normalization.num_batches_tracked"] += 1
        if h.size(0) == 1:
            raise ValueError("Expected more than 1 value per channel when training")
        if update_stats:
            out = F.batch_norm(h, rm, rv, None, None, True, 0.1, 1e-5)
        else:
            out = F.batch_norm(h, None, None, None, None, True, 0.1, 1e-5)
    else:
        out = F.batch_norm(h, rm, rv, None, None, False, 0.1, 1e-5)
    return out * scale + bias


def _features_head(sd, prefix, h, style, training, update_stats):
    """Linear(no bias) -> AdaIN -> ReLU -> Linear(no bias) -> AdaIN -> ReLU -> Linear.

    model/nerf_models/adain_style_nerf_model.py:57-71."""
    h = F.linear(h, sd[prefix + "features_head.0.weight"])
    h = F.relu(_adain(sd, prefix + "features_head.1.", h, style, training, update_stats))
    h = F.linear(h, sd[prefix + "features_head.3.weight"])
    h = F.relu(_adain(sd, prefix + "features_head.4.", h, style, training, update_stats))
    return F.linear(h, sd[prefix + "features_head.6.weight"], sd[prefix + "features_head.6.bias"])


def _backbone(sd, prefix, cfg, enc):
    h = enc
    for i in range(cfg["backbone_layers_count"]):
        if i == cfg["skip_layer_idx"]:
            h = torch.cat([h, enc], dim=-1)
        h = F.relu(F.linear(h, sd[prefix + f"backbone_layers.{i}.weight"],
                            sd[prefix + f"backbone_layers.{i}.bias"]))
    return h


def adain_nerf_forward(sd, prefix, cfg, bbox, empty_alpha, x, style, training, update_stats):
    """AdaInStyleNerfModel.forward on flat (M, 3) positions: second closed-interval AABB mask on
    the (bent) positions, x/size -> PE -> backbone with skip -> sigma head and style feature head;
    masked-out rows give f = 0, sigma = empty_space_alpha.

    model/nerf_models/adain_style_nerf_model.py:106-199."""
    m = x.size(0)
    feats = torch.zeros((m, cfg["output_features"]), dtype=torch.get_default_dtype(), device=x.device)
    sigma = torch.ones((m, 1), dtype=torch.get_default_dtype(), device=x.device) * empty_alpha
    mask = _in_box(x, bbox)
    xs, ss = x[mask, :], style[mask, :]
    size = bbox[:, 1] - bbox[:, 0]
    pe_cfg = cfg["position_encoder"]
    enc = positional_encoding(xs / size, pe_cfg["octaves"], pe_cfg["append_original"])
    h = _backbone(sd, prefix, cfg, enc)
    sg = F.linear(h, sd[prefix + "alpha_head.weight"], sd[prefix + "alpha_head.bias"])
    ft = _features_head(sd, prefix, h, ss, training, update_stats)
    feats[mask, :] = ft
    sigma[mask, :] = sg
    return feats, sigma.squeeze(-1)


def skybox_nerf_forward(sd, prefix, cfg, bbox, origins, directions, style, training, update_stats):
    """SkyboxAdaInStyleNerfModelV3: input [o/size, d/|d|] (6-D) -> PE -> backbone -> feature head;
    sigma == 10 everywhere, no AABB filter.

    model/nerf_models/skybox_adain_style_nerf_model_v3.py:74-149."""
    size = bbox[:, 1] - bbox[:, 0]
    o = origins / size
    d = directions / (directions.pow(2).sum(-1, keepdim=True).sqrt())
    pe_cfg = cfg["position_encoder"]
    enc = positional_encoding(torch.cat([o, d], dim=-1), pe_cfg["octaves"], pe_cfg["append_original"])
    h = _backbone(sd, prefix, cfg, enc)
    ft = _features_head(sd, prefix, h, style, training, update_stats)
    sigma = torch.ones_like(ft[..., :1]) * 10.0
    return ft, sigma.squeeze(-1)


def object_model_forward(sd: Dict[str, Tensor], prefix: str, model_cfg: dict, positions: Tensor,
                         origins: Tensor, directions: Tensor, style: Tensor, deformation: Tensor,
                         canonical_pose: bool, training: bool, update_stats: bool = True):
    """RayBendingStyleNerfModel.forward: broadcast o/d/style/deformation to every sample, AABB
    cull, boolean compaction, bender, NeRF, scatter back into (0 | empty_space_alpha | 0)-filled
    tensors.  model/nerf_models/ray_bending_style_nerf_model.py:137-219.

    positions (..., R, P, 3); origins (..., R, 3); directions (..., R, 3); style (..., 1, S);
    deformation (..., 1, D)  ->  features (..., R, P, F), sigma_raw (..., R, P), delta (..., R, P, 3)."""
    bbox = _bbox_tensor(model_cfg, positions.device)
    nerf_cfg = model_cfg["nerf_model"]
    bender_cfg = model_cfg["ray_bender_model"]
    empty_alpha = model_cfg["empty_space_alpha"]
    pcount = positions.size(-2)
    lead = list(positions.shape[:-1])
    exp = lambda v: v.unsqueeze(-2).expand(lead + [v.size(-1)])
    flat = lambda v: v.reshape(-1, v.size(-1))
    f_pos = flat(positions)
    f_org = flat(exp(origins))
    f_dir = flat(exp(directions))
    f_sty = flat(style.unsqueeze(-2).expand(lead + [style.size(-1)]))
    f_def = flat(deformation.unsqueeze(-2).expand(lead + [deformation.size(-1)]))
    total = f_pos.size(0)
    dev = positions.device
    out_f = torch.zeros((total, nerf_cfg["output_features"]), dtype=torch.get_default_dtype(), device=dev)
    out_s = torch.ones((total,), dtype=torch.get_default_dtype(), device=dev) * empty_alpha
    out_d = torch.zeros((total, 3), dtype=torch.get_default_dtype(), device=dev)

    mask = _in_box(f_pos, bbox)
    xs, os_, ds, ss, es = f_pos[mask, :], f_org[mask, :], f_dir[mask, :], f_sty[mask, :], f_def[mask, :]

    if bender_cfg["architecture"].endswith("positional_ray_bender_model"):
        delta = bender_forward(sd, prefix + "ray_bender.", bender_cfg, bbox, xs, es)
    else:  # ZeroedRayBender, model/nerf_models/zeroed_ray_bender_model.py:28-37
        delta = xs * 0.0
    if canonical_pose:
        delta = delta * 0.0
    bent = xs + delta

    if nerf_cfg["architecture"].endswith("skybox_adain_style_nerf_model_v3"):
        ft, sg = skybox_nerf_forward(sd, prefix + "nerf_model.", nerf_cfg, bbox, os_, ds, ss,
                                     training, update_stats)
    else:
        ft, sg = adain_nerf_forward(sd, prefix + "nerf_model.", nerf_cfg, bbox, empty_alpha, bent, ss,
                                    training, update_stats)
    out_f[mask, :] = ft
    out_s[mask] = sg
    out_d[mask, :] = delta
    return (out_f.reshape(lead + [-1]), out_s.reshape(lead), out_d.reshape(lead + [3]))


# --------------------------------------------------------------------------------------------
# Compositing (model/object_composer.py)
# --------------------------------------------------------------------------------------------

def position_distances(t: Tensor, directions: Tensor) -> Tensor:
    """dt_i = (t_{i+1} - t_i) |d|, last = 1e10 |d|.  model/object_composer.py:153-178."""
    first = t[..., 1:] - t[..., :-1]
    last = torch.ones(list(first.shape[:-1]) + [1], dtype=torch.get_default_dtype(), device=t.device) * 1e10
    dist = torch.cat([first, last], dim=-1)
    return dist * torch.linalg.norm(directions[..., None, :], dim=-1)


def alphas_from_raw(raw: Tensor, dist: Tensor, perturb: bool, noise: Optional[Tensor] = None):
    """alpha = 1 - exp(-relu(sigma_raw [+ N(0,1)]) dt).  model/object_composer.py:180-197."""
    used = None
    if perturb:
        used = torch.randn(raw.size(), device=raw.device) if noise is None else noise
        raw = raw + used
    return 1.0 - torch.exp(-F.relu(raw) * dist), used


def weights_from_alphas(alphas: Tensor) -> Tensor:
    """w_i = alpha_i prod_{j<i} (1 - alpha_j + 1e-10).  model/object_composer.py:199-214."""
    shift = 1.0 - alphas + 1e-10
    shifted = torch.cat([torch.ones_like(shift[..., 0:1]), shift[..., :-1]], dim=-1)
    return alphas * torch.cumprod(shifted, dim=-1)


def integrate(features, raw_alphas, directions, t, displacements, divergences, perturb,
              noise: Optional[Tensor] = None):
    """Alpha-composite one (possibly merged) sample list per ray.  model/object_composer.py:724-784."""
    dist = position_distances(t, directions)
    alphas, used = alphas_from_raw(raw_alphas, dist, perturb, noise)
    weights = weights_from_alphas(alphas)
    integrated = torch.sum(weights.unsqueeze(-1) * features, dim=-2)
    depth = torch.sum(weights * t, dim=-1)
    opacity = torch.sum(weights, dim=-1)
    disparity = 1.0 / torch.clamp(depth / opacity, min=1e-10)
    integrated_divergence = torch.mean(alphas.detach() * torch.abs(divergences), dim=-1)
    disp_mag = torch.mean(weights.detach() * torch.norm(displacements, dim=-1), dim=-1)
    return {
        "integrated_features": integrated,
        "opacity": opacity,
        "weights": weights,
        "depth": depth,
        "disparity": disparity,
        "integrated_displacements_magnitude": disp_mag,
        "integrated_divergence": integrated_divergence,
    }, used


def fix_overlaps(layout: ObjectLayout, all_raw, all_t, all_pos, all_disp, all_div, ray_origins):
    """Static samples inside [t_dyn[0], t_dyn[P_static - 1]) get sigma = -10, t = 0, x = o,
    delta = div = 0; intervals found with searchsorted(left) on the static's ORIGINAL t.  The upper
    bound indexes the dynamic object's samples with the *static* object's ``P - 1`` - a reference
    quirk that is reproduced.  model/object_composer.py:220-397."""
    out_raw = [v.clone() for v in all_raw]
    out_t = [v.clone() for v in all_t]
    out_pos = [v.clone() for v in all_pos]
    out_disp = [v.clone() for v in all_disp]
    out_div = [v.clone() for v in all_div]
    for s in range(layout.static_objects):
        ps = all_t[s].size(-1)
        idx = torch.arange(ps, device=all_t[s].device)
        for dyn in range(layout.dynamic_objects):
            d = layout.static_objects + dyn
            bounds = all_t[d][..., (0, ps - 1)]
            iv = torch.searchsorted(all_t[s].contiguous(), bounds.contiguous())
            mask = torch.logical_and(idx >= iv[..., 0:1], idx < iv[..., 1:2])
            out_raw[s][mask] = out_raw[s][mask] * 0.0 - 10.0
            out_t[s][mask] *= 0.0
            pmask = mask.unsqueeze(-1).expand_as(out_pos[s])
            out_pos[s][pmask] = ray_origins.unsqueeze(-2).expand_as(out_pos[s])[pmask]
            out_disp[s][mask] *= 0.0
            out_div[s][mask] *= 0.0
    return out_raw, out_t, out_pos, out_disp, out_div


def compose(config, layout, ray_origins, all_f, all_raw, all_t, all_pos, all_disp, all_div,
            stable_merge: bool = False):
    """Concatenate all objects along P, sort by t, gather everything.  object_composer.py:399-447.

    The reference calls ``torch.sort`` without ``stable=True`` (:435): the order inside groups of
    equal t is unspecified (and differs between torch's CPU and GPU sorts).  ``stable_merge=True``
    selects the tie rule the HIP renderer defines - stable in concatenation (object) order; it only
    changes results on rays where an in-box sample sits in a tie (e.g. boxes sharing a face)."""
    if config["model"]["fix_object_overlaps"]:
        all_raw, all_t, all_pos, all_disp, all_div = fix_overlaps(layout, all_raw, all_t, all_pos,
                                                                  all_disp, all_div, ray_origins)
    f = torch.cat(all_f, dim=-2)
    raw = torch.cat(all_raw, dim=-1)
    t = torch.cat(all_t, dim=-1)
    disp = torch.cat(all_disp, dim=-2)
    div = torch.cat(all_div, dim=-1)
    t, order = torch.sort(t, dim=-1, stable=True) if stable_merge else torch.sort(t, dim=-1)
    raw = torch.gather(raw, -1, order)
    div = torch.gather(div, -1, order)
    f = torch.gather(f, -2, order.unsqueeze(-1).expand_as(f))
    disp = torch.gather(disp, -2, order.unsqueeze(-1).expand_as(disp))
    return f, raw, t, disp, div


def approximate_divergence(positions: Tensor, displacements: Tensor, training: bool,
                           noise: Optional[Tensor] = None):
    """Hutchinson estimate e^T (d delta / dx) e; zeros in eval or when delta carries no graph.

    model/object_composer.py:582-601."""
    if not training or not displacements.requires_grad:
        return torch.zeros_like(displacements[..., 0]), None
    e = torch.randn_like(displacements) if noise is None else noise
    g = torch.autograd.grad(displacements, positions, e, create_graph=True)[0]
    return (g * e).sum(dim=-1), e


def composer_forward(config: dict, sd: Dict[str, Tensor], ray_origins: Tensor, ray_directions: Tensor,
                     focal_normals: Tensor, w2o: Tensor, style: Tensor, deformation: Tensor,
                     object_in_scene: Tensor, perturb: bool, canonical_pose: bool = False,
                     training: bool = False, noise: Optional[dict] = None,
                     record_noise: Optional[dict] = None, update_stats: bool = True,
                     stable_merge: bool = False) -> dict:
    """ObjectComposer.forward.  model/object_composer.py:786-892 (+ forward_object :486-580).

    ray_origins (..., 3); ray_directions (..., R, 3); w2o (..., 4, 4, K); style (..., S, K);
    deformation (..., D, K); object_in_scene (..., K).

    Noise protocol: with ``noise=None`` random draws come from torch's global CPU generator in
    the reference's exact order, so seeding both sides gives identical
    train-mode results; drawn tensors are stored in ``record_noise`` (if given) under the keys
    ``jitter_k``, ``alpha_k`` (coarse alpha noise feeding the resampler), ``pdf_k``,
    ``int_<type>_k``, ``int_<type>_global``.  Passing the same dict back as ``noise`` replays it."""
    layout = ObjectLayout(config)
    if w2o.size(-1) != layout.objects_count:
        raise Exception("wrong number of object transformation matrices")
    apply_activation = config["model"]["apply_activation"]
    noise = noise or {}
    rec = record_noise if record_noise is not None else {}
    rays = ray_directions.size(-2)

    per_object = []
    for k in range(layout.objects_count):
        m = layout.model_of_object[k]
        mcfg = config["model"]["object_models"][m]
        has_fine = mcfg.get("use_fine", True) is not False
        bbox = _bbox_tensor(mcfg, ray_directions.device)
        present = object_in_scene[..., k]
        o, d, _ = transform_rays(ray_origins, ray_directions, focal_normals, w2o[..., k])
        near, far = raywise_z_bounds(o, d, bbox, present)
        near = torch.clamp(near, min=mcfg["z_near_min"], max=mcfg["z_far_max"])
        far = torch.clamp(far, min=mcfg["z_near_min"], max=mcfg["z_far_max"])
        x, t, used = stratified_positions(o, d, near, far, mcfg["positions_count_coarse"], perturb,
                                          noise.get(f"jitter_{k}"))
        rec[f"jitter_{k}"] = used
        if training:
            x.requires_grad_(True) if not x.requires_grad else None
        sty = style[..., k].unsqueeze(-2)
        dfm = deformation[..., k].unsqueeze(-2)
        o_exp = o.unsqueeze(-2).expand(list(d.shape))
        feats, raw, disp = object_model_forward(sd, f"object_models_coarse.{m}.", mcfg, x, o_exp, d, sty,
                                                dfm, canonical_pose, training, update_stats)
        raw = raw.clone()
        raw[torch.logical_not(present)] = mcfg["empty_space_alpha"]
        if apply_activation:
            feats = torch.sigmoid(feats)
        dist = position_distances(t, d)
        c_alpha, used = alphas_from_raw(raw, dist, perturb, noise.get(f"alpha_{k}"))
        rec[f"alpha_{k}"] = used
        c_weights = weights_from_alphas(c_alpha)
        div, used = approximate_divergence(x, disp, training, noise.get(f"div_coarse_{k}"))
        rec[f"div_coarse_{k}"] = used
        res = {"coarse": (feats, raw, t, x, disp, div)}
        if has_fine:
            xf, tf, used = hierarchical_positions(o, d, mcfg["positions_count_fine"], t, c_weights, perturb,
                                                  noise.get(f"pdf_{k}"))
            rec[f"pdf_{k}"] = used
            if training:
                xf.requires_grad_(True) if not xf.requires_grad else None
            ff, rawf, dispf = object_model_forward(sd, f"object_models_fine.{m}.", mcfg, xf, o_exp, d, sty,
                                                   dfm, canonical_pose, training, update_stats)
            rawf = rawf.clone()
            rawf[torch.logical_not(present)] = mcfg["empty_space_alpha"]
            if apply_activation:
                ff = torch.sigmoid(ff)
            divf, used = approximate_divergence(xf, dispf, training, noise.get(f"div_fine_{k}"))
            rec[f"div_fine_{k}"] = used
            res["fine"] = (ff, rawf, tf, xf, dispf, divf)
        per_object.append(res)

    exp_origins = ray_origins.unsqueeze(-2).expand(list(ray_origins.shape[:-1]) + [rays, 3])
    results = {}
    for mtype in per_object[0].keys():
        results[mtype] = {}
        cols = [[], [], [], [], [], []]
        for k, res in enumerate(per_object):
            f, raw, t, x, disp, div = res[mtype]
            for lst, v in zip(cols, (f, raw, t, x, disp, div)):
                lst.append(v)
            out, used = integrate(f, raw, ray_directions, t, disp, div, perturb, noise.get(f"int_{mtype}_{k}"))
            rec[f"int_{mtype}_{k}"] = used
            out["extra_outputs"] = {}
            results[mtype][f"object_{k}"] = out
        cf, craw, ct, cdisp, cdiv = compose(config, layout, exp_origins, *cols, stable_merge=stable_merge)
        out, used = integrate(cf, craw, ray_directions, ct, cdisp, cdiv, perturb,
                              noise.get(f"int_{mtype}_global"))
        rec[f"int_{mtype}_global"] = used
        results[mtype]["global"] = out
    results["pytorch_hook"] = torch.zeros((1,) * 9, device=ray_directions.device)
    return results


# --------------------------------------------------------------------------------------------
# EnvironmentModel-level orchestration (model/environment_model.py), scene-encoding mode
# --------------------------------------------------------------------------------------------

def expected_positions(positions: Tensor, displacements: Tensor, weights: Tensor, eps: float = 1e-8) -> Tensor:
    """Weight-averaged bent sample position (the first surface a ray meets); the weights are detached.

    model/object_composer.py:603-622."""
    weights = weights.detach()
    bent = positions + displacements
    total = (bent * weights.unsqueeze(-1)).sum(dim=-2)
    return total / (weights.unsqueeze(-1).sum(dim=-2) + eps)


def expected_positions_forward(config: dict, sd: Dict[str, Tensor], ray_origins: Tensor, ray_directions: Tensor,
                               focal_normals: Tensor, w2o: Tensor, style: Tensor, deformation: Tensor,
                               object_in_scene: Tensor, object_id: int, perturb: bool, canonical_pose: bool = False,
                               training: bool = False, noise: Optional[dict] = None,
                               record_noise: Optional[dict] = None) -> dict:
    """ObjectComposer.forward_expected_positions (model/object_composer.py:624-722): one object instance,
    w2o (..., 4, 4), style (..., S), deformation (..., D), object_in_scene (...).  Returns
    {"coarse": (expected positions (..., R, 3), opacity (..., R)) [, "fine": ...]}.  The sample distances use the
    OBJECT-frame directions here and the coarse alpha noise feeds both the coarse weights and the resampler.
    Noise keys: jitter, alpha, pdf, alpha_fine."""
    layout = ObjectLayout(config)
    m = layout.model_of_object[object_id]
    mcfg = config["model"]["object_models"][m]
    has_fine = mcfg.get("use_fine", True) is not False
    noise = noise or {}
    rec = record_noise if record_noise is not None else {}
    bbox = _bbox_tensor(mcfg, ray_directions.device)
    o, d, _ = transform_rays(ray_origins, ray_directions, focal_normals, w2o)
    near, far = raywise_z_bounds(o, d, bbox, object_in_scene)
    near = torch.clamp(near, min=mcfg["z_near_min"], max=mcfg["z_far_max"])
    far = torch.clamp(far, min=mcfg["z_near_min"], max=mcfg["z_far_max"])
    x, t, used = stratified_positions(o, d, near, far, mcfg["positions_count_coarse"], perturb, noise.get("jitter"))
    rec["jitter"] = used
    sty, dfm = style.unsqueeze(-2), deformation.unsqueeze(-2)
    o_exp = o.unsqueeze(-2).expand(list(d.shape))
    absent = torch.logical_not(object_in_scene)

    def one_pass(prefix, positions, depths, key):
        _, raw, disp = object_model_forward(sd, prefix, mcfg, positions, o_exp, d, sty, dfm, canonical_pose, training)
        raw = raw.clone()
        raw[absent] = mcfg["empty_space_alpha"]
        alphas, used_alpha = alphas_from_raw(raw, position_distances(depths, d), perturb, noise.get(key))
        rec[key] = used_alpha
        weights = weights_from_alphas(alphas)
        return expected_positions(positions, disp, weights), weights.sum(dim=-1), weights

    exp_c, opacity_c, w_c = one_pass(f"object_models_coarse.{m}.", x, t, "alpha")
    results = {"coarse": (exp_c, opacity_c)}
    if has_fine:
        xf, tf, used = hierarchical_positions(o, d, mcfg["positions_count_fine"], t, w_c, perturb, noise.get("pdf"))
        rec["pdf"] = used
        exp_f, opacity_f, _ = one_pass(f"object_models_fine.{m}.", xf, tf, "alpha_fine")
        results["fine"] = (exp_f, opacity_f)
    return results


def merge_dictionaries(dicts: List[dict], dim: int) -> dict:
    """Recursive cat of result dicts; drops ``pytorch_hook``.  model/environment_model.py:523-545."""
    out = {}
    for key in dicts[0].keys():
        if key == "pytorch_hook":
            continue
        if torch.is_tensor(dicts[0][key]):
            out[key] = torch.cat([d[key] for d in dicts], dim=dim)
        else:
            out[key] = merge_dictionaries([d[key] for d in dicts], dim)
    return out


def batchified_composer_call(config, sd, origins, directions, normals, w2o, style, deformation,
                             in_scene, perturb, chunk: int = 0, canonical_pose=False, training=False):
    """Ray-chunked composer call.  model/environment_model.py:474-521 (chunk 1000 in full-frame
    rendering, :584)."""
    dim = directions.dim() - 2
    total = directions.size(dim)
    chunk = total if chunk == 0 else chunk
    outs = []
    for start in range(0, total, chunk):
        cur = directions[..., start:start + chunk, :]
        outs.append(composer_forward(config, sd, origins, cur, normals, w2o, style, deformation, in_scene,
                                     perturb, canonical_pose=canonical_pose, training=training))
    return merge_dictionaries(outs, dim)


def world_rays_from_cameras(config, camera_rotations, camera_translations, focals, image_size,
                            pixel_rows: Optional[Tensor] = None, pixel_cols: Optional[Tensor] = None,
                            upsample_factor: float = 1.0):
    """a1 + a2 (all pixels or an explicit pixel list) + a3 (camera -> world).

    model/environment_model.py:1080-1112.  camera_* (..., C, 3); focals (..., C)."""
    f = focals * config["data"]["focal_length_multiplier"]
    height = int(image_size[0] * upsample_factor)
    width = int(image_size[1] * upsample_factor)
    lead = list(camera_rotations.shape[:-1])
    dirs, origins, normals = create_camera_rays(lead, height, width, f * upsample_factor)
    if pixel_rows is None:
        dirs = dirs.reshape(lead + [height * width, 3])
    else:
        dirs = dirs[..., pixel_rows, pixel_cols, :]
    c2w = euler_to_matrix(camera_rotations, camera_translations)
    return transform_rays(origins, dirs, normals, c2w)


def object_matrices(rotations_o2w: Tensor, translations_o2w: Tensor):
    """(..., 3, K) pose parameters -> w2o, o2w (..., 1, 4, 4, K) with a singleton camera dim.

    model/environment_model.py:206-232."""
    k = rotations_o2w.size(-1)
    o2w = torch.stack([euler_to_matrix(rotations_o2w[..., i], translations_o2w[..., i]) for i in range(k)], -1)
    w2o = torch.stack([euler_to_matrix(rotations_o2w[..., i], translations_o2w[..., i]).inverse()
                       for i in range(k)], -1)
    return w2o.unsqueeze(-4), o2w.unsqueeze(-4)


def render_from_scene_encoding(config, sd, camera_rotations, camera_translations, focals, image_size,
                               object_rotations_o2w, object_translations_o2w, object_style,
                               object_deformation, object_in_scene, perturb=False, strides=None,
                               chunk: int = 0, canonical_pose=False, training=False):
    """Scene encoding -> composer result dict (the renderer part of
    EnvironmentModel.forward_from_scene_encoding, model/environment_model.py:1041-1158).

    ``strides=None`` renders every pixel in raster order; a list renders the strided grids."""
    rows = cols = None
    if strides:
        rows, cols = strided_grid_pixels(image_size[0], image_size[1], strides)
    origins, dirs, normals = world_rays_from_cameras(config, camera_rotations, camera_translations, focals,
                                                     image_size, rows, cols)
    w2o, _ = object_matrices(object_rotations_o2w, object_translations_o2w)
    return batchified_composer_call(config, sd, origins, dirs, normals, w2o, object_style.unsqueeze(-3),
                                    object_deformation.unsqueeze(-3), object_in_scene.unsqueeze(-2), perturb,
                                    chunk, canonical_pose, training)


# --------------------------------------------------------------------------------------------
# Pixel samplers (utils/lib_3d/ray_helper.py), loop-for-loop restatements returning flat indices
# --------------------------------------------------------------------------------------------

def _sampler_weight_mask(boxes: Tensor, weights, height: int, width: int, guard: bool) -> Tensor:
    """ray_helper.py:300-330 (patch sampler, no zero-area guard) / :655-675 (weighted sampler, guarded)."""
    flat = boxes.reshape(-1, 4, boxes.size(-1)).clone()
    flat[:, 0, :] = torch.floor(flat[:, 0, :] * width)
    flat[:, 2, :] = torch.ceil(flat[:, 2, :] * width)
    flat[:, 1, :] = torch.floor(flat[:, 1, :] * height)
    flat[:, 3, :] = torch.ceil(flat[:, 3, :] * height)
    masks = torch.zeros((flat.size(0), height, width))
    for n in range(flat.size(0)):
        for k in range(flat.size(-1)):
            left, top, right, bottom = (int(flat[n, i, k].item()) for i in range(4))
            area = (right - left) * (bottom - top)
            if guard and area == 0:
                continue
            masks[n, top:bottom, left:right] += weights[k] / area
    return masks.reshape(-1, height * width)


def sample_pixels_weighted(boxes: Tensor, weights, height: int, width: int, samples: int) -> Tensor:
    """RayHelper.sample_rays_weighted, ray_helper.py:611-728 -> (N, samples) flat pixel indices."""
    mask = _sampler_weight_mask(boxes, weights, height, width, True)
    out = []
    for n in range(mask.size(0)):
        cur = mask[n] / mask[n].sum()
        cdf = torch.cumsum(cur, dim=0)
        u = torch.rand((samples,))
        out.append(torch.clamp(torch.searchsorted(cdf, u), max=cdf.size(0) - 1))
    return torch.stack(out, 0)


def strided_patch_pixels(boxes: Tensor, weights, height: int, width: int, patch_size: int, strides) -> Tensor:
    """RayHelper.sample_rays_strided_patch (align_grid=True), ray_helper.py:236-431 -> (N, sum p_i^2)."""
    s0, sm = strides[0], strides[-1]
    sizes = [(patch_size * s0) // s for s in strides]
    half = sizes[-1] // 2
    mask = _sampler_weight_mask(boxes, weights, height, width, False)
    backward = list(range(sm // 2, sm)) + list(range(0, sm // 2))
    forward = list(range(sm // 2 + sm, sm, -1)) + [0] + list(range(sm - 1, sm // 2, -1))
    out = []
    for n in range(mask.size(0)):
        cur = mask[n] / mask[n].sum()
        cdf = torch.cumsum(cur, dim=0)
        u = torch.rand((1,))
        flat = int(torch.clamp(torch.searchsorted(cdf, u), max=cdf.size(0) - 1)[0].item())
        row, col = flat // width, flat % width
        row = min(height - sm * (half - 1) - 1, max(half * sm, row))
        col = min(width - sm * (half - 1) - 1, max(half * sm, col))
        start = [row - half * sm, col - half * sm]
        for a in range(2):
            diff = start[a] % sm
            if diff != sm // 2:
                start[a] = start[a] - backward[diff] if start[a] >= sm // 2 else start[a] + forward[diff]
        idx = []
        for s, size in zip(strides, sizes):
            off = sm // 2 - s // 2
            for r in range(start[0] - off, start[0] - off + s * size, s):
                for c in range(start[1] - off, start[1] - off + s * size, s):
                    idx.append(r * width + c)
        out.append(torch.as_tensor(idx, dtype=torch.int64))
    return torch.stack(out, 0)


def psnr(a: Tensor, b: Tensor) -> float:
    """PSNR = -10 log10(mean((a-b)^2) + 1e-8) on [0,1]-ranged data.  evaluation/metrics/psnr.py:10-34."""
    mse = torch.mean((a - b) ** 2)
    return float(-10.0 * torch.log10(mse + 1e-8))
