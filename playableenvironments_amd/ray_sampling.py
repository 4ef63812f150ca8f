"""Pixel selection for ray batches (the reference's RayHelper.sample_* family), without per-pixel
Python loops or per-object ``.item()`` syncs.

Each function returns int32 ``rows, cols`` of shape (N, R) - one pixel list per frame - that
``camera_rays`` turns into world rays on the GPU, plus the normalised (row/H, col/W) positions the
reference returns (utils/lib_3d/ray_helper.py:1157-1178).  Random draws use ``torch.rand`` /
``torch.randperm`` frame by frame in the reference's order, so a seeded generator reproduces the
reference's choices.
"""
from __future__ import annotations

import collections.abc
from typing import List, Sequence, Tuple

import torch


def _weight_masks(bounding_boxes: torch.Tensor, weights: Sequence[float], height: int, width: int,
                  guard_zero_area: bool) -> torch.Tensor:
    """(N, 4, K) normalised [left, top, right, bottom] boxes -> (N, H*W) sampling weights: every object
    adds weights[k] / area over its pixel-aligned box (ray_helper.py:300-330 / :655-675)."""
    n, _, k = bounding_boxes.shape
    bb = bounding_boxes.detach().to(torch.float32)
    left = torch.floor(bb[:, 0, :] * width)
    right = torch.ceil(bb[:, 2, :] * width)
    top = torch.floor(bb[:, 1, :] * height)
    bottom = torch.ceil(bb[:, 3, :] * height)
    # python slicing semantics of mask[top:bottom, left:right] (negative / oversize bounds never occur:
    # the boxes are clamped to [0, 1] by compute_object_bounding_boxes)
    left, right = left.clamp(0, width), right.clamp(0, width)
    top, bottom = top.clamp(0, height), bottom.clamp(0, height)
    area = (right - left) * (bottom - top)                                    # (N, K)
    w = _object_weights(weights, bb.device)
    per_object = w.unsqueeze(0) / area                                         # inf / nan for empty boxes, as the reference
    if guard_zero_area:
        per_object = torch.where(area != 0, per_object, torch.zeros_like(per_object))
    rows = torch.arange(height, device=bb.device).view(1, height, 1, 1)
    cols = torch.arange(width, device=bb.device).view(1, 1, width, 1)
    inside = (rows >= top.view(n, 1, 1, k)) & (rows < bottom.view(n, 1, 1, k)) & \
             (cols >= left.view(n, 1, 1, k)) & (cols < right.view(n, 1, 1, k))
    mask = torch.zeros((n, height, width), dtype=torch.float32, device=bb.device)
    for obj in range(k):  # sequential accumulation in object order, like the reference's += loop
        mask = mask + torch.where(inside[..., obj], per_object[:, obj].view(n, 1, 1), torch.zeros((), device=bb.device))
    return mask.reshape(n, height * width)


_OBJECT_WEIGHTS = {}


def _object_weights(weights: Sequence[float], device) -> torch.Tensor:
    """The per-object sampling weights as a device tensor (uploaded once: a host-to-device copy per call would stall
    the host behind the queued kernels)."""
    key = (tuple(float(v) for v in weights), str(device))
    if key not in _OBJECT_WEIGHTS:
        _OBJECT_WEIGHTS[key] = torch.as_tensor(list(weights), dtype=torch.float32, device=device)
    return _OBJECT_WEIGHTS[key]


def _sample_cdf(weights: torch.Tensor, count: int) -> torch.Tensor:
    """Per frame: normalise, cumsum, ``count`` uniform draws, searchsorted, clamp (ray_helper.py:353-362)."""
    out = []
    for i in range(weights.size(0)):
        cur = weights[i] / weights[i].sum()
        cdf = torch.cumsum(cur, dim=0)
        u = torch.rand((count,), device=weights.device)
        idx = torch.searchsorted(cdf, u)
        out.append(torch.clamp(idx, max=cdf.size(0) - 1))
    return torch.stack(out, dim=0)


def positions_from_indices(indices: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """(..., R) flat pixel indices -> (..., R, 2) normalised (row / H, col / W)  (ray_helper.py:1157-1178)."""
    rows = indices // width
    cols = indices % width
    return torch.stack([rows / height, cols / width], dim=-1)


def sample_pixels_weighted(bounding_boxes: torch.Tensor, weights: Sequence[float], height: int, width: int,
                           samples_per_image: int) -> torch.Tensor:
    """RayHelper.sample_rays_weighted (ray_helper.py:611-728): (N, samples) flat pixel indices drawn
    with replacement from the bounding-box weight image."""
    mask = _weight_masks(bounding_boxes, weights, height, width, guard_zero_area=True)
    return _sample_cdf(mask, samples_per_image)


def sample_pixels_uniform(frames: int, height: int, width: int, samples_per_image: int, device) -> torch.Tensor:
    """RayHelper.sample_rays (ray_helper.py:730-795): a fresh random permutation prefix per frame."""
    return torch.stack([torch.randperm(height * width, device=device)[:samples_per_image] for _ in range(frames)], 0)


def strided_patch_pixels(bounding_boxes: torch.Tensor, weights: Sequence[float], height: int, width: int,
                         patch_size: int, strides, align_grid: bool = True) -> torch.Tensor:
    """RayHelper.sample_rays_strided_patch (ray_helper.py:236-431; ``align_grid=False`` raises, as the reference does at
    :269-270): one
    box-weighted random centre per frame, clamped so the patch stays inside the image and aligned to the
    grid of the largest stride; then a ``p_i x p_i`` grid per stride (p_i = patch * s_0 / s_i), strides
    concatenated smallest first, row-major.  Returns (N, sum p_i^2) flat pixel indices."""
    if not align_grid:
        raise Exception("Align grid is required for patched ray sampling.")
    if patch_size % 2 != 0:
        raise Exception("Patch size must be a multiple of 2")
    if not isinstance(strides, collections.abc.Sequence):
        strides = [strides]
    s0, sm = strides[0], strides[-1]
    if (patch_size * s0) % (2 * sm) != 0:
        raise Exception("Patch size is not compatible with the chosen strides. Make patch size divisible by a higher power of 2")
    sizes = [(patch_size * s0) // s for s in strides]
    half = sizes[-1] // 2
    mask = _weight_masks(bounding_boxes, weights, height, width, guard_zero_area=False)
    centres = _sample_cdf(mask, 1)[:, 0]                                       # (N,) flat pixel index, stays on the device
    return patch_pixels_around(centres, height, width, patch_size, strides)


def patch_pixels_around(centres: torch.Tensor, height: int, width: int, patch_size: int, strides) -> torch.Tensor:
    """The pixel grids of ``strided_patch_pixels`` for given patch centres (N,) (flat pixel indices)."""
    s0, sm = strides[0], strides[-1]
    sizes = [(patch_size * s0) // s for s in strides]
    half = sizes[-1] // 2
    dev = centres.device
    row = torch.div(centres, width, rounding_mode="floor")
    col = centres - row * width
    # min(hi, max(lo, x)): the patch stays inside the image
    row = torch.clamp(torch.clamp(row, min=half * sm), max=height - sm * (half - 1) - 1)
    col = torch.clamp(torch.clamp(col, min=half * sm), max=width - sm * (half - 1) - 1)
    back, fwd = _alignment_tables(sm, dev)

    def align(start: torch.Tensor) -> torch.Tensor:
        # snap the patch start to the grid of the largest stride (offset sm // 2), towards the reference's side
        diff = start % sm
        moved = torch.where(start >= sm // 2, start - back[diff], start + fwd[diff])
        return torch.where(diff != sm // 2, moved, start)

    start_r, start_c = align(row - half * sm), align(col - half * sm)         # (N,)
    parts = []
    for s, size in zip(strides, sizes):
        off = sm // 2 - s // 2
        steps = torch.arange(size, device=dev) * s
        r = (start_r - off).unsqueeze(1) + steps                               # (N, size)
        c = (start_c - off).unsqueeze(1) + steps
        parts.append((r.unsqueeze(2) * width + c.unsqueeze(1)).reshape(r.size(0), size * size))
    return torch.cat(parts, dim=1)


def strided_patch_rows_cols(bounding_boxes: torch.Tensor, weights: Sequence[float], height: int, width: int,
                            patch_size: int, strides, align_grid: bool = True, _u: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``strided_patch_pixels`` as (rows, cols) int32 tensors of shape (N, sum p_i^2).  Device tensors take ONE launch
    (``pr_patch_pixels``: the weight image's cumulative sum in closed form, one uniform draw per frame from torch's device
    generator) instead of ~80 small tensor ops; host tensors go through ``strided_patch_pixels`` (index-for-index the
    reference's sampler under a shared seed - the draws of the two routes differ)."""
    if not bounding_boxes.is_cuda:
        return split_indices(strided_patch_pixels(bounding_boxes, weights, height, width, patch_size, strides, align_grid), width)
    if not align_grid:
        raise Exception("Align grid is required for patched ray sampling.")
    if patch_size % 2 != 0:
        raise Exception("Patch size must be a multiple of 2")
    if not isinstance(strides, collections.abc.Sequence):
        strides = [strides]
    strides = [int(v) for v in strides]
    if (patch_size * strides[0]) % (2 * strides[-1]) != 0:
        raise Exception("Patch size is not compatible with the chosen strides. Make patch size divisible by a higher power of 2")
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    bb = bounding_boxes.detach().to(torch.float32).contiguous()
    n, _, k = bb.shape
    dev = bb.device
    rays = sum(((patch_size * strides[0]) // s) ** 2 for s in strides)
    u = torch.rand((n,), device=dev) if _u is None else _u.to(device=dev, dtype=torch.float32).contiguous()
    rows = torch.empty((n, rays), dtype=torch.int32, device=dev)
    cols = torch.empty((n, rays), dtype=torch.int32, device=dev)
    w = _object_weights(weights, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pr_patch_pixels(n, k, height, width, patch_size, len(strides), (C.c_int32 * len(strides))(*strides),
                                       bb.data_ptr(), w.data_ptr(), u.data_ptr(), rows.data_ptr(), cols.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream), "pr_patch_pixels")
    return rows, cols


_ALIGNMENT_TABLES = {}


def _alignment_tables(sm: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per residue ``start % sm``: how far a patch start moves back (start >= sm // 2) or forward to reach the
    residue sm // 2 (ray_helper.py:372-396).  Uploaded once per (stride, device)."""
    key = (sm, str(device))
    if key not in _ALIGNMENT_TABLES:
        backward = list(range(sm // 2, sm)) + list(range(0, sm // 2))
        forward = list(range(sm // 2 + sm, sm, -1)) + [0] + list(range(sm - 1, sm // 2, -1))
        _ALIGNMENT_TABLES[key] = (torch.tensor(backward, device=device), torch.tensor(forward, device=device))
    return _ALIGNMENT_TABLES[key]


def split_indices(indices: torch.Tensor, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return (indices // width).to(torch.int32), (indices % width).to(torch.int32)


# ---------------------------------------------------------------------------------------------------------------------
# Samplers of the pose / keypoint consistency paths (model/environment_model.py:1197-1505).
# They work on the camera-frame direction grid, as the reference does: (..., H, W, 3), the grid of
# RayHelper.create_camera_rays.  Positions are (row / H, col / W) normalised to [0, 1].
# ---------------------------------------------------------------------------------------------------------------------
#: COCO keypoint pairs joined by a skeleton segment (ray_helper.py:815-832)
SKELETON_SEGMENTS = ((0, 11), (0, 12), (5, 6), (5, 7), (5, 11), (5, 12), (6, 8), (6, 11), (6, 12), (7, 9), (8, 10), (11, 12),
                     (11, 13), (12, 14), (13, 15), (14, 16))


def sample_rays_at(ray_directions: torch.Tensor, sampled_positions: torch.Tensor, correct_range: bool = True,
                   original_image_size: Tuple[int, int] = None) -> torch.Tensor:
    """Bilinear lookup of the (..., H, W, 3) direction grid at (..., n, 2) positions -> (..., n, 3)
    (RayHelper.sample_rays_at, ray_helper.py:1014-1052; grid_sample with align_corners)."""
    lead = list(ray_directions.shape[:-3])
    grid = ray_directions.reshape([-1] + list(ray_directions.shape[-3:])).permute(0, 3, 1, 2)
    pos = sampled_positions.reshape([-1] + list(sampled_positions.shape[-2:]))
    if correct_range:
        size = torch.tensor(original_image_size, dtype=ray_directions.dtype, device=ray_directions.device)
        pos = pos * (size / (size - 1 + 1e-8))
    pos = (pos[..., [1, 0]].unsqueeze(-2) - 0.5) * 2
    out = torch.nn.functional.grid_sample(grid, pos, align_corners=True).squeeze(-1).permute(0, 2, 1)
    return out.reshape(lead + list(out.shape[1:]))


def sample_rays_at_object(ray_directions: torch.Tensor, images: torch.Tensor, samples_per_image: int,
                          bounding_box: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``samples_per_image`` pixels drawn uniformly (with replacement) inside one (..., 4) [left, top, right, bottom] box per
    image: directions (..., n, 3), image values (..., n, C) and positions (..., n, 2)  (RayHelper.sample_rays_at_object,
    ray_helper.py:910-1012).  One vectorised weight mask instead of a loop with four ``.item()`` reads per image."""
    lead = list(ray_directions.shape[:-3])
    height, width = ray_directions.size(-3), ray_directions.size(-2)
    channels = images.size(-3)
    boxes = bounding_box.reshape(-1, 4, 1)
    # weight 1 inside the pixel-aligned box; boxes of zero area keep an all-zero mask (-> NaN cdf, index 0 ... as the reference)
    mask = _weight_masks(boxes, (1.0,), height, width, guard_zero_area=True)
    mask = torch.where(mask != 0, torch.ones_like(mask), mask)
    indices = _sample_cdf(mask, samples_per_image)                                              # (M, n)
    dirs = ray_directions.reshape(-1, height * width, 3)
    obs = images.reshape(-1, channels, height * width).transpose(1, 2)
    rows = torch.arange(dirs.size(0), device=dirs.device).unsqueeze(1)
    out_d, out_o = dirs[rows, indices], obs[rows, indices]
    positions = positions_from_indices(indices, height, width)
    return (out_d.reshape(lead + list(out_d.shape[1:])), out_o.reshape(lead + list(out_o.shape[1:])),
            positions.reshape(lead + list(positions.shape[1:])))


def sample_rays_at_keypoints(ray_directions: torch.Tensor, keypoints: torch.Tensor, max_samples_per_image: int
                             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Samples on the skeleton segments of COCO keypoints: ray_directions (..., O, C, H, W, 3), keypoints (..., O, C, 17, 3)
    as (row, col, confidence) in [0, 1] -> directions (..., O, C, n, 3), positions (..., O, C, n, 2), confidences
    (..., O, C, n).  One random fraction per (sequence, sample), shared by the observations and cameras of the sequence
    (RayHelper.sample_rays_at_keypoints, ray_helper.py:797-908)."""
    lead = list(ray_directions.shape[:-5])
    observations, cameras = ray_directions.size(-5), ray_directions.size(-4)
    dirs = ray_directions.reshape([-1] + list(ray_directions.shape[-5:]))
    kp = keypoints.reshape([-1] + list(keypoints.shape[-4:]))
    first = torch.as_tensor([a for a, _ in SKELETON_SEGMENTS], device=kp.device)
    second = torch.as_tensor([b for _, b in SKELETON_SEGMENTS], device=kp.device)
    begin, end = kp[..., first, :], kp[..., second, :]                                          # (M, O, C, 16, 3)
    repeat = -(-max_samples_per_image // len(SKELETON_SEGMENTS))
    begin = begin.repeat(1, 1, 1, repeat, 1)[..., :max_samples_per_image, :]
    end = end.repeat(1, 1, 1, repeat, 1)[..., :max_samples_per_image, :]
    fractions = torch.rand((kp.size(0), 1, 1, max_samples_per_image), dtype=kp.dtype, device=kp.device)
    fractions = fractions.repeat(1, observations, cameras, 1).unsqueeze(-1)
    points = begin + (end - begin) * fractions
    positions, scores = points[..., :2], points[..., -1]
    sampled = sample_rays_at(dirs, positions, correct_range=False)
    return (sampled.reshape(lead + list(sampled.shape[1:])), positions.reshape(lead + list(positions.shape[1:])),
            scores.reshape(lead + list(scores.shape[1:])))
