"""Host orchestration of the renderer: the scene-encoding path of the reference's EnvironmentModel.

Keeps the tensor-in / dict-out signatures of ``EnvironmentModel.forward(mode="scene_encodings")``,
``forward_from_scene_encoding`` (model/environment_model.py:1041-1158),
``render_full_frame_from_scene_encoding`` (:618-651), ``batchified_composer_call`` (:474-521),
``merge_dictionaries`` (:523-545) and ``fold_dictionary`` (:547-579), so evaluators / play loops /
the playable model that drive the reference through these entry points can drive this class.

Everything between "scene encoding" and "composer result" runs on the GPU: rays come from the
``pr_camera_rays`` kernel, the composer is the HIP renderer; the few per-(frame, object) 4x4
matrices and box projections are tiny PyTorch-ROCm ops.  The observation-driven modes need the
CNN encoders, which are out of scope for this package: they raise.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib, ray_sampling
from .object_composer import ObjectComposer, ObjectIDsHelper


def euler_to_matrix(rotations: torch.Tensor, translations: torch.Tensor) -> torch.Tensor:
    """(..., 3) Euler angles (x, y, z; radians) and (..., 3) translations -> (..., 4, 4) with
    R = Ry (Rx Rz)  (utils/lib_3d/transformations_3d.py:69-96)."""
    cx, sx = torch.cos(rotations[..., 0]), torch.sin(rotations[..., 0])
    cy, sy = torch.cos(rotations[..., 1]), torch.sin(rotations[..., 1])
    cz, sz = torch.cos(rotations[..., 2]), torch.sin(rotations[..., 2])
    zero, one = torch.zeros_like(cx), torch.ones_like(cx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], -1).reshape(cx.shape + (3, 3))
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).reshape(cx.shape + (3, 3))
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], -1).reshape(cx.shape + (3, 3))
    rot = torch.matmul(ry, torch.matmul(rx, rz))
    out = torch.zeros(cx.shape + (4, 4), dtype=rot.dtype, device=rot.device)
    out[..., :3, :3] = rot
    out[..., :3, 3] = translations
    out[..., 3, 3] = 1.0
    return out


def rigid_inverse(m: torch.Tensor) -> torch.Tensor:
    """Inverse of (..., 4, 4) rigid transforms [R t; 0 1] = [R^T  -R^T t; 0 1].

    The reference calls ``torch.inverse`` on these matrices (environment_model.py:221, :1078); on a GPU that is a batched
    LU with a host synchronisation per call (five per frame), which serialises the host behind the previous frame's
    render.  Every matrix on this path comes out of ``euler_to_matrix``, so the closed form applies; it agrees with the
    LU inverse to fp32 rounding and differentiates through plain tensor ops."""
    rt = m[..., :3, :3].transpose(-1, -2)
    t = -torch.matmul(rt, m[..., :3, 3:4])
    return torch.cat([torch.cat([rt, t], dim=-1), m[..., 3:4, :]], dim=-2)


def strided_grid_pixels(height: int, width: int, strides) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rows, cols) of RayHelper.sample_all_rays_strided_grid (utils/lib_3d/ray_helper.py:433-482,
    :533-582): pixel ``i*s + s//2`` for every stride, smallest stride first, row-major per stride."""
    if not isinstance(strides, collections.abc.Sequence):
        strides = [strides]
    rows, cols = [], []
    for s in strides:
        if height % s != 0:
            raise Exception("The image height is not divisible by the stride")
        if width % s != 0:
            raise Exception("The image width is not divisible by the stride")
        r = torch.arange(height // s, dtype=torch.int32) * s + s // 2
        c = torch.arange(width // s, dtype=torch.int32) * s + s // 2
        rr, cc = torch.meshgrid(r, c, indexing="ij")
        rows.append(rr.reshape(-1))
        cols.append(cc.reshape(-1))
    return torch.cat(rows), torch.cat(cols)


def camera_rays(c2w: torch.Tensor, focals: torch.Tensor, height: int, width: int, rows: torch.Tensor,
                cols: torch.Tensor):
    """World-frame rays of the selected pixels through ``pr_camera_rays``.

    c2w (..., 4, 4); focals (...) (already rescaled); rows / cols int (R) shared by all frames, or
    (..., R) with one pixel list per frame.
    Returns origins (..., 3), directions (..., R, 3), focal normals (..., 3)."""
    if not c2w.is_cuda:
        raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
    lead = list(c2w.shape[:-2])
    n = int(math.prod(lead)) if lead else 1
    dev = c2w.device
    m = c2w.detach().to(torch.float32).reshape(n, 4, 4)[:, :3, :].contiguous()
    f = torch.broadcast_to(focals.detach().to(torch.float32), lead).reshape(n).contiguous()
    per_frame = rows.dim() > 1
    r = rows.size(-1)
    if per_frame:
        rows = torch.broadcast_to(rows, lead + [r]).reshape(n, r)
        cols = torch.broadcast_to(cols, lead + [r]).reshape(n, r)
    rows = rows.to(device=dev, dtype=torch.int32).contiguous()
    cols = cols.to(device=dev, dtype=torch.int32).contiguous()
    origins = torch.empty((n, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((n, r, 3), dtype=torch.float32, device=dev)
    normals = torch.empty((n, 3), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):      # the launch goes to the current device: it has to be the tensors' device
        _lib.check(lib.pr_camera_rays(n, r, height, width, 1 if per_frame else 0, m.data_ptr(), f.data_ptr(), rows.data_ptr(),
                                      cols.data_ptr(), origins.data_ptr(), dirs.data_ptr(), normals.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "pr_camera_rays")
    return origins.reshape(lead + [3]), dirs.reshape(lead + [r, 3]), normals.reshape(lead + [3])


def camera_rays_at_positions(c2w: torch.Tensor, focals: torch.Tensor, height: int, width: int, positions: torch.Tensor,
                             correct_range: bool = False):
    """World-frame rays through CONTINUOUS image positions, without the (H, W, 3) direction grid the reference samples
    with grid_sample (RayHelper.create_camera_rays + sample_rays_at + transform_rays; ray_helper.py:15-52, :1014-1052,
    :1203-1227): a pinhole grid is linear in the pixel coordinates, so its bilinear lookup at pixel coordinate (v, u) is
    ((u - W/2) / f, -(v - H/2) / f, -1).  Used by the pose / keypoint consistency paths (optical-flow targets, skeleton
    samples).  Differentiable torch ops; agrees with the grid lookup to fp32 rounding.

    c2w (..., 4, 4); focals (...) (already rescaled); positions (..., n, 2) as (row, col) normalised to [0, 1];
    ``correct_range``: positions were produced as pixel / size (RayHelper.sample_rays_at's correction).
    Returns origins (..., 3), directions (..., n, 3), focal normals (..., 3)."""
    size = torch.tensor([height, width], dtype=positions.dtype, device=positions.device)
    pos = positions * (size / (size - 1 + 1e-8)) if correct_range else positions
    v = pos[..., 0] * (height - 1)                      # align_corners: 0 -> first pixel, 1 -> last pixel
    u = pos[..., 1] * (width - 1)
    f = focals.unsqueeze(-1)
    d_cam = torch.stack([(u - width / 2) / f, -(v - height / 2) / f, -torch.ones_like(u)], dim=-1)
    rot = c2w[..., :3, :3].unsqueeze(-3)
    directions = torch.sum(d_cam.unsqueeze(-2) * rot, -1)
    return c2w[..., :3, 3], directions, -c2w[..., :3, 2]


class EnvironmentModel(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.focal_length_multiplier = config["data"]["focal_length_multiplier"]
        self.use_weighted_sampling = config["model"].get("use_weighted_sampling", False)
        self.sampling_weights = config["model"].get("sampling_weights", None)
        self.object_composer = ObjectComposer(config)
        self.object_id_helper = ObjectIDsHelper(config)
        self.current_step = 0
        # per-device constants of the host path (pixel lists of full-frame / strided-grid renders, box points)
        self._pixel_cache: Dict = {}
        self._edge_point_cache: Dict = {}
        self._axes_point_cache: Dict = {}

    def set_step(self, current_step: int):
        self.current_step = current_step
        self.object_composer.set_step(current_step)

    # ------------------------------------------------------------------ modes
    def forward(self, *args, mode="observations", **kwargs):
        if mode == "scene_encodings":
            return self.forward_from_scene_encoding(*args, **kwargs)
        if mode in ("observations", "observations_scene_encoding_only", "pose_consistency", "keypoint_consistency"):
            raise NotImplementedError(
                f"forward mode '{mode}' needs the CNN object encoders of the reference, which this renderer package "
                "does not contain; encode the scene with the reference model and call mode='scene_encodings'")
        raise Exception(f"Unknown forward mode '{mode}'")

    # ------------------------------------------------------------------ tiny per-(frame, object) math
    def compute_transformation_matrix_w2o_o2w(self, object_rotation_parameters_o2w: torch.Tensor,
                                              object_translation_parameters_o2w: torch.Tensor):
        """(..., 3, K) poses -> w2o, o2w of shape (..., 1, 4, 4, K) (singleton cameras dim).
        model/environment_model.py:206-232."""
        if object_rotation_parameters_o2w.size(-1) != self.object_id_helper.objects_count:
            raise Exception(f"poses given for {object_rotation_parameters_o2w.size(-1)} objects instead of "
                            f"{self.object_id_helper.objects_count}")
        # all K objects in one batch of (..., K, 4, 4) matrices, then back to the reference's trailing object dimension
        o2w = euler_to_matrix(object_rotation_parameters_o2w.movedim(-1, -2), object_translation_parameters_o2w.movedim(-1, -2))
        w2o = rigid_inverse(o2w)
        return w2o.movedim(-3, -1).unsqueeze(-4), o2w.movedim(-3, -1).unsqueeze(-4)

    @staticmethod
    def _project(points: torch.Tensor, o2w: torch.Tensor, w2c: torch.Tensor, focals: torch.Tensor):
        """Object-frame points (K, P, 3) of all K objects, o2w (..., 4, 4, K), w2c (..., C, 4, 4), focals (..., C) ->
        image-plane coordinates (..., C, K, P, 2) relative to the image centre (x right, y down) and the camera-frame
        z (..., C, K, P, 1).  environment_model.py:272-292, all objects in one pass."""
        m = o2w.movedim(-1, -3).unsqueeze(-3)                                                   # (..., K, 1, 4, 4)
        world = torch.sum(points.unsqueeze(-2) * m[..., :3, :3], -1) + m[..., :3, -1]           # (..., K, P, 3)
        world = world.unsqueeze(-4)                                                             # (..., 1, K, P, 3)
        c = w2c.unsqueeze(-3).unsqueeze(-3)                                                     # (..., C, 1, 1, 4, 4)
        cam = torch.sum(world.unsqueeze(-2) * c[..., :3, :3], -1) + c[..., :3, -1]              # (..., C, K, P, 3)
        proj = -cam[..., :2] / cam[..., 2:3] * focals.unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
        proj = torch.stack([proj[..., 0], -proj[..., 1]], dim=-1)
        return proj, cam[..., 2:3]

    def _image_scale(self, width: int, height: int, like: torch.Tensor) -> torch.Tensor:
        """(4, 1) [width, height, width, height] on the device of ``like`` (uploaded once: a per-call host-to-device copy
        would stall the host behind the queued render)."""
        key = ("scale", width, height, like.dtype, str(like.device))
        if key not in self._pixel_cache:
            self._pixel_cache[key] = torch.as_tensor([width, height, width, height], dtype=like.dtype, device=like.device).unsqueeze(-1)
        return self._pixel_cache[key]

    def _edge_points(self, device) -> torch.Tensor:
        """(K, 68, 3) box corner + edge points of every object instance (the boxes are fixed buffers: built once per device)."""
        key = str(device)
        if key not in self._edge_point_cache:
            helper = self.object_id_helper
            self._edge_point_cache[key] = torch.stack(
                [self.object_composer.object_models_coarse[helper.model_idx_by_object_idx(k)].bounding_box.get_edge_points()
                 for k in range(helper.objects_count)], dim=0).to(device)
        return self._edge_point_cache[key]

    def compute_object_bounding_boxes(self, transformation_matrix_o2w, transformation_matrix_w2c, focals, height, width):
        """Image-plane boxes (..., C, 4, K) [left, top, right, bottom] and projected box points
        (..., C, 68, 2, K), normalised to [0, 1].  model/environment_model.py:234-327."""
        if transformation_matrix_o2w.dim() > transformation_matrix_w2c.dim():
            transformation_matrix_o2w = transformation_matrix_o2w[..., 0, :, :, :]
        proj, z = self._project(self._edge_points(transformation_matrix_o2w.device), transformation_matrix_o2w,
                                transformation_matrix_w2c, focals)                             # (..., C, K, 68, 2)
        behind = (z > 0).expand_as(proj)
        hi = torch.where(behind, 1e20, proj)
        lo = torch.where(behind, -1e20, proj)
        lo_xy, hi_xy = hi.min(dim=-2)[0], lo.max(dim=-2)[0]                                    # (..., C, K, 2) [x, y]
        boxes = torch.cat([lo_xy, hi_xy], dim=-1).movedim(-2, -1)                              # (..., C, 4, K) left top right bottom
        points = proj.movedim(-3, -1)                                                          # (..., C, 68, 2, K)
        scale = self._image_scale(width, height, boxes)
        boxes = (boxes + scale / 2) / scale
        pscale = scale[:2]
        points = (points + pscale / 2) / pscale
        return torch.clamp(boxes, min=0.0, max=1.0), torch.clamp(points, min=0.0, max=1.0)

    def compute_object_axes_projection(self, transformation_matrix_o2w, transformation_matrix_w2c, focals, height, width):
        """Projected origin + unit axes of every object (..., C, 4, 2, K), normalised, not clamped.
        model/environment_model.py:329-404."""
        if transformation_matrix_o2w.dim() > transformation_matrix_w2c.dim():
            transformation_matrix_o2w = transformation_matrix_o2w[..., 0, :, :, :]
        key = str(transformation_matrix_o2w.device)
        if key not in self._axes_point_cache:
            self._axes_point_cache[key] = torch.tensor([(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)],
                                                       device=transformation_matrix_o2w.device)
        pts = self._axes_point_cache[key].unsqueeze(0).expand(self.object_id_helper.objects_count, 4, 3)
        out = self._project(pts, transformation_matrix_o2w, transformation_matrix_w2c, focals)[0].movedim(-3, -1)
        pscale = self._image_scale(width, height, out)[:2]
        return (out + pscale / 2) / pscale

    def compute_ray_object_distances(self, ray_origins: torch.Tensor, ray_directions: torch.Tensor,
                                     transformation_matrix_o2w: torch.Tensor) -> torch.Tensor:
        """Squared distance between every ray (a line) and every object's box centre.

        ray_origins (..., C, 3); ray_directions (..., C, R, 3); transformation_matrix_o2w (..., 4, 4, K) ->
        (..., C, R, K).  model/environment_model.py:653-706."""
        origins = ray_origins.unsqueeze(-2)
        unit = ray_directions / torch.norm(ray_directions, dim=-1, keepdim=True)
        out = []
        for k in range(self.object_id_helper.objects_count):
            model = self.object_composer.object_models_coarse[self.object_id_helper.model_idx_by_object_idx(k)]
            m = transformation_matrix_o2w[..., k]
            centre = model.bounding_box.get_center_offset(device=origins.device)
            centre = torch.sum(centre.unsqueeze(-2) * m[..., :3, :3], -1) + m[..., :3, -1]      # object -> world
            to_object = origins - centre.unsqueeze(-2).unsqueeze(-2)
            along = torch.sum(to_object * unit, dim=-1)
            out.append((to_object - along.unsqueeze(-1) * unit).pow(2).sum(-1))
        return torch.stack(out, dim=-1)

    # ------------------------------------------------------------------ composer plumbing
    def batchified_composer_call(self, ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                 deformation, object_in_scene, perturb, samples_per_image_batching: int = 0,
                                 video_indexes=None, canonical_pose: bool = False):
        """model/environment_model.py:474-521.  The reference chunks rays (1000 per call in full-frame
        rendering) because it materialises (rays, samples, 192) tensors; the fused renderer does not
        need to, so ``samples_per_image_batching`` is accepted and ignored - the composer splits a
        call only if its scratch would exceed its workspace budget, which is exact."""
        results = self.object_composer(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                       deformation, object_in_scene, perturb, video_indexes=video_indexes,
                                       canonical_pose=canonical_pose)
        return self.merge_dictionaries([results], dimension=ray_directions.dim() - 2)

    def merge_dictionaries(self, dictionaries: List[Dict], dimension: int):
        merged = {}
        for key in list(dictionaries[0].keys()):
            if key == "pytorch_hook":
                continue
            if torch.is_tensor(dictionaries[0][key]):
                parts = [d[key] for d in dictionaries]
                merged[key] = parts[0] if len(parts) == 1 else torch.cat(parts, dim=dimension)
            else:
                merged[key] = self.merge_dictionaries([d[key] for d in dictionaries], dimension)
        return merged

    def fold_dictionary(self, dictionary: Dict, height: int, width: int):
        """Folds the first dimension of size ``height*width`` of every tensor into (height, width).
        model/environment_model.py:547-579 (size matching, as the reference)."""
        target = height * width
        for key in dictionary:
            cur = dictionary[key]
            if type(cur) is dict:
                dictionary[key] = self.fold_dictionary(cur, height, width)
            elif torch.is_tensor(cur):
                sizes = list(cur.size())
                for idx, size in enumerate(sizes):
                    if size == target:
                        dictionary[key] = cur.reshape(sizes[:idx] + [height, width] + sizes[idx + 1:])
                        break
        return dictionary

    # ------------------------------------------------------------------ scene encoding -> rays -> composer
    def forward_from_scene_encoding(self, camera_rotations, camera_translations, focals, image_size,
                                    object_rotation_parameters_o2w, object_translation_parameters_o2w, object_style,
                                    object_deformation, object_in_scene, samples_per_image: int, perturb: bool,
                                    samples_per_image_batching: int = 0, upsample_factor: float = 1.0,
                                    patch_size: int = 0, patch_stride=0, canonical_pose: bool = False,
                                    _ray_range: Tuple[int, int] = None) -> Dict:
        """model/environment_model.py:1041-1158; argument shapes documented there.

        camera_* (..., O, C, 3); focals (..., O, C); object_* (..., O, 3|S|D, K); object_in_scene (..., O, K).
        ``_ray_range`` (extension): render only the rays [begin, end) of the pixel list - one rank's contiguous share of
        a frame in ``render_sharded``."""
        rescaled_focals = focals * self.focal_length_multiplier
        height = int(image_size[0] * upsample_factor)
        width = int(image_size[1] * upsample_factor)

        c2w = euler_to_matrix(camera_rotations, camera_translations)
        w2o, o2w = self.compute_transformation_matrix_w2o_o2w(object_rotation_parameters_o2w,
                                                              object_translation_parameters_o2w)
        w2c = rigid_inverse(c2w)
        boxes, box_points = self.compute_object_bounding_boxes(o2w, w2c, rescaled_focals * upsample_factor, height, width)
        axes = self.compute_object_axes_projection(o2w, w2c.detach(), rescaled_focals.detach(), height, width)

        lead = list(camera_rotations.shape[:-1])
        flat_boxes = boxes.reshape(-1, 4, boxes.size(-1))
        if patch_size != 0 and samples_per_image != 0:
            idx = ray_sampling.strided_patch_pixels(flat_boxes, self.sampling_weights, height, width, patch_size, patch_stride)
            rows, cols = ray_sampling.split_indices(idx.reshape(lead + [-1]), width)
        elif samples_per_image == 0:
            # static pixel lists (every pixel, or the strided grids): built once per (size, strides, device)
            strides = tuple(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else (int(patch_stride),)
            key = (height, width, strides if patch_stride else None, str(c2w.device))
            if key not in self._pixel_cache:
                if patch_stride:
                    rows, cols = strided_grid_pixels(height, width, patch_stride)
                else:
                    r = torch.arange(height * width, dtype=torch.int32)
                    rows, cols = r // width, r % width
                self._pixel_cache[key] = (rows.to(c2w.device), cols.to(c2w.device))
            rows, cols = self._pixel_cache[key]
        elif self.use_weighted_sampling:
            idx = ray_sampling.sample_pixels_weighted(flat_boxes, self.sampling_weights, height, width, samples_per_image)
            rows, cols = ray_sampling.split_indices(idx.reshape(lead + [-1]), width)
        else:
            idx = ray_sampling.sample_pixels_uniform(flat_boxes.size(0), height, width, samples_per_image, boxes.device)
            rows, cols = ray_sampling.split_indices(idx.reshape(lead + [-1]), width)

        if _ray_range is not None:
            rows, cols = rows[..., _ray_range[0]:_ray_range[1]], cols[..., _ray_range[0]:_ray_range[1]]
        origins, directions, normals = camera_rays(c2w, rescaled_focals * upsample_factor, height, width, rows, cols)

        results = self.batchified_composer_call(origins, directions, normals, w2o, object_style.unsqueeze(-3),
                                                object_deformation.unsqueeze(-3), object_in_scene.unsqueeze(-2),
                                                perturb, samples_per_image_batching, canonical_pose=canonical_pose)
        results["object_rotation_parameters"] = object_rotation_parameters_o2w
        results["object_translation_parameters"] = object_translation_parameters_o2w
        results["reconstructed_bounding_boxes"] = boxes
        results["reconstructed_3d_bounding_boxes"] = box_points
        results["projected_axes"] = axes
        results["scene_encoding"] = {
            "camera_rotations": camera_rotations,
            "camera_translations": camera_translations,
            "focals": focals,
            "object_rotation_parameters": object_rotation_parameters_o2w,
            "object_translation_parameters": object_translation_parameters_o2w,
            "object_style": object_style,
            "object_deformation": object_deformation,
            "object_in_scene": object_in_scene,
        }
        return results

    def render_full_frame_from_scene_encoding(self, camera_rotations, camera_translations, focals, image_size,
                                              object_rotation_parameters_o2w, object_translation_parameters_o2w,
                                              object_style, object_deformation, object_in_scene, perturb: bool,
                                              samples_per_image_batching: int = 1000, upsample_factor: float = 1.0,
                                              canonical_pose: bool = False) -> Dict:
        """Every pixel of the frame, folded back to (height, width).  model/environment_model.py:618-651."""
        flat = self(camera_rotations, camera_translations, focals, image_size, object_rotation_parameters_o2w,
                    object_translation_parameters_o2w, object_style, object_deformation, object_in_scene, 0, perturb,
                    samples_per_image_batching, upsample_factor=upsample_factor, canonical_pose=canonical_pose,
                    mode="scene_encodings")
        height = int(image_size[0] * upsample_factor)
        width = int(image_size[1] * upsample_factor)
        return self.fold_dictionary(flat, height, width)

    # ------------------------------------------------------------------ multi-GPU: one render shared by all ranks
    def render_sharded(self, camera_rotations, camera_translations, focals, image_size, object_rotation_parameters_o2w,
                       object_translation_parameters_o2w, object_style, object_deformation, object_in_scene,
                       perturb: bool = False, patch_stride=0, upsample_factor: float = 1.0, canonical_pose: bool = False,
                       shard: str = "auto", fields=("integrated_features", "opacity", "depth"), entries=("global",),
                       dst=None, group=None) -> Dict:
        """One evaluation render of ``forward_from_scene_encoding`` (every pixel, or the strided grids of ``patch_stride``)
        shared by the ranks of a torch.distributed group (one process per GPU; RCCL over xGMI): every rank holds the
        whole scene encoding (a few KB) and the replicated weights, renders its share and the rendered maps are
        exchanged with ONE collective per requested field (BASELINE.json configs[3]).

        shard = "frames": the leading (batch) dimension is split over the ranks (``parallel.shard_range``; batches
        that do not divide evenly give ragged shards); "rays": every rank renders a contiguous range of the pixel list
        of all frames (a single frame across the node); "auto": frames when the batch has at least one frame per rank.
        Rays are independent and eval-mode BatchNorm uses the running statistics, so the assembled result is bit-identical
        to the single-GPU render.  Returns ``{type: {entry: {field: tensor}}}`` with the reference's shapes on ``dst``
        (None on the other ranks), on every rank when ``dst`` is None.  Without an initialised process group (or with one
        rank) it is the plain render."""
        import torch.distributed as dist
        from . import parallel
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if perturb and world > 1:
            raise ValueError("render_sharded is an evaluation render (perturb=False): per-rank noise would not reproduce "
                             "the single-GPU result")
        batch = camera_rotations.size(0)
        if shard == "auto":
            shard = "frames" if batch >= world else "rays"
        if shard not in ("frames", "rays"):
            raise ValueError(f"unknown shard mode {shard!r} (expected 'auto', 'frames' or 'rays')")
        args = [camera_rotations, camera_translations, focals, image_size, object_rotation_parameters_o2w,
                object_translation_parameters_o2w, object_style, object_deformation, object_in_scene]
        kwargs = dict(upsample_factor=upsample_factor, patch_stride=patch_stride, canonical_pose=canonical_pose,
                      mode="scene_encodings")
        if shard == "frames":
            if batch < world:
                raise ValueError(f"shard='frames' needs at least one frame per rank ({batch} frames, {world} ranks): use 'rays'")
            local_args = [parallel.shard_frames(a, rank, world, 0) if torch.is_tensor(a) else a for a in args]
            total, dim = batch, 0
            ray_range = None
        else:
            local_args = args
            height, width = int(image_size[0] * upsample_factor), int(image_size[1] * upsample_factor)
            if patch_stride:
                strides = patch_stride if isinstance(patch_stride, collections.abc.Sequence) else [patch_stride]
                total = sum((height // s) * (width // s) for s in strides)
            else:
                total = height * width
            dim = camera_rotations.dim() - 1
            ray_range = parallel.shard_range(total, rank, world)
        with torch.no_grad():
            local = self(*local_args, 0, perturb, 0, _ray_range=ray_range, **kwargs)
        out: Dict = {}
        receives = dst is None or rank == dst
        for ty in ("coarse", "fine"):
            if ty not in local:
                continue
            for entry in entries:
                for field in fields:
                    full = parallel.gather_ray_shards(local[ty][entry][field], total, dim, dst=dst, group=group)
                    if receives:
                        out.setdefault(ty, {}).setdefault(entry, {})[field] = full
        return out if receives else None
