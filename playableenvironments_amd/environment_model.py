"""Host orchestration of the renderer: the scene-encoding path of the reference's EnvironmentModel.

Keeps the tensor-in / dict-out signatures of ``EnvironmentModel.forward(mode="scene_encodings")``,
``forward_from_scene_encoding`` (model/environment_model.py:1041-1158),
``render_full_frame_from_scene_encoding`` (:618-651), ``batchified_composer_call`` (:474-521),
``merge_dictionaries`` (:523-545) and ``fold_dictionary`` (:547-579), so evaluators / play loops /
the playable model that drive the reference through these entry points can drive this class.

Everything between "scene encoding" and "composer result" runs on the GPU: rays come from the
``pr_camera_rays`` kernel, the composer is the HIP renderer; the few per-(frame, object) 4x4
matrices and box projections are tiny PyTorch-ROCm ops.

The observation-driven modes (``forward_from_observations`` :847-1039, ``render_full_frame_from_observations``
:581-616, ``forward_scene_encoding_from_observations`` :772-845, ``forward_pose_consistency`` :1197-1361,
``forward_keypoint_consistency`` :1363-1505) are the same orchestration around the renderer plus the reference's CNN
object encoders / pose estimators, which stay stock PyTorch modules and are out of scope for this package (SURVEY.md
section 8): they are INJECTED - ``EnvironmentModel(config, object_encoders=..., object_parameters_encoders=...)`` or
``set_encoders`` - with the call contracts of the reference's modules (documented at ``set_encoders``).
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import math
import warnings
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib, ray_sampling
from .modules import REGISTRATION_EPOCH as _REGISTRATION_EPOCH, CameraParametersStorage, ModuleList, Tracked
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


def _pose_matrices_kernel(rotations: torch.Tensor, translations: torch.Tensor):
    lead = list(rotations.shape[:-1])
    n = int(math.prod(lead)) if lead else 1
    rot = rotations.detach().to(torch.float32).reshape(n, 3).contiguous()
    tr = torch.broadcast_to(translations.detach().to(torch.float32), rotations.shape).reshape(n, 3).contiguous()
    # (two allocations: as outputs of an autograd node, views of ONE buffer would make an in-place edit of either raise)
    m = torch.empty((n, 4, 4), dtype=torch.float32, device=rotations.device)
    inv = torch.empty((n, 4, 4), dtype=torch.float32, device=rotations.device)
    with torch.cuda.device(rotations.device):
        _lib.check(_lib.load().pr_pose_matrices(n, rot.data_ptr(), tr.data_ptr(), m.data_ptr(), inv.data_ptr(),
                                                torch.cuda.current_stream(rotations.device).cuda_stream), "pr_pose_matrices")
    return m.reshape(lead + [4, 4]), inv.reshape(lead + [4, 4]), rot, tr


class _PoseMatrices(torch.autograd.Function):
    """pr_pose_matrices with pr_pose_matrices_backward: poses that carry a graph (object poses from trainable encoders,
    learnable camera offsets) - as torch ops the conversion and its backward are ~150 launches of a few microseconds."""

    @staticmethod
    def forward(ctx, rotations, translations):
        m, inv, rot, tr = _pose_matrices_kernel(rotations, translations)
        ctx.save_for_backward(rot, tr)
        ctx.shapes = (rotations.shape, translations.shape, rotations.dtype, translations.dtype)
        return m, inv

    @staticmethod
    def backward(ctx, g_m, g_inv):
        rot, tr = ctx.saved_tensors
        n = rot.size(0)
        g_rot, g_tr = torch.empty_like(rot), torch.empty_like(tr)
        gm = g_m.to(torch.float32).reshape(n, 4, 4).contiguous() if g_m is not None else None
        gi = g_inv.to(torch.float32).reshape(n, 4, 4).contiguous() if g_inv is not None else None
        with torch.cuda.device(rot.device):
            _lib.check(_lib.load().pr_pose_matrices_backward(n, rot.data_ptr(), tr.data_ptr(), gm.data_ptr() if gm is not None else None,
                                                             gi.data_ptr() if gi is not None else None, g_rot.data_ptr(), g_tr.data_ptr(),
                                                             torch.cuda.current_stream(rot.device).cuda_stream),
                       "pr_pose_matrices_backward")
        rot_shape, tr_shape, rot_dtype, tr_dtype = ctx.shapes
        g_rot = g_rot.reshape(rot_shape).to(rot_dtype)
        g_tr = g_tr.reshape(rot_shape)
        if tuple(tr_shape) != tuple(rot_shape):                  # translations were broadcast against the rotations
            g_tr = g_tr.sum_to_size(tr_shape)
        return g_rot, g_tr.to(tr_dtype)


def pose_matrices(rotations: torch.Tensor, translations: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``euler_to_matrix`` and ``rigid_inverse`` of the result, (..., 4, 4) each.  Device tensors take ONE kernel
    (``pr_pose_matrices``; with a graph: an autograd node whose backward is one kernel too) instead of ~30 torch launches;
    CPU tensors take the torch ops (which ``check_against_reference.py`` pins against the reference)."""
    if rotations.is_cuda:
        if torch.is_grad_enabled() and (rotations.requires_grad or translations.requires_grad):
            return _PoseMatrices.apply(rotations, translations)
        return _pose_matrices_kernel(rotations, translations)[:2]
    m = euler_to_matrix(rotations, translations)
    return m, rigid_inverse(m)


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
    if not torch.is_tensor(focals):
        focals = torch.as_tensor(focals, dtype=torch.float32, device=c2w.device)
    if torch.is_grad_enabled() and (c2w.requires_grad or focals.requires_grad):
        # learnable camera parameters (camera_parameters_offsets): the rays carry a graph back to them
        return _CameraRays.apply(c2w, focals, height, width, rows, cols)
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
    if n * r == 0:
        # an empty pixel list (a rank whose share of a small frame's tiles is empty): origins / normals from the matrices, no launch
        origins, normals = m[:, :, 3].clone(), -m[:, :, 2]
        return origins.reshape(lead + [3]), dirs.reshape(lead + [r, 3]), normals.reshape(lead + [3])
    lib = _lib.load()
    with torch.cuda.device(dev):      # the launch goes to the current device: it has to be the tensors' device
        _lib.check(lib.pr_camera_rays(n, r, height, width, 1 if per_frame else 0, m.data_ptr(), f.data_ptr(), rows.data_ptr(),
                                      cols.data_ptr(), origins.data_ptr(), dirs.data_ptr(), normals.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "pr_camera_rays")
    return origins.reshape(lead + [3]), dirs.reshape(lead + [r, 3]), normals.reshape(lead + [3])


def _camera_rays_prepared(camera34: torch.Tensor, focals: torch.Tensor, lead, height: int, width: int, rows: torch.Tensor,
                          cols: torch.Tensor):
    """``camera_rays`` on pr_scene_setup's outputs: camera34 (n, 3, 4) c2w rows, focals (n) - no marshalling copies."""
    n = camera34.size(0)
    dev = camera34.device
    per_frame = rows.dim() > 1
    r = rows.size(-1)
    if per_frame:
        rows = torch.broadcast_to(rows, list(lead) + [r]).reshape(n, r)
        cols = torch.broadcast_to(cols, list(lead) + [r]).reshape(n, r)
    if rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != dev:
        rows = rows.to(device=dev, dtype=torch.int32).contiguous()
        cols = cols.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty(n * (6 + 3 * r), dtype=torch.float32, device=dev)
    origins, normals, dirs = out[:3 * n].view(n, 3), out[3 * n:6 * n].view(n, 3), out[6 * n:].view(n, r, 3)
    if n * r == 0:
        origins.copy_(camera34[:, :, 3])
        normals.copy_(-camera34[:, :, 2])
    else:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().pr_camera_rays(n, r, height, width, 1 if per_frame else 0, camera34.data_ptr(), focals.data_ptr(),
                                                  rows.data_ptr(), cols.data_ptr(), origins.data_ptr(), dirs.data_ptr(), normals.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "pr_camera_rays")
    return origins.view(list(lead) + [3]), dirs.view(list(lead) + [r, 3]), normals.view(list(lead) + [3])


class _CameraRays(torch.autograd.Function):
    """``camera_rays`` with a backward pass: d_cam = ((col - W/2) / f, -(row - H/2) / f, -1), d_world = R d_cam, o = t,
    focal normal = -R[:, 2] (ray_helper.py:15-52, 1203-1227), differentiated with respect to the camera matrix and the
    focal length - the path learnable camera offsets (model/layers/camera_parameters_storage.py) are trained through."""

    @staticmethod
    def forward(ctx, c2w, focals, height, width, rows, cols):
        with torch.no_grad():
            origins, directions, normals = camera_rays(c2w, focals, height, width, rows, cols)
        ctx.save_for_backward(c2w, focals, rows, cols)
        ctx.size = (height, width)
        return origins, directions, normals

    @staticmethod
    def backward(ctx, g_origins, g_directions, g_normals):
        c2w, focals, rows, cols = ctx.saved_tensors
        height, width = ctx.size
        lead = list(c2w.shape[:-2])
        f = torch.broadcast_to(focals.to(torch.float32), lead).unsqueeze(-1)                     # (..., 1)
        u = (cols.to(torch.float32) - width / 2).to(c2w.device)                                   # (R) or (..., R)
        v = -(rows.to(torch.float32) - height / 2).to(c2w.device)
        g_c2w = torch.zeros_like(c2w, dtype=torch.float32)
        g_f = None
        if g_directions is not None:
            g_d = g_directions.to(torch.float32)
            d_cam = torch.stack([u / f, v / f, -torch.ones_like(u / f)], dim=-1)                 # (..., R, 3)
            g_c2w[..., :3, :3] = torch.einsum("...ri,...rj->...ij", g_d, d_cam)                   # d_world_i = R_ij d_cam_j
            g_cam = torch.einsum("...ri,...ij->...rj", g_d, c2w[..., :3, :3].to(torch.float32))   # R^T g
            g_f = -((g_cam[..., 0] * u + g_cam[..., 1] * v) / (f * f)).sum(-1)
        if g_origins is not None:
            g_c2w[..., :3, 3] = g_origins.to(torch.float32)
        if g_normals is not None:
            g_c2w[..., :3, 2] -= g_normals.to(torch.float32)
        if g_f is not None:
            # focals broadcast against the cameras' leading dimensions (missing or size-1 dimensions alike)
            g_f = g_f.sum_to_size(focals.shape).to(focals.dtype) if focals.dim() else g_f.sum().to(focals.dtype)
        return g_c2w.to(c2w.dtype), g_f, None, None, None, None


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


class _LazyGraph(torch.autograd.Function):
    """forward = ``fast(*tensors)`` without a graph (one HIP kernel); backward = torch.autograd through ``slow(*tensors)``
    recomputed on the saved inputs, only if a gradient ever arrives.  For the projected boxes / axes of a training call: the
    shipped losses do not read them, so the ~40 small tensor ops (and their graph) of the differentiable formulation are not
    issued per step; a loss that does read them gets exactly the gradients of that formulation."""

    @staticmethod
    def forward(ctx, fast, slow, *tensors):
        ctx.slow = slow
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            out = fast(*tensors)
        return tuple(out) if isinstance(out, (tuple, list)) else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        inputs = [t.detach().requires_grad_(need) for t, need in zip(ctx.saved_tensors, ctx.needs_input_grad[2:])]
        with torch.enable_grad():
            outs = ctx.slow(*inputs)
            outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
            wanted = [t for t in inputs if t.requires_grad]
            got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True) if pairs and wanted else []
        it = iter(got)
        return (None, None) + tuple(next(it) if t.requires_grad else None for t in inputs)


def _copy_out(results):
    """A copy of a (nested) result dictionary whose tensors do not alias the originals - the recorded frame's static outputs ->
    tensors the caller owns.  One allocation and ONE multi-tensor copy launch per dtype (``torch._foreach_copy_``) instead of a
    ``clone()`` per tensor: a result dictionary holds ~60 tensors, and 60 launches of ~5 us are as long as the small frames
    themselves.  The copies of a dtype are views of one flat buffer (disjoint regions)."""
    tensors = []

    def collect(x):
        if torch.is_tensor(x):
            tensors.append(x)
        elif isinstance(x, dict):
            for v in x.values():
                collect(v)
        elif isinstance(x, (list, tuple)):
            for v in x:
                collect(v)
    collect(results)
    copies = {}
    groups = {}
    for t in tensors:
        if id(t) in copies:
            continue
        if t.is_contiguous() and t.numel() > 0:
            groups.setdefault((t.dtype, t.device), []).append(t)
            copies[id(t)] = None
        else:
            copies[id(t)] = t.clone()
    for (dtype, device), members in groups.items():
        # every view starts at a multiple of 64 elements (>= 256 bytes for 4-byte types): aligned like separate allocations
        offsets, at = [], 0
        for t in members:
            offsets.append(at)
            at += (t.numel() + 63) // 64 * 64
        flat = torch.empty(at, dtype=dtype, device=device)
        views = [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offsets, members)]
        torch._foreach_copy_(views, members)
        for t, v in zip(members, views):
            copies[id(t)] = v

    def rebuild(x):
        if torch.is_tensor(x):
            return copies[id(x)]
        if isinstance(x, dict):
            return {k: rebuild(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(rebuild(v) for v in x)
        return x
    return rebuild(results)


class _SceneSetupGraph(torch.autograd.Function):
    """Attaches the graph of a TRAINING call to the outputs of ``pr_scene_setup`` (one launch: pose matrices, projected boxes /
    points / axes, the renderer's inputs in the renderer's layouts): the object poses, style and deformation codes carry a graph
    (trainable encoders produce them), the cameras do not.  Backward: ``pr_scene_setup_backward`` - ONE launch from the renderer's
    input gradients to d rotations / d translations / d style / d deformation; the projected boxes, points and axes get their
    gradients (no shipped loss reads them) from the tensor formulation, recomputed only if one ever arrives (``_LazyGraph``)."""

    @staticmethod
    def forward(ctx, model, prepared, meta, object_rotations, object_translations, object_style, object_deformation):
        ctx.model, ctx.meta = model, meta
        ctx.set_materialize_grads(False)      # outputs no loss reads (boxes, points, axes) arrive as None, not as zero tensors
        ctx.save_for_backward(object_rotations, object_translations, object_style, object_deformation)
        r = prepared["renderer"]
        # (fresh tensor objects: the arena's views handed out as outputs of this node)
        return (prepared["boxes"].view_as(prepared["boxes"]), prepared["box_points"].view_as(prepared["box_points"]),
                prepared["axes"].view_as(prepared["axes"]), r["w2o"].view_as(r["w2o"]), r["style"].view_as(r["style"]),
                r["deformation"].view_as(r["deformation"]))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_boxes, g_points, g_axes, g_w2o, g_style, g_deformation):
        rot, tr, style, dfm = ctx.saved_tensors
        frames, cameras, K, S, D = ctx.meta["frames"], ctx.meta["cameras"], ctx.meta["objects"], ctx.meta["S"], ctx.meta["D"]
        need = ctx.needs_input_grad[3:]
        dev = rot.device
        f32 = dict(dtype=torch.float32, device=dev)
        out = torch.empty(frames * K * (6 + S + D), **f32)          # (one allocation for the four gradients)
        g_rot = out[:frames * K * 3].view(rot.shape)
        g_tr = out[frames * K * 3:frames * K * 6].view(tr.shape)
        g_sty = out[frames * K * 6:frames * K * (6 + S)].view(style.shape)
        g_dfm = out[frames * K * (6 + S):].view(dfm.shape)
        keep = [t.contiguous() for t in (g_w2o, g_style, g_deformation) if t is not None]
        it = iter(keep)
        ptrs = [next(it).data_ptr() if t is not None else None for t in (g_w2o, g_style, g_deformation)]
        with torch.cuda.device(dev):
            _lib.check(_lib.load().pr_scene_setup_backward(frames, cameras, K, S, D, rot.data_ptr(), tr.data_ptr(), ptrs[0], ptrs[1], ptrs[2],
                                                           g_rot.data_ptr(), g_tr.data_ptr(), g_sty.data_ptr(), g_dfm.data_ptr(),
                                                           torch.cuda.current_stream(dev).cuda_stream), "pr_scene_setup_backward")
        grads = [g_rot, g_tr, g_sty, g_dfm]
        if g_boxes is not None or g_points is not None or g_axes is not None:
            # a loss that reads the projected boxes / box points / axes: their gradients through the tensor formulation
            extra = ctx.model._projection_gradients(ctx.meta, rot, tr, g_boxes, g_points, g_axes)
            grads[0] = grads[0] + extra[0]
            grads[1] = grads[1] + extra[1]
        return (None, None, None) + tuple(g if n else None for g, n in zip(grads, need))


class EnvironmentModel(Tracked, nn.Module):

    def __init__(self, config, object_encoders=None, object_parameters_encoders=None, image_decoder=None, grid_sampler=None):
        super().__init__()
        self.config = config
        self.focal_length_multiplier = config["data"]["focal_length_multiplier"]
        self.use_weighted_sampling = config["model"].get("use_weighted_sampling", False)
        self.sampling_weights = config["model"].get("sampling_weights", None)
        self.enable_camera_parameters_offsets = config["model"].get("enable_camera_parameters_offsets", False)
        # always constructed, like the reference (environment_model.py:36-39): its entries are part of the checkpoint
        cameras = config.get("training", {}).get("batching", {}).get("allowed_cameras", [0])
        self.training_cameras_count = len(cameras)
        self.camera_parameters_offsets = CameraParametersStorage(config["model"].get("camera_parameters_memory_size", 1),
                                                                 self.training_cameras_count)
        # config["model"]["image_decoder"] / ["grid_sampler"] (environment_model.py:53-56, 125-149): CNNs outside the renderer's
        # path; like the encoders they are injected (stock PyTorch modules with the reference's call contracts)
        self.use_image_decoder = "image_decoder" in config["model"]
        self.image_decoder = image_decoder
        self.grid_sampler = grid_sampler
        self.object_composer = ObjectComposer(config)
        self.object_id_helper = ObjectIDsHelper(config)
        # the reference's attribute names (environment_model.py:44-50); empty until encoders are injected
        self.object_parameters_encoders = ModuleList()
        self.object_encoders = ModuleList()
        if object_encoders is not None or object_parameters_encoders is not None:
            self.set_encoders(object_encoders, object_parameters_encoders)
        elif all("architecture" in e for e in config["model"].get("object_encoders", [{}])) and \
                all("architecture" in e for e in config["model"].get("object_parameters_encoder", [{}])):
            # a full configuration (the reference's YAML): build the encoders like the reference's constructor does
            # (environment_model.py:44-50), from this package's modules
            from .encoders import create_encoders
            self.set_encoders(*create_encoders(config))
        self.current_step = 0
        #: evaluation calls without a graph run their scene set-up - pose matrices, projected boxes / points / axes, the renderer's
        #: input layouts - as ONE launch (pr_scene_setup) instead of four kernels and ~10 small copies; results are bit-identical
        self.fused_scene_setup = True
        #: Automatic capture-and-replay of EVALUATION frames.  A call of ``forward_from_scene_encoding`` /
        #: ``forward_from_observations`` (so ``render_full_frame_*``, what the reference's evaluators and dataset creators call:
        #: evaluation/reconstructed_dataset_creator.py:121, evaluation/evaluator.py:58,95) under ``torch.no_grad()`` in eval mode
        #: without perturbation and with a static pixel selection (``samples_per_image == 0``: every pixel or the strided grids) is
        #: recorded per (mode, argument shapes, options, weights) as a HIP graph the SECOND time it is seen (the first call runs
        #: eagerly and warms the recording up) and replayed afterwards: the unchanged evaluator / play-loop code then costs ~0.1 -
        #: 1.3 ms of host time per frame instead of 0.4 - 11 ms (the observation mode's CNN encoders are ~300 small launches).
        #: "clone" (DEFAULT): every tensor of the result dictionary is copied out of the recording - for the caller
        #: indistinguishable from the eager call (same kernels, same launch order: the renderer's part is bit-identical);
        #: "alias": the dictionary holds the recording's static tensors, valid until the next call with the same shapes (an
        #: evaluator that writes its images before it renders the next batch); None: always eager.  A recording is dropped when the
        #: weights, a module registration, the precision, the annealing step or any switch it baked in changes (``_replay_signature``);
        #: a call that cannot be recorded (an injected module that reads back to the host) is detected once and stays eager - and so
        #: is a recording that holds MEMSET nodes (``pr_graph_node_census``; the renderer and this package's encoders record none,
        #: an injected decoder with a large ``mean`` / ``sum`` does): on ROCm 7.0.2 those stop executing after a host
        #: synchronisation between replays unless ``frame_graph.GRAPH_RUNTIME_SWITCH`` is in the environment.
        self.frame_replay = "clone"
        #: recordings kept (one per mode / shape / option combination; the oldest is dropped)
        self.frame_replay_slots = 4
        self._replays: Dict = {}
        self._in_replay = False
        # per-device constants of the host path (pixel lists of full-frame / strided-grid renders, box points)
        self._pixel_cache: Dict = {}
        self._edge_point_cache: Dict = {}
        self._axes_point_cache: Dict = {}

    def set_step(self, current_step: int):
        self.current_step = current_step
        self.object_composer.set_step(current_step)

    def set_encoders(self, object_encoders=None, object_parameters_encoders=None):
        """Injects the (stock PyTorch) modules that turn observations into the renderer's inputs; one per OBJECT MODEL, in
        the order of ``config["model"]["object_models"]``, with the call contracts of the reference's modules:

        * ``object_encoders[m](observations, bounding_box (..., O, C, 4), camera_rotations, camera_translations,
          global_frame_indexes, video_frame_indexes, video_indexes) -> (style (..., O, S), deformation (..., O, D),
          attention, crops)``  (model/object_encoder_v4.py:80-178; environment_model.py:449-450);
        * ``object_parameters_encoders[m]``: static models ``(observations) -> (rotations (..., O, 3, n_m), translations
          (..., O, 3, n_m))``; dynamic models ``(observations, transformation_matrix_w2c, camera_rotations, focals,
          bounding_boxes (..., O, C, 4, n_m), bounding_boxes_validity (..., O, C, n_m)) -> (rotations, translations)``
          (model/classic_object_parameters_encoder.py:129-237; environment_model.py:178-191)."""
        if object_encoders is not None:
            self.object_encoders = ModuleList(list(object_encoders))
        if object_parameters_encoders is not None:
            self.object_parameters_encoders = ModuleList(list(object_parameters_encoders))
        return self

    def create_object_encoders(self) -> List[nn.Module]:
        """One style / deformation encoder per object model, from ``config["model"]["object_encoders"]``
        (model/environment_model.py:109-123; built by this package's ``encoders`` module)."""
        from .encoders import create_encoders
        return create_encoders(self.config)[0]

    def create_object_parameters_encoders(self) -> List[nn.Module]:
        """One pose encoder per object model, from ``config["model"]["object_parameters_encoder"]``
        (model/environment_model.py:93-107)."""
        from .encoders import create_encoders
        return create_encoders(self.config)[1]

    def set_image_decoder(self, image_decoder, grid_sampler):
        """``grid_sampler(integrated_features (..., R, F), sampled_positions (..., R, 2)) -> grid`` and
        ``image_decoder(grid) -> (..., output features, height, width)`` (environment_model.py:733-741)."""
        self.image_decoder, self.grid_sampler = image_decoder, grid_sampler
        return self

    def compute_decoded_image(self, composition_results: Dict, sampled_positions: torch.Tensor):
        """Decodes the composited coarse features into an image, stored as ``coarse.global.decoded_images``
        (model/environment_model.py:708-741; same exceptions)."""
        if not self.use_image_decoder:
            raise Exception("Image decoding was requested, but the use of the image decoder was not configured")
        if "fine" in composition_results:
            raise Exception("Image decoding is being used only on the coarse features, but fine features are being computed anyway. "
                            "Please disable the fine nerf models.")
        if self.image_decoder is None or self.grid_sampler is None:
            raise RuntimeError("config['model']['image_decoder'] is set: inject the decoder CNN and the grid sampler with "
                               "EnvironmentModel(config, ..., image_decoder=..., grid_sampler=...) or set_image_decoder(...) "
                               "(they are not part of this package)")
        integrated_features = composition_results["coarse"]["global"]["integrated_features"]
        sampled_grid = self.grid_sampler(integrated_features, sampled_positions)
        composition_results["coarse"]["global"]["decoded_images"] = self.image_decoder(sampled_grid)

    def _require_encoders(self):
        want = self.object_id_helper.object_models_count
        if len(self.object_encoders) != want or len(self.object_parameters_encoders) != want:
            raise RuntimeError(
                f"the observation-driven modes need one object encoder and one object-parameters encoder per object model "
                f"({want}); this package does not contain the reference's CNN encoders - inject them with "
                "EnvironmentModel(config, object_encoders=..., object_parameters_encoders=...) or set_encoders(...), or "
                "encode the scene elsewhere and call mode='scene_encodings'")

    def get_object_encoder_parameters(self):
        """model/environment_model.py:67-68 (the trainers give the object encoders their own parameter group)."""
        return self.object_encoders.parameters()

    def get_camera_offsets_parameters(self):
        """model/environment_model.py:70-71 (the trainers' second optimiser)."""
        return self.camera_parameters_offsets.parameters()

    def get_main_parameters(self, additional_excluded_parameters=None):
        """Every parameter that belongs neither to the object encoders nor to the camera offsets, minus the names in
        ``additional_excluded_parameters`` (how the autoencoder subclasses take their CNN out: their override calls this one) -
        what ``Trainer.get_optimizer`` hands to Adam (model/environment_model.py:73-91, training/trainer.py:105-111)."""
        excluded = set(["object_encoders." + name for name, _ in self.object_encoders.named_parameters()] +
                       ["camera_parameters_offsets." + name for name, _ in self.camera_parameters_offsets.named_parameters()])
        if additional_excluded_parameters is not None:
            excluded = excluded.union(additional_excluded_parameters)
        return [p for name, p in self.named_parameters() if name not in excluded]

    def _corrected_cameras(self, camera_rotations, camera_translations, focals, global_frame_indexes, focal_quirk=False):
        """Adds the learnable per-frame camera offsets (environment_model.py:891-897, 1235-1241, 1408-1414).
        ``focal_quirk``: the scene-encoding-only mode adds the ROTATION offsets to the focals (environment_model.py:798);
        kept as the reference has it, broadcasting rules and all."""
        if not self.enable_camera_parameters_offsets:
            return camera_rotations, camera_translations, focals
        rotation_offsets, translation_offsets, focal_offsets = self.camera_parameters_offsets(global_frame_indexes)
        return (camera_rotations + rotation_offsets, camera_translations + translation_offsets,
                focals + (rotation_offsets if focal_quirk else focal_offsets))

    # ------------------------------------------------------------------ modes
    def forward(self, *args, mode="observations", **kwargs):
        """model/environment_model.py:743-770."""
        if mode == "observations":
            return self.forward_from_observations(*args, **kwargs)
        if mode == "scene_encodings":
            return self.forward_from_scene_encoding(*args, **kwargs)
        if mode == "observations_scene_encoding_only":
            return self.forward_scene_encoding_from_observations(*args, **kwargs)
        if mode == "pose_consistency":
            return self.forward_pose_consistency(*args, **kwargs)
        if mode == "keypoint_consistency":
            return self.forward_keypoint_consistency(*args, **kwargs)
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
        o2w, w2o = pose_matrices(object_rotation_parameters_o2w.movedim(-1, -2), object_translation_parameters_o2w.movedim(-1, -2))
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

    @staticmethod
    def _project_on_device(points: torch.Tensor, o2w: torch.Tensor, w2c: torch.Tensor, focals: torch.Tensor, height: int,
                           width: int, with_boxes: bool):
        """``_project`` + normalisation (+ box reduction and clamps) as ONE kernel (pr_project_points) for device tensors without
        a graph: points (K, P, 3), o2w (..., 4, 4, K), w2c (..., C, 4, 4), focals (..., C) -> projected (..., C, P, 2, K)
        and, with_boxes, boxes (..., C, 4, K)."""
        lead = list(w2c.shape[:-3])
        cameras, K, P = w2c.size(-3), points.size(0), points.size(1)
        n = int(math.prod(lead)) if lead else 1
        f32 = dict(dtype=torch.float32, device=w2c.device)
        mo = o2w.detach().to(torch.float32).movedim(-1, -3).reshape(n, K, 4, 4).contiguous()
        mc = w2c.detach().to(torch.float32).reshape(n, cameras, 4, 4).contiguous()
        fc = torch.broadcast_to(focals.detach().to(torch.float32), lead + [cameras]).reshape(n, cameras).contiguous()
        projected = torch.empty((n, cameras, P, 2, K), **f32)
        boxes = torch.empty((n, cameras, 4, K), **f32) if with_boxes else None
        with torch.cuda.device(w2c.device):
            _lib.check(_lib.load().pr_project_points(n, cameras, K, P, points.contiguous().data_ptr(), mo.data_ptr(), mc.data_ptr(),
                                                     fc.data_ptr(), height, width, projected.data_ptr(),
                                                     boxes.data_ptr() if with_boxes else None,
                                                     torch.cuda.current_stream(w2c.device).cuda_stream), "pr_project_points")
        projected = projected.reshape(lead + [cameras, P, 2, K])
        return projected, (boxes.reshape(lead + [cameras, 4, K]) if with_boxes else None)

    @staticmethod
    def _no_graph(*tensors) -> bool:
        return all(t.is_cuda for t in tensors) and not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))

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

    def _scene_setup(self, camera_rotations, camera_translations, focals, object_rotations, object_translations, object_style,
                     object_deformation, object_in_scene, height: int, width: int, upsample_factor: float):
        """pr_scene_setup for an evaluation call (no graph, fp32 device tensors in the reference's layouts); None when the call
        does not qualify (the tensor route then runs)."""
        tensors = (camera_rotations, camera_translations, focals, object_rotations, object_translations, object_style, object_deformation)
        if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
            return None
        if not (torch.is_tensor(object_in_scene) and object_in_scene.is_cuda and object_in_scene.dtype == torch.bool
                and object_in_scene.is_contiguous()):
            return None
        lead = list(camera_rotations.shape[:-1])                    # (..., O, C)
        K = self.object_id_helper.objects_count
        cameras = lead[-1]
        outer = lead[:-1]
        S, D = object_style.size(-2), object_deformation.size(-2)
        if (list(camera_translations.shape) != lead + [3] or list(focals.shape) != lead or
                list(object_rotations.shape) != outer + [3, K] or list(object_translations.shape) != outer + [3, K] or
                list(object_style.shape) != outer + [S, K] or list(object_deformation.shape) != outer + [D, K] or
                list(object_in_scene.shape) != outer + [K] or K > _lib.PR_MAX_OBJECTS):
            return None
        dev = camera_rotations.device
        frames = int(math.prod(outer)) if outer else 1
        n = frames * cameras
        points = self._edge_points(dev)
        key = ("axes_points", str(dev))
        if key not in self._pixel_cache:
            self._pixel_cache[key] = torch.tensor([(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)],
                                                  device=dev).unsqueeze(0).repeat(K, 1, 1).contiguous()
        axes_points = self._pixel_cache[key]
        f32 = dict(dtype=torch.float32, device=dev)
        P = points.size(1)
        # one arena for the fp32 outputs
        sizes = [n * 4 * K, n * P * 2 * K, n * 8 * K, n * 12, n, n * K * 12, n * K * S, n * K * D]
        arena = torch.empty(sum(sizes), **f32)
        parts, at = [], 0
        for size in sizes:
            parts.append(arena[at:at + size])
            at += size
        boxes, projected, axes, cam34, render_focals, w2o34, sty, dfm = parts
        present = torch.empty((n, K), dtype=torch.uint8, device=dev)
        q = _lib.SceneSetup()
        q.frames, q.cameras, q.objects, q.box_points_per_object = frames, cameras, K, P
        q.style_features, q.deformation_features, q.height, q.width = S, D, height, width
        q.focal_multiplier, q.upsample_factor, q.axes_with_upsampled_focals = float(self.focal_length_multiplier), float(upsample_factor), 0
        q.camera_rotations, q.camera_translations, q.focals = camera_rotations.data_ptr(), camera_translations.data_ptr(), focals.data_ptr()
        q.object_rotations, q.object_translations = object_rotations.data_ptr(), object_translations.data_ptr()
        q.style, q.deformation, q.object_in_scene = object_style.data_ptr(), object_deformation.data_ptr(), object_in_scene.data_ptr()
        q.box_points, q.axes_points = points.data_ptr(), axes_points.data_ptr()
        q.boxes, q.projected_points, q.axes = boxes.data_ptr(), projected.data_ptr(), axes.data_ptr()
        q.camera34, q.render_focals = cam34.data_ptr(), render_focals.data_ptr()
        q.w2o34, q.style_nks, q.deformation_nkd, q.present = w2o34.data_ptr(), sty.data_ptr(), dfm.data_ptr(), present.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().pr_scene_setup(C.byref(q), torch.cuda.current_stream(dev).cuda_stream), "pr_scene_setup")
        return {"boxes": boxes.view(lead + [4, K]), "box_points": projected.view(lead + [P, 2, K]), "axes": axes.view(lead + [4, 2, K]),
                "camera34": cam34.view(n, 3, 4), "render_focals": render_focals,
                "renderer": {"w2o": w2o34.view(n, K, 3, 4), "style": sty.view(n, K, S), "deformation": dfm.view(n, K, D),
                             "present": present, "frames": n, "S": S, "D": D}}

    def _projection_gradients(self, meta, rotations, translations, g_boxes, g_points, g_axes):
        """d (projected boxes, box points, axes) / d (object rotations, translations) for ``_SceneSetupGraph``: the differentiable
        tensor formulation of the projections, recomputed on the saved inputs (only when a loss reads those outputs)."""
        cam_rot, cam_tr, focals = meta["cameras_tensors"]
        with torch.enable_grad():
            rot = rotations.detach().requires_grad_(True)
            tr = translations.detach().requires_grad_(True)
            rescaled = focals * self.focal_length_multiplier
            render_focals = rescaled if meta["upsample"] == 1.0 else rescaled * meta["upsample"]
            _, w2c = pose_matrices(cam_rot, cam_tr)
            _, o2w = self.compute_transformation_matrix_w2o_o2w(rot, tr)
            boxes, points = self.compute_object_bounding_boxes(o2w, w2c, render_focals, meta["height"], meta["width"], _lazy=True)
            axes = self.compute_object_axes_projection(o2w, w2c, rescaled, meta["height"], meta["width"], _lazy=True)
            pairs = [(o, g) for o, g in ((boxes, g_boxes), (points, g_points), (axes, g_axes)) if g is not None]
            got = torch.autograd.grad([o for o, _ in pairs], [rot, tr], [g.reshape(o.shape) for o, g in pairs], allow_unused=True)
        return [g if g is not None else torch.zeros_like(t) for g, t in zip(got, (rotations, translations))]

    def compute_object_bounding_boxes(self, transformation_matrix_o2w, transformation_matrix_w2c, focals, height, width, _lazy=False):
        """Image-plane boxes (..., C, 4, K) [left, top, right, bottom] and projected box points
        (..., C, 68, 2, K), normalised to [0, 1].  model/environment_model.py:234-327."""
        if transformation_matrix_o2w.dim() > transformation_matrix_w2c.dim():
            transformation_matrix_o2w = transformation_matrix_o2w[..., 0, :, :, :]
        if self._no_graph(transformation_matrix_o2w, transformation_matrix_w2c, focals):
            points, boxes = self._project_on_device(self._edge_points(transformation_matrix_o2w.device), transformation_matrix_o2w,
                                                    transformation_matrix_w2c, focals, height, width, with_boxes=True)
            return boxes, points
        if not _lazy and transformation_matrix_o2w.is_cuda and transformation_matrix_w2c.is_cuda and torch.is_tensor(focals) and focals.is_cuda:
            # a graph is wanted: the values from the kernel, the gradients (if a loss ever reads the boxes) from the tensor ops below
            return _LazyGraph.apply(
                lambda a, b, c: self.compute_object_bounding_boxes(a, b, c, height, width),
                lambda a, b, c: self.compute_object_bounding_boxes(a, b, c, height, width, _lazy=True),
                transformation_matrix_o2w, transformation_matrix_w2c, focals)
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

    def compute_object_axes_projection(self, transformation_matrix_o2w, transformation_matrix_w2c, focals, height, width, _lazy=False):
        """Projected origin + unit axes of every object (..., C, 4, 2, K), normalised, not clamped.
        model/environment_model.py:329-404."""
        if transformation_matrix_o2w.dim() > transformation_matrix_w2c.dim():
            transformation_matrix_o2w = transformation_matrix_o2w[..., 0, :, :, :]
        key = str(transformation_matrix_o2w.device)
        if key not in self._axes_point_cache:
            self._axes_point_cache[key] = torch.tensor([(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)],
                                                       device=transformation_matrix_o2w.device)
        pts = self._axes_point_cache[key].unsqueeze(0).expand(self.object_id_helper.objects_count, 4, 3)
        if self._no_graph(transformation_matrix_o2w, transformation_matrix_w2c, focals):
            return self._project_on_device(pts, transformation_matrix_o2w, transformation_matrix_w2c, focals, height, width,
                                           with_boxes=False)[0]
        if not _lazy and transformation_matrix_o2w.is_cuda and transformation_matrix_w2c.is_cuda and torch.is_tensor(focals) and focals.is_cuda:
            return _LazyGraph.apply(
                lambda a, b, c: self.compute_object_axes_projection(a, b, c, height, width),
                lambda a, b, c: self.compute_object_axes_projection(a, b, c, height, width, _lazy=True),
                transformation_matrix_o2w, transformation_matrix_w2c, focals)
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
                                 video_indexes=None, canonical_pose: bool = False, _decoder_layout=None, _prepared=None):
        """model/environment_model.py:474-521.  The reference chunks rays (1000 per call in full-frame rendering) because it
        materialises (rays, samples, 192) tensors; the fused renderer does not need to.  In EVALUATION mode
        ``samples_per_image_batching`` is therefore accepted and ignored - rays are independent and the BatchNorm layers use
        their running statistics, so one call returns what the chunked calls return (the composer splits a call only if its
        scratch would exceed its workspace budget).  In TRAINING mode the chunks are part of the semantics: the reference runs
        the composer once per chunk, so every chunk normalises with ITS OWN batch statistics, and the running statistics and
        ``num_batches_tracked`` advance once per chunk - reproduced here chunk for chunk (TensorBatchifier.batchify:
        consecutive ranges of ``samples_per_image_batching`` rays, the last one shorter)."""
        extra = {} if _decoder_layout is None else {"_decoder_layout": _decoder_layout}
        if _prepared is not None:
            extra["_prepared"] = _prepared        # (pr_scene_setup's outputs: the composer skips its own marshalling)
        dimension = ray_directions.dim() - 2
        rays = ray_directions.size(dimension)
        if self.object_composer.training and 0 < samples_per_image_batching < rays:
            if _decoder_layout is not None:
                raise ValueError("decoder-layout emission needs all rays of the call in one composer call "
                                 "(training mode with samples_per_image_batching > 0 runs one call per ray chunk)")
            chunks = []
            for begin in range(0, rays, samples_per_image_batching):
                current = ray_directions.narrow(dimension, begin, min(samples_per_image_batching, rays - begin))
                chunks.append(self.object_composer(ray_origins, current, focal_normals, transformation_matrix_w2o, style,
                                                   deformation, object_in_scene, perturb, video_indexes=video_indexes,
                                                   canonical_pose=canonical_pose, **({} if _prepared is None else {"_prepared": _prepared})))
            return self.merge_dictionaries(chunks, dimension=dimension)
        results = self.object_composer(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                       deformation, object_in_scene, perturb, video_indexes=video_indexes,
                                       canonical_pose=canonical_pose, **extra)
        return self.merge_dictionaries([results], dimension=dimension)

    @staticmethod
    def decoder_layout(height: int, width: int, samples_per_image: int, patch_size: int, patch_stride, features_by_layer):
        """Ray groups of a strided render for ``ObjectComposer.forward(_decoder_layout=...)``: one group per stride (the
        strided grids of a full frame, or the strided patches of a training call - RayHelper.sample_all_rays_strided_grid /
        sample_rays_strided_patch, smallest stride first), group i owning the next ``features_by_layer[i]`` channels
        (split_features_by_layer, model/environment_model_multiresolution_backpropagated_autoencoder.py:29-61)."""
        strides = list(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else [int(patch_stride)]
        counts = [int(c) for c in features_by_layer]
        if not patch_stride or len(strides) != len(counts):
            raise ValueError(f"decoder layout: {len(counts)} feature groups need as many strides, got patch_stride={patch_stride}")
        if patch_size != 0 and samples_per_image != 0:
            sides = [(patch_size * strides[0]) // s for s in strides]
            return {"rays": [p * p for p in sides], "width": sides, "channels": counts}
        if samples_per_image != 0:
            raise ValueError("decoder layout needs a strided full-frame render or a strided patch")
        return {"rays": [(height // s) * (width // s) for s in strides], "width": [width // s for s in strides], "channels": counts}

    def merge_dictionaries(self, dictionaries: List[Dict], dimension: int):
        merged = {}
        for key in list(dictionaries[0].keys()):
            if key == "pytorch_hook":
                continue
            if torch.is_tensor(dictionaries[0][key]):
                parts = [d[key] for d in dictionaries]
                merged[key] = parts[0] if len(parts) == 1 else torch.cat(parts, dim=dimension)
            elif isinstance(dictionaries[0][key], (list, tuple)):
                # per-stride maps of the decoder layout: whole-call outputs, never split along the rays
                if len(dictionaries) != 1:
                    raise ValueError(f"'{key}' cannot be merged across ray chunks")
                merged[key] = dictionaries[0][key]
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

    # ------------------------------------------------------------------ automatic frame replay
    def __getstate__(self):
        # copy.deepcopy / pickle: recorded graphs and per-device caches stay with the original
        state = dict(self.__dict__)
        state.update(_replays={}, _in_replay=False, _pixel_cache={}, _edge_point_cache={}, _axes_point_cache={})
        return state

    def _replay_signature(self, name: str):
        """Everything a recorded evaluation frame baked in besides its input buffers: the storages AND values of the renderer's
        weights (the packed MFMA copies are made outside the recording), the storages of everything else the call reads through a
        raw pointer (the encoders' / the decoder's parameters and buffers), and the switches that select kernels or host branches."""
        composer = self.object_composer
        owner = None if name == "scene_encodings" else self       # (the renderer-only mode does not read the encoders' weights)
        params = composer._parameter_list(owner)                    # cached lists; walked every call when the tree holds foreign modules
        root = composer if owner is None else self
        if composer._tree_is_tracked(root):
            tree = _REGISTRATION_EPOCH[0]                           # a replaced buffer / parameter / submodule moves it
        else:
            tree = tuple(b.data_ptr() for b in root.buffers())
        decoder = None
        if self.use_image_decoder and name == "scene_encodings":
            decoder = tuple(t.data_ptr() for m in (self.image_decoder, self.grid_sampler) if isinstance(m, nn.Module)
                            for t in list(m.parameters()) + list(m.buffers()))
        return (tuple((p.data_ptr(), p._version) for p in params), tree, decoder, composer.precision, bool(composer.gate_feature_head),
                composer.state_epoch, composer.weights_epoch,
                None if composer.object_entry_fields is None else tuple(composer.object_entry_fields),
                bool(self.fused_scene_setup), bool(composer.training), bool(composer.use_naive_mlp), float(self.focal_length_multiplier),
                int(composer.max_workspace_bytes))

    def _replayed(self, name: str, method, tensors, statics: tuple):
        """The evaluation call ``method(*tensors, *statics...)`` through a recorded graph (see ``frame_replay``).  Returns None when
        the call has to run eagerly: the first call of a (mode, shapes, options) combination (it doubles as the recording's warm-up:
        a one-off render costs what it cost before), or a combination whose recording failed."""
        from .frame_graph import CapturedCall
        key = (name, tuple((tuple(t.shape), t.dtype, str(t.device)) for t in tensors), statics)
        signature = self._replay_signature(name)
        entry = self._replays.get(key)
        if entry is not None and entry[0] != signature:
            del self._replays[key]                              # stale: weights / precision / step changed since (re-record below)
            entry = None
        if entry is None:
            while len(self._replays) >= max(1, int(self.frame_replay_slots)):        # a handful of frame shapes at most: drop the oldest one
                self._replays.pop(next(iter(self._replays)))
            self._replays[key] = (signature, None)              # seen once: the next call with this signature records
            return None
        if entry[1] is None:
            self._in_replay = True
            try:
                recorded = CapturedCall(lambda *ts: method(*ts), list(tensors), warmup=1, what=f"frame_replay ({name})")
            except Exception as error:             # a module in front of the renderer that cannot be recorded (host reads, ...)
                torch.cuda.synchronize()
                warnings.warn(f"frame_replay: recording the {name} evaluation call failed ({type(error).__name__}: {error}); calls of "
                              "this shape run eagerly from now on", RuntimeWarning)
                recorded = False
            finally:
                self._in_replay = False
            # (raw pointers recorded: the workspace and the packed weights stay alive with the entry)
            entry = (signature, recorded, self.object_composer._workspace, [e[1] for e in self.object_composer._packed.values()])
            self._replays[key] = entry
        if entry[1] is False:
            return None
        results = entry[1].replay(tensors)
        if self.frame_replay == "clone":
            return _copy_out(results)
        return results

    def _replay_wanted(self, tensors, perturb, samples_per_image) -> bool:
        if self.frame_replay is None or self._in_replay or self.__dict__.get("_is_replica"):
            return False
        if self.frame_replay not in ("alias", "clone"):
            raise ValueError(f"unknown frame_replay {self.frame_replay!r} (expected None, 'alias' or 'clone')")
        return (not torch.is_grad_enabled() and not self.training and not self.object_composer.training and not perturb and
                samples_per_image == 0 and all(torch.is_tensor(t) and t.is_cuda for t in tensors) and
                not torch.cuda.is_current_stream_capturing())

    def _replicate_for_data_parallel(self):
        # nn.DataParallel: per-call replicas never record, and must not share the recordings / per-device constants of the original
        replica = super()._replicate_for_data_parallel()
        replica.__dict__.update(_replays={}, _in_replay=False, _pixel_cache={}, _edge_point_cache={}, _axes_point_cache={})
        return replica

    # ------------------------------------------------------------------ scene encoding -> rays -> composer
    def forward_from_scene_encoding(self, camera_rotations, camera_translations, focals, image_size,
                                    object_rotation_parameters_o2w, object_translation_parameters_o2w, object_style,
                                    object_deformation, object_in_scene, samples_per_image: int, perturb: bool,
                                    samples_per_image_batching: int = 0, upsample_factor: float = 1.0,
                                    patch_size: int = 0, patch_stride=0, canonical_pose: bool = False,
                                    _ray_range: Tuple[int, int] = None, _decoder_features=None) -> Dict:
        """model/environment_model.py:1041-1158; argument shapes documented there.

        camera_* (..., O, C, 3); focals (..., O, C); object_* (..., O, 3|S|D, K); object_in_scene (..., O, K).
        ``_ray_range`` (extension): render only the rays [begin, end) of the pixel list - one rank's contiguous share of
        a frame in ``render_sharded`` - or, given an int64 tensor, the listed rays (its share of interleaved tiles).  ``_decoder_features`` (extension): the decoder's feature count per stride, e.g.
        [64, 128] - the compositing kernel then also writes ``[type]["global"]["decoder_features"]``, the channels-first
        per-stride maps ``autoencoder_model.forward_decoder`` takes (see ``decoder_layout``)."""
        scene = (camera_rotations, camera_translations, focals, object_rotation_parameters_o2w, object_translation_parameters_o2w,
                 object_style, object_deformation, object_in_scene)
        if _ray_range is None and self._replay_wanted(scene, perturb, samples_per_image):
            stride_key = tuple(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else patch_stride
            statics = (tuple(image_size), samples_per_image_batching, upsample_factor, patch_size, stride_key, canonical_pose,
                       None if _decoder_features is None else tuple(_decoder_features))
            replayed = self._replayed(
                "scene_encodings",
                lambda a, b, c, d, e, f, g, h: self.forward_from_scene_encoding(
                    a, b, c, image_size, d, e, f, g, h, samples_per_image, perturb, samples_per_image_batching, upsample_factor,
                    patch_size, patch_stride, canonical_pose, _decoder_features=_decoder_features),
                scene, statics)
            if replayed is not None:
                return replayed
        height = int(image_size[0] * upsample_factor)
        width = int(image_size[1] * upsample_factor)
        prepared = None
        scene_graph = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in scene)
        cameras_graph = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in scene[:3])
        if self.fused_scene_setup and not cameras_graph:
            # (learnable camera offsets - a graph on the cameras - take the tensor route below)
            prepared = self._scene_setup(*[t.detach() if torch.is_tensor(t) else t for t in scene[:-1]], object_in_scene,
                                         height, width, upsample_factor)
        if prepared is not None and scene_graph:
            # a training call: the same launch, with the graph of the object poses / style / deformation attached to its outputs
            renderer = prepared["renderer"]
            meta = dict(frames=renderer["frames"] // camera_rotations.shape[-2], cameras=camera_rotations.shape[-2],
                        objects=self.object_id_helper.objects_count, S=renderer["S"], D=renderer["D"], height=height, width=width,
                        upsample=upsample_factor, cameras_tensors=(camera_rotations.detach(), camera_translations.detach(), focals.detach()))
            boxes_g, points_g, axes_g, w2o_g, sty_g, dfm_g = _SceneSetupGraph.apply(
                self, prepared, meta, object_rotation_parameters_o2w, object_translation_parameters_o2w, object_style, object_deformation)
            prepared = dict(prepared, boxes=boxes_g, box_points=points_g, axes=axes_g,
                            renderer=dict(renderer, w2o=w2o_g, style=sty_g, deformation=dfm_g))
        if prepared is not None:
            # one launch (pr_scene_setup): pose matrices, projected boxes / points / axes, the renderer's inputs in its layouts
            boxes, box_points, axes = prepared["boxes"], prepared["box_points"], prepared["axes"]
            c2w = w2o = None
        else:
            rescaled_focals = focals * self.focal_length_multiplier
            # (x * 1.0 is x bit for bit: the common case issues no launch for it)
            render_focals = rescaled_focals if upsample_factor == 1.0 else rescaled_focals * upsample_factor
            c2w, w2c = pose_matrices(camera_rotations, camera_translations)
            w2o, o2w = self.compute_transformation_matrix_w2o_o2w(object_rotation_parameters_o2w,
                                                                  object_translation_parameters_o2w)
            boxes, box_points = self.compute_object_bounding_boxes(o2w, w2c, render_focals, height, width)
            axes = self.compute_object_axes_projection(o2w, w2c.detach(), rescaled_focals.detach(), height, width)

        lead = list(camera_rotations.shape[:-1])
        flat_boxes = boxes.reshape(-1, 4, boxes.size(-1))
        if patch_size != 0 and samples_per_image != 0:
            rows, cols = ray_sampling.strided_patch_rows_cols(flat_boxes, self.sampling_weights, height, width, patch_size, patch_stride)
            rows, cols = rows.reshape(lead + [-1]), cols.reshape(lead + [-1])
        elif samples_per_image == 0:
            # static pixel lists (every pixel, or the strided grids): built once per (size, strides, device)
            strides = tuple(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else (int(patch_stride),)
            key = (height, width, strides if patch_stride else None, str(camera_rotations.device))
            if key not in self._pixel_cache:
                if patch_stride:
                    rows, cols = strided_grid_pixels(height, width, patch_stride)
                else:
                    r = torch.arange(height * width, dtype=torch.int32)
                    rows, cols = r // width, r % width
                self._pixel_cache[key] = (rows.to(device=camera_rotations.device, dtype=torch.int32).contiguous(),
                                          cols.to(device=camera_rotations.device, dtype=torch.int32).contiguous())
            rows, cols = self._pixel_cache[key]
        elif self.use_weighted_sampling:
            idx = ray_sampling.sample_pixels_weighted(flat_boxes, self.sampling_weights, height, width, samples_per_image)
            rows, cols = ray_sampling.split_indices(idx.reshape(lead + [-1]), width)
        else:
            idx = ray_sampling.sample_pixels_uniform(flat_boxes.size(0), height, width, samples_per_image, boxes.device)
            rows, cols = ray_sampling.split_indices(idx.reshape(lead + [-1]), width)

        if _ray_range is not None and torch.is_tensor(_ray_range):
            rows, cols = rows.index_select(-1, _ray_range), cols.index_select(-1, _ray_range)
        elif _ray_range is not None:
            rows, cols = rows[..., _ray_range[0]:_ray_range[1]], cols[..., _ray_range[0]:_ray_range[1]]
        if prepared is not None:
            origins, directions, normals = _camera_rays_prepared(prepared["camera34"], prepared["render_focals"], lead, height, width,
                                                                 rows, cols)
        else:
            origins, directions, normals = camera_rays(c2w, render_focals, height, width, rows, cols)

        layout = None
        if _decoder_features is not None:
            if _ray_range is not None:
                raise ValueError("decoder-layout emission needs all rays of the frame in one call")
            layout = self.decoder_layout(height, width, samples_per_image, patch_size, patch_stride, _decoder_features)
        results = self.batchified_composer_call(origins, directions, normals, w2o, object_style.unsqueeze(-3),
                                                object_deformation.unsqueeze(-3), object_in_scene.unsqueeze(-2),
                                                perturb, samples_per_image_batching, canonical_pose=canonical_pose,
                                                _decoder_layout=layout, _prepared=None if prepared is None else prepared["renderer"])
        if self.use_image_decoder:
            flat = rows.to(torch.int64) * width + cols.to(torch.int64)
            flat = flat.to(origins.device)
            flat = flat if flat.dim() > 1 else flat.expand(lead + [flat.numel()])
            self.compute_decoded_image(results, ray_sampling.positions_from_indices(flat, height, width))
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

    # ------------------------------------------------------------------ observation-driven modes (injected encoders)
    def compute_rotation_translation_o2w(self, observations, transformation_matrix_w2c, camera_rotations, focals,
                                         bounding_boxes, bounding_boxes_validity):
        """Object poses from the injected object-parameters encoders, concatenated along the object dimension
        (model/environment_model.py:161-204; here the camera argument is the w2c MATRIX (..., C, 4, 4), not a
        PoseParameters object)."""
        rotations, translations = [], []
        helper = self.object_id_helper
        for m in range(helper.object_models_count):
            encoder = self.object_parameters_encoders[m]
            if helper.is_static(m):
                r, t = encoder(observations)
            else:
                b, e = helper.dynamic_object_idx_range_by_model_idx(m)
                r, t = encoder(observations, transformation_matrix_w2c, camera_rotations, focals, bounding_boxes[..., b:e],
                               bounding_boxes_validity[..., b:e])
            rotations.append(r)
            translations.append(t)
        return torch.cat(rotations, dim=-1), torch.cat(translations, dim=-1)

    def compute_object_encodings(self, observations, camera_rotations, camera_translations, bounding_boxes,
                                 reconstructed_bounding_boxes, global_frame_indexes, video_frame_indexes, video_indexes,
                                 shuffle_style: bool):
        """Style / deformation codes of every object instance from the injected object encoders: static objects are
        cropped with their reconstructed box, dynamic ones with the annotated box; ``shuffle_style`` permutes the style codes
        along the observations dimension with a permutation that is not the identity (environment_model.py:406-472)."""
        video_indexes = video_indexes.unsqueeze(-1)
        video_indexes, _ = torch.broadcast_tensors(video_indexes, video_frame_indexes)
        helper = self.object_id_helper
        style, deformation, attention, crops = [], [], [], []
        for k in range(helper.objects_count):
            m = helper.model_idx_by_object_idx(k)
            box = reconstructed_bounding_boxes[..., k] if helper.is_static(m) else \
                bounding_boxes[..., helper.dynamic_object_idx_by_object_idx(k)]
            s, d, a, c = self.object_encoders[m](observations, box, camera_rotations, camera_translations, global_frame_indexes,
                                                 video_frame_indexes, video_indexes)
            if shuffle_style:
                count = s.size(-2)
                identity = torch.arange(count, device=observations.device, dtype=torch.int64)
                while True:
                    permutation = torch.randperm(count, device=observations.device)
                    if not torch.all(identity == permutation):
                        break
                s = s[..., permutation, :]
            style.append(s)
            deformation.append(d)
            attention.append(a)
            crops.append(c)
        return torch.stack(style, dim=-1), torch.stack(deformation, dim=-1), attention, crops

    def _object_in_scene(self, bounding_boxes_validity: torch.Tensor, quirk: bool) -> torch.Tensor:
        """(..., O, 1, n) presence flags, static objects first (always present), dynamic objects present when some
        camera detects them.  ``quirk``: forward_from_observations builds the static block TWICE over
        (environment_model.py:990-992: an already static_count-wide block is repeated static_count times), so its tensor has
        static_count^2 + dynamic_count entries; the composer reads the first K, i.e. with two static objects the dynamic
        objects' flags are never seen (they read as present).  Reproduced, because it decides what the trainers render."""
        helper = self.object_id_helper
        ones = torch.ones_like(bounding_boxes_validity[..., 0:1], dtype=torch.bool)
        static = torch.cat([ones] * helper.static_objects_count, dim=-1)
        blocks = [static] * helper.static_objects_count if quirk else [static]
        present = torch.cat(blocks + [bounding_boxes_validity], dim=-1)
        return present.max(dim=-2, keepdim=True)[0]

    def forward_scene_encoding_from_observations(self, observations, camera_rotations, camera_translations, focals,
                                                 bounding_boxes, bounding_boxes_validity, global_frame_indexes,
                                                 video_frame_indexes, video_indexes, shuffle_style: bool = False) -> Dict:
        """Scene encoding only (mode="observations_scene_encoding_only"; environment_model.py:772-845)."""
        self._require_encoders()
        camera_rotations, camera_translations, focals = self._corrected_cameras(camera_rotations, camera_translations, focals,
                                                                                global_frame_indexes, focal_quirk=True)
        rescaled_focals = focals * self.focal_length_multiplier
        height, width = observations.size(-2), observations.size(-1)
        c2w, w2c = pose_matrices(camera_rotations, camera_translations)
        rot, tr = self.compute_rotation_translation_o2w(observations, w2c.detach(), camera_rotations, rescaled_focals.detach(),
                                                        bounding_boxes, bounding_boxes_validity)
        _, o2w = self.compute_transformation_matrix_w2o_o2w(rot, tr)
        boxes, _ = self.compute_object_bounding_boxes(o2w, w2c.detach(), rescaled_focals.detach(), height, width)
        style, deformation, _, _ = self.compute_object_encodings(observations, camera_rotations, camera_translations,
                                                                 bounding_boxes, boxes, global_frame_indexes,
                                                                 video_frame_indexes, video_indexes, shuffle_style)
        present = self._object_in_scene(bounding_boxes_validity, quirk=False)
        return {"camera_rotations": camera_rotations, "camera_translations": camera_translations, "focals": focals,
                "object_rotation_parameters": rot, "object_translation_parameters": tr, "object_style": style,
                "object_deformation": deformation, "object_in_scene": present[..., 0, :]}

    def _select_pixels(self, boxes, lead, height, width, samples_per_image, patch_size, patch_stride, device, align_grid=True):
        """The four pixel-selection branches of the reference (environment_model.py:949-958): flat pixel indices (..., R)
        per frame, or a shared (R,) list for the static selections."""
        flat_boxes = boxes.reshape(-1, 4, boxes.size(-1))
        if patch_size != 0 and samples_per_image != 0:
            if flat_boxes.is_cuda:
                rows, cols = ray_sampling.strided_patch_rows_cols(flat_boxes, self.sampling_weights, height, width, patch_size,
                                                                  patch_stride, align_grid=align_grid)
                idx = rows.to(torch.int64) * width + cols.to(torch.int64)
            else:
                idx = ray_sampling.strided_patch_pixels(flat_boxes, self.sampling_weights, height, width, patch_size, patch_stride,
                                                        align_grid=align_grid)
            return idx.reshape(lead + [-1])
        if samples_per_image == 0:
            strides = tuple(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else (int(patch_stride),)
            key = ("flat", height, width, strides if patch_stride else None, str(device))
            if key not in self._pixel_cache:
                if patch_stride:
                    rows, cols = strided_grid_pixels(height, width, patch_stride)
                    idx = rows.to(torch.int64) * width + cols.to(torch.int64)
                else:
                    idx = torch.arange(height * width, dtype=torch.int64)
                self._pixel_cache[key] = idx.to(device)
            return self._pixel_cache[key]
        if self.use_weighted_sampling:
            return ray_sampling.sample_pixels_weighted(flat_boxes, self.sampling_weights, height, width,
                                                       samples_per_image).reshape(lead + [-1])
        return ray_sampling.sample_pixels_uniform(flat_boxes.size(0), height, width, samples_per_image, device).reshape(lead + [-1])

    def forward_from_observations(self, observations, camera_rotations, camera_translations, focals, bounding_boxes,
                                  bounding_boxes_validity, global_frame_indexes, video_frame_indexes, video_indexes,
                                  samples_per_image: int, perturb: bool, samples_per_image_batching: int = 0,
                                  shuffle_style: bool = False, upsample_factor: float = 1.0, patch_size: int = 0,
                                  patch_stride: int = 0, align_grid: bool = True, canonical_pose: bool = False,
                                  _decoder_features=None) -> Dict:
        """model/environment_model.py:847-1039; argument shapes documented there.  observations (..., O, C, 3, H, W);
        camera_* (..., O, C, 3); focals (..., O, C); bounding_boxes (..., O, C, 4, dynamic objects);
        bounding_boxes_validity (..., O, C, dynamic objects); *_indexes (bs, O) / (bs).

        What the trainers call (training/trainer.py).  Poses, style and deformation come from the injected encoders; the
        span between them and the result dictionary - camera rays, box projection, pixel selection with the ground-truth
        pixels gathered alongside, ray-object distances, the composer - is this package's (HIP renderer; batched,
        synchronisation-free host math).  The optional image decoder (config["model"]["image_decoder"]) is an injected
        module like the encoders (``align_grid=False`` with a patch raises in the reference too)."""
        self._require_encoders()
        batch = (observations, camera_rotations, camera_translations, focals, bounding_boxes, bounding_boxes_validity,
                 global_frame_indexes, video_frame_indexes, video_indexes)
        if not shuffle_style and self._replay_wanted(batch, perturb, samples_per_image):
            stride_key = tuple(patch_stride) if isinstance(patch_stride, collections.abc.Sequence) else patch_stride
            statics = (samples_per_image_batching, upsample_factor, patch_size, stride_key, align_grid, canonical_pose,
                       None if _decoder_features is None else tuple(_decoder_features))
            replayed = self._replayed(
                "observations",
                lambda *ts: self.forward_from_observations(*ts, samples_per_image, perturb, samples_per_image_batching, shuffle_style,
                                                           upsample_factor, patch_size, patch_stride, align_grid, canonical_pose,
                                                           _decoder_features=_decoder_features),
                batch, statics)
            if replayed is not None:
                return replayed
        camera_rotations, camera_translations, focals = self._corrected_cameras(camera_rotations, camera_translations, focals,
                                                                                global_frame_indexes)
        rescaled_focals = focals * self.focal_length_multiplier
        if upsample_factor != 1.0:
            h, w = observations.size(-2), observations.size(-1)
            flat = observations.reshape([-1] + list(observations.shape[-3:]))
            flat = torch.nn.functional.interpolate(flat, (int(h * upsample_factor), int(w * upsample_factor)), mode="bilinear")
            observations = flat.reshape(list(observations.shape[:-3]) + list(flat.shape[-3:]))
        height, width = observations.size(-2), observations.size(-1)
        lead = list(observations.shape[:-3])

        c2w, w2c = pose_matrices(camera_rotations, camera_translations)
        render_focals = rescaled_focals if upsample_factor == 1.0 else rescaled_focals * upsample_factor
        rot, tr = self.compute_rotation_translation_o2w(observations, w2c.detach(), camera_rotations, render_focals.detach(),
                                                        bounding_boxes, bounding_boxes_validity)
        w2o, o2w = self.compute_transformation_matrix_w2o_o2w(rot, tr)
        boxes, box_points = self.compute_object_bounding_boxes(o2w, w2c.detach(), render_focals.detach(), height, width)
        axes = self.compute_object_axes_projection(o2w, w2c.detach(), render_focals.detach(), height, width)

        # pixel selection; the ground-truth pixels and the normalised positions are gathered with the same indices
        idx = self._select_pixels(boxes, lead, height, width, samples_per_image, patch_size, patch_stride, observations.device,
                                  align_grid=align_grid)
        rows, cols = ray_sampling.split_indices(idx, width)
        hwc = observations.movedim(-3, -1).reshape(lead + [height * width, observations.size(-3)])
        full = idx if idx.dim() > 1 else idx.expand(lead + [idx.numel()])
        sampled_observations = torch.gather(hwc, -2, full.unsqueeze(-1).expand(list(full.shape) + [hwc.size(-1)]))
        sampled_positions = ray_sampling.positions_from_indices(full, height, width)

        origins, directions, normals = camera_rays(c2w, render_focals, height, width, rows, cols)
        style, deformation, attention, crops = self.compute_object_encodings(observations, camera_rotations, camera_translations,
                                                                             bounding_boxes, boxes, global_frame_indexes,
                                                                             video_frame_indexes, video_indexes, shuffle_style)
        distances = self.compute_ray_object_distances(origins, directions, o2w[..., 0, :, :, :])
        present = self._object_in_scene(bounding_boxes_validity, quirk=True)
        expanded_video_indexes, _ = torch.broadcast_tensors(video_indexes.unsqueeze(-1).unsqueeze(-1), origins[..., 0])
        layout = None if _decoder_features is None else \
            self.decoder_layout(height, width, samples_per_image, patch_size, patch_stride, _decoder_features)
        results = self.batchified_composer_call(origins, directions, normals, w2o, style.unsqueeze(-3), deformation.unsqueeze(-3),
                                                present, perturb, samples_per_image_batching, expanded_video_indexes,
                                                canonical_pose=canonical_pose, _decoder_layout=layout)
        if self.use_image_decoder:
            self.compute_decoded_image(results, sampled_positions)
        results["observations"] = sampled_observations
        results["positions"] = sampled_positions
        results["object_rotation_parameters"] = rot
        results["object_translation_parameters"] = tr
        results["ray_object_distances"] = distances
        results["reconstructed_bounding_boxes"] = boxes
        results["reconstructed_3d_bounding_boxes"] = box_points
        results["projected_axes"] = axes
        results["object_attention"] = attention
        results["object_crops"] = crops
        results["scene_encoding"] = {
            "camera_rotations": camera_rotations, "camera_translations": camera_translations, "focals": focals,
            "object_rotation_parameters": rot, "object_translation_parameters": tr, "object_style": style,
            "object_deformation": deformation, "object_in_scene": present[..., 0, :],
        }
        return results

    def render_full_frame_from_observations(self, observations, camera_rotations, camera_translations, focals,
                                            bounding_boxes, bounding_boxes_validity, global_frame_indexes, video_frame_indexes,
                                            video_indexes, perturb: bool, samples_per_image_batching: int = 1000,
                                            upsample_factor: float = 1.0, canonical_pose: bool = False) -> Dict:
        """Every pixel of the frames, folded back to (height, width).  model/environment_model.py:581-616."""
        flat = self(observations, camera_rotations, camera_translations, focals, bounding_boxes, bounding_boxes_validity,
                    global_frame_indexes, video_frame_indexes, video_indexes, 0, perturb, samples_per_image_batching,
                    upsample_factor=upsample_factor, canonical_pose=canonical_pose)
        height = int(observations.size(-2) * upsample_factor)
        width = int(observations.size(-1) * upsample_factor)
        return self.fold_dictionary(flat, height, width)

    # ------------------------------------------------------------------ pose / keypoint consistency (expected positions)
    @staticmethod
    def merge_expected_position_results(expected_position_results: List[Dict]) -> Dict:
        """environment_model.py:1160-1176: {key: [results of call 0, results of call 1, ...]}."""
        return {key: [r[key] for r in expected_position_results] for key in expected_position_results[0].keys()}

    @staticmethod
    def invert_expected_position_results(all_expected_position_results: List[Dict]) -> Dict:
        """environment_model.py:1178-1195: {model type: {"dynamic_object_i": results}}."""
        inverted = {key: {} for key in all_expected_position_results[0].keys()}
        for i, current in enumerate(all_expected_position_results):
            for key in current:
                inverted[key][f"dynamic_object_{i}"] = current[key]
        return inverted

    @staticmethod
    def camera_direction_grid(focals: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """Camera-frame pinhole directions (..., H, W, 3) = ((col - W/2) / f, -(row - H/2) / f, -1) of every pixel
        (RayHelper.create_camera_rays, ray_helper.py:15-52) - the grid the consistency samplers look up."""
        f = focals.unsqueeze(-1).unsqueeze(-1)
        rows, cols = torch.meshgrid(torch.arange(0, height, device=focals.device), torch.arange(0, width, device=focals.device),
                                    indexing="ij")
        dx = (cols - width / 2) / f
        dy = -(rows - height / 2) / f
        return torch.stack([dx, dy, -torch.ones_like(dx)], dim=-1)

    @staticmethod
    def _world_rays(c2w: torch.Tensor, camera_directions: torch.Tensor):
        """Camera-frame sample directions (..., n, 3) -> world origins (..., 3), directions (..., n, 3), focal normals
        (..., 3) (RayHelper.transform_rays of origin 0 / normal (0, 0, -1), ray_helper.py:1203-1227)."""
        rot = c2w[..., :3, :3]
        directions = torch.sum(camera_directions.unsqueeze(-2) * rot.unsqueeze(-3), -1)
        return c2w[..., :3, 3], directions, -rot[..., :, 2]

    def forward_pose_consistency(self, optical_flow, camera_rotations, camera_translations, focals, bounding_boxes,
                                 bounding_boxes_validity, global_frame_indexes, video_frame_indexes, video_indexes, object_style,
                                 object_deformation, object_rotation_parameters_o2w, object_translation_parameters_o2w,
                                 samples_per_image: int, perturb: bool) -> Dict:
        """model/environment_model.py:1197-1361: for every dynamic object, pixels sampled inside its box in frame t, their
        optical-flow targets in frame t + 1, and the expected surface position (object frame) the renderer finds along
        both rays - ``{type: {"dynamic_object_i": [(positions, opacity) of frame t, (positions, opacity) of frame t+1]}}``.
        optical_flow (..., O, C, 2, H, W) normalised, (row, col) channels; the other arguments as in the reference."""
        camera_rotations, camera_translations, focals = self._corrected_cameras(camera_rotations, camera_translations, focals,
                                                                                global_frame_indexes)
        rescaled_focals = focals * self.focal_length_multiplier
        height, width = optical_flow.size(-2), optical_flow.size(-1)
        object_style = object_style.unsqueeze(-3)
        object_deformation = object_deformation.unsqueeze(-3)
        if video_indexes is not None:
            video_indexes = video_indexes.unsqueeze(-1).unsqueeze(-1)
        grid = self.camera_direction_grid(rescaled_focals, height, width)                      # (..., O, C, H, W, 3)
        c2w = euler_to_matrix(camera_rotations, camera_translations)                           # (..., O, C, 4, 4)
        w2o, _ = self.compute_transformation_matrix_w2o_o2w(object_rotation_parameters_o2w, object_translation_parameters_o2w)
        helper = self.object_id_helper
        all_results = []
        for dyn in range(helper.dynamic_objects_count):
            k = helper.object_idx_by_dynamic_object_idx(dyn)
            box, valid = bounding_boxes[..., dyn], bounding_boxes_validity[..., dyn]
            m, sty, dfm = w2o[..., k], object_style[..., k], object_deformation[..., k]
            prev_dirs, prev_flow, prev_pos = ray_sampling.sample_rays_at_object(grid[..., :-1, :, :, :, :], optical_flow[..., :-1, :, :, :, :],
                                                                               samples_per_image, box[..., :-1, :, :])
            next_pos = prev_flow + prev_pos
            # the optical flow comes from an unknown higher resolution: no range correction (environment_model.py:1316)
            next_dirs = ray_sampling.sample_rays_at(grid[..., 1:, :, :, :, :], next_pos, correct_range=False)
            po, pd, pn = self._world_rays(c2w[..., :-1, :, :, :], prev_dirs)
            no, nd, nn_ = self._world_rays(c2w[..., 1:, :, :, :], next_dirs)
            previous = self.object_composer.forward_expected_positions(po, pd, pn, m[..., :-1, :, :, :], sty[..., :-1, :, :],
                                                                       dfm[..., :-1, :, :], object_in_scene=valid[..., :-1, :],
                                                                       object_id=k, perturb=perturb, video_indexes=video_indexes)
            following = self.object_composer.forward_expected_positions(no, nd, nn_, m[..., 1:, :, :, :], sty[..., 1:, :, :],
                                                                        dfm[..., 1:, :, :], object_in_scene=valid[..., 1:, :],
                                                                        object_id=k, perturb=perturb, video_indexes=video_indexes)
            all_results.append(self.merge_expected_position_results([previous, following]))
        results = self.invert_expected_position_results(all_results)
        results["pytorch_backward_hook"] = results["coarse"]["dynamic_object_0"][0]
        return results

    def forward_keypoint_consistency(self, observations, camera_rotations, camera_translations, focals, bounding_boxes,
                                     bounding_boxes_validity, global_frame_indexes, video_frame_indexes, video_indexes,
                                     object_style, object_deformation, object_rotation_parameters_o2w,
                                     object_translation_parameters_o2w, keypoints, keypoints_validity, max_samples_per_image: int,
                                     perturb: bool) -> Dict:
        """model/environment_model.py:1363-1505: for every dynamic object, rays through random points of its COCO skeleton
        segments -> ``{type: {"dynamic_object_i": (expected positions, keypoint confidences, opacity, sampled positions)}}``.
        keypoints (..., O, C, 17, 3, dynamic objects) as (row, col, confidence) in [0, 1]."""
        camera_rotations, camera_translations, focals = self._corrected_cameras(camera_rotations, camera_translations, focals,
                                                                                global_frame_indexes)
        rescaled_focals = focals * self.focal_length_multiplier
        height, width = observations.size(-2), observations.size(-1)
        object_style = object_style.unsqueeze(-3)
        object_deformation = object_deformation.unsqueeze(-3)
        if video_indexes is not None:
            video_indexes = video_indexes.unsqueeze(-1).unsqueeze(-1)
        grid = self.camera_direction_grid(rescaled_focals, height, width)
        c2w = euler_to_matrix(camera_rotations, camera_translations)
        w2o, _ = self.compute_transformation_matrix_w2o_o2w(object_rotation_parameters_o2w, object_translation_parameters_o2w)
        helper = self.object_id_helper
        all_results = []
        for dyn in range(helper.dynamic_objects_count):
            k = helper.object_idx_by_dynamic_object_idx(dyn)
            dirs, positions, confidences = ray_sampling.sample_rays_at_keypoints(grid, keypoints[..., dyn], max_samples_per_image)
            o, d, n = self._world_rays(c2w, dirs)
            expected = self.object_composer.forward_expected_positions(o, d, n, w2o[..., k], object_style[..., k],
                                                                       object_deformation[..., k],
                                                                       object_in_scene=bounding_boxes_validity[..., dyn], object_id=k,
                                                                       perturb=perturb, video_indexes=video_indexes)
            for key in list(expected):
                expected[key] = (expected[key][0], confidences, expected[key][1], positions)
            all_results.append(expected)
        results = self.invert_expected_position_results(all_results)
        results["pytorch_backward_hook"] = results["coarse"]["dynamic_object_0"][0]
        return results

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
        of all frames (a single frame across the node); "tiles": every rank renders the 8 x 8 pixel tiles t = rank (mod world)
        of the frame (``parallel.tile_shard_lists``: contiguous ranges give one rank the sky and another the players, tiles
        give every rank a sample of the whole frame - balanced evaluated samples); "auto": frames when the batch has at
        least one frame per rank, tiles otherwise.
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
            shard = "frames" if batch >= world else "tiles"
        if shard not in ("frames", "rays", "tiles"):
            raise ValueError(f"unknown shard mode {shard!r} (expected 'auto', 'frames', 'rays' or 'tiles')")
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
                grids = [(height // s, width // s) for s in strides]
            else:
                grids = [(height, width)]
            total = sum(h * w for h, w in grids)
            dim = camera_rotations.dim() - 1
            if shard == "tiles":
                key = ("tiles", tuple(grids), world, str(camera_rotations.device))
                if key not in self._pixel_cache:
                    self._pixel_cache[key] = [l.to(camera_rotations.device) for l in parallel.tile_shard_lists(grids, world)]
                lists = self._pixel_cache[key]
                ray_range = lists[rank]
            else:
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
                    if shard == "tiles":
                        full = parallel.gather_indexed_shards(local[ty][entry][field], lists, dim, dst=dst, group=group)
                    else:
                        full = parallel.gather_ray_shards(local[ty][entry][field], total, dim, dst=dst, group=group)
                    if receives:
                        out.setdefault(ty, {}).setdefault(entry, {})[field] = full
        return out if receives else None
