"""Producers of the renderer's inputs: the object encoders and object-parameters encoders of the
reference with their region-of-interest crop on a HIP kernel.

* ``roi_pool`` - ``torchvision.ops.roi_pool`` (what the reference calls at model/object_encoder_v4.py:121,
  model/object_encoder_v5.py:121, model/object_parameters_encoder_v4.py:131) as ``pr_roi_pool_forward / _backward`` of
  libplayrender.so: one thread per output element, argmax kept for the backward scatter.  No CPU fallback.
* ``ObjectEncoderV4`` / ``ObjectEncoderV5`` (style + deformation codes from the crop), ``ObjectParametersEncoderV4`` (rotation
  about the up axis from the crop, translation from the ground-plane ray cast), ``ClassicObjectParametersEncoder`` (ground-
  plane ray cast only), ``StaticObjectParametersEncoder``: parameter and buffer names follow the reference modules, so their
  checkpoints load with ``load_state_dict``; the small ResNets stay on PyTorch-ROCm (MIOpen convolutions), as north_star
  prescribes for the CNN side.  The per-object Python loops of the reference are batched over the objects, and the
  4x4 inverses are closed-form (``rigid_inverse``): no host synchronisation on the path.
* ``create_encoders(config)`` builds the two module lists ``EnvironmentModel`` needs from the configuration's
  ``architecture`` strings, like the reference's ``create_object_encoders`` / ``create_object_parameters_encoders``
  (model/environment_model.py:93-123).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .modules import AvgPool2d, BatchNorm2d, Conv2d, LeakyReLU, Linear, Sequential, Tracked


# ---------------------------------------------------------------------------------------------------------------------
# roi_pool on the HIP kernel
class _RoiPool(torch.autograd.Function):

    @staticmethod
    def forward(ctx, inputs: torch.Tensor, boxes: torch.Tensor, pooled_height: int, pooled_width: int, spatial_scale: float):
        if not inputs.is_cuda:
            raise RuntimeError("roi_pool runs on the HIP kernel: it needs device tensors (there is no CPU fallback)")
        x = inputs.detach().to(torch.float32).contiguous()
        b = boxes.detach().to(torch.float32).contiguous()
        n, c, h, w = x.shape
        k = b.size(0)
        out = torch.empty((k, c, pooled_height, pooled_width), dtype=torch.float32, device=x.device)
        argmax = torch.empty((k, c, pooled_height, pooled_width), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().pr_roi_pool_forward(n, c, h, w, x.data_ptr(), k, b.data_ptr(), pooled_height, pooled_width,
                                                       C.c_float(spatial_scale), out.data_ptr(), argmax.data_ptr(),
                                                       torch.cuda.current_stream(x.device).cuda_stream), "pr_roi_pool_forward")
        ctx.save_for_backward(b, argmax)
        ctx.shape = (n, c, h, w)
        ctx.pooled = (pooled_height, pooled_width)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        b, argmax = ctx.saved_tensors
        n, c, h, w = ctx.shape
        grad_input = torch.zeros(ctx.shape, dtype=torch.float32, device=grad_output.device)
        g = grad_output.to(torch.float32).contiguous()
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().pr_roi_pool_backward(n, c, h, w, b.size(0), b.data_ptr(), ctx.pooled[0], ctx.pooled[1], g.data_ptr(),
                                                        argmax.data_ptr(), grad_input.data_ptr(),
                                                        torch.cuda.current_stream(g.device).cuda_stream), "pr_roi_pool_backward")
        return grad_input, None, None, None, None


def roi_pool(inputs: torch.Tensor, boxes: torch.Tensor, output_size: Sequence[int], spatial_scale: float = 1.0) -> torch.Tensor:
    """torchvision.ops.roi_pool(input (N,C,H,W), boxes (K,5) [image index, x1, y1, x2, y2], output_size) -> (K,C,ph,pw);
    differentiable with respect to ``inputs``."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    return _RoiPool.apply(inputs, boxes, int(output_size[0]), int(output_size[1]), float(spatial_scale))


# ---------------------------------------------------------------------------------------------------------------------
# building blocks
class ResidualBlock(Tracked, nn.Module):
    """model/layers/residual_block.py:13-68: conv3x3 -> avg-pool -> BN -> LeakyReLU(0.2) -> conv3x3 -> BN, plus a
    conv1x1 -> avg-pool -> BN shortcut when the shape changes."""

    def __init__(self, in_planes: int, out_planes: int, downsample_factor: int = 1, last_affine: bool = True,
                 drop_final_activation: bool = False):
        super().__init__()
        self.conv1 = Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = BatchNorm2d(out_planes)
        self.relu = LeakyReLU(0.2, inplace=True)
        self.conv2 = Conv2d(out_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(out_planes, affine=last_affine)
        self.downsample_factor = downsample_factor
        self.drop_final_activation = drop_final_activation
        self.downsample = None
        if downsample_factor != 1 or in_planes != out_planes:
            self.downsample = Sequential(Conv2d(in_planes, out_planes, kernel_size=1, stride=1, bias=False),
                                            AvgPool2d(downsample_factor),
                                            BatchNorm2d(out_planes, affine=last_affine))

    def forward(self, x):
        out = self.relu(self.bn1(F.avg_pool2d(self.conv1(x), self.downsample_factor)))
        out = self.bn2(self.conv2(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        return out if self.drop_final_activation else self.relu(out)


def _expanded_boxes(boxes: torch.Tensor, rows: float, cols: float) -> torch.Tensor:
    """(..., 4) [left, top, right, bottom]: widened by ``cols`` box widths on both sides and by ``rows`` box heights at the
    top (never at the bottom: the feet stay where they are), clamped to the image (object_encoder_v4.py:62-78)."""
    size = boxes[..., 2:] - boxes[..., :2]
    left = boxes[..., 0] - size[..., 0] * cols
    right = boxes[..., 2] + size[..., 0] * cols
    top = boxes[..., 1] - size[..., 1] * rows
    return torch.clamp(torch.stack([left, top, right, boxes[..., 3]], dim=-1), min=0.0, max=1.0)


_CONSTANTS: Dict = {}


def _constant(values, dtype, device) -> torch.Tensor:
    """A small constant tensor on ``device``, uploaded ONCE per (values, dtype, device): ``torch.tensor([...], device=...)`` in a
    forward pass is a copy from pageable host memory - it waits for the stream's queued work on every call and cannot be
    recorded into a HIP graph.  (The cache holds a handful of 3- and 4-vectors.)"""
    key = (tuple(float(v) for v in values), dtype, str(device))
    cached = _CONSTANTS.get(key)
    if cached is None:
        cached = torch.tensor([float(v) for v in values], dtype=dtype, device=device)
        _CONSTANTS[key] = cached
    return cached


def _crop_first_camera(observations: torch.Tensor, boxes: torch.Tensor, input_size, rows: float, cols: float):
    """observations (..., C, 3, H, W), boxes (..., C, 4) normalised -> crops (M, 3, h, w) of the FIRST camera (the
    reference's encoders only look at it), M = prod(...), and the leading dims."""
    obs = observations[..., :1, :, :, :]
    box = _expanded_boxes(boxes[..., :1, :], rows, cols)
    height, width = obs.size(-2), obs.size(-1)
    scale = _constant([width, height, width, height], box.dtype, box.device)
    lead = list(obs.shape[:-3])                      # (..., 1)
    flat_obs = obs.reshape([-1] + list(obs.shape[-3:]))
    flat_box = (box * scale).reshape(-1, 4)
    index = torch.arange(flat_box.size(0), device=flat_box.device, dtype=flat_box.dtype).unsqueeze(-1)
    crops = roi_pool(flat_obs, torch.cat([index, flat_box], dim=-1), input_size)
    return crops, lead


class _CropEncoderBase(Tracked, nn.Module):
    def _read_expansion(self, model_config: Dict):
        self.expansion_factor_rows = 0.0
        self.expansion_factor_cols = 0.0
        if "expansion_factor" in model_config:
            self.expansion_factor_rows = model_config["expansion_factor"]["rows"]
            self.expansion_factor_cols = model_config["expansion_factor"]["cols"]


class ObjectEncoderV4(_CropEncoderBase):
    """model/object_encoder_v4.py:13-178: crop (+ camera pose as 6 constant channels) -> conv -> residual block with an
    attention channel -> 4 residual blocks -> global average -> style / deformation heads."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config = config
        self.input_size = model_config["input_size"]
        self.deformation_features = model_config["deformation_features"]
        self.style_features = model_config["style_features"]
        self._read_expansion(model_config)
        self.conv1 = Conv2d(3 + 6, 16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = BatchNorm2d(16)
        self.initial_backbone = Sequential(ResidualBlock(16, 16 + 1, downsample_factor=1, drop_final_activation=True))
        self.final_backbone = Sequential(ResidualBlock(16, 32, downsample_factor=2), ResidualBlock(32, 32, downsample_factor=1),
                                            ResidualBlock(32, 64, downsample_factor=2), ResidualBlock(64, 64, downsample_factor=1))
        self.style_head = Linear(64, self.style_features)
        self.deformation_head = Linear(64, self.deformation_features)

    def forward(self, observations, bounding_boxes, camera_rotations, camera_translations, global_frame_indexes,
                video_frame_indexes, video_indexes):
        crops, lead = _crop_first_camera(observations, bounding_boxes, self.input_size, self.expansion_factor_rows,
                                         self.expansion_factor_cols)
        pose = torch.cat([camera_rotations[..., :1, :], camera_translations[..., :1, :]], dim=-1).reshape(-1, 6)
        planes = pose.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, self.input_size[0], self.input_size[1])
        x = self.conv1(torch.cat([crops, planes], dim=-3))
        x = F.leaky_relu(self.bn1(F.avg_pool2d(x, 2)), 0.2)
        initial = self.initial_backbone(x)
        attention = torch.sigmoid(initial[:, -1:])
        features = F.leaky_relu(initial[:, :-1], 0.2) * attention
        cameras = lead[-1]
        features = features.reshape([-1, cameras] + list(features.shape[1:])).sum(dim=1) / cameras
        pooled = F.adaptive_avg_pool2d(self.final_backbone(features), (1, 1)).squeeze(-1).squeeze(-1)
        style = self.style_head(pooled).reshape(lead[:-1] + [self.style_features])
        deformation = self.deformation_head(pooled).reshape(lead[:-1] + [self.deformation_features])
        return (style, deformation, attention.reshape(lead + list(attention.shape[1:])), crops.reshape(lead + list(crops.shape[1:])))


def _resnet_trunk() -> Tuple[nn.Sequential, nn.Sequential]:
    initial = Sequential(ResidualBlock(64, 64, downsample_factor=2), ResidualBlock(64, 64, downsample_factor=1))
    final = Sequential(ResidualBlock(64, 128, downsample_factor=2), ResidualBlock(128, 128, downsample_factor=1),
                          ResidualBlock(128, 256, downsample_factor=2), ResidualBlock(256, 256, downsample_factor=1),
                          ResidualBlock(256, 512, downsample_factor=2), ResidualBlock(512, 512, downsample_factor=1))
    return initial, final


class ObjectEncoderV5(_CropEncoderBase):
    """model/object_encoder_v5.py: the deeper variant for the static objects - 7x7 stem, 8 residual blocks, no attention
    (the returned attention map is zero)."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config = config
        self.input_size = model_config["input_size"]
        self.deformation_features = model_config["deformation_features"]
        self.style_features = model_config["style_features"]
        self._read_expansion(model_config)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.initial_backbone, self.final_backbone = _resnet_trunk()
        self.style_head = Linear(512, self.style_features)
        self.deformation_head = Linear(512, self.deformation_features)

    def forward(self, observations, bounding_boxes, camera_rotations, camera_translations, global_frame_indexes,
                video_frame_indexes, video_indexes):
        crops, lead = _crop_first_camera(observations, bounding_boxes, self.input_size, self.expansion_factor_rows,
                                         self.expansion_factor_cols)
        x = F.leaky_relu(self.bn1(self.conv1(crops)), 0.2)
        initial = self.initial_backbone(x)
        attention = torch.zeros_like(initial[:, -1:])
        cameras = lead[-1]
        features = initial.reshape([-1, cameras] + list(initial.shape[1:])).sum(dim=1) / cameras
        pooled = F.adaptive_avg_pool2d(self.final_backbone(features), (1, 1)).squeeze(-1).squeeze(-1)
        style = self.style_head(pooled).reshape(lead[:-1] + [self.style_features])
        deformation = self.deformation_head(pooled).reshape(lead[:-1] + [self.deformation_features])
        return (style, deformation, attention.reshape(lead + list(attention.shape[1:])), crops.reshape(lead + list(crops.shape[1:])))


# ---------------------------------------------------------------------------------------------------------------------
# object-parameters (pose) encoders
def _ground_plane_feet(transformation_matrix_w2c, focals, bounding_boxes, height: int, width: int, up_axis: int):
    """The reference's classical localisation, batched over the objects: the ray through the bottom centre of every box of
    the first camera, intersected with the plane ``up_axis = 0`` (classic_object_parameters_encoder.py:168-196,
    object_parameters_encoder_v4.py:262-283).  transformation_matrix_w2c (..., C, 4, 4); focals (..., C); bounding_boxes
    (..., C, 4, n) normalised.  Returns ground points (..., 3, n), world ray directions (..., 3, n)."""
    from .environment_model import rigid_inverse
    eps = 1e-6
    c2w = rigid_inverse(transformation_matrix_w2c[..., 0, :, :])                         # (..., 4, 4)
    box = bounding_boxes[..., 0, :, :]                                                   # (..., 4, n)
    x = (box[..., 0, :] * width + box[..., 2, :] * width) / 2 - (width / 2)
    y = -(box[..., 3, :] * height - (height / 2))
    z = -focals[..., 0].unsqueeze(-1).expand_as(x)
    camera_dirs = torch.stack([x, y, z], dim=-2)                                         # (..., 3, n)
    rot = c2w[..., :3, :3]
    dirs = torch.matmul(rot, camera_dirs)                                                # (..., 3, n)
    origin = c2w[..., :3, 3].unsqueeze(-1)                                               # (..., 3, 1)
    steps = -origin[..., up_axis, :] / (dirs[..., up_axis, :] + eps)                     # (..., n)
    points = origin + steps.unsqueeze(-2) * dirs
    keep = _constant([0.0 if axis == up_axis else 1.0 for axis in range(3)], points.dtype, points.device)
    return points * keep.view(3, 1), dirs


class StaticObjectParametersEncoder(Tracked, nn.Module):
    """model/static_object_parameters_encoder.py:7-66: static objects sit at the midpoint of their configured ranges."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config, self.model_config = config, model_config
        self.objects_count = model_config["objects_count"]
        self.register_buffer("translation_range", torch.tensor(model_config["translation_range"], dtype=torch.float32))
        self.register_buffer("rotation_range", torch.tensor(model_config["rotation_range"], dtype=torch.float32))

    def forward(self, observations: torch.Tensor):
        lead = list(observations.shape[:-4])
        # (0 + 1) / 2 * (max - min) + min per object and axis, broadcast over the leading dims
        mid = lambda r: (0.5 * (r[..., 1] - r[..., 0]) + r[..., 0]).transpose(0, 1)       # (3, objects)
        rotation = mid(self.rotation_range).expand(lead + [3, self.objects_count]).clone()
        translation = mid(self.translation_range).expand(lead + [3, self.objects_count]).clone()
        return rotation, translation


class ClassicObjectParametersEncoder(Tracked, nn.Module):
    """model/classic_object_parameters_encoder.py:14-237: translation = the box's bottom centre cast on the ground plane
    (``zero_axis``, raised to the middle of that axis' configured range), rotation = the middle of the configured range;
    absent objects get zero translation."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config, self.model_config = config, model_config
        self.objects_count = model_config["objects_count"]
        self.zero_axis = model_config.get("zero_axis", 2)
        self.register_buffer("translation_range", torch.tensor(model_config["translation_range"], dtype=torch.float32))
        self.register_buffer("rotation_range", torch.tensor(model_config["rotation_range"], dtype=torch.float32))

    def forward(self, observations, transformation_matrix_w2c, camera_rotations, focals, bounding_boxes, bounding_boxes_validity,
                apply_ranges: bool = True):
        height, width = observations.size(-2), observations.size(-1)
        points, _ = _ground_plane_feet(transformation_matrix_w2c, focals, bounding_boxes, height, width, self.zero_axis)
        n = bounding_boxes.size(-1)
        if apply_ranges:
            lift = (self.translation_range[:n, self.zero_axis, 0] + self.translation_range[:n, self.zero_axis, 1]) / 2
            offset = torch.zeros((3, n), dtype=points.dtype, device=points.device)
            offset[self.zero_axis] = lift
            points = points + offset
            rotation = ((self.rotation_range[:n, :, 1] + self.rotation_range[:n, :, 0]) / 2).transpose(0, 1)   # (3, n)
        else:
            rotation = torch.zeros((3, n), dtype=points.dtype, device=points.device)
        valid = bounding_boxes_validity[..., 0, :].unsqueeze(-2)                                               # (..., 1, n)
        translations = torch.where(valid, points, torch.zeros_like(points))
        rotations = rotation.expand_as(translations).clone()
        if not translations.requires_grad:          # the reference marks its outputs as leaves that require gradients
            translations.requires_grad_(True)
        rotations.requires_grad_(True)
        return rotations, translations


class ObjectParametersEncoderV4(_CropEncoderBase):
    """model/object_parameters_encoder_v4.py:13-369: rotation about the y axis = camera yaw + atan2 of a 2-vector the
    ResNet predicts from the crop; translation = the ground-plane ray cast (xz plane), moved along the viewing direction
    by ``edge_to_center_distance / cos(offset)`` (the box's front edge is not the object's centre)."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config = config
        self.input_size = model_config["input_size"]
        self.edge_to_center_distance = model_config["edge_to_center_distance"]
        self._read_expansion(model_config)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.initial_backbone, self.final_backbone = _resnet_trunk()
        self.rotation_head = Linear(512, 2)
        torch.nn.init.uniform_(self.rotation_head.weight, a=-1e-5, b=1e-5)
        with torch.no_grad():
            self.rotation_head.bias.mul_(0.0)

    def compute_rotations(self, observations, camera_rotations, bounding_boxes, bounding_boxes_validity):
        """One object: observations (..., C, 3, H, W); bounding_boxes (..., C, 4); validity (..., C) -> rotations o2w and
        camera-to-object offsets, both (..., 3)."""
        crops, lead = _crop_first_camera(observations, bounding_boxes, self.input_size, self.expansion_factor_rows,
                                         self.expansion_factor_cols)
        x = F.leaky_relu(self.bn1(self.conv1(crops)), 0.2)
        features = self.initial_backbone(x)
        cameras = lead[-1]
        features = features.reshape([-1, cameras] + list(features.shape[1:])).sum(dim=1) / cameras
        pooled = F.adaptive_avg_pool2d(self.final_backbone(features), (1, 1)).squeeze(-1).squeeze(-1)
        vector = torch.tanh(self.rotation_head(pooled)) * 1.4
        yaw = torch.atan2(vector[..., 1], vector[..., 0]).reshape(lead[:-1])
        zeros = torch.zeros_like(yaw)
        offset = torch.stack([zeros, yaw, zeros], dim=-1)
        camera_yaw = torch.stack([zeros, camera_rotations[..., 0, 1], zeros], dim=-1)
        valid = bounding_boxes_validity[..., 0].unsqueeze(-1)
        return torch.where(valid, camera_yaw + offset, torch.zeros_like(offset)), torch.where(valid, offset, torch.zeros_like(offset))

    @staticmethod
    def normalize_range(tensor: torch.Tensor, min: float, max: float) -> torch.Tensor:
        """Values brought into [min, max] in steps of (max - min), as the reference's loops do - in closed form."""
        delta = max - min
        above = torch.clamp(torch.ceil((tensor - max) / delta), min=0.0)
        tensor = tensor - above * delta
        below = torch.clamp(torch.ceil((min - tensor) / delta), min=0.0)
        return tensor + below * delta

    def compute_translations(self, observations, transformation_matrix_w2c, c2o_rotation_offset, focals, bounding_boxes,
                             bounding_boxes_validity):
        height, width = observations.size(-2), observations.size(-1)
        points, dirs = _ground_plane_feet(transformation_matrix_w2c, focals, bounding_boxes, height, width, up_axis=1)
        flat = dirs * _constant([1.0, 0.0, 1.0], dirs.dtype, dirs.device).view(3, 1)
        flat = flat / torch.sqrt(flat.pow(2).sum(-2, keepdim=True))
        yaw = self.normalize_range(c2o_rotation_offset[..., 1, :], -(math.pi / 4), +(math.pi / 4))            # (..., n)
        points = points + flat * (self.edge_to_center_distance / torch.cos(yaw)).unsqueeze(-2)
        valid = bounding_boxes_validity[..., 0, :].unsqueeze(-2)
        return torch.where(valid, points, torch.zeros_like(points))

    def forward(self, observations, transformation_matrix_w2c, camera_rotations, focals, bounding_boxes, bounding_boxes_validity,
                apply_ranges: bool = True):
        rotations, offsets = [], []
        for k in range(bounding_boxes.size(-1)):
            r, o = self.compute_rotations(observations, camera_rotations, bounding_boxes[..., k], bounding_boxes_validity[..., k])
            rotations.append(r)
            offsets.append(o)
        rotations = torch.stack(rotations, dim=-1)
        offsets = torch.stack(offsets, dim=-1)
        translations = self.compute_translations(observations, transformation_matrix_w2c, offsets, focals, bounding_boxes,
                                                 bounding_boxes_validity)
        return rotations, translations


OBJECT_ENCODER_CLASSES = {
    "model.object_encoder_v4": ObjectEncoderV4,
    "model.object_encoder_v5": ObjectEncoderV5,
}
OBJECT_PARAMETERS_ENCODER_CLASSES = {
    "model.static_object_parameters_encoder": StaticObjectParametersEncoder,
    "model.classic_object_parameters_encoder": ClassicObjectParametersEncoder,
    "model.object_parameters_encoder_v4": ObjectParametersEncoderV4,
}


def create_encoders(config: Dict) -> Tuple[List[nn.Module], List[nn.Module]]:
    """(object_encoders, object_parameters_encoders), one module per object model, from the ``architecture`` strings of
    ``config["model"]["object_encoders"]`` and ``config["model"]["object_parameters_encoder"]`` (the reference resolves the
    same strings with importlib, model/environment_model.py:93-123)."""
    def build(entries, table, what):
        out = []
        for entry in entries:
            name = entry.get("architecture")
            if name not in table:
                raise Exception(f"{what} architecture {name!r} is not supported (known: {sorted(table)})")
            out.append(table[name](config, entry))
        return out
    model = config["model"]
    return (build(model["object_encoders"], OBJECT_ENCODER_CLASSES, "object encoder"),
            build(model["object_parameters_encoder"], OBJECT_PARAMETERS_ENCODER_CLASSES, "object parameters encoder"))
