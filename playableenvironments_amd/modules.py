"""Parameter containers of the object models, ``state_dict``-compatible with the reference.

These modules own the ``nn.Parameter`` / buffer storages under exactly the names the reference
uses (``object_models_coarse.2.nerf_model.backbone_layers.4.weight``,
``...features_head.1.ada_in.normalization.running_mean``, ``ray_bender.positional_encoder.current_step``
...), so reference checkpoints load with ``load_state_dict`` and the trainers'
``model.parameters()`` see the same tensors.  They deliberately have NO ``forward``: all arithmetic
of these networks runs in the fused HIP kernel (csrc/mlp.hip), which reads the parameter storages
in place.  Initialisation follows the reference constructors (cited per class).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

#: bumped whenever a module OF THIS PACKAGE (re-)registers a parameter, a buffer or a submodule (``layer.weight = nn.Parameter(...)``,
#: ``load_state_dict(assign=True)``, a replaced BatchNorm buffer, ``model.alpha_head = other``): ``ObjectComposer`` caches parameter
#: lists and raw-pointer structs of its module tree and rebuilds them when the tree may have changed under them.  Scoped to the
#: package's own classes (``Tracked`` below) - no process-wide ``torch.nn.modules.module.register_module_*_registration_hook``: other
#: people's modules constructed in the same process neither pay for nor disturb this bookkeeping.  A tree that contains a module
#: of another class (a user swapped a stock ``nn.Linear`` in) is never cached (``tree_is_tracked``).
REGISTRATION_EPOCH = [0]
_TREE_SLOTS = ("_parameters", "_buffers", "_modules")
_BUFFER_TYPE = getattr(nn, "Buffer", nn.Parameter)      # (torch >= 2.5: assigning an nn.Buffer registers a buffer)


class Tracked:
    """Mixin in front of ``nn.Module`` subclasses: every way a parameter / buffer / submodule can be (re-)registered or removed -
    ``__setattr__`` (which ``nn.Module`` uses for plain assignment and ``load_state_dict(assign=True)``), ``__delattr__``,
    ``register_parameter`` / ``register_buffer`` / ``add_module`` / ``register_module`` - moves ``REGISTRATION_EPOCH``.
    ``nn.DataParallel`` replicas (``_is_replica``) do not: they are per-call copies whose composers never cache."""

    def __setattr__(self, name, value):
        d = self.__dict__
        # (a plain tensor under a NEW name is an ordinary attribute - the composer's scratch, a noise seed word - not a registration)
        if not d.get("_is_replica") and (isinstance(value, (nn.Parameter, nn.Module, _BUFFER_TYPE)) or name in _TREE_SLOTS or
                                         any(name in (d.get(slot) or ()) for slot in _TREE_SLOTS)):
            REGISTRATION_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        REGISTRATION_EPOCH[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        REGISTRATION_EPOCH[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        REGISTRATION_EPOCH[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)

    def add_module(self, name, module):
        REGISTRATION_EPOCH[0] += 1
        super().add_module(name, module)

    def register_module(self, name, module):
        REGISTRATION_EPOCH[0] += 1
        super().register_module(name, module)

    def _replicate_for_data_parallel(self):
        epoch = REGISTRATION_EPOCH[0]
        replica = super()._replicate_for_data_parallel()       # (assigns the replica's _parameters / _buffers / _modules dictionaries)
        REGISTRATION_EPOCH[0] = epoch
        return replica


def tree_is_tracked(module: nn.Module) -> bool:
    """Every module of the tree is one of this package's tracked classes: its registrations cannot change unnoticed."""
    return all(isinstance(m, Tracked) for m in module.modules())


class Linear(Tracked, nn.Linear):
    pass


class BatchNorm1d(Tracked, nn.BatchNorm1d):
    pass


class ModuleList(Tracked, nn.ModuleList):
    # nn.ModuleList edits ``self._modules`` directly in these (``insert`` shifts entries by dictionary writes, ``pop`` / ``__delitem__``
    # rebuild the dictionary, ``__setitem__`` goes through ``setattr`` with a string index): count them as registrations
    def insert(self, index, module):
        REGISTRATION_EPOCH[0] += 1
        super().insert(index, module)

    def __delitem__(self, idx):
        REGISTRATION_EPOCH[0] += 1
        super().__delitem__(idx)

    def pop(self, key):
        REGISTRATION_EPOCH[0] += 1
        return super().pop(key)

    def __setitem__(self, idx, module):
        REGISTRATION_EPOCH[0] += 1
        super().__setitem__(idx, module)


class Sequential(Tracked, nn.Sequential):
    def insert(self, index, module):
        REGISTRATION_EPOCH[0] += 1
        return super().insert(index, module)

    def __delitem__(self, idx):
        REGISTRATION_EPOCH[0] += 1
        super().__delitem__(idx)

    def pop(self, key):
        REGISTRATION_EPOCH[0] += 1
        return super().pop(key)

    def __setitem__(self, idx, module):
        REGISTRATION_EPOCH[0] += 1
        super().__setitem__(idx, module)


class Conv2d(Tracked, nn.Conv2d):
    pass


class BatchNorm2d(Tracked, nn.BatchNorm2d):
    pass


class LeakyReLU(Tracked, nn.LeakyReLU):
    pass


class AvgPool2d(Tracked, nn.AvgPool2d):
    pass


class BoundingBox(Tracked, nn.Module):
    """Axis-aligned box, ``dimensions`` (3, 2) non-persistent buffer (utils/lib_3d/bounding_box.py:10-21)."""

    def __init__(self, dimensions):
        super().__init__()
        if len(dimensions) != 3:
            raise Exception(f"Dimenions should have dimension 3, but dimension ({len(dimensions)}) was passed")
        self.register_buffer("dimensions", torch.as_tensor(dimensions, dtype=torch.float32), persistent=False)

    def get_size(self) -> torch.Tensor:
        return self.dimensions[:, 1] - self.dimensions[:, 0]

    def get_center_offset(self, device=None) -> torch.Tensor:
        """Centre of the box relative to the canonical centre (0, 0, 0)  (bounding_box.py:23-33)."""
        centre = self.dimensions[:, 0] + (self.dimensions[:, 1] - self.dimensions[:, 0]) / 2
        return centre if device is None else centre.to(device)

    def get_corner_points(self) -> torch.Tensor:
        """(8, 3) corners in the reference's order (bounding_box.py:58-98): 0 = all-low, 6 = all-high."""
        lo, hi = self.dimensions[:, 0], self.dimensions[:, 1]
        pick = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
        both = torch.stack([lo, hi], 0)
        return torch.stack([torch.stack([both[p[a], a] for a in range(3)]) for p in pick], 0)

    def get_edge_points(self, points_per_edge: int = 5) -> torch.Tensor:
        """Corners + ``points_per_edge`` interior points on each of the 12 edges (bounding_box.py:100-131)."""
        edges = [0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 0, 4, 1, 5, 2, 6, 3, 7]
        corners = self.get_corner_points()
        ends = corners[torch.as_tensor(edges, device=corners.device)].reshape(12, 2, 3)
        frac = torch.linspace(0.0, 1.0, points_per_edge + 2, device=corners.device)[1:-1]
        pts = ends[:, 0].unsqueeze(-1) + (ends[:, 1] - ends[:, 0]).unsqueeze(-1) * frac
        pts = pts.transpose(1, 2).reshape(-1, 3)
        return torch.cat([corners, pts], dim=0)


class _Normalization(Tracked, nn.Module):
    """``ada_in`` level: holds ``normalization`` = BatchNorm1d(affine=False) (model/layers/adain.py:44-47)."""

    def __init__(self, features: int):
        super().__init__()
        self.normalization = BatchNorm1d(features, affine=False)


class AffineTransformAdaIn(Tracked, nn.Module):
    """Style affine + AdaIN statistics (model/layers/adain.py:5-19): scale biased to 1, bias to 0."""

    def __init__(self, in_features: int, style_features_count: int):
        super().__init__()
        self.style_features_count = style_features_count
        self.affine_transform = Linear(style_features_count, 2 * in_features)
        self.ada_in = _Normalization(in_features)
        self.affine_transform.bias.data[:in_features] = 1
        self.affine_transform.bias.data[in_features:] = 0


class _Placeholder(Tracked, nn.Module):
    """Parameter-free slot (the ReLUs of the reference's AdaInSequential) that keeps the indices."""


def _features_head(width: int, style: int, out: int) -> nn.Sequential:
    # indices 0,1,3,4,6 carry parameters (model/nerf_models/adain_style_nerf_model.py:57-71)
    return Sequential(
        Linear(width, width, bias=False),
        AffineTransformAdaIn(width, style),
        _Placeholder(),
        Linear(width, width // 2, bias=False),
        AffineTransformAdaIn(width // 2, style),
        _Placeholder(),
        Linear(width // 2, out),
    )


class AdaInStyleNerfModel(Tracked, nn.Module):
    """Weights of model/nerf_models/adain_style_nerf_model.py:14-55 (input: 3-D position)."""
    input_dimensions = 3
    kind = 0

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.model_config = model_config
        self.layers_width = model_config["layers_width"]
        self.backbone_layers_count = model_config["backbone_layers_count"]
        self.output_features = model_config["output_features"]
        self.skip_layer_idx = model_config["skip_layer_idx"]
        self.style_features = model_config["style_features"]
        self.empty_space_alpha = model_config["empty_space_alpha"]
        if self.skip_layer_idx >= self.backbone_layers_count:
            raise Exception("Skip layer must refer to a valid backbone layer idx")
        pe = model_config["position_encoder"]
        if not pe["append_original"]:
            raise Exception("position_encoder.append_original=False is not supported by the HIP renderer")
        self.octaves = pe["octaves"]
        self.encoding_size = self.input_dimensions * (1 + 2 * self.octaves)
        self.bounding_box = BoundingBox(model_config["bounding_box"])
        self.backbone_layers = ModuleList()
        size = self.encoding_size
        for idx in range(self.backbone_layers_count):
            if idx == self.skip_layer_idx:
                size += self.encoding_size
            self.backbone_layers.append(Linear(size, self.layers_width))
            size = self.layers_width
        if self.kind == 0:
            self.alpha_head = Linear(self.layers_width, 1)
        self.features_head = _features_head(self.layers_width, self.style_features, self.output_features)


class SkyboxAdaInStyleNerfModelV3(AdaInStyleNerfModel):
    """Weights of model/nerf_models/skybox_adain_style_nerf_model_v3.py:14-65 (input: origin + unit
    direction, 6-D; no sigma head, sigma == 10)."""
    input_dimensions = 6
    kind = 1


class _AnnealableEncoderState(Tracked, nn.Module):
    """``positional_encoder`` level of the bender: int32 ``current_step`` buffer
    (model/annealable_positional_encoder.py:26-44)."""

    def __init__(self, octaves: int, num_steps: int):
        super().__init__()
        self.octaves_count = octaves
        self.num_steps = num_steps
        self.register_buffer("current_step", torch.zeros((), dtype=torch.int))

    def set_step(self, current_step: int):
        # in place: the buffer keeps its storage and its version counter moves, which is what the composer's host-side caches of
        # the octave weights key on (a fresh tensor per call could come back at a recycled address with version 0)
        self.current_step.fill_(current_step)

    def annealing_weights(self) -> torch.Tensor:
        """(1 - cos(pi clamp(step * octaves / num_steps - k, 0, 1))) / 2  (annealable_positional_encoder.py:59-63)."""
        alpha = self.current_step * self.octaves_count / self.num_steps
        k = torch.arange(self.octaves_count, dtype=torch.float32, device=self.current_step.device)
        return (1 - torch.cos(math.pi * torch.clamp(alpha - k, min=0.0, max=1.0))) / 2


class PositionalRayBender(Tracked, nn.Module):
    """Weights of model/nerf_models/positional_ray_bender_model.py:12-79."""
    has_weights = True

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.model_config = model_config
        self.layers_width = model_config["layers_width"]
        self.layers_count = model_config["layers_count"]
        self.skip_layer_idx = model_config["skip_layer_idx"]
        self.deformation_features = model_config["deformation_features"]
        pe = model_config["position_encoder"]
        if not pe["append_original"]:
            raise Exception("position_encoder.append_original=False is not supported by the HIP renderer")
        self.positional_encoder = _AnnealableEncoderState(pe["octaves"], pe["num_steps"])
        self.encoding_size = 3 * (1 + 2 * pe["octaves"])
        self.bounding_box = BoundingBox(model_config["bounding_box"])
        self.backbone_layers = ModuleList()
        size = self.encoding_size + self.deformation_features
        for idx in range(self.layers_count):
            if idx == self.skip_layer_idx:
                size += self.encoding_size + self.deformation_features
            self.backbone_layers.append(Linear(size, self.layers_width))
            size = self.layers_width
        self.output_head = Linear(self.layers_width, 3, bias=False)
        for layer in self.backbone_layers:
            torch.nn.init.kaiming_uniform_(layer.weight, a=0, mode="fan_in", nonlinearity="relu")
            torch.nn.init.zeros_(layer.bias)
        torch.nn.init.uniform_(self.backbone_layers[-1].weight, a=-1e-5, b=1e-5)

    def set_step(self, current_step: int):
        self.positional_encoder.set_step(current_step)


class ZeroedRayBender(Tracked, nn.Module):
    """model/nerf_models/zeroed_ray_bender_model.py:7-37: no parameters, displacement == 0."""
    has_weights = False

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.model_config = model_config

    def set_step(self, current_step: int):
        pass


_NERF_CLASSES = {
    "model.nerf_models.adain_style_nerf_model": AdaInStyleNerfModel,
    "model.nerf_models.skybox_adain_style_nerf_model_v3": SkyboxAdaInStyleNerfModelV3,
}
_BENDER_CLASSES = {
    "model.nerf_models.positional_ray_bender_model": PositionalRayBender,
    "model.nerf_models.zeroed_ray_bender_model": ZeroedRayBender,
}


class RayBendingStyleNerfModel(Tracked, nn.Module):
    """One object model: NeRF + ray bender + box (model/nerf_models/ray_bending_style_nerf_model.py:12-60).

    The ``architecture`` strings of the reference configs are mapped to the classes above (the
    reference resolves them with importlib; the dotted names are kept as keys so its YAML works)."""

    def __init__(self, config: Dict, model_config: Dict):
        super().__init__()
        self.config = config
        self.model_config = model_config
        self.empty_space_alpha = model_config["empty_space_alpha"]
        self.bounding_box = BoundingBox(model_config["bounding_box"])
        self.style_features = model_config["style_features"]
        self.deformation_features = model_config["deformation_features"]
        self.nerf_model_config = model_config["nerf_model"]
        self.ray_bender_model_config = model_config["ray_bender_model"]
        for sub in (self.nerf_model_config, self.ray_bender_model_config):
            sub["bounding_box"] = model_config["bounding_box"]
            sub["empty_space_alpha"] = model_config["empty_space_alpha"]
            sub["style_features"] = model_config["style_features"]
            sub["deformation_features"] = model_config["deformation_features"]
        try:
            nerf_cls = _NERF_CLASSES[self.nerf_model_config["architecture"]]
            bender_cls = _BENDER_CLASSES[self.ray_bender_model_config["architecture"]]
        except KeyError as e:
            raise Exception(f"object model architecture {e} is not supported by the HIP renderer")
        self.nerf_model = nerf_cls(config, self.nerf_model_config)
        self.ray_bender = bender_cls(config, self.ray_bender_model_config)

    def set_step(self, current_step: int):
        self.ray_bender.set_step(current_step)


OBJECT_MODEL_CLASSES = {"model.nerf_models.ray_bending_style_nerf_model": RayBendingStyleNerfModel}


class CameraParametersStorage(Tracked, nn.Module):
    """Learnable per-frame corrections of the measured cameras: 3 rotation, 3 translation and 1 focal offset per
    (frame, camera), zero-initialised, read only in training mode (zeros in evaluation), translations scaled by 10 and
    focals by 1000 (model/layers/camera_parameters_storage.py:9-67 over model/layers/indexed_storage.py:9-58).

    The reference keeps one 7-vector ``nn.Parameter`` per entry and picks them in a Python loop with ``.item()``; here the
    table is ONE ``(entries, 7)`` parameter read with one device gather.  The checkpoint format is the reference's
    (``storage.storage.{entry}`` -> ``(7,)``): rows are split / joined when a state dict is written / read.

    Optimiser semantics differ from the reference's per-entry parameters (both shipped configurations leave the offsets off):
    an entry that is not in the batch has ``grad = None`` there and is skipped by the optimiser, here its row of the table's
    gradient is zero, so Adam still applies its momentum / weight decay to it; and the optimiser state is one tensor, so a
    reference optimiser checkpoint does not resume.  Train the table with ``torch.optim.SparseAdam`` semantics (or mask the
    update to the rows of ``frame_indexes``) where that matters."""

    features_count = 7

    def __init__(self, storage_size: int, cameras_count: int):
        super().__init__()
        self.storage_size = int(storage_size)
        self.cameras_count = int(cameras_count)
        self.camera_adjusted_storage_size = self.storage_size * self.cameras_count
        self.table = nn.Parameter(torch.zeros(self.camera_adjusted_storage_size, self.features_count))

    def forward(self, frame_indexes: torch.Tensor):
        steps = torch.arange(self.cameras_count, device=frame_indexes.device) * self.storage_size
        entries = frame_indexes.unsqueeze(-1) + steps                       # (..., cameras): frame + camera * storage_size
        if self.training:
            rows = self.table[entries.long()]
            rotation, translation, focal = rows[..., :3], rows[..., 3:6], rows[..., 6]
        else:
            shape = list(entries.shape)
            rotation = torch.zeros(shape + [3], dtype=torch.float32, device=frame_indexes.device)
            translation = torch.zeros(shape + [3], dtype=torch.float32, device=frame_indexes.device)
            focal = torch.zeros(shape, dtype=torch.float32, device=frame_indexes.device)
        return rotation, translation * 10, focal * 1000

    # ---- the reference's checkpoint layout
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        rows = self.table if keep_vars else self.table.detach().clone()
        for entry, row in enumerate(rows.unbind(0)):
            destination[f"{prefix}storage.storage.{entry}"] = row

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        rows = []
        for entry in range(self.camera_adjusted_storage_size):
            key = f"{prefix}storage.storage.{entry}"
            if key in state_dict:
                value = state_dict[key]
                if tuple(value.shape) != (self.features_count,):
                    error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(value.shape)} from "
                                      f"checkpoint, the shape in current model is ({self.features_count},).")
                    continue
                rows.append((entry, value))
            elif strict:
                missing_keys.append(key)
        if strict:
            own = {f"{prefix}storage.storage.{e}" for e in range(self.camera_adjusted_storage_size)}
            unexpected_keys.extend(key for key in state_dict if key.startswith(prefix) and key not in own)
        if rows:
            with torch.no_grad():       # one scatter instead of one copy per entry (tables hold tens of thousands of frames)
                index = torch.tensor([entry for entry, _ in rows], device=self.table.device)
                values = torch.stack([value.detach().to(device=self.table.device, dtype=self.table.dtype) for _, value in rows])
                self.table.index_copy_(0, index, values)
