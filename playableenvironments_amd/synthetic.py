"""Seeded synthetic scene encodings (cameras, object poses, style, deformation) and weights.

There is no dataset on the GPU box; tests and ``bench.py`` render these scenes (SURVEY.md
section 8d).  Everything is produced from ``numpy.random.default_rng(seed)`` so the CPU oracle and
the HIP renderer see identical inputs.  Shapes follow EnvironmentModel.forward_from_scene_encoding
(model/environment_model.py:1041-1065 of the reference): leading dims are (batch, observations),
cameras_count = 1.
"""
from __future__ import annotations

import math
import re
from typing import Dict

import numpy as np
import torch


def _t(a) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a), dtype=torch.float32)


def tennis_scene(batch: int = 1, observations: int = 1, seed: int = 1234, image_size=(256, 256),
                 style_features: int = 64, deformation_features: int = 32, objects: int = 4) -> Dict:
    """Tennis world: z up, court in the xy plane, camera behind the near baseline looking along +y.

    camera rot = (1.25 + U(-.05,.05), U(-.1,.1), 0), trans = (U(-1,1), -28 + U(-2,2), 9 + U(-1,1));
    focal so that the frame spans the shipped FOV (1400 px at 512x288 before the 0.51417 multiplier);
    players at (U(-2.5,2.5), -9 / +11 + U(-2,2), 0.01) so both stay inside the narrow frustum."""
    rng = np.random.default_rng(seed)
    lead = (batch, observations)
    cam_rot = np.zeros(lead + (1, 3), np.float32)
    cam_rot[..., 0] = 1.25 + rng.uniform(-0.05, 0.05, lead + (1,))
    cam_rot[..., 1] = rng.uniform(-0.1, 0.1, lead + (1,))
    cam_tr = np.zeros(lead + (1, 3), np.float32)
    cam_tr[..., 0] = rng.uniform(-1, 1, lead + (1,))
    cam_tr[..., 1] = -28 + rng.uniform(-2, 2, lead + (1,))
    cam_tr[..., 2] = 9 + rng.uniform(-1, 1, lead + (1,))
    focals = np.full(lead + (1,), 1400.0 * image_size[0] / 288.0, np.float32)

    rot = np.zeros(lead + (3, objects), np.float32)
    tr = np.zeros(lead + (3, objects), np.float32)
    if objects == 4:
        for k, sign in ((2, -1.0), (3, 1.0)):
            tr[..., 0, k] = rng.uniform(-2.5, 2.5, lead)
            tr[..., 1, k] = (-9.0 if sign < 0 else 11.0) + rng.uniform(-2, 2, lead)
            tr[..., 2, k] = 0.01
    style = rng.standard_normal(lead + (style_features, objects)).astype(np.float32)
    deformation = rng.standard_normal(lead + (deformation_features, objects)).astype(np.float32)
    in_scene = np.ones(lead + (objects,), bool)
    return {
        "camera_rotations": _t(cam_rot), "camera_translations": _t(cam_tr), "focals": _t(focals),
        "image_size": tuple(image_size),
        "object_rotation_parameters": _t(rot), "object_translation_parameters": _t(tr),
        "object_style": _t(style), "object_deformation": _t(deformation),
        "object_in_scene": torch.as_tensor(in_scene),
    }


def single_player_scene(seed: int = 1234, image_size=(128, 128), style_features=64,
                        deformation_features=32) -> Dict:
    """BASELINE.json configs[0]: one tennis player, camera zoomed so every ray crosses the box
    (worst-case MLP load)."""
    rng = np.random.default_rng(seed)
    lead = (1, 1)
    # Camera on the -y side of the player, 8 m away, looking along +y at the box centre (z = 1.075);
    # the 1.5 x 2.15 m box face must cover the whole frustum -> narrow field of view.
    cam_rot = np.zeros(lead + (1, 3), np.float32)
    cam_rot[..., 0] = math.pi / 2
    cam_tr = np.zeros(lead + (1, 3), np.float32)
    cam_tr[..., 1] = -8.0
    cam_tr[..., 2] = 1.075
    # half-width of the frustum at the far face (y = +0.5 -> depth 8.5) must stay below 0.75
    focal_px = 0.5 * image_size[1] * 8.5 / 0.70
    focals = np.full(lead + (1,), focal_px / 0.51417, np.float32)
    rot = np.zeros(lead + (3, 1), np.float32)
    tr = np.zeros(lead + (3, 1), np.float32)
    style = rng.standard_normal(lead + (style_features, 1)).astype(np.float32)
    deformation = rng.standard_normal(lead + (deformation_features, 1)).astype(np.float32)
    return {
        "camera_rotations": _t(cam_rot), "camera_translations": _t(cam_tr), "focals": _t(focals),
        "image_size": tuple(image_size),
        "object_rotation_parameters": _t(rot), "object_translation_parameters": _t(tr),
        "object_style": _t(style), "object_deformation": _t(deformation),
        "object_in_scene": torch.ones(lead + (1,), dtype=torch.bool),
    }


def minecraft_scene(batch: int = 1, observations: int = 1, seed: int = 1234, image_size=(256, 256),
                    style_features: int = 32, deformation_features: int = 32) -> Dict:
    """Minecraft world: y up.  camera rot = (-0.30, 1.57 + U(-.1,.1), 0), trans = (12, 4, 0) + U(-1,1)^3,
    focal 750 px at 512x288 before the 0.5 multiplier; players at (U(-3,3), 0, U(-3,3)) with a random
    y rotation.  Objects: background, skybox, player, player."""
    rng = np.random.default_rng(seed)
    lead = (batch, observations)
    objects = 4
    cam_rot = np.zeros(lead + (1, 3), np.float32)
    cam_rot[..., 0] = -0.30
    cam_rot[..., 1] = 1.57 + rng.uniform(-0.1, 0.1, lead + (1,))
    cam_tr = np.zeros(lead + (1, 3), np.float32)
    cam_tr[..., :] = np.array([12.0, 4.0, 0.0], np.float32) + rng.uniform(-1, 1, lead + (1, 3))
    focals = np.full(lead + (1,), 750.0 * image_size[0] / 288.0, np.float32)
    rot = np.zeros(lead + (3, objects), np.float32)
    tr = np.zeros(lead + (3, objects), np.float32)
    for k in (2, 3):
        tr[..., 0, k] = rng.uniform(-3, 3, lead)
        tr[..., 2, k] = rng.uniform(-3, 3, lead)
        rot[..., 1, k] = rng.uniform(-math.pi, math.pi, lead)
    style = rng.standard_normal(lead + (style_features, objects)).astype(np.float32)
    deformation = rng.standard_normal(lead + (deformation_features, objects)).astype(np.float32)
    return {
        "camera_rotations": _t(cam_rot), "camera_translations": _t(cam_tr), "focals": _t(focals),
        "image_size": tuple(image_size),
        "object_rotation_parameters": _t(rot), "object_translation_parameters": _t(tr),
        "object_style": _t(style), "object_deformation": _t(deformation),
        "object_in_scene": torch.ones(lead + (objects,), dtype=torch.bool),
    }


def observation_batch(scene, boxes_seed: int = 3, dynamic_objects: int = 2):
    """Synthetic dataset tensors for the observation-driven modes on top of a synthetic scene's cameras: smooth images,
    annotated boxes of the dynamic objects, index tensors.  Returns a dict keyed like Batch.to_tuple()
    (dataset/batching.py:252-264)."""
    g = torch.Generator().manual_seed(boxes_seed)
    cam = scene["camera_rotations"]
    lead = list(cam.shape[:-1])                                                        # (bs, O, C)
    h, w = scene["image_size"]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    base = torch.stack([xx, yy, xx * yy], 0)                                           # (3, H, W)
    tint = torch.rand(lead + [3, 1, 1], generator=g)
    observations = (base * 0.5 + tint * 0.5).contiguous()
    centre = 0.3 + 0.4 * torch.rand(lead + [2, dynamic_objects], generator=g)
    half = 0.05 + 0.1 * torch.rand(lead + [2, dynamic_objects], generator=g)
    boxes = torch.cat([centre - half, centre + half], dim=-2).clamp(0, 1)              # [left, top, right, bottom]
    validity = torch.ones(lead + [dynamic_objects], dtype=torch.bool)
    bs, obs_count = lead[0], lead[1]
    frame = torch.arange(bs * obs_count).reshape(bs, obs_count)
    return {"observations": observations, "camera_rotations": cam, "camera_translations": scene["camera_translations"],
            "focals": scene["focals"], "bounding_boxes": boxes, "bounding_boxes_validity": validity,
            "global_frame_indexes": frame, "video_frame_indexes": frame.clone(), "video_indexes": torch.arange(bs)}


def randomize_module_state(module: torch.nn.Module, seed: int = 0, step: int = 60000,
                           alpha_bias: float = 0.0, bender_scale: float = 1.0) -> None:
    """Deterministic non-trivial weights for an (untrained) composer: the module keeps its default
    initialisation drawn under ``torch.manual_seed(seed)`` by the caller; here the BatchNorm running
    statistics are set to mean ~ N(0, .1), var ~ U(.5, 1.5) so the eval-mode AdaIN affine is not the
    identity, the sigma-head bias can be raised so that occlusion matters, the last bender layer can
    be scaled up so that displacements are visible, and the annealing step is set."""
    g = torch.Generator().manual_seed(seed + 977)
    with torch.no_grad():
        for name, buf in module.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
        for name, p in module.named_parameters():
            if name.endswith("alpha_head.bias"):
                p.add_(alpha_bias)
        if bender_scale != 1.0:
            # the reference initialises the LAST bender backbone layer to U(-1e-5, 1e-5)
            # (model/nerf_models/positional_ray_bender_model.py:66-79): scale that one
            last = {}
            for name, p in module.named_parameters():
                m = re.match(r"(.*ray_bender\.backbone_layers\.)(\d+)\.weight$", name)
                if m:
                    last[m.group(1)] = max(last.get(m.group(1), -1), int(m.group(2)))
            params = dict(module.named_parameters())
            for prefix, idx in last.items():
                params[f"{prefix}{idx}.weight"].mul_(bender_scale)
    if hasattr(module, "set_step"):
        module.set_step(step)
