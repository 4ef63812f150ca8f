"""Renderer configurations in the reference's own dictionary schema.

The product modules accept the same ``config`` dictionary the reference builds from its YAML
(``config["model"]["object_models"][i]["nerf_model"]...``, utils/configuration.py:30-242), so a
dictionary loaded from a reference YAML drops in unchanged.  Because the reference YAML files do
not travel with this repository, the two shipped renderer configurations are rebuilt here
programmatically; every value is the one in the cited YAML line.

  tennis    : configs/tennis/193_tennis_v7_...yaml   (:29 focal multiplier, :37-40 flags,
              :60 static models, :120-370 object models)
  minecraft : configs/minecraft/013_minecraft_v1_...yaml (:28, :36, :56, :108-290)
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional, Sequence

NERF_ADAIN = "model.nerf_models.adain_style_nerf_model"
NERF_SKYBOX = "model.nerf_models.skybox_adain_style_nerf_model_v3"
BENDER_ZERO = "model.nerf_models.zeroed_ray_bender_model"
BENDER_POSITIONAL = "model.nerf_models.positional_ray_bender_model"
OBJECT_MODEL = "model.nerf_models.ray_bending_style_nerf_model"


def _nerf(architecture: str, width=256, layers=8, features=192, skip=4, octaves=10) -> dict:
    return {
        "architecture": architecture,
        "layers_width": width,
        "backbone_layers_count": layers,
        "output_features": features,
        "skip_layer_idx": skip,
        "position_encoder": {"octaves": octaves, "append_original": True},
    }


def _bender(positional: bool, width=128, layers=6, skip=3, octaves=6, num_steps=60000) -> dict:
    if not positional:
        return {"architecture": BENDER_ZERO}
    return {
        "architecture": BENDER_POSITIONAL,
        "layers_width": width,
        "layers_count": layers,
        "skip_layer_idx": skip,
        "position_encoder": {"octaves": octaves, "append_original": True, "num_steps": num_steps},
    }


def _object(name, bbox, coarse, fine, z_near, z_far, style, nerf, bender, use_fine=False,
            deformation=32, empty_alpha=-3.5) -> dict:
    return {
        "name": name,
        "architecture": OBJECT_MODEL,
        "bounding_box": [list(map(float, b)) for b in bbox],
        "positions_count_coarse": coarse,
        "positions_count_fine": fine,
        "use_fine": use_fine,
        "empty_space_alpha": empty_alpha,
        "z_near_min": z_near,
        "z_far_max": z_far,
        "deformation_features": deformation,
        "style_features": style,
        "nerf_model": nerf,
        "ray_bender_model": bender,
    }


def _wrap(focal_multiplier, fix_overlaps, static_models, objects: List[dict], counts: Sequence[int],
          sampling_weights, strides=(4, 8), encoders: Optional[dict] = None) -> dict:
    cfg = _wrap_renderer(focal_multiplier, fix_overlaps, static_models, objects, counts, sampling_weights, strides)
    if encoders is not None:
        # the sections EnvironmentModel builds its CNN object encoders / pose estimators from (playableenvironments_amd.encoders)
        cfg["model"]["object_encoders"] = copy.deepcopy(encoders["object_encoders"])
        cfg["model"]["object_parameters_encoder"] = copy.deepcopy(encoders["object_parameters_encoder"])
    return cfg


_ZERO_RANGE = [[[-0.0, 0.0], [-0.0, 0.0], [-0.0, 0.0]]]

#: object_encoders / object_parameters_encoder sections of the shipped tennis configuration (configs/tennis/193_...yaml)
TENNIS_ENCODERS = {
    "object_encoders": [
        {"background": None, "architecture": "model.object_encoder_v5", "input_size": [64, 256], "style_features": 64, "deformation_features": 32},
        {"background_backplate": None, "architecture": "model.object_encoder_v5", "input_size": [32, 256], "style_features": 64, "deformation_features": 32},
        {"player_1": None, "architecture": "model.object_encoder_v4", "input_size": [64, 64], "style_features": 64, "deformation_features": 32},
        {"player_2": None, "architecture": "model.object_encoder_v4", "input_size": [64, 64], "style_features": 64, "deformation_features": 32},
    ],
    "object_parameters_encoder": [
        {"background": None, "architecture": "model.static_object_parameters_encoder", "objects_count": 1,
         "translation_range": _ZERO_RANGE, "rotation_range": _ZERO_RANGE},
        {"background_backplate": None, "architecture": "model.static_object_parameters_encoder", "objects_count": 1,
         "translation_range": [[[-0.0, 0.0], [20.085, 20.085], [-0.0, 0.0]]], "rotation_range": _ZERO_RANGE},
        {"player_1": None, "architecture": "model.classic_object_parameters_encoder", "objects_count": 1,
         "translation_range": [[[-7.5, 7.5], [-20.0, 0.0], [0.01, 0.01]]], "rotation_range": _ZERO_RANGE},
        {"player_2": None, "architecture": "model.classic_object_parameters_encoder", "objects_count": 1,
         "translation_range": [[[-7.5, 7.5], [-0.0, 20.0], [0.01, 0.01]]], "rotation_range": _ZERO_RANGE},
    ],
}

#: the same sections of the shipped minecraft configuration (configs/minecraft/013_...yaml)
MINECRAFT_ENCODERS = {
    "object_encoders": [
        {"background": None, "architecture": "model.object_encoder_v5", "input_size": [64, 256], "style_features": 32, "deformation_features": 32},
        {"skybox": None, "architecture": "model.object_encoder_v5", "input_size": [144, 256], "style_features": 32, "deformation_features": 32},
        {"player_1": None, "architecture": "model.object_encoder_v4", "input_size": [64, 64], "style_features": 32, "deformation_features": 32,
         "expansion_factor": {"rows": 2.8, "cols": 2}},
    ],
    "object_parameters_encoder": [
        {"background": None, "architecture": "model.static_object_parameters_encoder", "objects_count": 1,
         "translation_range": _ZERO_RANGE, "rotation_range": _ZERO_RANGE},
        {"skybox": None, "architecture": "model.static_object_parameters_encoder", "objects_count": 1,
         "translation_range": _ZERO_RANGE, "rotation_range": _ZERO_RANGE},
        {"player_1": None, "architecture": "model.object_parameters_encoder_v4", "objects_count": 2, "input_size": [64, 64],
         "edge_to_center_distance": 0.0, "expansion_factor": {"rows": 2.8, "cols": 2}},
    ],
}


def _wrap_renderer(focal_multiplier, fix_overlaps, static_models, objects: List[dict], counts: Sequence[int],
                   sampling_weights, strides=(4, 8)) -> dict:
    return {
        "data": {"focal_length_multiplier": focal_multiplier},
        "model": {
            "apply_activation": False,
            "fix_object_overlaps": fix_overlaps,
            "static_object_models": static_models,
            "use_weighted_sampling": True,
            "sampling_weights": list(sampling_weights),
            "object_parameters_encoder": [{"objects_count": int(c)} for c in counts],
            "object_encoders": [{} for _ in objects],
            "object_models": objects,
            # derived by utils/configuration.py:146-158 from downsampling_layers_count [2, 1]
            "autoencoder": {"downsample_factor": list(strides)},
        },
    }


def tennis_config(hierarchical: Optional[Sequence[int]] = None, encoders: bool = False) -> dict:
    """Shipped tennis renderer: background (P=4), backplate (P=4), two players (P=32, ray bender).

    ``hierarchical=(Pc, Pf)`` is the benchmark override of BASELINE.json configs[1]: every object
    gets ``use_fine`` with ``Pc`` coarse + ``Pf`` resampled positions.
    ``encoders=True`` adds the shipped object-encoder / object-parameters-encoder sections, from which
    ``EnvironmentModel(config)`` builds its CNN encoders itself (otherwise they are injected)."""
    s = 64
    objs = [
        _object("background", [[-30.0, 30.0], [-40.0, 20.585], [-0.5, 0.0]], 4, 4, 5.0, 70.0, s,
                _nerf(NERF_ADAIN), _bender(False)),
        _object("background_backplate", [[-30.0, 30.0], [0.0, 0.5], [-0.0, 30.0]], 4, 4, 5.0, 70.0, s,
                _nerf(NERF_ADAIN), _bender(False)),
        _object("player_1", [[-0.75, 0.75], [-0.5, 0.5], [-0.0, 2.15]], 32, 32, 5.0, 70.0, s,
                _nerf(NERF_ADAIN), _bender(True)),
        _object("player_2", [[-0.75, 0.75], [-0.5, 0.5], [-0.0, 2.15]], 32, 32, 5.0, 70.0, s,
                _nerf(NERF_ADAIN), _bender(True)),
    ]
    if hierarchical is not None:
        pc, pf = hierarchical
        for o in objs:
            o["positions_count_coarse"], o["positions_count_fine"], o["use_fine"] = int(pc), int(pf), True
    return _wrap(0.51417, False, 2, objs, [1, 1, 1, 1], [0.55, 0.15, 0.15, 0.15], encoders=TENNIS_ENCODERS if encoders else None)


def tennis_single_player_config(positions: int = 32) -> dict:
    """BASELINE.json configs[0]: one tennis ``player_1`` object (NeRF + ray bender), P=32."""
    objs = [_object("player_1", [[-0.75, 0.75], [-0.5, 0.5], [-0.0, 2.15]], positions, positions, 5.0, 70.0,
                    64, _nerf(NERF_ADAIN), _bender(True))]
    return _wrap(0.51417, False, 0, objs, [1], [1.0])


def minecraft_config(encoders: bool = False) -> dict:
    """Shipped minecraft renderer: background (P=16), skybox (P=1, 6-D input, opaque), one player
    model shared by two object instances (P=32, ray bender); overlap fix on (default)."""
    s = 32
    objs = [
        _object("background", [[-10.0, 10.0], [-0.6, 2.0], [-10.0, 10.0]], 16, 16, 0.05, 30.0, s,
                _nerf(NERF_ADAIN), _bender(False)),
        _object("skybox", [[-200.0, 200.0], [-200.0, 200.0], [-200.0, 200.0]], 1, 1, 90.0, 91.0, s,
                _nerf(NERF_SKYBOX), _bender(False)),
        _object("player_1", [[-0.6, 0.6], [-0.0, 2.1], [-1.2, 1.2]], 32, 32, 0.05, 30.0, s,
                _nerf(NERF_ADAIN), _bender(True)),
    ]
    return _wrap(0.5, True, 2, objs, [1, 1, 2], [0.0, 0.70, 0.15, 0.15], encoders=MINECRAFT_ENCODERS if encoders else None)


def reduced_config(base: dict, width=32, layers=4, skip=2, features=16, octaves=4,
                   bender_width=16, bender_layers=3, bender_skip=1, bender_octaves=3,
                   positions: Optional[Dict[str, Sequence[int]]] = None) -> dict:
    """Shrinks every network of ``base`` (used for the small golden fixtures).

    Widths must stay multiples of 32 for the backbone and 16 for the bender hidden layers
    (MFMA tile granularity of the HIP kernels)."""
    cfg = copy.deepcopy(base)
    for o in cfg["model"]["object_models"]:
        n = o["nerf_model"]
        n.update(layers_width=width, backbone_layers_count=layers, skip_layer_idx=skip, output_features=features)
        n["position_encoder"]["octaves"] = octaves
        b = o["ray_bender_model"]
        if b["architecture"] == BENDER_POSITIONAL:
            b.update(layers_width=bender_width, layers_count=bender_layers, skip_layer_idx=bender_skip)
            b["position_encoder"]["octaves"] = bender_octaves
        if positions and o["name"] in positions:
            pc, pf = positions[o["name"]]
            o["positions_count_coarse"], o["positions_count_fine"] = int(pc), int(pf)
    return cfg


def enable_fine(base: dict, coarse: Optional[int] = None, fine: Optional[int] = None) -> dict:
    """Returns a copy with ``use_fine`` on for every object (the reference requires all-or-nothing,
    SURVEY.md section 7 item 8)."""
    cfg = copy.deepcopy(base)
    for o in cfg["model"]["object_models"]:
        o["use_fine"] = True
        if coarse is not None:
            o["positions_count_coarse"] = int(coarse)
        if fine is not None:
            o["positions_count_fine"] = int(fine)
    return cfg
