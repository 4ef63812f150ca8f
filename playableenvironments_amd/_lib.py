"""ctypes binding of libplayrender.so (the C ABI declared in include/playrender.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C playableenvironments_amd/csrc``).
There is no CPU fallback: if the shared object is missing, ``load()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PR_MAX_OBJECTS = 8
PR_MAX_LAYERS = 12
PR_MAX_OCTAVES = 16

PR_FLAG_PERTURB = 1
PR_FLAG_CANONICAL_POSE = 2
PR_FLAG_FIX_OVERLAPS = 4
PR_FLAG_NAIVE_MLP = 8
PR_FLAG_TRAIN_BN = 16
PR_FLAG_SAVE_FOR_BACKWARD = 32
PR_FLAG_GATE_HEAD = 64
PR_FLAG_DEVICE_NOISE = 128
PR_FLAG_DIVERGENCE_GRAD = 256
PR_FLAG_SIGMOID_FEATURES = 512
PR_FLAG_SPLIT_BACKWARD = 1024
PR_PRECISION_FP32 = 0
PR_PRECISION_F16X3 = 1
PR_PRECISION_F16 = 2
PR_PROFILE_CATEGORIES = 8   # host array length of pr_profile_collect

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Linear(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("out_features", C.c_int32), ("in_features", C.c_int32)]


class ObjectModel(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("has_bender", C.c_int32), ("positions", C.c_int32),
        ("style_features", C.c_int32), ("deformation_features", C.c_int32), ("output_features", C.c_int32),
        ("layers_width", C.c_int32), ("backbone_count", C.c_int32), ("skip_layer_idx", C.c_int32),
        ("octaves", C.c_int32), ("bender_width", C.c_int32), ("bender_count", C.c_int32),
        ("bender_skip", C.c_int32), ("bender_octaves", C.c_int32),
        ("bender_octave_weights", C.c_float * PR_MAX_OCTAVES),
        ("bbox", C.c_float * 6),
        ("empty_space_alpha", C.c_float), ("z_near_min", C.c_float), ("z_far_max", C.c_float), ("bn_eps", C.c_float),
        ("backbone", Linear * PR_MAX_LAYERS),
        ("alpha_head", Linear),
        ("head0", Linear),
        ("affine1", Linear), ("bn1_mean", C.c_void_p), ("bn1_var", C.c_void_p), ("bn1_batches", C.c_void_p),
        ("head3", Linear),
        ("affine4", Linear), ("bn4_mean", C.c_void_p), ("bn4_var", C.c_void_p), ("bn4_batches", C.c_void_p),
        ("head6", Linear),
        ("bender", Linear * PR_MAX_LAYERS),
        ("bender_out", Linear),
    ]


class Object(C.Structure):
    _fields_ = [("coarse", ObjectModel), ("packed_coarse", C.c_void_p), ("fine", ObjectModel), ("packed_fine", C.c_void_p)]


class Noise(C.Structure):
    _fields_ = [
        ("jitter", C.c_void_p * PR_MAX_OBJECTS), ("alpha", C.c_void_p * PR_MAX_OBJECTS),
        ("pdf", C.c_void_p * PR_MAX_OBJECTS), ("integrate", C.c_void_p * PR_MAX_OBJECTS),
        ("integrate_global", C.c_void_p), ("divergence", C.c_void_p * PR_MAX_OBJECTS),
    ]


class Entry(C.Structure):
    _fields_ = [
        ("integrated_features", C.c_void_p), ("opacity", C.c_void_p), ("weights", C.c_void_p), ("depth", C.c_void_p),
        ("disparity", C.c_void_p), ("integrated_displacements_magnitude", C.c_void_p), ("integrated_divergence", C.c_void_p),
    ]


ENTRY_FIELDS = [f[0] for f in Entry._fields_]


PR_MAX_DECODER_GROUPS = 4


class DecoderLayout(C.Structure):
    _fields_ = [("groups", C.c_int32), ("rays", C.c_int32 * PR_MAX_DECODER_GROUPS), ("width", C.c_int32 * PR_MAX_DECODER_GROUPS),
                ("channel_begin", C.c_int32 * PR_MAX_DECODER_GROUPS), ("channel_end", C.c_int32 * PR_MAX_DECODER_GROUPS),
                ("map", C.c_void_p * PR_MAX_DECODER_GROUPS)]


class Outputs(C.Structure):
    _fields_ = [
        ("object", Entry * PR_MAX_OBJECTS), ("global_", Entry),
        ("sample_t", C.c_void_p * PR_MAX_OBJECTS), ("sample_sigma", C.c_void_p * PR_MAX_OBJECTS),
        ("sample_slot", C.c_void_p * PR_MAX_OBJECTS), ("evaluated_samples", C.c_void_p), ("normalised_samples", C.c_void_p),
        ("sample_delta", C.c_void_p * PR_MAX_OBJECTS), ("head_samples", C.c_void_p),
        ("decoder", DecoderLayout),
    ]


class Call(C.Structure):
    _fields_ = [
        ("frames", C.c_int32), ("rays", C.c_int32), ("objects", C.c_int32), ("static_objects", C.c_int32),
        ("use_fine", C.c_int32), ("flags", C.c_uint32), ("precision", C.c_int32), ("reserved_", C.c_int32),
        ("ray_origins", C.c_void_p), ("ray_directions", C.c_void_p), ("w2o", C.c_void_p), ("style", C.c_void_p),
        ("deformation", C.c_void_p), ("object_in_scene", C.c_void_p),
        ("linspace_coarse", C.c_void_p * PR_MAX_OBJECTS), ("linspace_fine", C.c_void_p * PR_MAX_OBJECTS),
        ("positions_fine", C.c_int32 * PR_MAX_OBJECTS),
        ("noise_coarse", Noise), ("noise_fine", Noise),
        ("noise_seed", C.c_uint64), ("noise_ray_offset", C.c_int32), ("noise_total_rays", C.c_int32),
        ("noise_seed_device", C.c_void_p),
    ]


class EntryGrads(C.Structure):
    _fields_ = [("integrated_features", C.c_void_p), ("opacity", C.c_void_p), ("depth", C.c_void_p),
                ("integrated_displacements_magnitude", C.c_void_p), ("weights", C.c_void_p),
                ("integrated_divergence", C.c_void_p)]


GRAD_FIELDS = [f[0] for f in EntryGrads._fields_]


class OutputGrads(C.Structure):
    _fields_ = [("object", EntryGrads * PR_MAX_OBJECTS), ("global_", EntryGrads),
                ("sample_t", C.c_void_p * PR_MAX_OBJECTS), ("sample_delta", C.c_void_p * PR_MAX_OBJECTS)]


class LinearGrad(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class ModelGrads(C.Structure):
    _fields_ = [
        ("backbone", LinearGrad * PR_MAX_LAYERS), ("alpha_head", LinearGrad), ("head0", LinearGrad), ("affine1", LinearGrad),
        ("head3", LinearGrad), ("affine4", LinearGrad), ("head6", LinearGrad), ("bender", LinearGrad * PR_MAX_LAYERS),
        ("bender_out", LinearGrad),
    ]


class InputGrads(C.Structure):
    _fields_ = [("w2o", C.c_void_p), ("style", C.c_void_p), ("deformation", C.c_void_p), ("model", ModelGrads * PR_MAX_OBJECTS),
                ("model_fine", ModelGrads * PR_MAX_OBJECTS), ("ray_origins", C.c_void_p), ("ray_directions", C.c_void_p)]


# every exported symbol of include/playrender.h : (restype, argtypes)
class SceneSetup(C.Structure):
    """pr_scene_setup_t (include/playrender.h)."""
    _fields_ = [("frames", C.c_int32), ("cameras", C.c_int32), ("objects", C.c_int32), ("box_points_per_object", C.c_int32),
                ("style_features", C.c_int32), ("deformation_features", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("focal_multiplier", C.c_float), ("upsample_factor", C.c_float), ("axes_with_upsampled_focals", C.c_int32),
                ("camera_rotations", C.c_void_p), ("camera_translations", C.c_void_p), ("focals", C.c_void_p),
                ("object_rotations", C.c_void_p), ("object_translations", C.c_void_p), ("style", C.c_void_p),
                ("deformation", C.c_void_p), ("object_in_scene", C.c_void_p), ("box_points", C.c_void_p), ("axes_points", C.c_void_p),
                ("boxes", C.c_void_p), ("projected_points", C.c_void_p), ("axes", C.c_void_p), ("camera34", C.c_void_p),
                ("render_focals", C.c_void_p), ("w2o34", C.c_void_p), ("style_nks", C.c_void_p), ("deformation_nkd", C.c_void_p),
                ("present", C.c_void_p)]


SYMBOLS = {
    "pr_packed_size": (C.c_int, [C.POINTER(ObjectModel), C.POINTER(C.c_size_t)]),
    "pr_pack_model": (C.c_int, [C.POINTER(ObjectModel), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pr_pack_models": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(ObjectModel)), c_int32_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                 C.c_void_p]),
    "pr_workspace_size": (C.c_int, [C.POINTER(Call), C.POINTER(Object), C.POINTER(C.c_size_t)]),
    "pr_render_forward": (C.c_int, [C.POINTER(Call), C.POINTER(Object), C.POINTER(Outputs), C.POINTER(Outputs),
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "pr_backward_workspace_size": (C.c_int, [C.POINTER(Call), C.POINTER(Object), C.POINTER(C.c_size_t)]),
    "pr_render_backward": (C.c_int, [C.POINTER(Call), C.POINTER(Object), C.POINTER(OutputGrads), C.POINTER(OutputGrads),
                                     C.POINTER(InputGrads), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pr_camera_rays": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_expected_positions": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_noise_fill": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pr_roi_pool_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_roi_pool_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_profile_enable": (C.c_int, [C.c_int]),
    "pr_profile_collect": (C.c_int, [C.POINTER(C.c_double), c_int32_p]),
    "pr_probe_mfma_f32": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "pr_probe_mfma_f16": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "pr_abi_version": (C.c_int, []),
    "pr_pose_matrices": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_pose_matrices_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "pr_project_points": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_scene_setup": (C.c_int, [C.POINTER(SceneSetup), C.c_void_p]),
    "pr_scene_setup_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10),
    "pr_patch_pixels": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pr_graph_node_census": (C.c_int, [C.c_void_p, c_int32_p]),
    "pr_last_error": (C.c_char_p, []),
    "pr_device_info": (C.c_int, [c_int32_p, c_int32_p, C.c_char_p, C.c_size_t]),
}

_LIB: Optional[C.CDLL] = None


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libplayrender.so")


def load() -> C.CDLL:
    """Loads libplayrender.so (once) and sets the prototypes.  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C playableenvironments_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pr_abi_version() != 5:
        raise RuntimeError(f"libplayrender ABI version {lib.pr_abi_version()} != 5 (rebuild: make -C playableenvironments_amd/csrc)")
    _LIB = lib
    return lib


class PlayRenderError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().pr_last_error()
        raise PlayRenderError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
