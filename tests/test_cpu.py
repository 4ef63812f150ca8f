"""CPU suite (python -m pytest tests -m "not gpu"): oracle vs reference-generated golden vectors,
host logic, C-ABI surface, multi-process sharding on gloo.  No compute call needs a GPU."""
import ast
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest
import torch

from oracle import render_oracle as ro
from oracle.make_golden import recipe_config
from playableenvironments_amd import ObjectComposer, _lib, configs, synthetic
from playableenvironments_amd.environment_model import strided_grid_pixels
from playableenvironments_amd.object_composer import ObjectIDsHelper
from tests.helpers import compare_results

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def load_fixture(path):
    z = np.load(path)
    recipe = ast.literal_eval(bytes(z["recipe"]).decode())
    inputs = [torch.from_numpy(z[f"in/{i}"]) for i in range(7)]
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    noise = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("noise/")}
    out = {}
    for k in z.files:
        if k.startswith("out/"):
            node = out
            parts = k[4:].split("/")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = torch.from_numpy(z[k])
    return recipe, inputs, sd, noise, out, bool(int(z["perturb"]))


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 7


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_golden(path):
    """The oracle must reproduce the reference's recorded outputs bit for bit (same torch build) or
    to fp32 round-off (other builds): tolerance rtol 1e-5 / atol 1e-6, NaNs in the same places."""
    recipe, inputs, sd, noise, want, perturb = load_fixture(path)
    cfg = recipe_config(recipe)
    with torch.no_grad():
        got = ro.composer_forward(cfg, sd, *inputs, perturb, noise=noise)
    rep = compare_results(want, got, rtol=1e-5, atol=1e-6)
    bad = {k: v for k, v in rep.items() if not v[1]}
    assert not bad, bad


def test_stable_merge_only_differs_on_tied_rays():
    """The renderer defines cross-object ties as stable in object order; against the reference's
    unspecified order this may change rays that contain an in-box sample inside a tie, nothing else."""
    recipe, inputs, sd, noise, want, perturb = load_fixture(os.path.join(ROOT, "tests", "golden", "tennis_small_eval.npz"))
    cfg = recipe_config(recipe)
    with torch.no_grad():
        got = ro.composer_forward(cfg, sd, *inputs, perturb, stable_merge=True)
    diff = (got["coarse"]["global"]["opacity"] - want["coarse"]["global"]["opacity"]).abs()
    assert (diff > 1e-6).float().mean() < 0.05


def test_known_answer_bounding_box():
    """The reference's only known-answer block (utils/lib_3d/bounding_box.py:134-148): unit box,
    points strictly inside are inside, points beyond a face are outside; faces are inclusive."""
    box = torch.tensor([[0.0, 1.0], [0.0, 1.0], [0.0, 1.0]])
    inside = torch.tensor([[0.5, 0.5, 0.5], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.1, 0.9, 0.5], [1.0, 0.0, 0.3]])
    outside = torch.tensor([[1.1, 0.5, 0.5], [-0.1, 0.5, 0.5], [0.5, 1.5, 0.5], [0.5, 0.5, -1e-6], [2.0, 2.0, 2.0]])
    assert ro._in_box(inside, box).all() and not ro._in_box(outside, box).any()


def test_euler_roundtrip_identity():
    m = ro.euler_to_matrix(torch.zeros(3), torch.tensor([1.0, 2.0, 3.0]))
    assert torch.equal(m[:3, :3], torch.eye(3)) and torch.equal(m[:3, 3], torch.tensor([1.0, 2.0, 3.0]))
    r = ro.euler_to_matrix(torch.tensor([0.3, -1.1, 0.7]), torch.zeros(3))[:3, :3]
    assert torch.allclose(r @ r.T, torch.eye(3), atol=1e-6)


def test_weights_sum_to_opacity_and_stay_in_unit_interval():
    torch.manual_seed(0)
    raw = torch.randn(50, 40) * 3
    t = torch.sort(torch.rand(50, 40) * 20 + 5, dim=-1)[0]
    d = torch.randn(50, 3)
    a, _ = ro.alphas_from_raw(raw, ro.position_distances(t, d), False)
    w = ro.weights_from_alphas(a)
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()


# ------------------------------------------------------------------------------------------------
# host logic of the product package
# ------------------------------------------------------------------------------------------------
def test_state_dict_layout_matches_reference_names():
    comp = ObjectComposer(configs.tennis_config())
    sd = comp.state_dict()
    assert len(sd) == 156 and sum(p.numel() for p in comp.parameters()) == 2867716  # SURVEY.md 3.4 / 2.2 [probed]
    for key, shape in {
        "object_models_coarse.0.nerf_model.backbone_layers.4.weight": (256, 319),
        "object_models_coarse.2.nerf_model.alpha_head.weight": (1, 256),
        "object_models_coarse.2.nerf_model.features_head.1.affine_transform.weight": (512, 64),
        "object_models_coarse.2.nerf_model.features_head.4.ada_in.normalization.running_var": (128,),
        "object_models_coarse.3.nerf_model.features_head.6.weight": (192, 128),
        "object_models_coarse.2.ray_bender.backbone_layers.3.weight": (128, 199),
        "object_models_coarse.2.ray_bender.output_head.weight": (3, 128),
        "object_models_coarse.2.ray_bender.positional_encoder.current_step": (),
    }.items():
        assert tuple(sd[key].shape) == shape, key
    mc = ObjectComposer(configs.minecraft_config()).state_dict()
    assert tuple(mc["object_models_coarse.1.nerf_model.backbone_layers.0.weight"].shape) == (256, 126)
    assert tuple(mc["object_models_coarse.1.nerf_model.backbone_layers.4.weight"].shape) == (256, 382)
    assert "object_models_coarse.1.nerf_model.alpha_head.weight" not in mc


@pytest.mark.parametrize("path", GOLDEN[:3], ids=[os.path.basename(p)[:-4] for p in GOLDEN[:3]])
def test_reference_state_dict_loads_strictly(path):
    recipe, _, sd, _, _, _ = load_fixture(path)
    comp = ObjectComposer(recipe_config(recipe))
    comp.load_state_dict(sd, strict=True)


def test_fine_models_follow_use_fine():
    comp = ObjectComposer(configs.tennis_config())
    assert all(m is None for m in comp.object_models_fine)
    comp = ObjectComposer(configs.tennis_config(hierarchical=(64, 128)))
    assert all(m is not None for m in comp.object_models_fine)
    assert any(k.startswith("object_models_fine.") for k in comp.state_dict())


def test_object_ids_helper_minecraft():
    h = ObjectIDsHelper(configs.minecraft_config())
    assert (h.objects_count, h.static_objects_count, h.dynamic_objects_count) == (4, 2, 2)
    assert [h.model_idx_by_object_idx(k) for k in range(4)] == [0, 1, 2, 2]
    assert h.object_idx_by_dynamic_object_idx(1) == 3
    layout = ro.ObjectLayout(configs.minecraft_config())
    assert layout.model_of_object == [0, 1, 2, 2] and layout.static_objects == 2


def test_annealing_weights_match_oracle():
    comp = ObjectComposer(configs.tennis_config())
    for step in (0, 5000, 20000, 60000, 100000):
        comp.set_step(step)
        enc = comp.object_models_coarse[2].ray_bender.positional_encoder
        want = ro.annealing_weights(torch.tensor(step, dtype=torch.int), 6, 60000)
        assert torch.equal(enc.annealing_weights(), want)


def test_strided_grid_matches_oracle_and_sizes():
    rows, cols = strided_grid_pixels(288, 512, [4, 8])
    r2, c2 = ro.strided_grid_pixels(288, 512, [4, 8])
    assert rows.numel() == 72 * 128 + 36 * 64 == 11520
    assert torch.equal(rows.long(), r2) and torch.equal(cols.long(), c2)
    assert rows[0] == 2 and cols[1] == 6 and rows[72 * 128] == 4
    with pytest.raises(Exception):
        strided_grid_pixels(100, 100, [8])


def test_renderer_refuses_cpu_tensors_and_training():
    comp = ObjectComposer(configs.tennis_single_player_config()).eval()
    o, d, n = torch.zeros(1, 1, 1, 3), torch.zeros(1, 1, 1, 8, 3), torch.zeros(1, 1, 1, 3)
    args = (o, d, n, torch.eye(4).reshape(1, 1, 1, 4, 4, 1), torch.zeros(1, 1, 1, 64, 1), torch.zeros(1, 1, 1, 32, 1),
            torch.ones(1, 1, 1, 1, dtype=torch.bool), False)
    with pytest.raises(RuntimeError):
        comp(*args)  # no CPU fallback


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(built_library):
    header = open(os.path.join(ROOT, "include", "playrender.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(pr_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(built_library, name) is not None
    assert built_library.pr_abi_version() == 5


def test_product_kernels_have_no_flat_memory_operations(built_library):
    """A device pointer that reaches a kernel through a table loses its address space; hipcc then emits flat loads / stores and, since a
    flat access may be an LDS access, waits for EVERY outstanding request in front of the first use of any loaded value - inside a
    software-pipelined K loop that is a full L2 round trip per step (the training forward, both head phases and the
    ray-bender head had them for four rounds).  `as_global()` states the address space where such a pointer is used; this test reads the
    built library's gfx950 code objects and holds every kernel to zero flat accesses."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_check
    if not os.path.exists(isa_check.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    counts = isa_check.memory_operations(_lib.library_path())
    kernels = [k for k in counts if "k_" in k]
    assert len(kernels) >= 60, len(kernels)                       # the disassembly found the library's kernels
    for wanted in ("k_mlp_mfma_group", "k_mlp_mfma_train_group_split", "k_mlp_head_group", "k_chain_bwd_group_f16", "k_head_bwd_group",
                   "k_gemm_tn_all_bf16", "k_gemm_tn_all_f16", "k_mlp_split_group", "k_composite"):
        assert any(wanted in k for k in kernels), wanted
    flat = {k: v for k, v in counts.items() if v["flat_load"] or v["flat_store"]}
    assert not flat, flat


def test_k_loops_never_wait_for_every_outstanding_request(built_library):
    """The software-pipelined K loops request the fragments of the next step(s) in front of the current step's MFMAs; a `s_waitcnt
    vmcnt(0)` inside such a loop waits for the requests just issued and turns the pipeline into one memory round trip per step.  hipcc
    produced exactly that from three harmless-looking things: flat pointers, requests under `if (two)`, and first
    requests in another order than the loop's.  Every K loop of the shipped product kernels is held to partial waits."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_check
    if not os.path.exists(isa_check.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    loops = isa_check.matrix_loops(_lib.library_path())
    held = ("k_mlp_mfma_group", "k_mlp_mfma_train_group", "k_mlp_mfma_train_group_split", "k_mlp_head_group", "k_mlp_split_group",
            "k_mlp_f16_group", "k_chain_bwd_group", "k_chain_bwd_group_f16", "k_head_bwd_group", "k_div_chain_group")
    for wanted in held:
        # (mangled: _ZN2pr<len><name>E...)
        names = [k for k in loops if f"{len(wanted)}{wanted}E" in k]
        assert names, wanted
        for k in names:
            assert loops[k], k
            for loop in loops[k]:
                assert not any("vmcnt(0)" in w for w in loop["waits"]), (k, loop)


def test_plain_c_client_links_and_calls_the_abi(built_library, tmp_path):
    """The drop-in boundary is a C ABI: ``include/playrender.h`` has to compile as plain C99 (what a cgo / JNI / N-API stub includes, no
    C++ or torch type in any signature) and a C program linked against ``libplayrender.so`` has to reach the host-side entry points -
    the version, a size query on a zeroed model (refused with a message), the host-only graph census on a NULL graph (refused).  No
    device call: runs without a GPU."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    source = tmp_path / "client.c"
    source.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "playrender.h"
int main(void) {
    pr_object_model_t model;
    size_t bytes = 0;
    int32_t counts[4] = {7, 7, 7, 7};
    memset(&model, 0, sizeof model);
    if (pr_abi_version() != PR_ABI_VERSION) return 1;
    if (pr_packed_size(&model, &bytes) == 0) return 2;            /* a zeroed description is not a model */
    if (pr_last_error() == NULL || strlen(pr_last_error()) == 0) return 3;
    if (pr_graph_node_census(NULL, counts) == 0) return 4;
    if (pr_profile_enable(0) != 0) return 5;
    printf("abi %d, sizeof(pr_call_t) %zu, sizeof(pr_object_model_t) %zu, refusal: %s\n", pr_abi_version(), sizeof(pr_call_t),
           sizeof(pr_object_model_t), pr_last_error());
    return 0;
}
""")
    lib_dir = os.path.dirname(_lib.library_path())
    binary = tmp_path / "client"
    build = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(source),
                            "-L", lib_dir, "-lplayrender", f"-Wl,-rpath,{lib_dir}", "-o", str(binary)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(binary)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr[-2000:])
    assert f"sizeof(pr_call_t) {C.sizeof(_lib.Call)}," in run.stdout and f"sizeof(pr_object_model_t) {C.sizeof(_lib.ObjectModel)}," in run.stdout


def test_struct_sizes_match_header_layout():
    """ctypes mirrors must have the C sizes (pointer = 8, int32 = 4, natural alignment)."""
    assert C.sizeof(_lib.Linear) == 24
    assert C.sizeof(_lib.Entry) == 56
    assert C.sizeof(_lib.ObjectModel) == 14 * 4 + 16 * 4 + 6 * 4 + 4 * 4 + 24 * 12 + 24 + 24 + (24 + 24) + 24 + (24 + 24) + 24 + 24 * 12 + 24


def _model_struct_host(cfg_model, positions):
    """ObjectModel with non-NULL dummy pointers: enough for the host-only size computations."""
    comp = ObjectComposer({"data": {"focal_length_multiplier": 1.0},
                           "model": {"apply_activation": False, "fix_object_overlaps": False, "static_object_models": 0,
                                     "object_parameters_encoder": [{"objects_count": 1}], "object_encoders": [{}],
                                     "object_models": [cfg_model]}})
    return comp, comp._model_struct(comp.object_models_coarse[0], positions)


def test_packed_and_workspace_sizes_on_host(built_library):
    cfg = configs.tennis_config()
    comp, s = _model_struct_host(cfg["model"]["object_models"][2], 32)
    size = C.c_size_t()
    assert built_library.pr_packed_size(C.byref(s), C.byref(size)) == 0
    # fragment-ordered copy = the padded weights for the forward kernels, plus every matrix once more as W^T fragments for
    # the backward chains (biases and the small heads are not repeated), plus - split-precision training, round 4 - the W^T
    # fragments and the forward segments of phase 1 once more as bf16 triples (1.5 x the fp32 fragments each)
    raw = sum(p.numel() for n, p in comp.named_parameters() if "affine_transform" not in n) * 4
    assert 4.2 * raw <= size.value <= 5.4 * raw, size.value / raw
    call = _lib.Call()
    call.frames, call.rays, call.objects = 1, 1000, 1
    for f in ("ray_origins", "ray_directions", "w2o", "style", "deformation", "object_in_scene"):
        setattr(call, f, 256)
    call.linspace_coarse[0] = 256
    objs = (_lib.Object * 1)()
    objs[0].coarse = s
    objs[0].packed_coarse = 256
    ws = C.c_size_t()
    assert built_library.pr_workspace_size(C.byref(call), objs, C.byref(ws)) == 0
    per_sample = ws.value / (1000 * 32)
    assert 192 * 4 <= per_sample <= 192 * 4 + 64  # feature row + dense per-sample state
    call.rays = 0
    assert built_library.pr_workspace_size(C.byref(call), objs, C.byref(ws)) != 0
    assert b"empty call" in built_library.pr_last_error()


def test_abi_error_paths_without_a_device(built_library):
    """Status codes and messages of the host-side validation (no device work happens before it)."""
    lib = built_library
    cfg = configs.tennis_config()
    _, s = _model_struct_host(cfg["model"]["object_models"][2], 32)
    call = _lib.Call()
    call.frames, call.rays, call.objects = 1, 64, 1
    for f in ("ray_origins", "ray_directions", "w2o", "style", "deformation", "object_in_scene"):
        setattr(call, f, 256)
    call.linspace_coarse[0] = 256
    objs = (_lib.Object * 1)()
    objs[0].coarse = s
    objs[0].packed_coarse = 256
    size = C.c_size_t()
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == 0
    out = _lib.Outputs()
    # workspace too small / misaligned: refused before anything is enqueued
    assert lib.pr_render_forward(C.byref(call), objs, C.byref(out), None, 256, size.value - 1, None) == -2
    assert b"workspace too small" in lib.pr_last_error()
    assert lib.pr_render_forward(C.byref(call), objs, C.byref(out), None, 257, size.value, None) == -1
    assert b"aligned" in lib.pr_last_error()
    assert lib.pr_render_forward(None, objs, C.byref(out), None, 256, size.value, None) == -1
    # the backward pass needs a forward call that saved its intermediates
    bsize = C.c_size_t()
    assert lib.pr_backward_workspace_size(C.byref(call), objs, C.byref(bsize)) == -1
    assert b"PR_FLAG_SAVE_FOR_BACKWARD" in lib.pr_last_error()
    eval_size = size.value
    call.flags = _lib.PR_FLAG_SAVE_FOR_BACKWARD                                        # differentiable eval-mode call
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == 0 and size.value > 4 * eval_size
    assert lib.pr_backward_workspace_size(C.byref(call), objs, C.byref(bsize)) == 0 and bsize.value > 0
    call.flags = _lib.PR_FLAG_SAVE_FOR_BACKWARD | _lib.PR_FLAG_TRAIN_BN
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == 0 and size.value > 4 * eval_size
    assert lib.pr_backward_workspace_size(C.byref(call), objs, C.byref(bsize)) == 0 and bsize.value > 0
    call.precision = _lib.PR_PRECISION_F16X3
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == -1          # split kernel: no saved activations
    assert b"PR_PRECISION_FP32" in lib.pr_last_error()
    call.precision = _lib.PR_PRECISION_F16                                            # ... nor the single-product fp16 tier
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == -1
    assert b"PR_PRECISION_FP32" in lib.pr_last_error()
    call.flags = 0
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == 0 and size.value == eval_size
    call.precision, call.flags = 7, 0
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == -1
    assert b"precision" in lib.pr_last_error()
    call.precision = 0
    call.objects = 9
    assert lib.pr_workspace_size(C.byref(call), objs, C.byref(size)) == -1
    # overlap fix with a static object that has more positions than a dynamic one: the reference raises IndexError
    call.objects, call.static_objects, call.flags = 2, 1, _lib.PR_FLAG_FIX_OVERLAPS
    two = (_lib.Object * 2)()
    _, s_static = _model_struct_host(cfg["model"]["object_models"][2], 48)
    two[0].coarse, two[1].coarse = s_static, s
    two[0].packed_coarse = two[1].packed_coarse = 256
    call.linspace_coarse[1] = 256
    assert lib.pr_workspace_size(C.byref(call), two, C.byref(size)) == -1
    assert b"IndexError" in lib.pr_last_error()
    # expected positions / camera rays argument checks
    assert lib.pr_expected_positions(1, 8, 1, 1, 4, 256, 256, 256, 256, 256, None, 256, None) == -1
    assert lib.pr_camera_rays(0, 8, 4, 4, 0, 256, 256, 256, 256, 256, 256, 256, None) == -1


def test_unsupported_configuration_is_rejected_loudly(built_library):
    """The C ABI refuses a model it has no kernel for (a struct edited behind the host layer's back) ..."""
    cfg = configs.tennis_config()
    _, s = _model_struct_host(cfg["model"]["object_models"][0], 4)
    s.layers_width = 512
    size = C.c_size_t()
    assert built_library.pr_packed_size(C.byref(s), C.byref(size)) != 0
    assert b"layers_width" in built_library.pr_last_error()


def _edited(cfg, path, value):
    import copy
    out = copy.deepcopy(cfg)
    cur = out
    for key in path[:-1]:
        cur = cur[key]
    cur[path[-1]] = value
    return out


@pytest.mark.parametrize("path, value, needle", [
    (("model", "object_models", 2, "nerf_model", "layers_width"), 512, "['nerf_model']['layers_width'] = 512"),
    (("model", "object_models", 2, "nerf_model", "layers_width"), 257, "['nerf_model']['layers_width'] = 257"),
    (("model", "object_models", 0, "nerf_model", "backbone_layers_count"), 13, "['backbone_layers_count'] = 13"),
    (("model", "object_models", 0, "nerf_model", "position_encoder", "octaves"), 17, "['position_encoder']['octaves'] = 17"),
    (("model", "object_models", 1, "nerf_model", "position_encoder", "append_original"), False, "['append_original'] = False"),
    (("model", "object_models", 2, "ray_bender_model", "layers_width"), 288, "['ray_bender_model']['layers_width'] = 288"),
    (("model", "object_models", 2, "ray_bender_model", "layers_count"), 13, "['ray_bender_model']['layers_count'] = 13"),
    (("model", "object_models", 2, "ray_bender_model", "position_encoder", "octaves"), 20, "['ray_bender_model']['position_encoder']['octaves'] = 20"),
    (("model", "object_models", 2, "ray_bender_model", "position_encoder", "append_original"), False, "['ray_bender_model']['position_encoder']"),
    (("model", "object_models", 3, "nerf_model", "output_features"), 64, "['output_features'] differ"),
    (("model", "object_parameters_encoder", 2, "objects_count"), 7, "add up to 10 object instances"),
])
def test_kernel_limits_are_checked_at_construction(path, value, needle):
    """... and the host layer says so at CONSTRUCTION, with the configuration key in the message - not at the first render
    (model/nerf_models/adain_style_nerf_model.py:24-45 and model/positional_encoder.py:41-65 accept any width / depth / octave
    count; these are the limits of the kernels behind include/playrender.h, listed in INTEGRATION.md)."""
    cfg = configs.tennis_config()
    with pytest.raises(ValueError) as err:
        ObjectComposer(_edited(cfg, path, value))
    assert needle in str(err.value), str(err.value)
    with pytest.raises(ValueError):
        from playableenvironments_amd.environment_model import EnvironmentModel
        EnvironmentModel(_edited(cfg, path, value))


def test_construction_limits_agree_with_the_library(built_library):
    """The host-side limits cannot drift from the library's: for a sweep of widths / depths / octave counts the constructor accepts
    exactly the models ``pr_packed_size`` (csrc/mlp.hip compute_dims) accepts."""
    from playableenvironments_amd.object_composer import validate_config_limits
    base = configs.tennis_config()
    _, s0 = _model_struct_host(base["model"]["object_models"][2], 8)
    cases = [("layers_width", w) for w in (1, 2, 31, 200, 224, 225, 256, 257, 288)] + \
            [("backbone_layers_count", c) for c in (1, 2, 12, 13)] + [("octaves", o) for o in (0, 10, 16, 17)] + \
            [("output_features", f) for f in (1, 3, 256, 257)] + [("bender_width", w) for w in (1, 64, 256, 257)] + \
            [("bender_count", c) for c in (1, 2, 12, 13)] + [("bender_octaves", o) for o in (0, 6, 16, 17)]
    host_keys = {"layers_width": ("nerf_model", "layers_width"), "backbone_layers_count": ("nerf_model", "backbone_layers_count"),
                 "octaves": ("nerf_model", "position_encoder", "octaves"), "output_features": ("nerf_model", "output_features"),
                 "bender_width": ("ray_bender_model", "layers_width"), "bender_count": ("ray_bender_model", "layers_count"),
                 "bender_octaves": ("ray_bender_model", "position_encoder", "octaves")}
    for field, value in cases:
        cfg = _edited(base, ("model", "object_models", 2) + host_keys[field], value)
        if field == "output_features":      # (keep the objects' feature counts equal: that is a composer rule, not a kernel limit)
            for i in range(4):
                cfg = _edited(cfg, ("model", "object_models", i, "nerf_model", "output_features"), value)
        if field in ("backbone_layers_count", "bender_count"):      # (keep the skip index inside the stack)
            cfg = _edited(cfg, ("model", "object_models", 2) + host_keys[field][:1] + ("skip_layer_idx",), 1)
        try:
            validate_config_limits(cfg)
            host_ok = True
        except ValueError:
            host_ok = False
        s = type(s0).from_buffer_copy(s0)        # (ctypes structure: a byte copy)
        struct_field = {"backbone_layers_count": "backbone_count"}.get(field, field)
        assert struct_field in dict(type(s)._fields_)
        setattr(s, struct_field, value)
        if field == "backbone_layers_count":
            s.skip_layer_idx = 1
        if field == "bender_count":
            s.bender_skip = 1
        size = C.c_size_t()
        lib_ok = built_library.pr_packed_size(C.byref(s), C.byref(size)) == 0
        assert host_ok == lib_ok, (field, value, host_ok, built_library.pr_last_error())


# ------------------------------------------------------------------------------------------------
# multi-process sharding (gloo, world_size 2)
# ------------------------------------------------------------------------------------------------
def _gloo_worker(rank, world, port, total, results):
    import torch.distributed as dist
    from playableenvironments_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(total * 3, dtype=torch.float32).reshape(1, total, 3)
        b, e = parallel.shard_range(total, rank, world)
        local = full[:, b:e] * 1.0
        out = parallel.gather_ray_shards(local, total, dim=1, dst=0)
        everyone = parallel.gather_ray_shards(local, total, dim=1, dst=None)
        ok = torch.equal(everyone, full) and ((rank == 0 and torch.equal(out, full)) or (rank != 0 and out is None))
        frames = parallel.shard_frames(torch.arange(5), rank, world)
        ok = ok and frames.tolist() == ([0, 1, 2] if rank == 0 else [3, 4])
        # interleaved 8 x 8 tiles of a 20 x 27 frame + its stride-4 grid (ragged edge tiles, unequal shard sizes)
        lists = parallel.tile_shard_lists([(20, 27), (5, 6)], world)
        count = 20 * 27 + 5 * 6
        image = torch.arange(count * 2, dtype=torch.float32).reshape(1, count, 2)
        mine = image.index_select(1, lists[rank])
        everyone = parallel.gather_indexed_shards(mine, lists, dim=1, dst=None)
        first = parallel.gather_indexed_shards(mine, lists, dim=1, dst=0)
        ok = ok and torch.equal(everyone, image) and ((rank == 0 and torch.equal(first, image)) or (rank != 0 and first is None))
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 11])
def test_ray_shard_gather_world2_gloo(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 200) + total
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, total, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert results[0] and results[1]


class _FakeComposer(torch.nn.Module):
    """A cheap deterministic per-ray function behind EnvironmentModel's host logic, for the CPU tests of render_sharded
    (the HIP composer has no CPU path): every output row depends on its own ray and frame only, like the renderer's."""

    def __init__(self, real):
        super().__init__()
        self.real = real

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.real, name)

    def forward(self, ray_origins, ray_directions, focal_normals, w2o, style, deformation, object_in_scene, perturb,
                video_indexes=None, canonical_pose=False):
        per_frame = (style.mean(dim=(-1, -2)) + w2o[..., 0, 3, :].sum(-1)).unsqueeze(-1)      # (..., 1)
        feats = torch.cat([ray_directions, ray_origins.unsqueeze(-2).expand_as(ray_directions)], -1) * per_frame.unsqueeze(-1)
        entry = {"integrated_features": feats, "opacity": ray_directions.sum(-1) + per_frame, "depth": ray_directions[..., 0] * 2}
        return {"coarse": {"global": entry}, "pytorch_hook": torch.zeros((1,) * 9)}


def _cpu_camera_rays(c2w, focals, height, width, rows, cols):
    lead = list(c2w.shape[:-2])
    dirs, origins, normals = ro.create_camera_rays(lead, height, width, focals)
    idx = rows.to(torch.int64) * width + cols.to(torch.int64)
    idx = idx.expand(lead + [idx.size(-1)]) if idx.dim() == 1 else idx
    picked = torch.gather(dirs.reshape(lead + [height * width, 3]), -2, idx.unsqueeze(-1).expand(list(idx.shape) + [3]))
    return ro.transform_rays(origins, picked, normals, c2w)


def _sharded_render_worker(rank, world, port, results):
    import torch.distributed as dist
    from playableenvironments_amd import environment_model as em
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        em.camera_rays = _cpu_camera_rays
        cfg = configs.tennis_config()
        model = em.EnvironmentModel(cfg)
        model.object_composer = _FakeComposer(model.object_composer)
        model.eval()
        ok = True
        for batch, shard, stride in ((3, "auto", 0), (1, "auto", [4, 8]), (3, "rays", 0), (2, "frames", [4, 8]), (1, "rays", [4, 8]),
                                     (2, "tiles", 0), (1, "tiles", [4, 8])):
            scene = synthetic.tennis_scene(batch=batch, observations=2, seed=21, image_size=(16, 24))
            args = [scene[k] for k in ("camera_rotations", "camera_translations", "focals")] + [scene["image_size"]] + \
                   [scene[k] for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                                       "object_deformation", "object_in_scene")]
            with torch.no_grad():
                whole = model(*args, 0, False, patch_stride=stride, mode="scene_encodings")
            everyone = model.render_sharded(*args, False, patch_stride=stride, shard=shard)
            first = model.render_sharded(*args, False, patch_stride=stride, shard=shard, dst=0, fields=("opacity",))
            for field in ("integrated_features", "opacity", "depth"):
                ok = ok and torch.equal(everyone["coarse"]["global"][field], whole["coarse"]["global"][field])
            if rank == 0:
                ok = ok and torch.equal(first["coarse"]["global"]["opacity"], whole["coarse"]["global"]["opacity"])
                ok = ok and set(first["coarse"]["global"]) == {"opacity"}
            else:
                ok = ok and first is None
        try:
            model.render_sharded(*args, True)
            ok = False
        except ValueError:
            pass
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_render_sharded_world2_gloo():
    """EnvironmentModel.render_sharded on two gloo ranks: frame shards (ragged: 3 frames over 2 ranks), ray shards of a
    single frame (strided grids, odd split), forced modes; the assembled maps equal the unsharded render exactly, on
    every rank (dst=None) or on rank 0 only."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_sharded_render_worker, args=(r, 2, port, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert results[0] and results[1]


def _gloo_grad_worker(rank, world, port, results):
    import torch.distributed as dist
    from playableenvironments_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3, bias=False))
        grads = []
        for i, p in enumerate(net.parameters()):
            g = torch.full_like(p, float(rank + 1) * (i + 1))
            grads.append(g)
            if not (rank == 1 and i == 0):          # one rank has no gradient for one parameter
                p.grad = g.clone()
        calls = parallel.allreduce_gradients(net.parameters(), bucket_bytes=64)   # tiny buckets: several collectives
        ok = calls > 1
        for i, p in enumerate(net.parameters()):
            want = (1.0 * (i + 1) + (0.0 if i == 0 else 2.0 * (i + 1))) / 2.0
            ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
        # gradients that are consecutive views of one flat buffer (what ObjectComposer's backward hands out) are reduced
        # in place, without flattening copies
        flat = torch.empty(sum(p.numel() for p in net.parameters()))
        offset = 0
        for i, p in enumerate(net.parameters()):
            p.grad = flat[offset:offset + p.numel()].view(p.shape)
            p.grad.fill_(float(rank + 1) * (i + 1))
            offset += p.numel()
        before = flat.data_ptr()
        calls = parallel.allreduce_gradients(net.parameters(), bucket_bytes=64)
        ok = ok and calls == -(-flat.numel() // 16) and parallel._shared_gradient_buffer(list(net.parameters())) is not None
        for i, p in enumerate(net.parameters()):
            ok = ok and torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) and \
                p.grad.untyped_storage().data_ptr() == before
        net[1].running_mean.fill_(float(rank + 3))
        parallel.broadcast_buffers(net, src=0)
        ok = ok and float(net[1].running_mean[0]) == 3.0
        # pipelined gather of per-rank feature maps (bench.py, evaluation): order and contents
        pipe = parallel.AsyncFeatureGather(depth=1)
        for i in range(3):
            pipe.submit(torch.full((2, 5), float(10 * rank + i)))
        stacks = pipe.drain()
        ok = ok and len(stacks) == 3
        for i, st in enumerate(stacks):
            ok = ok and tuple(st.shape) == (2 * world, 5) and st[::2, 0].tolist() == [float(i), float(10 + i)]
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    """Data-parallel training exchange (C5): bucketed in-place gradient average + buffer broadcast."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_gloo_grad_worker, args=(r, 2, port, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert results[0] and results[1]


class _HookedRender(torch.autograd.Function):
    """What ObjectComposer's autograd node does with its parameter gradients: views of ONE flat buffer, handed to the composer's
    ``gradient_hooks`` before they are returned to autograd.  out = sum_i (i + 1) * sum(x) * sum(p_i)."""

    @staticmethod
    def forward(ctx, composer, x, *params):
        ctx.composer, ctx.params, ctx.scale = composer, params, float(x.sum())
        return sum((i + 1) * ctx.scale * p.sum() for i, p in enumerate(params))

    @staticmethod
    def backward(ctx, grad):
        flat = torch.zeros(sum(p.numel() for p in ctx.params))
        views, offset = [], 0
        for i, p in enumerate(ctx.params):
            v = flat[offset:offset + p.numel()].view(p.shape)
            v.fill_((i + 1) * ctx.scale * float(grad))
            views.append(v)
            offset += p.numel()
        for hook in ctx.composer.gradient_hooks:
            hook(flat)
        return (None, None) + tuple(views)


class _HookedComposer:
    def __init__(self):
        torch.manual_seed(0)
        self.params = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
        self.gradient_hooks = []

    def _parameter_list(self):
        return self.params

    def __call__(self, x):
        return _HookedRender.apply(self, x, *self.params)


def _overlap_worker(rank, world, port, results):
    import torch.distributed as dist
    from playableenvironments_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comp = _HookedComposer()
        overlap = parallel.OverlappedGradientAllReduce(comp)
        xs = [torch.full((2,), float(rank + 1 + j)) for j in range(3)]       # rank-dependent inputs: rank-dependent gradients

        def mean_scale(js):            # gradient of p_i = (i + 1) * sum_j sum(x_j), averaged over the ranks
            return sum(sum(2.0 * (r + 1 + j) for j in js) for r in range(world)) / world

        def check(total, what):
            for i, p in enumerate(comp.params):
                assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * total)), (what, i, p.grad.flatten()[:2], (i + 1) * total)

        # (1) one call into empty gradients: reduced in flight, in place
        comp(xs[0]).backward()
        assert overlap.finish() == 1 and overlap.launched == 1 and overlap.deferred == 0
        check(mean_scale([0]), "one call")
        assert parallel._shared_gradient_buffer(comp.params) is not None
        # (2) gradient accumulation: a second backward() into the gradients of (1)
        comp(xs[1]).backward()
        assert overlap.finish() == 1 and overlap.launched == 1 and overlap.deferred == 1
        check(mean_scale([0]) + mean_scale([1]), "accumulation")
        # (3) several composer calls in ONE graph (train-mode ray chunks): the first is reduced in flight, the others corrected
        for p in comp.params:
            p.grad = None
        (comp(xs[0]) + comp(xs[1]) + comp(xs[2])).backward()
        assert overlap.finish() == 3 and overlap.launched == 2 and overlap.deferred == 3
        check(mean_scale([0, 1, 2]), "three calls in one graph")
        # (4) a tensor hook on a parameter: autograd does not adopt the buffer's view - nothing is reduced in flight
        for p in comp.params:
            p.grad = None
        handle = comp.params[1].register_hook(lambda g: g * 1.0)
        comp(xs[2]).backward()
        assert overlap.finish() == 1 and overlap.launched == 2
        check(mean_scale([2]), "tensor hook")
        handle.remove()
        # (5) sums instead of averages; nothing pending: finish() is a no-op
        for p in comp.params:
            p.grad = None
        overlap.remove()
        overlap = parallel.OverlappedGradientAllReduce(comp, average=False)
        (comp(xs[0]) + comp(xs[1])).backward()
        assert overlap.finish() == 2 and overlap.finish() == 0
        check(mean_scale([0, 1]) * world, "sum")
        # (6) gradients replaced behind the collective's back: refused, not silently wrong
        for p in comp.params:
            p.grad = None
        comp(xs[0]).backward()
        comp.params[0].grad = comp.params[0].grad.clone()
        try:
            overlap.finish()
            raise AssertionError("finish() accepted gradients that are not the reduced buffer")
        except RuntimeError as e:
            assert "not the buffer" in str(e)
        results[rank] = True
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_allreduce_guards_world2_gloo():
    """parallel.OverlappedGradientAllReduce (the all-reduce started inside backward()) on two gloo ranks, driven by an autograd node
    that hands out its gradients the way ObjectComposer's does: the overlapped single-call case, gradient accumulation, several
    composer calls in one graph (train-mode ray chunks), a parameter with a tensor hook, sums - always the gradients
    ``allreduce_gradients`` after ``backward()`` would leave."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = 29650 + (os.getpid() % 150)
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert results.get(0) and results.get(1)


def test_tile_shard_lists_partition_the_pixel_list():
    """8 x 8 tiles dealt round robin: every ray belongs to exactly one rank, a tile is never split, consecutive tiles go to
    consecutive ranks (also across the grids of a strided render), and one rank is the identity."""
    from playableenvironments_amd import parallel
    for grids, world in (([(16, 24)], 2), ([(20, 27), (5, 6)], 3), ([(64, 64)], 8), ([(7, 5)], 4)):
        lists = parallel.tile_shard_lists(grids, world)
        total = sum(h * w for h, w in grids)
        assert sorted(torch.cat(lists).tolist()) == list(range(total))
        offset, first = 0, 0
        for h, w in grids:
            per_row = (w + 7) // 8
            for k, l in enumerate(lists):
                inside = l[(l >= offset) & (l < offset + h * w)] - offset
                tiles = (inside // w // 8) * per_row + (inside % w) // 8
                assert bool(((tiles + first) % world == k).all())
                assert tiles.tolist() == sorted(tiles.tolist())            # tile-major order inside a rank
            offset += h * w
            first += ((h + 7) // 8) * per_row
    assert torch.equal(parallel.tile_shard_lists([(9, 9)], 1)[0].sort()[0], torch.arange(81))
    sizes = [l.numel() for l in parallel.tile_shard_lists([(256, 256)], 8)]
    assert sizes == [8192] * 8


def test_flattened_parameters_train_like_separate_ones():
    """parallel.flatten_parameters: names, values and state_dict survive, the views follow the arena (and its version counter),
    and Adam on the arena produces bit for bit what Adam produces on the separate tensors."""
    from playableenvironments_amd import parallel

    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3), torch.nn.BatchNorm1d(3))
    a, b = make(), make()
    b[0].bias.requires_grad_(False)
    a[0].bias.requires_grad_(False)
    before = {k: v.clone() for k, v in b.state_dict().items()}
    arena = parallel.flatten_parameters(b)
    assert arena.numel() == sum(p.numel() for p in b.parameters() if p.requires_grad)
    assert list(b.state_dict()) == list(before) and all(torch.equal(v, before[k]) for k, v in b.state_dict().items())
    assert [n for n, _ in a.named_parameters()] == [n for n, _ in b.named_parameters()]
    opt_a = torch.optim.Adam([p for p in a.parameters() if p.requires_grad], lr=1e-2)
    opt_b = torch.optim.Adam([arena], lr=1e-2)
    x = torch.randn(16, 5)
    for step in range(3):
        for model, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad(set_to_none=True)       # (the documented recipe: nothing else clears the views' gradients)
            model(x).square().mean().backward()
            if model is b:
                version = b[2].weight._version
                parallel.flat_gradient(arena, b)
                assert arena.grad is not None and all(p.grad is None for p in b.parameters() if p.requires_grad)
            opt.step()
        assert b[2].weight._version > version          # the views see the arena's in-place update
        for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p, q), (step, n)
    b.load_state_dict(before)                          # checkpoints load into the views
    assert torch.equal(arena.detach()[:35].view(7, 5), before["0.weight"])


def test_direct_dictionary_edits_move_the_registration_epoch():
    """modules.Tracked sees ``__setattr__`` / ``register_*`` / ``add_module``; the edits that write a module's ``_parameters`` /
    ``_modules`` dictionaries directly have to report themselves: ``parallel.flatten_parameters`` (of ANY module - flattening the
    encoders alone must reach the composer of the parent model, whose cached parameter lists feed the evaluation recordings'
    signature) and ``ModuleList`` / ``Sequential`` ``insert`` / ``pop`` / ``del`` / item assignment."""
    from playableenvironments_amd import modules, parallel
    from playableenvironments_amd import environment_model as em
    cfg = configs.reduced_config(configs.minecraft_config(encoders=True), width=32, layers=3, skip=1, features=16, octaves=2,
                                 bender_width=16, bender_layers=2, bender_skip=1, bender_octaves=2)
    model = em.EnvironmentModel(cfg)
    composer = model.object_composer
    cached = composer._parameter_list(model)                  # the whole model's parameters, as the recordings' signature reads them
    assert composer._parameter_list(model) is cached
    epoch = modules.REGISTRATION_EPOCH[0]
    parallel.flatten_parameters(model.object_encoders)        # a module that does not contain the composer
    assert modules.REGISTRATION_EPOCH[0] > epoch
    fresh = composer._parameter_list(model)
    assert fresh is not cached and all(a is b for a, b in zip(fresh, model.parameters())) and len(fresh) == len(list(model.parameters()))
    for make in (lambda: modules.ModuleList([modules.Linear(2, 2) for _ in range(3)]),
                 lambda: modules.Sequential(*[modules.Linear(2, 2) for _ in range(3)])):
        for edit in (lambda c: c.insert(1, modules.Linear(2, 2)), lambda c: c.pop(0), lambda c: c.__delitem__(1),
                     lambda c: c.__setitem__(0, modules.Linear(2, 2))):
            container = make()
            epoch = modules.REGISTRATION_EPOCH[0]
            edit(container)
            assert modules.REGISTRATION_EPOCH[0] > epoch, (type(container).__name__, edit)


def test_weights_epoch_follows_the_owning_optimisers_step():
    """``ObjectComposer.weights_epoch`` (part of the packed-weight and recorded-frame keys: torch's fused optimisers do not move version
    counters) moves once more AFTER the step of the optimiser that holds the composer's parameters - a render between ``backward()``
    and ``step()`` must not leave pre-step packed weights under the post-step key.  The hook is torch's optimiser post-step hook,
    installed by the first backward pass that produced parameter gradients; steps of optimisers that own other tensors, and steps
    while no backward pass is pending, leave the epoch alone; views of a ``flatten_parameters`` arena count as the arena's storage."""
    from playableenvironments_amd import object_composer as oc, parallel
    cfg = configs.reduced_config(configs.tennis_config(), width=32, layers=3, skip=1, features=16, octaves=2,
                                 bender_width=16, bender_layers=2, bender_skip=1, bender_octaves=2)
    composer = ObjectComposer(cfg)
    other = torch.nn.Linear(3, 3)
    mine = torch.optim.SGD(composer.parameters(), lr=0.0)
    theirs = torch.optim.SGD(other.parameters(), lr=0.0)
    for p in list(composer.parameters()) + list(other.parameters()):
        p.grad = torch.zeros_like(p)
    epoch = composer.weights_epoch
    mine.step()
    assert composer.weights_epoch == epoch                  # nothing pending: optimiser steps are not watched
    oc._watch_optimizer_steps(composer)                      # (what _RenderFunction.backward does after it produced parameter gradients)
    theirs.step()
    assert composer.weights_epoch == epoch and composer in oc._AWAITING_STEP
    mine.step()
    assert composer.weights_epoch == epoch + 1 and composer not in oc._AWAITING_STEP
    mine.step()
    assert composer.weights_epoch == epoch + 1
    arena = parallel.flatten_parameters(composer)
    arena.grad = torch.zeros_like(arena)
    flat = torch.optim.SGD([arena], lr=0.0)             # (the arena is what the optimiser holds; the composer's parameters are views of it)
    oc._watch_optimizer_steps(composer)
    theirs.step()
    assert composer.weights_epoch == epoch + 1
    flat.step()
    assert composer.weights_epoch == epoch + 2


def test_shard_range_partitions():
    from playableenvironments_amd.parallel import shard_range
    for total in (0, 1, 7, 8, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


# ------------------------------------------------------------------------------------------------
# pixel samplers (a2): the product's vectorised index logic against the oracle's loop restatement
# ------------------------------------------------------------------------------------------------
def _random_boxes(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.rand(n, 4, k, generator=g) * 0.5
    boxes[:, 2:] = boxes[:, :2] + 0.1 + torch.rand(n, 2, k, generator=g) * 0.4
    return boxes.clamp(0, 1)


@pytest.mark.parametrize("patch,strides", [(16, [4, 8]), (8, [2, 4]), (12, [4]), (64, [4, 8])])
def test_strided_patch_sampler_matches_oracle(patch, strides):
    from playableenvironments_amd import ray_sampling as rs
    h, w = (288, 512) if patch == 64 else (96, 160)
    boxes = _random_boxes(4, 4, patch)
    weights = [0.55, 0.15, 0.15, 0.15]
    torch.manual_seed(5)
    want = ro.strided_patch_pixels(boxes, weights, h, w, patch, strides)
    torch.manual_seed(5)
    got = rs.strided_patch_pixels(boxes, weights, h, w, patch, strides)
    assert torch.equal(want, got)
    sm = strides[-1]
    rows, cols = got // w, got % w
    assert int(rows.min()) >= 0 and int(rows.max()) < h and int(cols.min()) >= 0 and int(cols.max()) < w
    # the coarsest grid is aligned to the centres of the (s_M x s_M) pixel cells
    last = (patch * strides[0] // sm) ** 2
    assert ((rows[:, -last:] % sm) == sm // 2).all() and ((cols[:, -last:] % sm) == sm // 2).all()
    assert got.shape[1] == sum((patch * strides[0] // s) ** 2 for s in strides)


def test_weighted_sampler_matches_oracle_and_prefers_boxes():
    from playableenvironments_amd import ray_sampling as rs
    boxes = _random_boxes(3, 4, 11)
    boxes[0, :, 2] = torch.tensor([0.3, 0.3, 0.3, 0.3])  # an empty box must be skipped (zero-area guard)
    weights = [0.0, 0.7, 0.15, 0.15]
    torch.manual_seed(7)
    want = ro.sample_pixels_weighted(boxes, weights, 64, 80, 500)
    torch.manual_seed(7)
    got = rs.sample_pixels_weighted(boxes, weights, 64, 80, 500)
    assert torch.equal(want, got)
    r, c = (got[1] // 80).float() / 64, (got[1] % 80).float() / 80
    inside = ((c >= boxes[1, 0, 1:].min() - 0.02) & (r >= boxes[1, 1, 1:].min() - 0.02)).float().mean()
    assert inside > 0.99


# --------------------------------------------------------------------------------------------
# Train-mode gradient fixtures (reference autograd), tests/golden/grads
# --------------------------------------------------------------------------------------------
GRAD_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "grads", "*.npz")))


def load_gradient_fixture(path):
    z = np.load(path)
    recipe, inputs, sd, noise, out, perturb = load_fixture(path)
    probes = {tuple(k[6:].split("/")): torch.from_numpy(z[k]) for k in z.files if k.startswith("probe/")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    return recipe, inputs, sd, noise, out, perturb, probes, grads


def probe_loss(results, probes):
    total = 0.0
    for (entry, key), w in probes.items():
        t = results["coarse"][entry][key]
        total = total + (t * w.to(t.device)).sum()
    return total


def test_gradient_fixtures_present():
    assert len(GRAD_GOLDEN) >= 3


@pytest.mark.parametrize("path", GRAD_GOLDEN, ids=[os.path.basename(p)[:-4] for p in GRAD_GOLDEN])
def test_oracle_autograd_reproduces_reference_gradients(path):
    """torch.autograd through the oracle against the gradients the REFERENCE produced for the same functional."""
    recipe, inputs, sd, noise, want, perturb, probes, grads = load_gradient_fixture(path)
    cfg = recipe_config(recipe)
    names = [k for k in grads if k not in ("w2o", "style", "deformation")]
    sd = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sd[k].requires_grad_(True)
    leaf = [inputs[i].clone().requires_grad_(True) for i in (3, 4, 5)]
    torch.manual_seed(0)
    got = ro.composer_forward(cfg, sd, *inputs[:3], *leaf, inputs[6], perturb, training=True, noise=noise,
                              update_stats=False)
    rep = compare_results({"coarse": want["coarse"]}, {"coarse": got["coarse"]}, rtol=1e-5, atol=1e-6)
    # integrated_divergence needs the Hutchinson noise, which the fixtures do not carry
    assert not {k: v for k, v in rep.items() if not v[1] and not k.endswith("integrated_divergence")}
    probe_loss(got, probes).backward()
    mine = {k: sd[k].grad for k in names}
    mine.update(dict(zip(("w2o", "style", "deformation"), (t.grad for t in leaf))))
    for k, a in grads.items():
        b = mine[k] if mine[k] is not None else torch.zeros_like(a)
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, k


# --------------------------------------------------------------------------------------------
# renderer <-> decoder wire format (next row f-1): closed-form checks (the reference comparison runs in
# oracle/check_against_reference.py, exact equality)
# --------------------------------------------------------------------------------------------
def test_wire_format_folds_and_patches():
    from playableenvironments_amd import wire_format as wf
    h, w, strides = 16, 24, [4, 8]
    rows, cols = strided_grid_pixels(h, w, strides)
    image = torch.arange(h * w, dtype=torch.float32).reshape(h, w)
    rays = image[rows.long(), cols.long()].reshape(1, -1, 1)          # one value per ray, strided-grid order
    folded = wf.fold_strided_grid_samples(rays, strides, (h, w), dim=1)
    assert [tuple(f.shape) for f in folded] == [(1, 4, 6, 1), (1, 2, 3, 1)]
    for f, s in zip(folded, strides):
        assert torch.equal(f[0, ..., 0], image[s // 2::s, s // 2::s])
    d = wf.fold_strided_tensors({"a": {"b": rays.clone()}, "c": torch.zeros(5)}, h, w, strides)
    assert isinstance(d["a"]["b"], list) and torch.equal(d["a"]["b"][1], folded[1]) and torch.is_tensor(d["c"])
    with pytest.raises(Exception):
        wf.fold_strided_grid_samples(rays, [5], (h, w), dim=1)
    # strided patch: 8x8 @ stride 4 and 4x4 @ stride 8, 192 channels -> decoder inputs [64 @ /4, 128 @ /8]
    feats = torch.randn(2, 8 * 8 + 4 * 4, 192)
    splitted, patches = wf.decoder_patches(feats, 8, [4, 8], [64, 128])
    assert [tuple(p.shape) for p in patches] == [(2, 64, 8, 8), (2, 128, 4, 4)]
    assert torch.equal(patches[0][1, 5, 2, 3], feats[1, 2 * 8 + 3, 5])
    assert torch.equal(patches[1][0, 7, 1, 2], feats[0, 64 + 1 * 4 + 2, 64 + 7])
    assert torch.equal(splitted[1], feats[:, 64:, 64:])


def test_wire_format_reproduces_the_reference_subclass_decoder_inputs():
    """tests/golden/dropin (recorded by oracle/check_dropin.py from the reference's own multiresolution subclass): the
    ray-major features its composer returned, through this package's glue, are the tensors its decoder received - exactly."""
    import glob
    from playableenvironments_amd import wire_format as wf
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dropin", "*.npz")))
    assert len(paths) >= 2
    for path in paths:
        z = np.load(path)
        meta = ast.literal_eval(bytes(z["meta"]).decode())
        feats = torch.from_numpy(z["integrated_features"])
        want = [torch.from_numpy(z[f"decoder_input_{i}"]) for i in range(len(meta["strides"]))]
        _, patches = wf.decoder_patches(feats, meta["patch_size"], meta["strides"], [w.shape[-3] for w in want])
        for have, w in zip(patches, want):
            assert torch.equal(have, w), path


def test_wire_format_grid_samplers():
    from playableenvironments_amd import wire_format as wf
    feats = torch.randn(2, 5, 12, 20)
    r = torch.tensor([0, 3, 11, 7])
    c = torch.tensor([0, 19, 4, 10])
    pos = torch.stack([r.float() / 12, c.float() / 20], -1).unsqueeze(0).repeat(2, 1, 1)
    got = wf.sample_features_at(feats, pos, original_image_size=(12, 20))
    assert torch.allclose(got, feats[:, :, r, c].permute(0, 2, 1), atol=1e-5)        # pixel centres are hit exactly
    obs = torch.randn(1, 3, 32, 48)
    rows = (torch.arange(4) * 4 + 2 + 8).float() / 32
    cols = (torch.arange(4) * 4 + 2 + 12).float() / 48
    rr, cc = torch.meshgrid(rows, cols, indexing="ij")
    ppos = torch.stack([rr.reshape(-1), cc.reshape(-1)], -1).unsqueeze(0)
    region = wf.sample_original_region_from_patch_samples(obs, ppos, 4)
    assert torch.equal(region, obs[:, :, 8:24, 12:28])


def test_ray_object_distances_closed_form():
    """EnvironmentModel.compute_ray_object_distances: squared point-line distance to the box centres (the exact
    comparison with the reference method runs in oracle/check_against_reference.py)."""
    from playableenvironments_amd.environment_model import EnvironmentModel
    cfg = configs.minecraft_config()
    model = EnvironmentModel(cfg)
    k = model.object_id_helper.objects_count
    o2w = torch.eye(4).reshape(1, 4, 4, 1).repeat(1, 1, 1, k).clone()
    o2w[0, 0, 3, :] = torch.arange(k, dtype=torch.float32)             # object k shifted by k along x
    origins = torch.tensor([[[0.0, 5.0, 0.0]]])                         # (1, C=1, 3)
    dirs = torch.tensor([[[[0.0, -2.0, 0.0], [1.0, 0.0, 0.0]]]])       # straight down / along x
    d = model.compute_ray_object_distances(origins, dirs, o2w)
    assert tuple(d.shape) == (1, 1, 2, k)
    centres = torch.stack([model.object_composer.object_models_coarse[model.object_id_helper.model_idx_by_object_idx(i)]
                           .bounding_box.get_center_offset() for i in range(k)]) + torch.stack(
                               [torch.tensor([float(i), 0.0, 0.0]) for i in range(k)])
    want_down = centres[:, 0] ** 2 + centres[:, 2] ** 2                # line x = z = 0
    want_x = (centres[:, 1] - 5.0) ** 2 + centres[:, 2] ** 2           # line y = 5, z = 0
    assert torch.allclose(d[0, 0, 0], want_down, atol=1e-5) and torch.allclose(d[0, 0, 1], want_x, atol=1e-5)


def test_pose_math_against_reference_fixture():
    """Host pose math of EnvironmentModel (objects batched, closed-form rigid inverse, no host synchronisation) against
    outputs of the reference's per-object torch.inverse path (tests/golden/host/pose_math_minecraft.npz, written by
    oracle/make_golden.py)."""
    from playableenvironments_amd.environment_model import EnvironmentModel, euler_to_matrix, rigid_inverse
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "host", "pose_math_minecraft.npz")).items()}
    cfg = configs.minecraft_config()
    model = EnvironmentModel(cfg)
    w2o, o2w = model.compute_transformation_matrix_w2o_o2w(g["object_rotation_parameters"], g["object_translation_parameters"])
    c2w = euler_to_matrix(g["camera_rotations"], g["camera_translations"])
    w2c = rigid_inverse(c2w)
    focals = g["focals"] * cfg["data"]["focal_length_multiplier"]
    boxes, points = model.compute_object_bounding_boxes(o2w, w2c, focals, 288, 512)
    axes = model.compute_object_axes_projection(o2w, w2c, focals, 288, 512)
    for name, got in (("w2o", w2o), ("o2w", o2w), ("w2c", w2c), ("boxes", boxes), ("box_points", points), ("axes", axes)):
        assert got.shape == g[name].shape, name
        assert torch.allclose(got, g[name], rtol=1e-4, atol=1e-5), name
    # the inverse really is one: w2o o2w = I for every (frame, object)
    prod = torch.einsum("...ijk,...jlk->...ilk", w2o, o2w)
    assert torch.allclose(prod, torch.eye(4).reshape(4, 4, 1).expand_as(prod), atol=1e-5)
    # gradients flow to the pose parameters through the closed form
    rot = g["object_rotation_parameters"].clone().requires_grad_(True)
    model.compute_transformation_matrix_w2o_o2w(rot, g["object_translation_parameters"])[0].square().sum().backward()
    assert torch.isfinite(rot.grad).all() and float(rot.grad.abs().max()) > 0


# --------------------------------------------------------------------------------------------
# forward_expected_positions: outputs and gradients recorded from the reference (tests/golden/expected_positions)
# --------------------------------------------------------------------------------------------
EXPECTED_GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "expected_positions", "*.npz")))


def load_expected_positions_fixture(path):
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k])
    recipe = ast.literal_eval(bytes(z["recipe"]).decode())
    inputs = [t(f"in/{i}") for i in range(7)]
    sd = {k[3:]: t(k) for k in z.files if k.startswith("sd/")}
    noise = {k[6:]: t(k) for k in z.files if k.startswith("noise/")}
    types = sorted({k.split("/")[1] for k in z.files if k.startswith("out/")})
    out = {ty: (t(f"out/{ty}/0"), t(f"out/{ty}/1")) for ty in types}
    probes = {ty: (t(f"probe/{ty}/0"), t(f"probe/{ty}/1")) for ty in types}
    grads = {k[5:]: t(k) for k in z.files if k.startswith("grad/")}
    return recipe, inputs, sd, noise, out, probes, grads, bool(int(z["perturb"])), int(z["object_id"])


def pose_parameter_gradients(w2o: torch.Tensor, grad_w2o: torch.Tensor):
    """Projects an elementwise d loss / d w2o onto the rigid motions: gradients with respect to Euler angles and a
    translation composed on the right of ``w2o`` (what the reference's pose parameters see)."""
    lead = list(w2o.shape[:-2])
    rot = torch.zeros(lead + [3], requires_grad=True)
    tr = torch.zeros(lead + [3], requires_grad=True)
    (torch.matmul(w2o, ro.euler_to_matrix(rot, tr)) * grad_w2o).sum().backward()
    return rot.grad, tr.grad


def test_expected_positions_fixtures_present():
    assert len(EXPECTED_GOLDEN) >= 2


@pytest.mark.parametrize("path", EXPECTED_GOLDEN, ids=[os.path.basename(p)[:-4] for p in EXPECTED_GOLDEN])
def test_oracle_expected_positions_reproduce_reference_outputs_and_gradients(path):
    recipe, inputs, sd, noise, want, probes, grads, perturb, obj = load_expected_positions_fixture(path)
    cfg = recipe_config(recipe)
    o, d, n, w2o, sty, dfm, ins = inputs
    names = [k for k in grads if k not in ("w2o", "style", "deformation")]
    sd = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sd[k].requires_grad_(True)
    leaf = [t[..., obj].clone().requires_grad_(True) for t in (w2o, sty, dfm)]
    got = ro.expected_positions_forward(cfg, sd, o, d, n, *leaf, ins[..., obj], obj, perturb, training=True, noise=noise)
    assert set(got) == set(want)
    for ty in want:
        for a, b in zip(want[ty], got[ty]):
            assert torch.equal(a, b.detach()), ty
    sum((t * p).sum() for ty in got for t, p in zip(got[ty], probes[ty])).backward()
    mine = {k: sd[k].grad for k in names}
    mine.update(dict(zip(("w2o", "style", "deformation"), (t.grad for t in leaf))))
    largest = max(float(a.abs().max()) for a in grads.values())
    for k, a in grads.items():
        b = mine[k] if mine[k] is not None else torch.zeros_like(a)
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-7 * largest, k
    assert float(grads["w2o"].abs().max()) > 0 and float(grads["deformation"].abs().max()) > 0


CONSISTENCY_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "consistency", "*.npz")))


@pytest.mark.parametrize("path", CONSISTENCY_GOLDEN, ids=[os.path.basename(p)[:-4] for p in CONSISTENCY_GOLDEN])
def test_consistency_forwards_reproduce_reference_outputs(path, monkeypatch):
    """forward_pose_consistency / forward_keypoint_consistency of the product's host logic with the ORACLE behind its composer,
    on the pixels the reference drew: the reference's expected positions, opacities, confidences - value for value."""
    from oracle.check_against_reference import _OracleComposerAdapter
    from playableenvironments_amd import environment_model as em
    from tests.helpers import run_consistency_fixture, stand_in_encoders
    assert len(CONSISTENCY_GOLDEN) >= 2
    z = np.load(path)
    meta = ast.literal_eval(bytes(z["meta"]).decode())
    cfg = recipe_config(ast.literal_eval(bytes(z["recipe"]).decode()))
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, meta["world"]))
    model.object_composer.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
    model.object_composer = _OracleComposerAdapter(cfg, model.object_composer.eval())
    pairs = run_consistency_fixture(z, model.eval(), "cpu", monkeypatch)
    for k, (want, got) in pairs.items():
        assert want.shape == got.shape, k
        assert torch.allclose(want, got, rtol=1e-5, atol=5e-6), (k, float((want - got).abs().max()))   # (positions up to 2.2)
    assert len(pairs) == 16


def test_consistency_path_samplers():
    """sample_rays_at / sample_rays_at_object / sample_rays_at_keypoints (ray_helper.py:797-1052; exact equality with the
    reference functions is checked in oracle/check_against_reference.py): closed-form properties."""
    from playableenvironments_amd import ray_sampling as rs
    h, w = 12, 20
    rows, cols = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([cols, rows, -torch.ones_like(rows)], dim=-1).unsqueeze(0)             # direction = (col, row, -1)
    # at pixel centres (normalised with the range correction) the lookup returns the grid itself
    pos = torch.tensor([[[3 / h, 7 / w], [0.0, 0.0], [(h - 1) / h, (w - 1) / w]]])
    got = rs.sample_rays_at(grid, pos, correct_range=True, original_image_size=(h, w))
    assert torch.allclose(got[0, :, :2], torch.tensor([[7.0, 3.0], [0.0, 0.0], [w - 1.0, h - 1.0]]), atol=1e-4)
    # in between, the lookup is linear in the position (what a pinhole direction grid is)
    mid = rs.sample_rays_at(grid, torch.tensor([[[0.5, 0.5]]]), correct_range=False)
    assert torch.allclose(mid[0, 0, :2], torch.tensor([(w - 1) / 2, (h - 1) / 2]), atol=1e-4)
    # samples at an object: all inside the pixel-aligned box, values gathered at the same pixels
    images = torch.stack([rows, cols]).unsqueeze(0)                                            # (1, 2, H, W): value = (row, col)
    box = torch.tensor([[0.25, 0.5, 0.6, 0.9]])                                                 # left, top, right, bottom
    torch.manual_seed(0)
    d, o, p = rs.sample_rays_at_object(grid, images, 200, box)
    assert d.shape == (1, 200, 3) and o.shape == (1, 200, 2) and p.shape == (1, 200, 2)
    assert torch.equal(d[..., 0], o[..., 1]) and torch.equal(d[..., 1], o[..., 0])
    assert float(o[..., 1].min()) >= 5 and float(o[..., 1].max()) < 12 and float(o[..., 0].min()) >= 6 and float(o[..., 0].max()) < 11
    assert torch.allclose(p[..., 0] * h, o[..., 0]) and torch.allclose(p[..., 1] * w, o[..., 1])
    # keypoints: every sample lies on its skeleton segment, the fraction is shared by observations and cameras
    kp = torch.rand(2, 3, 2, 17, 3)
    grid6 = grid.reshape(1, 1, 1, h, w, 3).expand(2, 3, 2, h, w, 3)
    torch.manual_seed(1)
    dirs, positions, scores = rs.sample_rays_at_keypoints(grid6, kp, 20)
    assert dirs.shape == (2, 3, 2, 20, 3) and positions.shape == (2, 3, 2, 20, 2) and scores.shape == (2, 3, 2, 20)
    for s in range(20):
        a, b = rs.SKELETON_SEGMENTS[s % 16]
        frac = (scores[..., s] - kp[..., a, 2]) / (kp[..., b, 2] - kp[..., a, 2])
        assert torch.allclose(frac, frac[:, :1, :1].expand_as(frac), atol=1e-3)
        want = kp[..., a, :2] + (kp[..., b, :2] - kp[..., a, :2]) * frac.unsqueeze(-1)
        assert torch.allclose(positions[..., s, :], want, atol=1e-4)


def test_camera_rays_at_continuous_positions():
    """environment_model.camera_rays_at_positions (no direction grid) against the reference's route - build the pinhole
    grid, sample it bilinearly at the positions, rotate into the world (oracle restatements of create_camera_rays /
    transform_rays, ray_sampling.sample_rays_at)."""
    from playableenvironments_amd import ray_sampling as rs
    from playableenvironments_amd.environment_model import camera_rays_at_positions, euler_to_matrix
    torch.manual_seed(0)
    lead, h, w = [2, 3, 1], 36, 64
    focals = 40.0 + 10.0 * torch.rand(lead)
    c2w = euler_to_matrix(torch.rand(lead + [3]) - 0.5, torch.randn(lead + [3]))
    positions = torch.rand(lead + [25, 2])
    grid, o, n = ro.create_camera_rays(lead, h, w, focals)
    # with the range correction the positions are pixel / size, i.e. at most (size - 1) / size
    pixel_positions = positions * torch.tensor([(h - 1) / h, (w - 1) / w])
    for correct, pos in ((False, positions), (True, pixel_positions)):
        sampled = rs.sample_rays_at(grid, pos, correct_range=correct, original_image_size=(h, w))
        wo, wd, wn = ro.transform_rays(o, sampled, n, c2w)
        go, gd, gn = camera_rays_at_positions(c2w, focals, h, w, pos, correct_range=correct)
        assert torch.allclose(wd, gd, rtol=1e-5, atol=1e-5) and torch.allclose(wo, go) and torch.allclose(wn, gn)


# ------------------------------------------------------------------------------------------------
# producers of the renderer's inputs: encoders, roi_pool restatement, Batch
# ------------------------------------------------------------------------------------------------
def test_roi_pool_oracle_known_answers():
    """The CPU restatement of torchvision.ops.roi_pool: hand-computed bins, an empty bin, per-bin clipping at the image
    border, and ATen's adaptive_max_pool2d (an independent implementation) on boxes inside the image."""
    from oracle import roi_pool_oracle as rp
    x = torch.arange(36, dtype=torch.float32).reshape(1, 1, 6, 6)
    out, arg = rp.roi_pool(x, torch.tensor([[0.0, 1.0, 1.0, 4.0, 4.0]]), (2, 2))
    assert out.reshape(-1).tolist() == [14.0, 16.0, 26.0, 28.0] and arg.reshape(-1).tolist() == [14, 16, 26, 28]
    out, arg = rp.roi_pool(x, torch.tensor([[0.0, 4.0, 4.0, 9.0, 9.0]]), (2, 2))     # leaves the image: clipped, then empty
    assert out.reshape(-1).tolist() == [35.0, 0.0, 0.0, 0.0] and arg.reshape(-1).tolist() == [35, -1, -1, -1]
    out, _ = rp.roi_pool(x, torch.tensor([[0.0, 2.4, 2.6, 2.4, 2.6]]), (1, 1))        # rounds to the single pixel (3, 2)
    assert float(out) == 20.0
    g = rp.roi_pool_backward(torch.ones(1, 1, 2, 2), torch.tensor([[[[14, 16], [26, 14]]]]), torch.tensor([[0.0, 0, 0, 0, 0]]), (1, 1, 6, 6))
    assert float(g.reshape(-1)[14]) == 2.0 and float(g.sum()) == 4.0
    assert rp.check_against_adaptive_max_pool(seed=1, cases=25)


def test_encoders_build_from_config_and_keep_reference_names():
    from playableenvironments_amd import encoders
    from playableenvironments_amd.environment_model import EnvironmentModel
    cfg = configs.minecraft_config(encoders=True)
    model = EnvironmentModel(cfg)
    assert [type(m).__name__ for m in model.object_encoders] == ["ObjectEncoderV5", "ObjectEncoderV5", "ObjectEncoderV4"]
    assert [type(m).__name__ for m in model.object_parameters_encoders] == ["StaticObjectParametersEncoder",
                                                                            "StaticObjectParametersEncoder", "ObjectParametersEncoderV4"]
    keys = set(model.state_dict())
    for name in ("object_encoders.2.conv1.weight", "object_encoders.2.initial_backbone.0.downsample.0.weight",
                 "object_encoders.2.final_backbone.3.bn2.running_var", "object_encoders.0.style_head.bias",
                 "object_parameters_encoders.2.rotation_head.weight", "object_parameters_encoders.0.translation_range",
                 "object_composer.object_models_coarse.2.nerf_model.backbone_layers.4.weight"):
        assert name in keys, name
    assert tuple(model.object_encoders[2].conv1.weight.shape) == (16, 9, 3, 3)
    assert float(model.object_parameters_encoders[2].rotation_head.weight.detach().abs().max()) <= 1e-5
    assert len(EnvironmentModel(configs.minecraft_config()).object_encoders) == 0          # renderer-only configuration
    with pytest.raises(Exception, match="not supported"):
        bad = configs.tennis_config(encoders=True)
        bad["model"]["object_encoders"][0]["architecture"] = "model.unknown_encoder"
        encoders.create_encoders(bad)
    # closed-form range normalisation == the reference's subtract-until-inside loops
    v = torch.tensor([-4.0, -0.9, -0.78, 0.0, 0.78, 0.8, 3.0, 7.1])
    got = encoders.ObjectParametersEncoderV4.normalize_range(v, -0.785, 0.785)
    want = v.clone()
    for i in range(want.numel()):
        while want[i] > 0.785:
            want[i] -= 1.57
        while want[i] < -0.785:
            want[i] += 1.57
    assert torch.allclose(got, want, atol=1e-6)


def test_static_and_classic_parameter_encoders_closed_form():
    """Pose estimators that need no network: the static encoder returns the middle of its ranges; the classic encoder
    casts the bottom centre of the box onto the ground plane - checked against a hand-built camera looking straight down."""
    from playableenvironments_amd import encoders
    from playableenvironments_amd.environment_model import euler_to_matrix, rigid_inverse
    cfg = configs.tennis_config(encoders=True)
    static = encoders.StaticObjectParametersEncoder(cfg, cfg["model"]["object_parameters_encoder"][1])
    rot, tr = static(torch.zeros(2, 3, 1, 3, 8, 8))
    assert tuple(tr.shape) == (2, 3, 3, 1) and torch.allclose(tr[0, 0, :, 0], torch.tensor([0.0, 20.085, 0.0])) and float(rot.abs().max()) == 0
    classic = encoders.ClassicObjectParametersEncoder(cfg, cfg["model"]["object_parameters_encoder"][2])
    # camera 10 m above the origin looking along -z (identity rotation): pixel (u, v) hits the ground at 10 * (u, -v) / f
    c2w = euler_to_matrix(torch.zeros(1, 1, 1, 3), torch.tensor([[[[0.0, 0.0, 10.0]]]]))
    boxes = torch.tensor([0.55, 0.2, 0.65, 0.6]).reshape(1, 1, 1, 4, 1)               # bottom centre at (0.6 W, 0.6 H)
    rot, tr = classic(torch.zeros(1, 1, 1, 3, 100, 200), rigid_inverse(c2w), torch.zeros(1, 1, 1, 3), torch.full((1, 1, 1), 50.0), boxes,
                      torch.ones(1, 1, 1, 1, dtype=torch.bool))
    want = torch.tensor([10.0 * (0.6 * 200 - 100) / 50.0, -10.0 * (0.6 * 100 - 50) / 50.0, 0.01])
    assert torch.allclose(tr[0, 0, :, 0], want, atol=1e-4), tr
    _, gone = classic(torch.zeros(1, 1, 1, 3, 100, 200), rigid_inverse(c2w), torch.zeros(1, 1, 1, 3), torch.full((1, 1, 1), 50.0), boxes,
                      torch.zeros(1, 1, 1, 1, dtype=torch.bool))
    assert float(gone.abs().max()) == 0.0


def test_batch_packs_into_one_arena_and_keeps_the_reference_tuple():
    from playableenvironments_amd import batching
    from tests.helpers import observation_batch
    b = observation_batch(synthetic.tennis_scene(batch=2, observations=3, image_size=(24, 32)))
    flow = torch.randn(2, 3, 1, 2, 24, 32)
    kp = torch.rand(2, 3, 1, 17, 3, 2)
    batch = batching.batch_from_tensors(b["observations"], b["camera_rotations"], b["camera_translations"], b["focals"], b["bounding_boxes"],
                                        b["bounding_boxes_validity"], b["global_frame_indexes"], b["video_frame_indexes"],
                                        b["video_indexes"], optical_flows=flow, keypoints=kp,
                                        keypoints_validity=torch.ones(2, 3, 1, 2, dtype=torch.bool))
    assert batch.size == 3 and batch.has_flow() and batch.has_keypoints() and not batch.has_object_poses()
    assert batch.pin_memory() is batch
    base = batch._arena.data_ptr()
    for name in ("observations", "bounding_boxes_validity", "optical_flows", "keypoints", "video_indexes"):
        t = getattr(batch, name)
        assert base <= t.data_ptr() < base + batch._arena.numel() and (t.data_ptr() - base) % 256 == 0
    t = batch.to_tuple(cuda=False)
    assert len(t) == 12 and torch.equal(t[0], b["observations"]) and torch.equal(t[8], b["bounding_boxes_validity"])
    assert t[8].dtype == torch.bool and t[9].dtype == torch.int64
    assert torch.equal(batch.to_keypoints_typle(cuda=False)[0], kp)
    args = batch.observation_mode_arguments(cuda=False)
    assert len(args) == 9 and torch.equal(args[1], b["camera_rotations"]) and torch.equal(args[8], b["video_indexes"])
    with pytest.raises(Exception, match="Object poses"):
        batch.to_object_poses_tuple(cuda=False)


def test_camera_parameters_storage_checkpoint_layout_and_values():
    """Learnable camera offsets (model/layers/camera_parameters_storage.py): one table here, the reference's
    one-parameter-per-entry checkpoint keys outside; entry = frame + camera * storage_size; x10 / x1000 scales; zeros
    outside training.  (The comparison with the imported reference module is in oracle/check_against_reference.py.)"""
    from playableenvironments_amd.modules import CameraParametersStorage
    from playableenvironments_amd.environment_model import EnvironmentModel
    store = CameraParametersStorage(4, 2)
    sd = store.state_dict()
    assert list(sd) == [f"storage.storage.{i}" for i in range(8)] and all(v.shape == (7,) for v in sd.values())
    sd = {k: torch.full((7,), float(i + 1)) for i, k in enumerate(sd)}
    store.load_state_dict(sd, strict=True)
    frames = torch.tensor([[1, 3]])
    rot, tr, focal = store.train()(frames)
    assert rot.shape == (1, 2, 2, 3) and focal.shape == (1, 2, 2)
    assert torch.equal(rot[0, :, :, 0], torch.tensor([[2., 6.], [4., 8.]]))          # (frame, camera) -> entry + 1
    assert torch.equal(tr, rot * 10) and torch.equal(focal, rot[..., 0] * 1000)
    assert all(float(o.abs().max()) == 0 for o in store.eval()(frames))
    with pytest.raises(RuntimeError, match="Missing key"):
        store.load_state_dict({k: v for k, v in sd.items() if not k.endswith(".3")}, strict=True)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        store.load_state_dict(dict(sd, **{"storage.storage.8": torch.zeros(7)}), strict=True)

    cfg = configs.reduced_config(configs.tennis_config(), width=32, layers=2, skip=1, features=8, octaves=2,
                                 bender_width=16, bender_layers=2, bender_skip=1, bender_octaves=2)
    cfg["model"]["enable_camera_parameters_offsets"] = True
    cfg["model"]["camera_parameters_memory_size"] = 3
    model = EnvironmentModel(cfg)
    assert [k for k in model.state_dict() if k.startswith("camera_parameters_offsets.")] == \
        [f"camera_parameters_offsets.storage.storage.{i}" for i in range(3)]
    with torch.no_grad():
        model.camera_parameters_offsets.table.copy_(torch.arange(21.).reshape(3, 7) * 1e-3)
    rotations, translations, focals = torch.zeros(1, 2, 1, 3), torch.zeros(1, 2, 1, 3), torch.full((1, 2, 1), 100.)
    model.train()
    r, t, f = model._corrected_cameras(rotations, translations, focals, torch.tensor([[2, 0]]))
    assert torch.allclose(r[0, 0, 0], torch.tensor([14., 15., 16.]) * 1e-3) and torch.allclose(t[0, 1, 0], torch.tensor([3., 4., 5.]) * 1e-2)
    assert torch.allclose(f[0, :, 0], torch.tensor([100. + 20., 100. + 6.]))
    model.eval()
    r, t, f = model._corrected_cameras(rotations, translations, focals, torch.tensor([[2, 0]]))
    assert torch.equal(r, rotations) and torch.equal(f, focals)


def test_trainer_loss_fixtures_present():
    """tests/golden/consumers/*_loss_info.npz: the loss_info dictionary of the reference's TrainerMultiresolutionBackpropagatedDecoder.
    compute_losses on its own model (oracle/check_consumers.py write), which the same script reproduces - entry by entry, with the
    gradient of the total loss - on the reference's subclass over this package's EnvironmentModel.  Data only; the check itself
    needs the reference and runs in the build container."""
    import numpy as np
    for world in ("tennis", "minecraft"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "consumers", f"{world}_loss_info.npz"))
        keys = [k for k in z.files if k.startswith("info/")]
        assert len(keys) == 58 and "info/loss" in keys and "info/bounding_box_loss" in keys and "info/coarse_reconstruction_loss" in keys
        assert abs(float(z["info/loss"]) - float(z["total_loss"])) < 1e-6
        assert all(np.isfinite(float(z[k])) for k in keys)


def test_adam_step_rejects_bad_arguments_without_a_gpu():
    """pr_adam_step validates before anything is enqueued (callable error paths on a box without a GPU)."""
    lib = _lib.load()
    assert lib.pr_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, 1.0, None, None, None, None) == 0     # nothing to do
    assert lib.pr_adam_step(None, None, None, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, 1.0, None, None, None, None) == -1
    assert b"NULL pointer" in lib.pr_last_error()
    assert lib.pr_adam_step(256, 256, 256, 256, 8, 1e-3, 1.0, 0.999, 1e-8, 0.0, 0, 0, 1.0, None, None, None, None) == -1
    assert b"betas" in lib.pr_last_error()
    assert lib.pr_adam_step(256, 256, 256, 256, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, 0.0, None, None, None, None) == -1
    assert b"step" in lib.pr_last_error()
    from playableenvironments_amd import parallel
    with pytest.raises(ValueError, match="amsgrad"):
        parallel.ArenaAdam([torch.nn.Parameter(torch.zeros(2))], amsgrad=True)


def test_scene_setup_rejects_bad_arguments_without_a_gpu():
    """pr_scene_setup (the fused scene set-up of an evaluation call) validates its sizes and pointers before anything is
    enqueued: callable error paths on a box without a GPU, like the renderer's (test_abi_error_paths)."""
    lib = _lib.load()
    assert lib.pr_scene_setup(None, None) == -1
    assert b"NULL argument" in lib.pr_last_error()
    q = _lib.SceneSetup()
    q.frames, q.cameras, q.objects, q.box_points_per_object = 1, 1, 4, 8
    q.style_features, q.deformation_features, q.height, q.width = 32, 32, 288, 512
    q.focal_multiplier, q.upsample_factor = 0.5, 1.0
    assert lib.pr_scene_setup(C.byref(q), None) == -1               # every pointer is NULL
    assert b"NULL pointer" in lib.pr_last_error()
    q.objects = 9                                                     # more instances than the ABI holds
    assert lib.pr_scene_setup(C.byref(q), None) == -1
    assert b"bad sizes" in lib.pr_last_error()
    q.objects, q.frames = 4, 0                                        # an empty call is a no-op
    assert lib.pr_scene_setup(C.byref(q), None) == 0


def test_bench_line_is_compact_and_complete():
    """The ONE stdout line of bench.py (what the driver parses into BENCH_rNN.json): built from a full record of every leg - round
    4's committed 22.6 KB record, which the driver could not parse as a line - it must stay under bench.LINE_BUDGET bytes, be
    json, and carry the contract keys with `roofline` and `cpu_baseline`; an 8-rank record (rank_devices, both collectives) too."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.LINE_BUDGET <= 4096
    with open(os.path.join(root, "profiles", "r04_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    # the widest case: 8 ranks, both collectives, the identical-frames secondary
    wide = dict(full, n_gpus=8)
    wide["distributed"] = {"world_size": 8, "backend": "nccl", "nccl_version": "2.26.6", "launched_by": "torch.distributed.run",
                           "rank_devices": [f"rank {r}: cuda:{r} AMD Instinct MI355X" for r in range(8)]}
    wide["feature_gather"] = {"world_size": 8, "bytes_per_rank": 50331648, "tensor": [1, 1, 1, 65536, 192],
                              "all_gather": {"ms": 3.2, "receive_GB_per_s_per_receiving_rank": 110.1, "receiving_ranks": 8},
                              "gather_dst0": {"ms": 3.0, "receive_GB_per_s_per_receiving_rank": 117.4, "receiving_ranks": 1}, "link_note": "x" * 300}
    wide["identical_frames"] = {"value": 2.5, "unit": "Mrays/s", "ms_per_step": 207.0, "note": "y" * 200}
    # round 6: the shader clock / package power of the measured region ride along (tools/gpu_telemetry.py)
    full["roofline"] = dict(full["roofline"], clock={"sclk_mhz": 2395.1, "power_w": 1157.3, "power_cap_w": 1400.0}, clock_note="z" * 300)
    full["split_precision"] = dict(full["split_precision"], clock={"sclk_mhz": 2062.0, "power_w": 1284.0})
    for record in (full, wide):
        text = bench.compact_line(record, "bench_full.json")
        assert len(text.encode()) < bench.LINE_BUDGET and "\n" not in text
        line = json.loads(text)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline", "distributed", "summary", "full"):
            assert key in line, key
        assert line["value"] == record["value"] and line["ms_per_step"] == record["ms_per_step"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in line["roofline"], key
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
        assert line["summary"]["train_step_ms"] == record["train_step"]["ms_per_step"]
    assert json.loads(bench.compact_line(full))["roofline"]["clock"]["sclk_mhz"] == 2395.1
    assert json.loads(bench.compact_line(full))["summary"]["split_precision_sclk_mhz"] == 2062.0
    assert len(json.loads(bench.compact_line(wide))["distributed"]["rank_devices"]) == 8
    assert json.loads(bench.compact_line(wide))["feature_gather"]["all_gather_GB_per_s"] == 110.1
    # a record bloated beyond the budget still yields a parsable line (optional blocks go first, the contract keys never)
    bloated = dict(wide)
    bloated["config"] = dict(wide["config"], workload="w" * 1500)
    bloated["distributed"] = dict(wide["distributed"], rank_devices=["d" * 300 for _ in range(8)])
    text = bench.compact_line(bloated)
    assert len(text.encode()) < bench.LINE_BUDGET and json.loads(text)["roofline"]["frac"] == full["roofline"]["frac"]
