"""GPU parity suite (python -m pytest tests -m gpu, on an MI355X): the HIP renderer, called through
the C ABI, against the CPU oracle and the reference-generated golden vectors.

Tolerance (fp32 path, exact-fp32 MFMA): rtol 1e-4, atol 1e-5 on every result field, NaNs in the
same places.  Sample depths and AABB decisions must be bit-identical."""
import os
import sys

import pytest
import warnings

import torch

from oracle import render_oracle as ro
from oracle.make_golden import recipe_config
from playableenvironments_amd import ObjectComposer, configs, synthetic
from playableenvironments_amd import environment_model as em
from tests.helpers import (arbitrate, bender_kink_margin, compare_results, composer_inputs, grid_pixels, oracle_in_float64, poison_device_memory,
                           to_double)
from tests.test_cpu import GOLDEN, load_fixture

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_library):
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a GPU: the renderer has no CPU fallback")


def build(cfg, seed=0, step=20000, alpha_bias=2.0, precision="fp32"):
    torch.manual_seed(seed)
    comp = ObjectComposer(cfg)
    comp.precision = precision
    synthetic.randomize_module_state(comp, seed=seed, step=step, alpha_bias=alpha_bias, bender_scale=1e4)
    return comp.eval()


def run_both(cfg, comp, inputs, perturb=False, canonical=False, export=False):
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    rec = {}
    with torch.no_grad():
        torch.manual_seed(123)
        want = ro.composer_forward(cfg, sd, *inputs, perturb, canonical_pose=canonical, record_noise=rec, stable_merge=True)
        comp = comp.cuda()
        got = comp(*[v.cuda() for v in inputs], perturb, canonical_pose=canonical, _noise=rec if perturb else None,
                   _export=export)
    torch.cuda.synchronize()
    run_both.noise = rec            # (the draws of this call, for a float64 replay)
    return want, got


def run_exact(cfg, state, inputs, perturb, noise, canonical=False, training=False):
    """The oracle's op graph in float64 on the same weights, inputs and replayed noise: the arbiter of the tolerances that
    are wider than fp32 round-off (tests/helpers.arbitrate)."""
    with oracle_in_float64(), torch.no_grad():
        return ro.composer_forward(cfg, to_double(state), *to_double(list(inputs)), perturb, canonical_pose=canonical,
                                   training=training, noise=to_double(noise), update_stats=False, stable_merge=True)


def assert_no_farther_than_the_oracle(exact, want, got, fields, factor=4.0):
    """|HIP - fp64| <= factor x |fp32 oracle - fp64| for the listed result fields (suffix match)."""
    rep = {k: v for k, v in arbitrate(exact, want, got, factor=factor).items() if k.endswith(fields)}
    assert rep
    bad = {k: f"HIP {v[0]:.3e} vs oracle {v[1]:.3e}" for k, v in rep.items() if not v[2]}
    assert not bad, bad
    return rep


def assert_close(want, got, rtol=RTOL, atol=ATOL):
    rep = compare_results(want, got, rtol=rtol, atol=atol)
    bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1]}
    assert not bad, bad


CASES = {
    "tennis": (lambda: configs.tennis_config(), lambda: synthetic.tennis_scene(), 24, 2.0),
    "tennis_two_frames": (lambda: configs.tennis_config(), lambda: synthetic.tennis_scene(batch=2, observations=2, seed=3), 12, 2.0),
    "minecraft": (lambda: configs.minecraft_config(), lambda: synthetic.minecraft_scene(), 24, 3.0),
    "minecraft_two_frames": (lambda: configs.minecraft_config(), lambda: synthetic.minecraft_scene(batch=2, seed=8), 14, 3.0),
    "tennis_hierarchical": (lambda: configs.tennis_config(hierarchical=(16, 32)), lambda: synthetic.tennis_scene(seed=5), 16, 2.0),
    # BASELINE.json configs[1] itself: 4 objects x (64 coarse + 128 resampled) positions, the benchmark's scene - 256 / 768
    # merged entries per ray, i.e. the four-wave compositing kernel with K > 1, the resampler's 64 -> 192 rank merge and
    # the segmented transmittance scan across objects
    "tennis_c2_64_128": (lambda: configs.tennis_config(hierarchical=(64, 128)), lambda: synthetic.tennis_scene(seed=1234), 24, 2.0),
    # the skybox ships with one position per ray, which the reference's resampler cannot handle
    # (empty pdf) -> give it 3 coarse + 2 fine positions
    "minecraft_hierarchical": (lambda: configs.reduced_config(configs.enable_fine(configs.minecraft_config()), width=256, layers=8,
                                                              skip=4, features=192, octaves=10, bender_width=128, bender_layers=6,
                                                              bender_skip=3, bender_octaves=6,
                                                              positions={"background": (16, 16), "skybox": (3, 2), "player_1": (32, 32)}),
                               lambda: synthetic.minecraft_scene(seed=6), 12, 3.0),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("perturb", [False, True], ids=["eval", "perturb"])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_composer_matches_oracle(name, perturb, precision):
    """Both MLP kernels (exact fp32 MFMA, and fp32 emulated with three fp16 MFMAs) against the oracle at
    the same tolerance (rtol 1e-4 / atol 1e-5)."""
    make_cfg, make_scene, n, bias = CASES[name]
    cfg, scene = make_cfg(), make_scene()
    comp = build(cfg, alpha_bias=bias, precision=precision)
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    want, got = run_both(cfg, comp, inputs, perturb=perturb)
    assert set(got) == set(want)
    if name == "tennis_c2_64_128" and perturb:
        # 128 randomly placed resampled depths per object between 64 coarse ones: the per-sample weights of the fine pass are
        # a sensitive function of the coarse pass (inverse CDF -> sample spacing -> alpha).  Measured on this case: the
        # ORACLE's own fine weights move by 7.5e-6 when its network weights change by one ulp; the kernels' dot products
        # differ from torch's by several ulp (summation order).  Every integrated field keeps the fp32 tolerance; the
        # per-sample weights are
        # ARBITRATED in float64: the kernels' weights may be as far from the exact result as the fp32 oracle's are (x 4).
        rep = compare_results(want, got, rtol=RTOL, atol=ATOL)
        bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1] and not k.endswith("weights")}
        assert not bad, bad
        state = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
        exact = run_exact(cfg, state, inputs, True, run_both.noise)
        assert_no_farther_than_the_oracle(exact, want, got, "weights")
        return
    assert_close(want, got)


@pytest.mark.parametrize("name", list(CASES))
def test_half_precision_tier_is_close_to_the_oracle(name):
    """precision="f16" (PR_PRECISION_F16: the split kernel's hi x hi product only - plain fp16 operands, fp32 accumulation) is the
    throughput tier, not a parity configuration: its rendered fields stay within rtol 2e-2 (atol 2e-2 of the field's peak) of the
    oracle and the feature image keeps >= 40 dB PSNR; the geometry (sample counts) is the exact path's; and it really is another
    kernel than "f16x3" (the results differ)."""
    make_cfg, make_scene, n, bias = CASES[name]
    cfg, scene = make_cfg(), make_scene()
    comp = build(cfg, alpha_bias=bias, precision="f16")
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    want, got = run_both(cfg, comp, inputs)
    comp.precision = "f16x3"
    with torch.no_grad():
        split = comp(*[v.cuda() for v in inputs], False)
    last = "fine" if "fine" in got else "coarse"
    differs = False
    for level in ("coarse", "fine"):
        if level not in got:
            continue
        for field in ("integrated_features", "opacity", "depth"):
            if field not in got[level]["global"]:
                continue
            w, g = want[level]["global"][field].double(), got[level]["global"][field].cpu().double()
            peak = float(w.abs().max())
            assert torch.allclose(g, w, rtol=2e-2, atol=2e-2 * peak), (level, field, float((g - w).abs().max()), peak)
            differs |= not torch.equal(got[level]["global"][field], split[level]["global"][field])
    w = want[last]["global"]["integrated_features"].double()
    g = got[last]["global"]["integrated_features"].cpu().double()
    psnr = 10.0 * torch.log10(w.abs().max() ** 2 / ((g - w) ** 2).mean())
    assert float(psnr) >= 40.0, float(psnr)
    assert differs


def test_single_player_all_rays_in_box():
    """BASELINE.json configs[0] shape: one player object, 32 samples/ray, every ray crosses the box."""
    cfg = configs.tennis_single_player_config()
    comp = build(cfg)
    inputs = composer_inputs(cfg, synthetic.single_player_scene(image_size=(40, 40)))
    want, got = run_both(cfg, comp, inputs, export=True)
    assert_close(want, got)
    ev = int(got["coarse"]["_samples"][0]["evaluated"][0])
    assert ev > 0.9 * 40 * 40 * 32


def test_geometry_is_bit_identical():
    """Sample depths and AABB decisions feed discontinuities: they must equal the fp32 torch path exactly."""
    cfg = configs.minecraft_config()
    comp = build(cfg, alpha_bias=3.0)
    inputs = composer_inputs(cfg, synthetic.minecraft_scene(seed=31), pixels=grid_pixels(256, 256, 40))
    o, d, n, w2o, sty, dfm, ins = inputs
    with torch.no_grad():
        got = comp.cuda()(*[v.cuda() for v in inputs], False, _export=True)
    ex = got["coarse"]["_samples"][0]
    lay = ro.ObjectLayout(cfg)
    for k in range(lay.objects_count):
        m = cfg["model"]["object_models"][lay.model_of_object[k]]
        bbox = ro._bbox_tensor(m)
        oo, dd, _ = ro.transform_rays(o, d, n, w2o[..., k])
        near, far = ro.raywise_z_bounds(oo, dd, bbox, ins[..., k])
        near = near.clamp(m["z_near_min"], m["z_far_max"])
        far = far.clamp(m["z_near_min"], m["z_far_max"])
        x, t, _ = ro.stratified_positions(oo, dd, near, far, m["positions_count_coarse"], False)
        assert torch.equal(ex["t"][k].cpu().reshape(t.shape), t)
        inb = ro._in_box(x, bbox)
        assert torch.equal((ex["slot"][k].cpu() >= 0).reshape(inb.shape), inb)
        assert int(ex["evaluated"][k]) == int(inb.sum())
        # compact rows are a permutation-free enumeration in flat order
        slots = ex["slot"][k].cpu().reshape(-1)
        assert torch.equal(slots[slots >= 0], torch.arange(int(inb.sum()), dtype=torch.int32))


def test_mfma_kernel_agrees_with_scalar_kernel():
    cfg = configs.tennis_config()
    comp = build(cfg).cuda()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=41), pixels=grid_pixels(256, 256, 20))]
    with torch.no_grad():
        a = comp(*inputs, False)
        comp.use_naive_mlp = True
        b = comp(*inputs, False)
    assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_matches_reference_golden_vectors(path, precision):
    """Reference-generated fixtures (reduced widths exercise every padding path).  The reference's
    tie order is unspecified, so rays whose merged list contains an in-box sample inside a group of
    equal t are excluded from the global comparison (they are covered by the stable-merge oracle test)."""
    recipe, inputs, sd, noise, want, perturb = load_fixture(path)
    cfg = recipe_config(recipe)
    comp = ObjectComposer(cfg)
    comp.load_state_dict(sd, strict=True)
    comp.precision = precision
    comp = comp.eval().cuda()
    with torch.no_grad():
        got = comp(*[v.cuda() for v in inputs], perturb, _noise=noise if perturb else None)
        stable = ro.composer_forward(cfg, sd, *inputs, perturb, noise=noise, stable_merge=True)
    torch.cuda.synchronize()
    assert_close(stable, got)
    # rays where the stable and the reference order agree must match the reference itself.  (Agreement is judged with a
    # tolerance far below the effect of a swapped tie: the oracle is re-run on THIS host's CPU, whose vector units may sum
    # in another order than the build container's did when the fixture was recorded.)
    for ty in want:
        a, b = stable[ty]["global"], want[ty]["global"]
        same = torch.isclose(a["opacity"], b["opacity"], rtol=0, atol=2e-6) & \
            torch.isclose(a["integrated_features"], b["integrated_features"], rtol=1e-5, atol=2e-6).all(-1)
        assert same.float().mean() > 0.8
        for key in ("integrated_features", "opacity", "depth"):
            a, b = want[ty]["global"][key], got[ty]["global"][key].cpu()
            m = same if a.dim() == same.dim() else same.unsqueeze(-1).expand_as(a)
            assert torch.allclose(a[m], b[m], rtol=RTOL, atol=ATOL, equal_nan=True), (ty, key)
        for k in range(ro.ObjectLayout(cfg).objects_count):
            rep = compare_results(want[ty][f"object_{k}"], got[ty][f"object_{k}"], RTOL, ATOL)
            assert all(v[1] for v in rep.values()), rep


def test_canonical_pose_zeroes_displacements():
    cfg = configs.tennis_config()
    comp = build(cfg)
    inputs = composer_inputs(cfg, synthetic.tennis_scene(seed=51), pixels=grid_pixels(256, 256, 20))
    want, got = run_both(cfg, comp, inputs, canonical=True)
    assert_close(want, got)
    assert float(got["coarse"]["global"]["integrated_displacements_magnitude"].abs().max()) == 0.0
    _, bent = run_both(cfg, comp, inputs, canonical=False)
    assert float(bent["coarse"]["object_2"]["integrated_displacements_magnitude"].abs().max()) > 0.0


def test_absent_object_contributes_nothing():
    cfg = configs.tennis_config()
    comp = build(cfg)
    scene = synthetic.tennis_scene(seed=52)
    scene["object_in_scene"][..., 2] = False
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 20))
    want, got = run_both(cfg, comp, inputs)
    assert_close(want, got)
    assert float(got["coarse"]["object_2"]["opacity"].abs().max()) == 0.0


@pytest.mark.parametrize("which", ["shipped", "c2_64_128"])
def test_rays_are_independent_full_size_properties(which):
    """Size-independent properties at the benchmark's ray count (256x256), for the shipped tennis configuration and for
    BASELINE.json configs[1] (64 + 128 hierarchical positions per object, coarse + fine networks): rendering a ray subset
    reproduces the corresponding slice bit for bit, so does the ray-chunked path; opacity = sum(weights) <= 1,
    weights >= 0."""
    cfg = configs.tennis_config() if which == "shipped" else configs.tennis_config(hierarchical=(64, 128))
    comp = build(cfg).cuda()
    scene = synthetic.tennis_scene(seed=1234)
    o, d, n, w2o, sty, dfm, ins = [v.cuda() for v in composer_inputs(cfg, scene)]
    with torch.no_grad():
        full = comp(o, d, n, w2o, sty, dfm, ins, False)
        idx = torch.arange(0, d.size(-2), 97, device="cuda")
        part = comp(o, d[..., idx, :], n, w2o, sty, dfm, ins, False)
        comp.max_workspace_bytes = (256 << 20) if which == "shipped" else (4 << 30)   # force the ray-chunked path
        chunked = comp(o, d, n, w2o, sty, dfm, ins, False)
    assert d.size(-2) == 65536
    types = ["coarse"] if which == "shipped" else ["coarse", "fine"]
    assert [t for t in ("coarse", "fine") if t in full] == types
    for ty in types:
        names = ["global"] + ([f"object_{k}" for k in range(4)] if ty == "fine" else [])
        for name in names:
            g, gp, gc = full[ty][name], part[ty][name], chunked[ty][name]
            for key in ("integrated_features", "opacity", "depth", "weights"):
                assert torch.equal(torch.nan_to_num(g[key][..., idx, :] if g[key].dim() > 4 else g[key][..., idx]),
                                   torch.nan_to_num(gp[key])), (ty, name, key)
                assert torch.equal(torch.nan_to_num(g[key]), torch.nan_to_num(gc[key])), (ty, name, key)
            w = g["weights"]
            assert float(w.min()) >= 0.0
            assert torch.allclose(w.sum(-1), g["opacity"], rtol=1e-5, atol=1e-6)
            assert float(g["opacity"].max()) <= 1.0 + 1e-5
            assert torch.isfinite(g["integrated_features"]).all()
    if which == "c2_64_128":
        assert full["coarse"]["global"]["weights"].shape[-1] == 256 and full["fine"]["global"]["weights"].shape[-1] == 768
        # the merged fine depths are sorted along the ray, so the weights of the global entry and of the objects carry
        # the same mass wherever only one object is hit: sum over objects of the per-object opacity >= global opacity
        per_object = sum(full["fine"][f"object_{k}"]["opacity"] for k in range(4))
        assert bool((per_object + 1e-5 >= full["fine"]["global"]["opacity"]).all())


def test_camera_rays_kernel_is_bit_identical():
    cfg = configs.minecraft_config()
    scene = synthetic.minecraft_scene(batch=2, seed=61, image_size=(64, 96))
    rows, cols = ro.strided_grid_pixels(64, 96, [4, 8])
    o, d, n = ro.world_rays_from_cameras(cfg, scene["camera_rotations"], scene["camera_translations"], scene["focals"],
                                         scene["image_size"], rows, cols)
    c2w = ro.euler_to_matrix(scene["camera_rotations"], scene["camera_translations"]).cuda()
    f = (scene["focals"] * cfg["data"]["focal_length_multiplier"]).cuda()
    go, gd, gn = em.camera_rays(c2w, f, 64, 96, rows, cols)
    assert torch.equal(go.cpu(), o) and torch.equal(gd.cpu(), d) and torch.equal(gn.cpu(), n)


def test_environment_model_scene_encoding_modes():
    """EnvironmentModel.forward(mode='scene_encodings') and render_full_frame_from_scene_encoding:
    result schema of the reference plus parity with the oracle.  Matrices are built on the GPU here
    (ulp-level differences in sin/cos and the 4x4 inverse), so a handful of rays may flip an AABB
    decision: at most 0.5% of the rays may exceed the tolerance."""
    cfg = configs.minecraft_config()
    model = em.EnvironmentModel(cfg)
    torch.manual_seed(0)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    model.eval()
    scene = synthetic.minecraft_scene(batch=1, observations=2, seed=71, image_size=(48, 64))
    sd = {k: v.clone() for k, v in model.object_composer.state_dict().items()}
    args = [scene[k] for k in ("camera_rotations", "camera_translations", "focals")] + [scene["image_size"]] + \
           [scene[k] for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                               "object_deformation", "object_in_scene")]
    with torch.no_grad():
        want = ro.render_from_scene_encoding(cfg, sd, *args, strides=[4, 8])
        model = model.cuda()
        gargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
        got = model(*gargs, 0, False, 1000, patch_stride=[4, 8], mode="scene_encodings")
        full = model.render_full_frame_from_scene_encoding(*gargs, False)
    assert {"coarse", "object_rotation_parameters", "object_translation_parameters", "reconstructed_bounding_boxes",
            "reconstructed_3d_bounding_boxes", "projected_axes", "scene_encoding"} == set(got)
    assert "pytorch_hook" not in got
    a = want["coarse"]["global"]["integrated_features"]
    b = got["coarse"]["global"]["integrated_features"].cpu()
    assert a.shape == b.shape == (1, 2, 1, 12 * 16 + 6 * 8, 192)
    bad = ((a - b).abs() > ATOL + RTOL * a.abs()).any(-1).float().mean()
    assert float(bad) <= 0.005, float(bad)
    assert tuple(full["coarse"]["global"]["integrated_features"].shape) == (1, 2, 1, 48, 64, 192)
    assert tuple(got["reconstructed_bounding_boxes"].shape) == (1, 2, 1, 4, 4)
    assert tuple(got["reconstructed_3d_bounding_boxes"].shape) == (1, 2, 1, 68, 2, 4)
    assert tuple(got["projected_axes"].shape) == (1, 2, 1, 4, 2, 4)
    with pytest.raises(RuntimeError, match="inject"):      # the observation-driven modes need the (injected) encoders
        model(*gargs, 0, False, mode="observations")


def test_training_shaped_patch_call_config5_shape():
    """BASELINE.json configs[4] forward shape: minecraft, patch 48 @ strides [4, 8] -> 48^2 + 24^2 = 2880 rays per
    frame, perturb=True, 3 observations.  Per-frame pixel lists go through pr_camera_rays (bitwise check) and the
    composer result has the reference's shapes."""
    from playableenvironments_amd import ray_sampling as rs
    cfg = configs.minecraft_config()
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    model = model.eval().cuda()
    scene = synthetic.minecraft_scene(batch=2, observations=3, seed=81, image_size=(288, 512))
    args = [scene[k] for k in ("camera_rotations", "camera_translations", "focals")] + [scene["image_size"]] + \
           [scene[k] for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                               "object_deformation", "object_in_scene")]
    gargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    with torch.no_grad():
        torch.manual_seed(4)
        out = model(*gargs, 150, True, patch_size=48, patch_stride=[4, 8], mode="scene_encodings")
    feats = out["coarse"]["global"]["integrated_features"]
    assert tuple(feats.shape) == (2, 3, 1, 2880, 192) and torch.isfinite(feats).all()
    assert tuple(out["coarse"]["object_3"]["weights"].shape) == (2, 3, 1, 2880, 32)
    # per-frame pixel lists: the kernel must reproduce the oracle's gather of the full ray grid bit for bit
    boxes = out["reconstructed_bounding_boxes"].reshape(-1, 4, 4).cpu()
    torch.manual_seed(9)
    idx = rs.strided_patch_pixels(boxes, cfg["model"]["sampling_weights"], 288, 512, 48, [4, 8])
    rows, cols = rs.split_indices(idx.reshape(2, 3, 1, -1), 512)
    c2w = ro.euler_to_matrix(scene["camera_rotations"], scene["camera_translations"])
    f = scene["focals"] * cfg["data"]["focal_length_multiplier"]
    go, gd, gn = em.camera_rays(c2w.cuda(), f.cuda(), 288, 512, rows, cols)
    dirs, o, n = ro.create_camera_rays([2, 3, 1], 288, 512, f)
    dirs = dirs.reshape(6, 288 * 512, 3)[torch.arange(6).unsqueeze(1), idx].reshape(2, 3, 1, -1, 3)
    wo, wd, wn = ro.transform_rays(o, dirs, n, c2w)
    assert torch.equal(gd.cpu(), wd) and torch.equal(go.cpu(), wo)


@pytest.mark.parametrize("name", ["tennis", "minecraft", "tennis_hierarchical"])
def test_train_mode_batchnorm_forward(name):
    """module.train(): both AdaIN BatchNorm layers normalise with the batch statistics of each object call
    and update running_mean / running_var / num_batches_tracked (momentum 0.1, unbiased variance) - compared
    with the oracle's torch batch_norm in training mode, with replayed perturbation noise."""
    make_cfg, make_scene, n, bias = CASES[name]
    cfg, scene = make_cfg(), make_scene()
    comp = build(cfg, alpha_bias=bias)
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    sd0 = {k: v.clone() for k, v in sd.items()}       # (the oracle updates the running statistics of `sd` in place)
    rec = {}
    with torch.no_grad():
        torch.manual_seed(5)
        want = ro.composer_forward(cfg, sd, *inputs, True, training=True, update_stats=True, record_noise=rec,
                                   stable_merge=True)
        comp = comp.cuda().train()
        got = comp(*[v.cuda() for v in inputs], True, _noise=rec)
    torch.cuda.synchronize()
    # Batch-normalising low-variance channels is ill-conditioned in fp32: perturbing the weights by one ulp moves
    # the ORACLE's train-mode features by 2e-5..3e-5 (1e-7 in eval mode; measured): everything that does
    # not pass through BatchNorm keeps 1e-4 / 1e-5 ...
    rep = compare_results(want, got, rtol=RTOL, atol=ATOL)
    bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1] and not k.endswith("integrated_features")}
    assert not bad, bad
    # ... and the features are ARBITRATED in float64: no farther from the exact result than the fp32 oracle is (x 4)
    exact = run_exact(cfg, sd0, inputs, True, rec, training=True)
    assert_no_farther_than_the_oracle(exact, want, got, "integrated_features")
    after = comp.state_dict()
    checked = 0
    for key, val in sd.items():
        if "ada_in.normalization" in key:
            a, b = val.float(), after[key].cpu().float()
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), (key, (a - b).abs().max())
            checked += 1
    assert checked >= 12
    comp.eval()


def test_train_mode_ray_chunks_normalise_per_chunk():
    """``samples_per_image_batching`` in TRAINING mode (environment_model.py:474-521): the reference evaluates its composer once
    per chunk of rays, so every chunk is normalised with its own batch statistics and the running statistics and
    ``num_batches_tracked`` advance once per chunk.  HIP renderer behind the product's ``batchified_composer_call`` against the
    oracle's chunked call (pinned against the reference's in check_against_reference.py): features at the train-mode tolerance,
    buffers to 1e-4, counters exactly - and different from the unchunked call, which is what evaluation mode may ignore."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    scene = synthetic.minecraft_scene(batch=2, seed=31, image_size=(48, 64))
    inputs = composer_inputs(cfg, scene, strides=[4, 8])                      # 240 rays
    comp = build(cfg, alpha_bias=3.0)
    sd = {k: v.detach().clone() for k, v in comp.state_dict().items()}
    with torch.no_grad():
        want = ro.batchified_composer_call(cfg, sd, *inputs, False, chunk=100, training=True)     # updates sd in place
    model = em.EnvironmentModel(cfg)
    model.object_composer.load_state_dict(comp.state_dict())
    model = model.cuda().train()
    with torch.no_grad():
        got = model.batchified_composer_call(*[v.cuda() for v in inputs], False, samples_per_image_batching=100)
    torch.cuda.synchronize()
    rep = compare_results(want, got, rtol=1e-3, atol=2e-4)
    bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1]}
    assert not bad, bad
    after = model.object_composer.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            assert int(after[k]) == int(v) and int(v) in (3, 6), (k, int(after[k]), int(v))
        elif "running_" in k:
            assert torch.allclose(after[k].cpu(), v, rtol=1e-4, atol=1e-6), k
    # one call over all rays is a different computation in training mode (one set of statistics, one counter step)
    model.object_composer.load_state_dict(comp.state_dict())
    with torch.no_grad():
        whole = model.batchified_composer_call(*[v.cuda() for v in inputs], False, samples_per_image_batching=0)
    counters = [int(v) for k, v in model.object_composer.state_dict().items() if k.endswith("num_batches_tracked")]
    assert set(counters) <= {1, 2}
    assert not torch.allclose(whole["coarse"]["global"]["integrated_features"], got["coarse"]["global"]["integrated_features"], rtol=1e-3, atol=2e-4)
    # evaluation mode: the chunk size changes nothing
    model.eval()
    with torch.no_grad():
        a = model.batchified_composer_call(*[v.cuda() for v in inputs], False, samples_per_image_batching=100)
        b = model.batchified_composer_call(*[v.cuda() for v in inputs], False, samples_per_image_batching=0)
    assert torch.equal(a["coarse"]["global"]["integrated_features"], b["coarse"]["global"]["integrated_features"])


def test_train_mode_guards():
    cfg = configs.tennis_config()
    comp = build(cfg).cuda()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=2), pixels=grid_pixels(256, 256, 8))]
    out = comp(*inputs, False)         # eval mode with gradients enabled: a differentiable call (frozen BatchNorm)
    assert out["coarse"]["global"]["integrated_features"].requires_grad
    comp.max_workspace_bytes = 1 << 20
    with pytest.raises(RuntimeError, match="activations"):
        comp(*inputs, False)           # differentiable calls are never split: the saved activations must fit
    del comp.max_workspace_bytes
    comp.train()
    # a camera that sees nothing -> no evaluated sample: torch's BatchNorm accepts an EMPTY batch (it raises for exactly one
    # value per channel only - torch.nn.functional._verify_batch_size), so the reference's call goes through with its running
    # statistics untouched and num_batches_tracked incremented (oracle/check_dropin.py met this with the reference's trainer patch)
    scene = synthetic.tennis_scene(seed=2)
    scene["camera_rotations"][..., 0] = -1.4
    blind = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 4))]
    before = {k: v.clone() for k, v in comp.state_dict().items() if "running_" in k or "num_batches" in k}
    assert len(before) >= 20
    for check in ("eager", "deferred"):
        comp.batchnorm_check = check
        with torch.no_grad():
            out = comp(*blind, False)
        assert torch.isfinite(out["coarse"]["global"]["integrated_features"]).all()
        assert int(comp.last_normalised_samples["coarse"].sum()) == 0
    after = comp.state_dict()
    for k, v in before.items():
        if "num_batches" in k:
            assert int(after[k]) == int(v) + 2, k
        else:
            assert torch.equal(after[k], v), k
    # exactly one normalised sample (a whole object call of one sample: not reachable with >= 2 positions per ray) is what
    # raises; the deferred check raises it at the next call of the composer, after which the composer is usable again
    K = len(cfg["model"]["object_models"])
    event = torch.cuda.Event()
    event.record()
    comp._pending_bn_check = (torch.tensor([7, 1, 5, 0][:K] + [9] * max(0, K - 4)), event, ["coarse"], K)
    with torch.no_grad():
        with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
            comp(*inputs, False)
        out = comp(*inputs, False)
        assert torch.isfinite(out["coarse"]["global"]["integrated_features"]).all()
    torch.cuda.synchronize()
    comp._raise_pending_batchnorm_check()      # the last call saw samples for every object: nothing pending to raise


# --------------------------------------------------------------------------------------------
# Backward pass (pr_render_backward) against the oracle's autograd.  The probed outputs include the compositing
# weights themselves (per object, and the merged list of the global entry).
# --------------------------------------------------------------------------------------------
GRAD_KEYS = ("integrated_features", "opacity", "depth", "integrated_displacements_magnitude")
SMALL_NETS = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1,
                  bender_octaves=3)


def _probe_loss(results, probes, K):
    total = 0.0
    for (ty, name, key), probe in probes.items():
        t = results[ty][name][key]
        total = total + (t * probe.to(t.device)).sum()
    return total


class ForwardFieldMismatch(AssertionError):
    """A forward field of a differentiable call beyond the tolerance AND farther from float64 than the fp32 oracle (x 4)."""
    def __init__(self, fields):
        super().__init__(fields)
        self.fields = fields


#: "a sample within rounding of a kink of the ray bender's Jacobian" (tests/helpers.bender_kink_margin: the smallest distance of a
#: pre-activation / raw displacement from its kink, relative to the layer's scale): summation order moves a pre-activation by ~1e-7 of
#: the layer's scale.  ONE margin for both precisions, from the measured settlements (every settlement is logged through SETTLEMENTS):
#: in the -m gpu suite and the seed-0 sweep slices none occurs; in the 80-case backward sweeps of seed 7 (profiles/r05_sweep_backward_80_
#: seed7*.log) exactly one case settles, the same one in fp32 and f16x3, 1 ray of 144, margin 6.2e-9 - the split-precision forward needs
#: no wider margin than fp32 (round 4 allowed it 1e-6 on an argument about 22-bit operands; nothing measured ever used it).
KINK_MARGIN = {"fp32": 2e-7, "f16x3": 2e-7}

#: every forward-field excess of a differentiable call that `_gradients` SETTLED instead of failing on (float64 arbitration, a
#: divergence kink), with its numbers: drained and asserted on by the tests (the harness must not classify its excesses silently)
SETTLEMENTS = []


def drain_settlements():
    out = list(SETTLEMENTS)
    SETTLEMENTS.clear()
    return out


def _gradients(cfg, scene, n, bias, perturb, canonical=False, frozen=(), keys=GRAD_KEYS, training=True, rays=False,
               min_divergence=1e-2, absent=None, exact=False, noise_seed=123, precision="fp32"):
    """(oracle autograd, HIP backward) gradients of a random linear functional of the output fields ``keys``; ``rays``: also
    with respect to the camera rays (ray_origins, ray_directions).  ``exact``: a third entry per tensor - the oracle's
    autograd in float64 on the same weights, inputs and replayed noise (the arbiter of ill-conditioned cases)."""
    # precision "f16x3": the split-precision BACKWARD (bf16 triples, PR_FLAG_SPLIT_BACKWARD) behind the exact fp32 forward
    comp = build(cfg, alpha_bias=bias, precision=precision).train(training)
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    o, d, nrm, w2o, sty, dfm, ins = inputs
    if absent is not None:           # (object index, frame index or None = every frame) marked absent
        ins = ins.clone()
        flat = ins.reshape(-1, ins.size(-1))
        if absent[1] is None:
            flat[:, absent[0]] = False
        else:
            flat[absent[1] % flat.size(0), absent[0]] = False
    K = w2o.size(-1)
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    sd64 = to_double(sd) if exact else None
    names = [k for k, _ in comp.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    ref_in = [t.clone().requires_grad_(True) for t in (w2o, sty, dfm)]
    ref_rays = [t.clone().requires_grad_(rays) for t in (o, d)]
    rec = {}
    torch.manual_seed(noise_seed)
    want = ro.composer_forward(cfg, sd, *ref_rays, nrm, *ref_in, ins, perturb, canonical_pose=canonical, training=training,
                               record_noise=rec, stable_merge=True)
    gen = torch.Generator().manual_seed(7)
    probes = {(ty, nm, key): torch.randn(want[ty][nm][key].shape, generator=gen)
              for ty in ("coarse", "fine") if ty in want
              for nm in [f"object_{k}" for k in range(K)] + ["global"] for key in keys}
    _probe_loss(want, probes, K).backward()
    exact_grads = {}
    if exact:
        with oracle_in_float64():
            for k in names:
                sd64[k].requires_grad_(True)
            in64 = [t.double().clone().requires_grad_(True) for t in (w2o, sty, dfm)]
            rays64 = [t.double().clone().requires_grad_(rays) for t in (o, d)]
            want64 = ro.composer_forward(cfg, sd64, *rays64, nrm.double(), *in64, ins, perturb, canonical_pose=canonical,
                                         training=training, noise=to_double(rec), update_stats=False, stable_merge=True)
            _probe_loss(want64, probes, K).backward()
        exact_grads = {k: sd64[k].grad for k in names}
        exact_grads.update(zip(("w2o", "style", "deformation"), (t.grad for t in in64)))
        if rays:
            exact_grads.update(zip(("ray_origins", "ray_directions"), (t.grad for t in rays64)))
    comp = comp.cuda()
    for k, p in comp.named_parameters():
        if any(f in k for f in frozen):
            p.requires_grad_(False)
    hip_in = [t.clone().cuda().requires_grad_(True) for t in (w2o, sty, dfm)]
    hip_rays = [t.clone().cuda().requires_grad_(rays) for t in (o, d)]
    got = comp(*hip_rays, nrm.cuda(), *hip_in, ins.cuda(), perturb, canonical_pose=canonical, _noise=rec)
    _probe_loss(got, probes, K).backward()
    torch.cuda.synchronize()
    # forward fields of the differentiable call, incl. the Hutchinson divergence (replayed probes)
    fields = {ty: want[ty] for ty in ("coarse", "fine") if ty in want}
    rep = compare_results(fields, {ty: got[ty] for ty in fields}, rtol=1e-3, atol=2e-4)
    bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1]}
    settled = set()
    if bad:
        # a field beyond the tolerance is ARBITRATED, not waved through: the oracle's op graph in float64 on the same weights, inputs
        # and replayed noise is the exact result, and the HIP field may be as far from it as the fp32 oracle's is (x 4) - the
        # Hutchinson estimate of a narrow ray bender cancels to a small fraction of its terms (randomized sweep, seed 7 case 0)
        with oracle_in_float64():
            exact_fwd = ro.composer_forward(cfg, to_double({k: v.detach() for k, v in sd.items()}), *[t.detach().double() for t in ref_rays],
                                            nrm.double(), *[t.detach().double() for t in ref_in], ins, perturb, canonical_pose=canonical,
                                            training=training, noise=to_double(rec), update_stats=False, stable_merge=True)
        verdicts = arbitrate({ty: exact_fwd[ty] for ty in fields}, fields, {ty: got[ty] for ty in fields})
        settled = {k for k in bad if verdicts[k][2]}
        if settled:
            SETTLEMENTS.append({"kind": "float64 arbitration", "precision": precision, "fields": sorted(settled),
                                "hip_minus_fp64": max(verdicts[k][0] for k in settled), "oracle_minus_fp64": max(verdicts[k][1] for k in settled)})
        bad = {k: f"{v} (HIP - fp64 {verdicts[k][0]:.3e}, fp32 oracle - fp64 {verdicts[k][1]:.3e})" for k, v in bad.items() if k not in settled}
    if bad and all(k.endswith("integrated_divergence") for k in bad):
        # The Hutchinson estimate probes the ray benders' JACOBIAN, which is discontinuous where a raw displacement meets its clamp
        # bound or a hidden unit's pre-activation crosses 0: a sample within fp32 rounding of such a kink lands on either side
        # depending on the summation order, and its ray's estimate moves by a finite amount.  Accepted only if
        # (a) nothing but divergence fields differs, (b) at most 0.5 % of the rays (at least 2) are affected, (c) the oracle run
        # really has a sample within rounding of a kink (tests/helpers.bender_kink_margin; thresholds below).
        def rays_off(path):
            ty, name, key = path.split(".")
            a, b = fields[ty][name][key].detach().double(), got[ty][name][key].detach().cpu().double()
            return int(((a - b).abs() > 1e-3 * a.abs() + 2e-4).sum()), a.numel()
        counts = {k: rays_off(k) for k in bad}
        if all(n <= max(2, int(0.005 * total)) for n, total in counts.values()):
            def run_oracle():
                ro.composer_forward(cfg, {k: v.detach() for k, v in sd.items()}, *[t.detach() for t in ref_rays], nrm,
                                    *[t.detach() for t in ref_in], ins, perturb, canonical_pose=canonical, training=training,
                                    noise=rec, update_stats=False, stable_merge=True)
            margin = bender_kink_margin(run_oracle)
            # "within rounding": KINK_MARGIN (below), measured
            if margin < KINK_MARGIN[precision]:
                SETTLEMENTS.append({"kind": "divergence kink", "precision": precision, "fields": sorted(bad), "margin": margin,
                                    "rays_off": max(c for c, _ in counts.values()), "rays": max(t for _, t in counts.values())})
                settled |= set(bad)
                bad = {}
            else:
                bad = {k: f"{v}; rays off {counts[k]}, kink margin {margin:.1e}" for k, v in bad.items()}
        else:
            bad = {k: f"{v}; rays off {counts[k]}" for k, v in bad.items()}
    if bad:
        raise ForwardFieldMismatch(bad)
    if not canonical and training:
        a = want["coarse"]["global"]["integrated_divergence"].detach()
        b = got["coarse"]["global"]["integrated_divergence"].detach().cpu()
        # (min_divergence: the suite's scenes are built to have a sizeable estimate; the random sweep passes 0)
        # (a sweep case whose bender objects are all absent has an estimate of exactly zero: seed 24 case 177)
        assert min_divergence == 0.0 or float(a.abs().max()) > min_divergence
        assert "coarse.global.integrated_divergence" in settled or float((a - b).abs().max()) <= 1e-3 * float(a.abs().max()) + 2e-4
    params = dict(comp.named_parameters())
    ref = {k: sd[k].grad for k in names}
    hip = {k: params[k].grad for k in names}
    for label, a, b in zip(("w2o", "style", "deformation"), ref_in, hip_in):
        ref[label], hip[label] = a.grad, b.grad
    if rays:
        for label, a, b in zip(("ray_origins", "ray_directions"), ref_rays, hip_rays):
            ref[label], hip[label] = a.grad, b.grad
    out = {}
    for k in ref:
        b = hip[k].detach().cpu() if hip[k] is not None else None
        a = ref[k] if ref[k] is not None else torch.zeros_like(b)
        b = b if b is not None else torch.zeros_like(a)
        out[k] = (a, b, exact_grads[k] if exact_grads.get(k) is not None else torch.zeros_like(a, dtype=torch.float64)) if exact else (a, b)
    return out


HIER_POSITIONS = {"background": (8, 12), "background_backplate": (8, 12), "player_1": (12, 20), "player_2": (12, 20)}


@pytest.mark.parametrize("name,perturb", [("tennis", False), ("tennis", True), ("minecraft", False), ("minecraft", True),
                                          ("tennis_frames", True), ("tennis_hierarchical", False),
                                          ("tennis_hierarchical", True)])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_backward_matches_oracle_autograd(name, perturb, precision):
    """Every parameter gradient, d style, d deformation and d transformation_matrix_w2o against torch.autograd
    through the oracle (train mode, replayed noise), on shallow networks where the comparison is well conditioned:
    max |difference| <= 1e-4 * max |reference| per tensor."""
    if name == "minecraft":
        cfg, scene, n, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(), 16, 3.0
    elif name == "tennis_frames":
        cfg, scene, n, bias = (configs.reduced_config(configs.tennis_config(), **SMALL_NETS),
                               synthetic.tennis_scene(batch=2, observations=2, seed=3), 12, 2.0)
    elif name == "tennis_hierarchical":
        # coarse + fine models, resampled depths detached: gradients of both passes
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, n, bias = synthetic.tennis_scene(seed=5), 14, 2.0
    else:
        cfg, scene, n, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(), 16, 2.0
    grads = _gradients(cfg, scene, n, bias, perturb, precision=precision)
    assert len(grads) > 50
    # the fine pass places its samples by inverse-CDF resampling of the coarse weights, so fp32 round-off of the coarse
    # pass moves fine samples: perturbing the resampling variates by 1e-6 moves the ORACLE's own gradients by 6e-5 and
    # a 1e-6 relative weight perturbation by up to 6e-2 (a ReLU / AABB flip; measured) -> 1e-3 there, 1e-4 elsewhere
    tol = 1e-3 if name == "tennis_hierarchical" else 1e-4
    bad = {}
    nonzero = 0
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        nonzero += scale > 0
        err = float((a - b).abs().max())
        if err > tol * scale + 1e-9:
            bad[k] = (err, scale)
    assert not bad, bad
    assert nonzero > 40


@pytest.mark.parametrize("name,perturb", [("tennis", False), ("tennis_frames", True), ("minecraft", True), ("tennis_hierarchical", False)])
def test_backward_to_the_camera_rays(name, perturb):
    """d loss / d ray_origins and d ray_directions - what learnable camera parameters are trained through (the reference's rays
    are torch tensors with a graph: ray_helper.py:15-52, 1203-1227) - against torch.autograd through the oracle: sample
    positions o + d t, the slab-test depths, the skybox's [o / size, d / |d|] input and the spacings dt |d| of every entry
    all depend on the rays.  (The other gradients of the same call are compared as well: asking for the ray gradients must
    not disturb them.)"""
    if name == "minecraft":
        cfg, scene, n, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(), 16, 3.0
    elif name == "tennis_frames":
        cfg, scene, n, bias = (configs.reduced_config(configs.tennis_config(), **SMALL_NETS),
                               synthetic.tennis_scene(batch=2, observations=2, seed=3), 12, 2.0)
    elif name == "tennis_hierarchical":
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, n, bias = synthetic.tennis_scene(seed=5), 14, 2.0
    else:
        cfg, scene, n, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(), 16, 2.0
    grads = _gradients(cfg, scene, n, bias, perturb, rays=True)
    tol = 1e-3 if name == "tennis_hierarchical" else 2e-4
    bad = {}
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        err = float((a - b).abs().max())
        # the origin of a frame sums the contributions of all its rays and samples: compare at the scale of the summands
        if err > tol * (scale if k != "ray_origins" else max(scale, float(grads["ray_directions"][0].abs().max()))) + 1e-9:
            bad[k] = (err, scale)
    assert not bad, bad
    assert float(grads["ray_origins"][0].abs().max()) > 0 and float(grads["ray_directions"][0].abs().max()) > 0


def _rgb_config(world):
    """Models that output colours directly: 3 features, apply_activation (sigmoid on the raw features of EVERY sample before
    compositing, object_composer.py:548-549).  An empty-space density close to 0 lets the perturbation noise lift samples
    outside the boxes: they composite sigmoid(0) = 0.5."""
    base = configs.tennis_config() if world == "tennis" else configs.minecraft_config()
    cfg = configs.reduced_config(base, width=64, layers=4, skip=2, features=3, octaves=4, bender_width=32, bender_layers=3,
                                 bender_skip=1, bender_octaves=3)
    cfg["model"]["apply_activation"] = True
    for o in cfg["model"]["object_models"]:
        o["empty_space_alpha"] = -0.5
    return cfg


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_rgb_models_with_sigmoid_features(world):
    """config["model"]["apply_activation"] = True (the reference allows it for output_features == 3 only): forward fields of
    both kernels against the oracle, evaluation and perturbed, and every gradient of a training call (sigmoid derivative in
    the compositing backward, 3-wide feature rows in the head products) against oracle autograd."""
    cfg = _rgb_config(world)
    scene = (synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene)(seed=3)
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 20))
    for precision in ("fp32", "f16x3"):
        for perturb in (False, True):
            want, got = run_both(cfg, build(cfg, precision=precision), inputs, perturb=perturb)
            assert_close(want, got)
            feats = want["coarse"]["global"]["integrated_features"]
            assert feats.shape[-1] == 3 and 0.0 < float(feats.max()) <= 1.0 + 1e-6
    for perturb in (False, True):
        grads = _gradients(cfg, scene, 16, 2.0 if world == "tennis" else 3.0, perturb, rays=True)
        bad = {}
        for k, (a, b) in grads.items():
            scale = max(float(a.abs().max()), float(grads["ray_directions"][0].abs().max()) if k == "ray_origins" else 0.0)
            err = float((a - b).abs().max())
            if not (err <= 2e-4 * scale + 1e-9):
                bad[k] = (err, scale)
        assert not bad, bad
        assert float(grads["object_models_coarse.0.nerf_model.features_head.6.weight"][0].abs().max()) > 0


def test_backward_with_padded_widths_on_poisoned_scratch():
    """Widths that are not multiples of 32 (backbone 48 -> padded 64 with a 24-wide second head layer, ray bender 16 -> padded
    32): the padding columns of the backward scratch are written by nobody, and 0 x garbage must stay 0.  The allocator is
    poisoned with NaNs first, so that a read of unwritten scratch cannot pass by luck (found by tests/gpu_fuzz.py: the
    divergence tangents and the chain's entry gradient of a 16-wide bender read such columns)."""
    cfg = configs.reduced_config(configs.tennis_config(), width=48, layers=4, skip=2, features=32, octaves=4, bender_width=16,
                                 bender_layers=3, bender_skip=1, bender_octaves=3)
    for perturb in (False, True):
        poison_device_memory()
        grads = _gradients(cfg, synthetic.tennis_scene(seed=5), 16, 2.0, perturb, keys=GRAD_KEYS + ("integrated_divergence",),
                           rays=True)
        bad = {}
        for k, (a, b) in grads.items():
            scale = max(float(a.abs().max()), float(grads["ray_directions"][0].abs().max()) if k == "ray_origins" else 0.0)
            err = float((a - b).abs().max())
            if not (err <= 5e-4 * scale + 1e-9):        # (NaN-safe)
                bad[k] = (err, scale)
        assert not bad, bad


def test_absent_objects_are_evaluated_like_the_reference():
    """The reference runs an absent object's network on its samples (slab depths 0 -> clamped to z_near_min) and overrides
    the densities with empty_space_alpha AFTERWARDS (object_composer.py:522-547): with perturbation noise that density can
    turn positive, and the sample then composites real network features; in training mode those samples are part of the
    BatchNorm batch statistics.  Static objects are never absent in the reference's own pipeline and an absent player has
    no sample in its small box, so this is a corner of the interface - found by tests/gpu_fuzz.py (skybox marked absent).
    An empty_space_alpha close to 0 makes the noise cross it often."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    for o in cfg["model"]["object_models"]:
        o["empty_space_alpha"] = -0.3
    scene = synthetic.minecraft_scene(batch=2, seed=11)
    inputs = list(composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 14)))
    ins = inputs[6].clone()
    ins.reshape(-1, ins.size(-1))[:, 1] = False          # the skybox everywhere
    ins.reshape(-1, ins.size(-1))[0, 0] = False          # the background in the first frame
    inputs[6] = ins
    comp = build(cfg, alpha_bias=3.0)
    want, got = run_both(cfg, comp, inputs, perturb=True, export=True)
    assert_close(want, got)
    sky = want["coarse"]["object_1"]["integrated_features"]
    assert float(sky.abs().amax(-1).gt(0).float().mean()) > 0.2          # the absent skybox does reach its own composite
    assert int(got["coarse"]["_samples"][0]["evaluated"][1]) == sky[..., 0].numel()
    # training mode: the batch statistics include the absent frames' samples (forward fields and gradients against the oracle)
    grads = _gradients(cfg, scene, 14, 3.0, True, absent=(0, 0))
    bad = {k: (float((a - b).abs().max()), float(a.abs().max())) for k, (a, b) in grads.items()
           if float((a - b).abs().max()) > 2e-4 * float(a.abs().max()) + 1e-9}
    assert not bad, bad


@pytest.mark.parametrize("name,perturb,alone", [("tennis", False, True), ("tennis", True, False), ("minecraft", True, True),
                                                ("minecraft", False, False), ("tennis_hierarchical", False, False)])
def test_backward_of_the_divergence_estimate(name, perturb, alone):
    """Gradient of integrated_divergence - the reference's DOUBLE backward through the ray bender
    (compute_approximate_divergence with create_graph=True, object_composer.py:582-601; mean(alpha.detach() |div|), :768-769)
    - against torch.autograd through the oracle with the same probes.  ``alone``: the loss reads integrated_divergence
    only, so every non-zero gradient below comes from the second-order pass (ray bender weights, object poses; nothing
    reaches the NeRF backbones, the style or the deformation codes: the estimate is piecewise linear in them)."""
    if name == "minecraft":
        cfg, scene, n, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(), 16, 3.0
    elif name == "tennis_hierarchical":
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, n, bias = synthetic.tennis_scene(seed=5), 14, 2.0
    else:
        cfg, scene, n, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(), 16, 2.0
    keys = ("integrated_divergence",) if alone else GRAD_KEYS + ("integrated_divergence",)
    grads = _gradients(cfg, scene, n, bias, perturb, keys=keys)
    tol = 1e-3 if name == "tennis_hierarchical" else 2e-4
    bad, bender, elsewhere = {}, 0, 0
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        if "ray_bender" in k and scale > 0:
            bender += 1
        elif scale > 0 and k not in ("w2o",):
            elsewhere += 1
        err = float((a - b).abs().max())
        if err > tol * scale + 1e-9:
            bad[k] = (err, scale)
    assert not bad, bad
    assert bender >= 4 and float(grads["w2o"][0].abs().max()) > 0      # 3 layers + the output head of every bender model
    if alone:
        assert elsewhere == 0, [k for k, (a, _) in grads.items() if float(a.abs().max()) > 0 and "ray_bender" not in k]


@pytest.mark.parametrize("name,perturb", [("tennis", False), ("minecraft", True), ("tennis_hierarchical", False)])
def test_backward_of_the_compositing_weights(name, perturb):
    """pr_entry_grads_t.weights: a loss that reads the compositing weights themselves - per object, and the merged list of
    the global entry (overlap fix included for minecraft) - against the oracle's autograd.  Per tensor: 1e-4 of its own
    largest entry plus an fp32-epsilon floor relative to the largest gradient of the call (tiny tensors such as a
    single bias are sums with cancellation)."""
    if name == "minecraft":
        cfg, scene, n, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(), 16, 3.0
    elif name == "tennis_hierarchical":
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, n, bias = synthetic.tennis_scene(seed=5), 14, 2.0
    else:
        cfg, scene, n, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(), 16, 2.0
    grads = _gradients(cfg, scene, n, bias, perturb, keys=("weights", "opacity"))
    largest = max(float(a.abs().max()) for a, _ in grads.values())
    tol = 1e-3 if name == "tennis_hierarchical" else 1e-4
    bad, nonzero = {}, 0
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        nonzero += scale > 0
        err = float((a - b).abs().max())
        if err > tol * scale + 2e-7 * largest:
            bad[k] = (err, scale)
    assert not bad, bad
    assert nonzero > 20


@pytest.mark.parametrize("name,perturb", [("tennis", False), ("minecraft", True), ("tennis_hierarchical", False)])
def test_backward_in_eval_mode_with_frozen_batchnorm(name, perturb):
    """module.eval() with gradients enabled (test-time optimisation of poses / style codes / weights): the BatchNorm
    layers normalise with their running statistics, which are constants of the graph - against the oracle's autograd
    with training=False.  The running statistics must not move."""
    if name == "minecraft":
        cfg, scene, n, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(), 16, 3.0
    elif name == "tennis_hierarchical":
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, n, bias = synthetic.tennis_scene(seed=5), 14, 2.0
    else:
        cfg, scene, n, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(), 16, 2.0
    grads = _gradients(cfg, scene, n, bias, perturb, training=False)
    largest = max(float(a.abs().max()) for a, _ in grads.values())
    tol = 1e-3 if name == "tennis_hierarchical" else 1e-4
    bad, nonzero = {}, 0
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        nonzero += scale > 0
        err = float((a - b).abs().max())
        if err > tol * scale + 2e-7 * largest:
            bad[k] = (err, scale)
    assert not bad, bad
    assert nonzero > 40


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_backward_full_size_networks(precision):
    """Shipped network sizes (8 x 256 backbone, 6 x 128 bender, F = 192, two frames).  Deep ReLU / BatchNorm stacks on
    a few hundred samples are ill-conditioned - the ORACLE's own gradients move by up to 6e-3 (relative, max norm)
    when its weights are perturbed by one ulp (measured) - so the case is ARBITRATED in float64: per
    gradient tensor, max |HIP - fp64| <= 4 x max |fp32 oracle autograd - fp64| (+ 1e-6 of the tensor's largest entry)."""
    cfg = configs.minecraft_config()
    grads = _gradients(cfg, synthetic.minecraft_scene(batch=2, seed=8), 14, 3.0, True, exact=True, precision=precision)
    bad, worst = {}, 0.0
    for k, (a, b, e) in grads.items():
        err_hip, err_ref = float((b.double() - e).abs().max()), float((a.double() - e).abs().max())
        scale = float(e.abs().max())
        if scale == 0.0 and err_hip == 0.0:
            continue
        worst = max(worst, err_hip / max(err_ref, 1e-6 * scale, 1e-300))
        if not err_hip <= 4.0 * err_ref + 1e-6 * scale:
            bad[k] = (err_hip, err_ref, scale)
    assert not bad, (worst, bad)
    assert len(grads) > 80


@pytest.mark.parametrize("profile", ["rising", "falling", "gaps", "vanishing"])
def test_split_weight_gradients_follow_the_rows_scales(profile):
    """The split-precision weight gradients (k_gemm_tn_all_f16) scale every 16-row half slab of their operands by powers of
    two that follow the rows' magnitudes, and multiply their accumulators down when a later half slab raises the running scale.  A loss
    whose per-ray weight spans fifteen orders of magnitude along the rays (rising: the running scale is raised again and again; falling:
    later half slabs sink below fp16's range and must degrade to nothing, not to garbage; gaps: stretches of rays without any gradient
    between them - all-zero half slabs; vanishing: every gradient around 1e-35, where a power of two that lifts a tile into fp16's range
    does not exist in fp32 - the scales are capped, 0 x inf never happens) against the exact-fp32 kernels on the same call: every gradient within 5e-4 of its tensor's
    largest entry (the sweeps' bar), nothing non-finite."""
    cfg = configs.reduced_config(configs.minecraft_config(), width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32,
                                 bender_layers=3, bender_skip=1, bender_octaves=3)
    scene = synthetic.minecraft_scene(batch=1, seed=21)
    n = 48                                                 # 2 304 rays: several 2 048-row splits per object
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))]
    rays = n * n
    ramp = torch.linspace(-9.0, 6.0, rays)
    if profile == "falling":
        ramp = ramp.flip(0)
    weight = torch.pow(torch.tensor(10.0), ramp)
    if profile == "gaps":
        weight = weight * ((torch.arange(rays) // 97) % 3 != 1).float()
    if profile == "vanishing":
        weight = torch.full((rays,), 1e-33)
    weight = weight.cuda()
    grads = {}
    for precision in ("fp32", "f16x3"):
        comp = build(cfg, alpha_bias=2.0, precision=precision).cuda().train()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = comp(*inputs, False)
        feat = out["coarse"]["global"]["integrated_features"]
        assert feat.shape[-2] == rays, feat.shape
        loss = (feat.square().sum(-1).reshape(-1) * weight).sum()
        loss.backward()
        torch.cuda.synchronize()
        grads[precision] = {k: p.grad.detach().clone() for k, p in comp.named_parameters() if p.grad is not None}
    assert len(grads["fp32"]) == len(grads["f16x3"]) > 20
    bad = {}
    for k, exact in grads["fp32"].items():
        split = grads["f16x3"][k]
        assert bool(torch.isfinite(split).all()), k
        scale = float(exact.abs().max())
        err = float((exact - split).abs().max())
        if err > 5e-4 * scale + 1e-44:
            bad[k] = (err, scale)
    assert not bad, bad


def test_backward_absent_object_and_frozen_parameters():
    """An object that is absent from one frame contributes no samples there; frozen parameters get no gradient buffer
    (NULL in pr_model_grads_t) while everything else still matches the oracle."""
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    scene = synthetic.tennis_scene(batch=2, seed=9)
    scene["object_in_scene"][0, ..., 2] = False
    grads = _gradients(cfg, scene, 12, 2.0, True, frozen=("object_models_coarse.1.", "ray_bender.backbone_layers.0"))
    frozen = 0
    for k, (a, b) in grads.items():
        if "object_models_coarse.1." in k or "ray_bender.backbone_layers.0" in k:
            assert float(b.abs().max()) == 0.0, k          # no gradient was produced for a frozen parameter
            frozen += 1
            continue
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-9, (k, float((a - b).abs().max()), scale)
    assert frozen > 10


def test_backward_canonical_pose_and_unused_outputs():
    """canonical_pose zeroes the displacement gradients; a loss that only reads global.integrated_features (the
    shipped training configuration) gives the same gradients as the oracle."""
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    grads = _gradients(cfg, synthetic.tennis_scene(seed=4), 12, 2.0, False, canonical=True)
    for k, (a, b) in grads.items():
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-9, k
        if "ray_bender" in k or k == "deformation":
            assert float(b.abs().max()) == 0.0, k


from tests.test_cpu import GRAD_GOLDEN, load_gradient_fixture, probe_loss  # noqa: E402


@pytest.mark.parametrize("path", GRAD_GOLDEN, ids=[os.path.basename(p)[:-4] for p in GRAD_GOLDEN])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_backward_matches_reference_gradient_fixtures(path, precision):
    """HIP forward + backward against gradients recorded from the reference's own autograd (tests/golden/grads):
    max |difference| <= 1e-4 * max |reference| per tensor."""
    recipe, inputs, sd, noise, want, perturb, probes, grads = load_gradient_fixture(path)
    cfg = recipe_config(recipe)
    comp = ObjectComposer(cfg)
    comp.load_state_dict(sd, strict=True)
    comp = comp.cuda().train()
    comp.precision = precision          # "f16x3": the split-precision backward products behind the fp32 forward
    leaf = [inputs[i].clone().cuda().requires_grad_(True) for i in (3, 4, 5)]
    got = comp(*[v.cuda() for v in inputs[:3]], *leaf, inputs[6].cuda(), perturb, _noise=noise if perturb else None)
    probe_loss(got, probes).backward()
    torch.cuda.synchronize()
    params = dict(comp.named_parameters())
    bad = {}
    for k, a in grads.items():
        if k in ("w2o", "style", "deformation"):
            b = leaf[("w2o", "style", "deformation").index(k)].grad
        else:
            b = params[k].grad
        b = b.detach().cpu() if b is not None else torch.zeros_like(a)
        err, scale = float((a - b).abs().max()), float(a.abs().max())
        if err > 1e-4 * scale + 1e-9:
            bad[k] = (err, scale)
    assert not bad, bad


def test_environment_model_training_step_gradients_reach_the_poses():
    """EnvironmentModel.forward(mode="scene_encodings") in training mode: the renderer's d loss / d w2o flows through
    the torch glue (4x4 inverse, Euler matrices) into the object pose parameters, and an optimiser step on the composer
    runs.  Pose / style gradients against the oracle's autograd on the same scene encoding (shallow networks)."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    model = em.EnvironmentModel(cfg)
    torch.manual_seed(0)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    model.train()
    scene = synthetic.minecraft_scene(batch=1, observations=2, seed=71, image_size=(48, 64))
    sd = {k: v.clone() for k, v in model.object_composer.state_dict().items()}
    keys = ("camera_rotations", "camera_translations", "focals", "object_rotation_parameters",
            "object_translation_parameters", "object_style", "object_deformation", "object_in_scene")

    def leaves(device):
        out = {k: scene[k].clone().to(device) for k in keys}
        for k in ("object_rotation_parameters", "object_translation_parameters", "object_style"):
            out[k].requires_grad_(True)
        return out

    ref = leaves("cpu")
    want = ro.render_from_scene_encoding(cfg, sd, ref["camera_rotations"], ref["camera_translations"], ref["focals"],
                                         scene["image_size"], ref["object_rotation_parameters"],
                                         ref["object_translation_parameters"], ref["object_style"], ref["object_deformation"],
                                         ref["object_in_scene"], strides=[4, 8], training=True)
    gen = torch.Generator().manual_seed(3)
    probe = torch.randn(want["coarse"]["global"]["integrated_features"].shape, generator=gen)
    (want["coarse"]["global"]["integrated_features"] * probe).sum().backward()

    model = model.cuda()
    opt = torch.optim.SGD(model.object_composer.parameters(), lr=1e-6)
    hip = leaves("cuda")
    got = model(hip["camera_rotations"], hip["camera_translations"], hip["focals"], scene["image_size"],
                hip["object_rotation_parameters"], hip["object_translation_parameters"], hip["object_style"],
                hip["object_deformation"], hip["object_in_scene"], 0, False, 1000, patch_stride=[4, 8], mode="scene_encodings")
    (got["coarse"]["global"]["integrated_features"] * probe.cuda()).sum().backward()
    opt.step()
    torch.cuda.synchronize()
    for k in ("object_rotation_parameters", "object_translation_parameters", "object_style"):
        a, b = ref[k].grad, hip[k].grad.cpu()
        assert float(a.abs().max()) > 0
        # the pose matrices are built on the GPU here (ulp-level differences feed discontinuous AABB decisions)
        assert float((a - b).norm()) <= 2e-2 * float(a.norm()), (k, float((a - b).norm()), float(a.norm()))


@pytest.mark.parametrize("case", ["tennis_player", "tennis_player_perturb", "hierarchical_perturb", "minecraft_background",
                                  "minecraft_player_canonical"])
def test_forward_expected_positions_matches_oracle(case):
    """ObjectComposer.forward_expected_positions (object_composer.py:624-722) for one object instance: expected bent
    surface point and opacity per ray against the oracle (pinned bitwise against the reference), replayed noise."""
    perturb = case.endswith("perturb")
    canonical = case.endswith("canonical")
    if case.startswith("hierarchical"):
        cfg, scene, obj, bias = configs.tennis_config(hierarchical=(16, 32)), synthetic.tennis_scene(seed=19), 3, 2.0
    elif case.startswith("minecraft"):
        cfg, scene, bias = configs.minecraft_config(), synthetic.minecraft_scene(seed=21), 3.0
        obj = 0 if "background" in case else 2
    else:
        cfg, scene, obj, bias = configs.tennis_config(), synthetic.tennis_scene(seed=17), 2, 2.0
    comp = build(cfg, alpha_bias=bias)
    o, d, n, w2o, sty, dfm, ins = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 20))
    args = (o, d, n, w2o[..., obj], sty[..., obj], dfm[..., obj], ins[..., obj])
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    rec = {}
    with torch.no_grad():
        torch.manual_seed(11)
        want = ro.expected_positions_forward(cfg, sd, *args, obj, perturb, canonical_pose=canonical, record_noise=rec)
        comp = comp.cuda()
        got = comp.forward_expected_positions(*[a.cuda() for a in args], obj, perturb, canonical_pose=canonical,
                                              _noise=rec if perturb else None)
    torch.cuda.synchronize()
    assert set(got) == set(want)
    for ty in want:
        for a, b, what in zip(want[ty], got[ty], ("expected_positions", "opacity")):
            b = b.cpu()
            assert a.shape == b.shape
            assert torch.allclose(a, b, rtol=RTOL, atol=1e-4 if what == "expected_positions" else ATOL), \
                (ty, what, float((a - b).abs().max()))
        assert float(want[ty][1].max()) > 0.1   # the object is actually hit


@pytest.mark.parametrize("case", ["tennis_player", "tennis_player_perturb", "minecraft_player", "tennis_hierarchical"])
def test_expected_positions_backward_matches_oracle_autograd(case):
    """Differentiable forward_expected_positions (pose / keypoint consistency losses): gradients of a random linear
    functional of (expected positions, opacity) with respect to every parameter, the style, the deformation and the
    pose against torch.autograd through the oracle; shallow networks, train mode, replayed noise."""
    perturb = case in ("tennis_player_perturb", "tennis_hierarchical")
    if case == "minecraft_player":
        cfg, scene, obj, bias = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS), synthetic.minecraft_scene(seed=19), 2, 3.0
    elif case == "tennis_hierarchical":
        cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
        scene, obj, bias = synthetic.tennis_scene(seed=5), 2, 2.0
    else:
        cfg, scene, obj, bias = configs.reduced_config(configs.tennis_config(), **SMALL_NETS), synthetic.tennis_scene(seed=17), 2, 2.0
    comp = build(cfg, alpha_bias=bias).train()
    h, w = scene["image_size"]
    o, d, nrm, w2o, sty, dfm, ins = composer_inputs(cfg, scene, pixels=grid_pixels(h, w, 16))
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    names = [k for k, _ in comp.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    # the pose enters as the reference produces it: a rigid transform generated by rotation / translation parameters
    # (here a perturbation of the scene's w2o around the identity), and the gradients are compared on those parameters
    single = (w2o[..., obj], sty[..., obj], dfm[..., obj])
    lead = list(single[0].shape[:-2])

    def pose_inputs(device):
        rot = torch.zeros(lead + [3], device=device, requires_grad=True)
        tr = torch.zeros(lead + [3], device=device, requires_grad=True)
        return rot, tr, torch.matmul(single[0].to(device), ro.euler_to_matrix(rot, tr))

    rot_ref, tr_ref, w2o_ref = pose_inputs("cpu")
    ref_in = [t.clone().requires_grad_(True) for t in single[1:]]
    rec = {}
    torch.manual_seed(11)
    want = ro.expected_positions_forward(cfg, sd, o, d, nrm, w2o_ref, *ref_in, ins[..., obj], obj, perturb, training=True,
                                         record_noise=rec)
    gen = torch.Generator().manual_seed(3)
    probes = {ty: [torch.randn(t.shape, generator=gen) for t in want[ty]] for ty in want}
    sum((t * p).sum() for ty in want for t, p in zip(want[ty], probes[ty])).backward()
    comp = comp.cuda()
    rot_hip, tr_hip, w2o_hip = pose_inputs("cuda")
    hip_in = [t.clone().cuda().requires_grad_(True) for t in single[1:]]
    got = comp.forward_expected_positions(o.cuda(), d.cuda(), nrm.cuda(), w2o_hip, *hip_in, ins[..., obj].cuda(), obj, perturb,
                                          _noise=rec)
    assert set(got) == set(want)
    for ty in want:
        for a, b in zip(want[ty], got[ty]):
            assert torch.allclose(a.detach(), b.detach().cpu(), rtol=1e-3, atol=2e-4), (ty, float((a - b.detach().cpu()).abs().max()))
    sum((t * p.cuda()).sum() for ty in got for t, p in zip(got[ty], probes[ty])).backward()
    torch.cuda.synchronize()
    params = dict(comp.named_parameters())
    pairs = {k: (sd[k].grad, params[k].grad) for k in names}
    for label, a, b in zip(("pose rotation", "pose translation", "style", "deformation"), [rot_ref, tr_ref] + ref_in,
                           [rot_hip, tr_hip] + hip_in):
        pairs[label] = (a.grad, b.grad)
    largest = max(float(a.abs().max()) for a, _ in pairs.values() if a is not None)
    tol = 1e-3 if case == "tennis_hierarchical" else 1e-4
    bad, nonzero = {}, 0
    for k, (a, b) in pairs.items():
        if a is None and b is None:
            continue
        b = b.detach().cpu() if b is not None else torch.zeros_like(a)
        a = a if a is not None else torch.zeros_like(b)
        scale = float(a.abs().max())
        nonzero += scale > 0
        err = float((a - b).abs().max())
        if err > tol * scale + 2e-7 * largest:      # fp32-epsilon floor: tensors whose true gradient is ~0
            bad[k] = (err, scale)
    assert not bad, bad
    assert nonzero > 15 and float(pairs["pose rotation"][0].abs().max()) > 0 and float(pairs["deformation"][0].abs().max()) > 0


from tests.test_cpu import EXPECTED_GOLDEN, load_expected_positions_fixture, pose_parameter_gradients  # noqa: E402


@pytest.mark.parametrize("path", EXPECTED_GOLDEN, ids=[os.path.basename(p)[:-4] for p in EXPECTED_GOLDEN])
def test_expected_positions_match_reference_fixtures(path):
    """forward_expected_positions in training mode against outputs and gradients recorded from the reference
    (tests/golden/expected_positions).  d loss / d w2o is compared on the rigid motions (Euler angles + translation,
    the only way the reference produces these matrices): its component that leaves the rigid transforms - the
    reference measures sample distances with the OBJECT-frame direction, whose length is 1 for every rigid pose - has
    no effect on any pose parameter and is not produced by the renderer."""
    recipe, inputs, sd, noise, want, probes, grads, perturb, obj = load_expected_positions_fixture(path)
    cfg = recipe_config(recipe)
    comp = ObjectComposer(cfg)
    comp.load_state_dict(sd, strict=True)
    comp = comp.cuda().train()
    o, d, n, w2o, sty, dfm, ins = inputs
    leaf = [t[..., obj].clone().cuda().requires_grad_(True) for t in (w2o, sty, dfm)]
    got = comp.forward_expected_positions(o.cuda(), d.cuda(), n.cuda(), *leaf, ins[..., obj].cuda(), obj, perturb, _noise=noise)
    assert set(got) == set(want)
    for ty in want:
        for a, b, atol in zip(want[ty], got[ty], (1e-4, ATOL)):
            assert torch.allclose(a, b.detach().cpu(), rtol=RTOL, atol=atol), (ty, float((a - b.detach().cpu()).abs().max()))
    sum((t * p.cuda()).sum() for ty in got for t, p in zip(got[ty], probes[ty])).backward()
    torch.cuda.synchronize()
    params = dict(comp.named_parameters())
    largest = max(float(a.abs().max()) for a in grads.values())
    bad = {}

    def check(k, a, b):
        err, scale = float((a - b).abs().max()), float(a.abs().max())
        if err > 1e-4 * scale + 2e-7 * largest:
            bad[k] = (err, scale)

    for k, a in grads.items():
        if k == "w2o":
            want_pose = pose_parameter_gradients(w2o[..., obj], a)
            got_pose = pose_parameter_gradients(w2o[..., obj], leaf[0].grad.detach().cpu())
            check("pose rotation", want_pose[0], got_pose[0])
            check("pose translation", want_pose[1], got_pose[1])
            continue
        b = leaf[("w2o", "style", "deformation").index(k)].grad if k in ("style", "deformation") else params[k].grad
        check(k, a, b.detach().cpu() if b is not None else torch.zeros_like(a))
    assert not bad, bad


# --------------------------------------------------------------------------------------------
# Edge cases of the call shapes
# --------------------------------------------------------------------------------------------
def _compare_with_oracle(cfg, comp, inputs, perturb=False):
    want, got = run_both(cfg, comp, inputs, perturb=perturb)
    rep = compare_results(want, got, rtol=RTOL, atol=ATOL)
    return {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1]}


def test_single_ray_and_single_frame_call():
    """R = 1: one ray through the whole pipeline (compaction, MLP tile padding, compositing) - hit and miss."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    comp = build(cfg, alpha_bias=3.0)
    scene = synthetic.minecraft_scene(seed=3)
    for pixel in ((128, 128), (0, 0), (255, 255)):
        inputs = composer_inputs(cfg, scene, pixels=(torch.tensor([pixel[0]]), torch.tensor([pixel[1]])))
        assert inputs[1].shape[-2] == 1
        bad = _compare_with_oracle(cfg, comp, inputs)
        assert not bad, (pixel, bad)


def test_every_object_absent():
    """object_in_scene all False: every sample carries the empty-space alpha, nothing is evaluated by the MLP; fields
    (including the NaN disparity of zero-opacity rays) as the oracle."""
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    comp = build(cfg)
    inputs = list(composer_inputs(cfg, synthetic.tennis_scene(seed=4), pixels=grid_pixels(256, 256, 8)))
    inputs[6] = torch.zeros_like(inputs[6])
    bad = _compare_with_oracle(cfg, comp, inputs)
    assert not bad, bad
    want, got = run_both(cfg, comp, inputs)
    assert float(got["coarse"]["global"]["opacity"].abs().max()) == 0.0


def test_eight_object_instances():
    """PR_MAX_OBJECTS = 8 instances in one call: two static models and six players sharing one model (K = 8 sample lists
    merged per ray, overlap fix on), against the oracle."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    cfg["model"]["object_parameters_encoder"] = [{"objects_count": 1}, {"objects_count": 1}, {"objects_count": 6}]
    comp = build(cfg, alpha_bias=3.0)
    assert comp.object_id_helper.objects_count == 8
    base = synthetic.minecraft_scene(seed=6)
    scene = dict(base)
    g = torch.Generator().manual_seed(1)
    for key in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation"):
        t = base[key]
        extra = t[..., 2:3].repeat_interleave(6, dim=-1)
        if key == "object_translation_parameters":       # spread the six players around the first one
            extra = extra + torch.cat([torch.zeros(list(t.shape[:-1]) + [1]), 1.5 * torch.randn(list(t.shape[:-1]) + [5], generator=g)], -1)
            extra[..., 1, :] = t[..., 1, 2:3]             # keep them on the ground
        elif key in ("object_style", "object_deformation"):
            extra = torch.randn(extra.shape, generator=g)
        scene[key] = torch.cat([t[..., :2], extra], dim=-1)
    scene["object_in_scene"] = torch.ones(list(base["object_in_scene"].shape[:-1]) + [8], dtype=torch.bool)
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 16))
    assert inputs[3].shape[-1] == 8
    bad = _compare_with_oracle(cfg, comp, inputs, perturb=True)
    assert not bad, bad
    with pytest.raises(Exception):
        cfg9 = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
        cfg9["model"]["object_parameters_encoder"] = [{"objects_count": 1}, {"objects_count": 1}, {"objects_count": 7}]
        c9 = build(cfg9).cuda()
        big = [t.cuda() for t in inputs]
        big[3] = torch.cat([big[3], big[3][..., :1]], -1)
        big[4] = torch.cat([big[4], big[4][..., :1]], -1)
        big[5] = torch.cat([big[5], big[5][..., :1]], -1)
        big[6] = torch.cat([big[6], big[6][..., :1]], -1)
        with torch.no_grad():
            c9(*big, False)            # nine instances: rejected (PR_MAX_OBJECTS)


def test_patch_pixel_kernel_matches_the_tensor_route():
    """pr_patch_pixels (one launch: closed-form cumulative sum of the box weight image + binary search) against the tensor-op route
    of ray_sampling (weight image, cumulative sum, searchsorted - the reference's RayHelper.sample_rays_strided_patch) for the
    SAME uniform draws: identical pixel lists, centres inside and outside the boxes, patches clamped at every image border."""
    from playableenvironments_amd import ray_sampling as rs
    torch.manual_seed(3)
    dev = torch.device("cuda")
    weights = [0.2, 0.1, 0.35, 0.35]
    # sizes whose pixel count is 64 k + 1 or leaves ranges of 64 k + 1 pixels inside the 64-way lookup (65 x 64, 33 x 33, 65 x 65,
    # 129 x 129): a step rounded from hi - lo instead of hi - lo + 1 never probes the last pixel of such a range
    for (h, w, patch, strides) in ((288, 512, 48, [4, 8]), (96, 160, 16, [2, 4]), (64, 64, 8, [4]), (65, 64, 8, [2, 4]),
                                   (33, 33, 8, [2]), (65, 65, 8, [4]), (129, 129, 16, [2, 4])):
        n, k = 64, 4
        lo = torch.rand((n, 2, k), device=dev) * 0.7
        ext = torch.rand((n, 2, k), device=dev) * 0.3 + 0.02
        boxes = torch.cat([lo, (lo + ext).clamp(max=1.0)], dim=1)              # (N, 4, K) [left, top, right, bottom]
        boxes[:, :, 3] = torch.tensor([0.0, 0.0, 1.0, 1.0], device=dev)        # one object covers the frame: corners get drawn too
        u = torch.rand((n,), device=dev)
        u[:4] = torch.tensor([0.0, 1e-7, 0.9999999, 0.5], device=dev)
        u[4] = 1.0                                                             # the last pixel of the image
        rows, cols = rs.strided_patch_rows_cols(boxes, weights, h, w, patch, strides, _u=u)
        mask = rs._weight_masks(boxes, weights, h, w, guard_zero_area=False).double()
        cdf = torch.cumsum(mask, dim=1)
        centres = torch.searchsorted(cdf, (u.double() * cdf[:, -1]).unsqueeze(1)).clamp(max=h * w - 1)[:, 0]
        want = rs.patch_pixels_around(centres, h, w, patch, strides)
        got = rows.to(torch.int64) * w + cols.to(torch.int64)
        same = (got == want).all(dim=1)
        # a draw that lands within rounding of a pixel's cumulative weight may resolve to the neighbouring pixel
        assert int((~same).sum()) <= 1, (int((~same).sum()), h, w)
        assert int(rows.min()) >= 0 and int(rows.max()) < h and int(cols.min()) >= 0 and int(cols.max()) < w
    # the centre lookup itself, sharply: a 2 x 2 patch at stride 1 starts one pixel up / left of the drawn centre (no alignment, clamped only
    # at the image border), so a wrong centre shows as a wrong pixel list - 40 random image sizes (odd, prime, 64 k + 1 ...), 256 draws each
    rng = np.random.default_rng(11)
    for h, w in [(int(a), int(b)) for a, b in rng.integers(5, 330, size=(36, 2))] + [(65, 65), (1 + 64 * 3, 7), (9, 1 + 64 * 5), (64, 65)]:
        n, k = 256, 4
        lo = torch.rand((n, 2, k), device=dev) * 0.7
        ext = torch.rand((n, 2, k), device=dev) * 0.3 + 0.02
        boxes = torch.cat([lo, (lo + ext).clamp(max=1.0)], dim=1)
        boxes[:, :, 3] = torch.tensor([0.0, 0.0, 1.0, 1.0], device=dev)
        u = torch.rand((n,), device=dev)
        u[:3] = torch.tensor([0.0, 1.0, 0.9999999], device=dev)
        rows, cols = rs.strided_patch_rows_cols(boxes, weights, h, w, 2, [1], _u=u)
        mask = rs._weight_masks(boxes, weights, h, w, guard_zero_area=False).double()
        cdf = torch.cumsum(mask, dim=1)
        centres = torch.searchsorted(cdf, (u.double() * cdf[:, -1]).unsqueeze(1)).clamp(max=h * w - 1)[:, 0]
        want = rs.patch_pixels_around(centres, h, w, 2, [1])
        got = rows.to(torch.int64) * w + cols.to(torch.int64)
        differing = int((~(got == want).all(dim=1)).sum())
        assert differing <= 2, (differing, h, w)          # (draws within rounding of a pixel's cumulative weight)


def test_recorded_training_step_equals_eager_steps():
    """frame_graph.GraphedStep: forward + loss + backward + Adam recorded once as a HIP graph and replayed (batches of
    back-to-back replays with host synchronisations between them) leaves the parameters where the same eager iterations leave
    them; with perturbation every replay draws its own noise seed on the device.  Runs in its own process: the runtime switch
    that keeps the memset nodes of a recorded graph alive on ROCm 7.0.2 has to be in the environment before the first HIP call."""
    import subprocess
    from playableenvironments_amd.frame_graph import GRAPH_RUNTIME_SWITCH
    env = dict(os.environ)
    env[GRAPH_RUNTIME_SWITCH[0]] = GRAPH_RUNTIME_SWITCH[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_step_check.py")
    done = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0 and "GRAPH STEP OK" in done.stdout, done.stdout[-2000:] + done.stderr[-4000:]


def test_frame_graph_replay_is_bit_identical():
    """FrameGraph: one minecraft frame captured as a HIP graph and replayed for other scene encodings - every field equal
    to the eager render of the same encoding, for both kernels; stale weights are refused."""
    from playableenvironments_amd.frame_graph import FrameGraph, SCENE_KEYS
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    model = model.eval().cuda()
    size = (64, 96)
    scenes = [{k: v.cuda() for k, v in synthetic.minecraft_scene(seed=s, image_size=size).items() if torch.is_tensor(v)}
              for s in (5, 6, 7)]
    for precision in ("fp32", "f16x3"):
        model.object_composer.precision = precision
        graph = FrameGraph(model, scenes[0], size)
        for scene in scenes[::-1]:
            got = graph.render(scene)
            with torch.no_grad():
                want = model(*[scene[k] for k in SCENE_KEYS[:3]], size, *[scene[k] for k in SCENE_KEYS[3:]], 0, False,
                             mode="scene_encodings")
            torch.cuda.synchronize()
            for entry in ("global", "object_0", "object_3"):
                for key in ("integrated_features", "opacity", "depth", "weights"):
                    assert torch.equal(got["coarse"][entry][key], want["coarse"][entry][key]), (precision, entry, key)
            assert torch.equal(got["reconstructed_bounding_boxes"], want["reconstructed_bounding_boxes"])
    # what the captured launches baked in: the packed weights of the precision, the annealing step, the parameter values
    model.object_composer.precision = "fp32"
    with pytest.raises(RuntimeError, match="changed since the frame was captured"):
        graph.render(scenes[0])
    with torch.no_grad():                      # the render at the other precision repacks: the captured buffers must survive it
        model(*[scenes[0][k] for k in SCENE_KEYS[:3]], size, *[scenes[0][k] for k in SCENE_KEYS[3:]], 0, False, mode="scene_encodings")
    model.object_composer.precision = "f16x3"
    again = graph.render(scenes[2])
    with torch.no_grad():
        want = model(*[scenes[2][k] for k in SCENE_KEYS[:3]], size, *[scenes[2][k] for k in SCENE_KEYS[3:]], 0, False, mode="scene_encodings")
    assert torch.equal(again["coarse"]["global"]["integrated_features"], want["coarse"]["global"]["integrated_features"])
    model.set_step(30000)
    with pytest.raises(RuntimeError, match="changed since the frame was captured"):
        graph.render(scenes[0])
    graph = FrameGraph(model, scenes[0], size)
    graph.render(scenes[1])
    with torch.no_grad():
        next(model.object_composer.parameters()).add_(1e-3)
    with pytest.raises(RuntimeError, match="changed since the frame was captured"):
        graph.render(scenes[0])
    model.train()
    with pytest.raises(ValueError):
        FrameGraph(model, scenes[0], size)


#: rays of the seeded native frame whose drop-in render (matrices / rays built on the GPU) leaves the oracle's tolerance: measured
#: (round 5, MI355X: none - the GPU-built pose matrices and rays reproduce the CPU-built ones closely enough for every box decision)
NATIVE_FRAME_FLIPPED_RAYS = {("tennis", "fp32"): 0, ("tennis", "f16x3"): 0, ("minecraft", "fp32"): 0, ("minecraft", "f16x3"): 0}


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_native_evaluation_frame_matches_oracle(world):
    """The frame the reference's evaluators and play loop render (environment_model_backpropagated_autoencoder.py:
    173-236): 288 x 512, strided grids [4, 8] = 72 x 128 + 36 x 64 = 11 520 rays, SHIPPED network sizes.  (1) composer level, the
    same CPU-built inputs on both sides: every field of every entry within the fp32 tolerance, both kernels; (2) the plain drop-in
    call ``forward_from_scene_encoding(..., patch_stride=[4, 8])`` (matrices and rays built on the GPU: ulp-level differences may
    flip a box decision on a handful of rays) against the oracle's own scene-encoding render; (3) the decoder-layout maps are
    the wire-format fold of the ray-major features, bit for bit; (4) the frame replayed from a FrameGraph is bit-identical."""
    from playableenvironments_amd.frame_graph import FrameGraph, SCENE_KEYS
    size, strides = (288, 512), [4, 8]
    cfg = configs.tennis_config() if world == "tennis" else configs.minecraft_config()
    make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
    scene = make(seed=1234, image_size=size)
    model = em.EnvironmentModel(cfg)
    torch.manual_seed(0)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=60000, alpha_bias=1.0, bender_scale=1e4)
    model.eval()
    comp = model.object_composer
    inputs = composer_inputs(cfg, scene, strides=strides)
    assert inputs[1].shape[-2] == 11520
    sd = {k: v.detach().clone() for k, v in comp.state_dict().items()}
    args = [scene[k] for k in SCENE_KEYS[:3]] + [size] + [scene[k] for k in SCENE_KEYS[3:]]
    with torch.no_grad():
        want = ro.composer_forward(cfg, sd, *inputs, False, stable_merge=True)     # (= the oracle's scene-encoding render)
    model = model.cuda()
    gargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    gscene = {k: scene[k].cuda() for k in SCENE_KEYS}
    for precision in ("fp32", "f16x3"):
        comp.precision = precision
        with torch.no_grad():
            got = comp(*[v.cuda() for v in inputs], False)
            env = model.forward_from_scene_encoding(*gargs, 0, False, 1200, patch_stride=strides, _decoder_features=[64, 128])
        torch.cuda.synchronize()
        assert_close(want, got)
        a = want["coarse"]["global"]["integrated_features"]
        b = env["coarse"]["global"]["integrated_features"].cpu()
        assert a.shape == b.shape == (1, 1, 1, 11520, 192)
        flipped = int(((a - b).abs() > ATOL + RTOL * a.abs()).any(-1).sum())
        print(f"native frame {world} {precision}: {flipped} of 11520 rays beyond the tolerance (GPU-built matrices and rays)")
        # the MEASURED count of this seeded frame (deterministic kernels; recorded on the MI355X box, round 5) + 25 %, not a blanket
        # 0.5 % (58 rays): a regression of the pose / ray set-up that flips more box decisions than this fails
        allowed = NATIVE_FRAME_FLIPPED_RAYS[(world, precision)]
        assert flipped <= allowed + -(-allowed // 4), (precision, flipped, allowed)
        # the maps the decoder consumes = the reference's fold_strided_grid_samples + split_features_by_layer + permute
        maps = env["coarse"]["global"]["decoder_features"]
        assert [tuple(m.shape[-3:]) for m in maps] == [(64, 72, 128), (128, 36, 64)]
        feats = env["coarse"]["global"]["integrated_features"]
        first = feats[..., :72 * 128, 0:64].reshape(1, 1, 1, 72, 128, 64).movedim(-1, -3)
        second = feats[..., 72 * 128:, 64:192].reshape(1, 1, 1, 36, 64, 128).movedim(-1, -3)
        assert torch.equal(maps[0], first) and torch.equal(maps[1], second)
        graph = FrameGraph(model, gscene, size, patch_stride=strides)
        other = {k: v.cuda() for k, v in make(seed=77, image_size=size).items() if torch.is_tensor(v)}
        for sc in (other, gscene):
            replayed = graph.render(sc)
            with torch.no_grad():
                eager = model.forward_from_scene_encoding(*[sc[k] for k in SCENE_KEYS[:3]], size, *[sc[k] for k in SCENE_KEYS[3:]],
                                                          0, False, 1200, patch_stride=strides)
            torch.cuda.synchronize()
            for entry in eager["coarse"]:
                for key in ("integrated_features", "opacity", "depth", "weights", "disparity"):
                    x, y = replayed["coarse"][entry][key], eager["coarse"][entry][key]
                    assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), (precision, entry, key)
            for key in ("reconstructed_bounding_boxes", "reconstructed_3d_bounding_boxes", "projected_axes"):
                assert torch.equal(replayed[key], eager[key]), key
        del graph
    comp.precision = "fp32"


def test_frame_graph_of_the_observation_mode_matches_the_eager_call():
    """FrameGraph(mode="observations"): the evaluators' ``render_full_frame_from_observations`` span - this package's CNN encoders
    and pose estimators, roi_pool crops, pose math, rays, renderer - captured once and replayed for another batch: every tensor
    of the result dictionary equals the eager call's (native 288 x 512 frame, strides [4, 8], two frames).  Not bit for bit: the
    stock PyTorch-ROCm convolutions of the encoders are not run-to-run deterministic - two EAGER calls on the same batch differ by
    5e-7 in the estimated poses (measured, tools/perf/dbg_obs_graph.py), which the render turns into 1e-5 .. 1e-4 - so the
    yardstick is the eager call's own repeatability; the renderer's part of a replay is bit-identical (the scene-encoding tests)."""
    from playableenvironments_amd.frame_graph import FrameGraph, OBSERVATION_KEYS
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.minecraft_config(encoders=True), **small)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda().eval()
    size = (288, 512)
    batches = [{k: v.cuda() for k, v in synthetic.observation_batch(synthetic.minecraft_scene(batch=2, seed=s, image_size=size),
                                                                    boxes_seed=s).items()} for s in (3, 4)]
    graph = FrameGraph(model, batches[0], mode="observations", patch_stride=[4, 8])

    def flat(d, prefix=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, prefix + k + "/")
            elif isinstance(v, (list, tuple)):
                for i, t in enumerate(v):
                    yield f"{prefix}{k}/{i}", t
            elif torch.is_tensor(v):
                yield prefix + k, v
    for b in (batches[1], batches[0]):
        replayed = dict(flat(graph.render(b)))
        with torch.no_grad():
            eager = dict(flat(model(*[b[k] for k in OBSERVATION_KEYS], 0, False, 1200, patch_stride=[4, 8])))
        torch.cuda.synchronize()
        assert sorted(replayed) == sorted(eager) and len(eager) > 50
        for k in eager:
            a, b = torch.nan_to_num(replayed[k].float(), nan=-7.0), torch.nan_to_num(eager[k].float(), nan=-7.0)
            assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-3, atol=1e-3), (k, float((a - b).abs().max()))
    with torch.no_grad():
        next(model.object_encoders[0].parameters()).add_(1e-3)           # the encoders' weights are part of the signature
    with pytest.raises(RuntimeError, match="changed since the frame was captured"):
        graph.render(batches[0])


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_fused_scene_setup_is_bit_identical(world):
    """pr_scene_setup (one launch: pose matrices, projected boxes / points / axes, the renderer's input layouts) against the route
    through pr_pose_matrices / pr_project_points and the composer's own marshalling: every tensor of the result dictionary of an
    evaluation call equal bit for bit - batches, several observations, two cameras, upsampling, the strided grids, absent objects."""
    cfg = configs.reduced_config(configs.tennis_config() if world == "tennis" else configs.minecraft_config(), **SMALL_NETS)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.eval().cuda()
    make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene

    def flat(d, prefix=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, prefix + k + "/")
            elif torch.is_tensor(v):
                yield prefix + k, v
    cases = [dict(batch=1, observations=1, cameras=1, kw=dict(patch_stride=[4, 8])),
             dict(batch=2, observations=3, cameras=1, kw=dict()),
             dict(batch=1, observations=2, cameras=2, kw=dict(patch_stride=[4, 8], upsample_factor=2.0)),
             dict(batch=2, observations=1, cameras=1, kw=dict(canonical_pose=True))]
    for case in cases:
        size = (32, 48)
        scene = make(batch=case["batch"], observations=case["observations"], seed=41, image_size=size)
        if case["cameras"] == 2:       # a second camera per observation: the first one moved
            for k in ("camera_rotations", "camera_translations", "focals"):
                other = scene[k] + (0.03 if k != "focals" else 5.0)
                scene[k] = torch.cat([scene[k], other], dim=2)
        scene["object_in_scene"][0, 0, -1] = False
        args = [scene[k].cuda() for k in ("camera_rotations", "camera_translations", "focals")] + [size] + \
               [scene[k].cuda() for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                                          "object_deformation", "object_in_scene")]
        with torch.no_grad():
            model.fused_scene_setup = True
            fused = dict(flat(model(*args, 0, False, mode="scene_encodings", **case["kw"])))
            model.fused_scene_setup = False
            plain = dict(flat(model(*args, 0, False, mode="scene_encodings", **case["kw"])))
        torch.cuda.synchronize()
        model.fused_scene_setup = True
        assert sorted(fused) == sorted(plain) and len(plain) > 40
        for k in plain:
            assert fused[k].shape == plain[k].shape, (case, k)
            assert torch.equal(torch.nan_to_num(fused[k].float(), nan=-7.0), torch.nan_to_num(plain[k].float(), nan=-7.0)), (case, k)


def _flat_tensors(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _flat_tensors(v, prefix + k + "/")
        elif isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if torch.is_tensor(t):
                    yield f"{prefix}{k}/{i}", t
        elif torch.is_tensor(v):
            yield prefix + k, v


def _same_results(a, b, what=""):
    a, b = dict(_flat_tensors(a)), dict(_flat_tensors(b))
    assert sorted(a) == sorted(b) and len(a) > 40, what
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (what, k)
        x, y = a[k], b[k]
        if x.is_floating_point():
            x, y = torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)
        assert torch.equal(x, y), (what, k, float((x.float() - y.float()).abs().max()))


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_fused_scene_setup_of_a_training_call(world):
    """A TRAINING call through pr_scene_setup + pr_scene_setup_backward (one launch each instead of ~25 + ~10 small ones) against
    the tensor route (``fused_scene_setup = False``: pr_pose_matrices with its autograd node, permuting copies, the composer's own
    marshalling): every output and every gradient - object rotations / translations, style, deformation, the renderer's
    parameters (outputs bit for bit; gradients up to the order of the renderer's atomic sums); and a loss that reads the projected
    boxes / axes gets the tensor formulation's gradients."""
    cfg = configs.reduced_config(configs.tennis_config() if world == "tennis" else configs.minecraft_config(), **SMALL_NETS)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda().train()
    make = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
    size = (64, 96)
    scene = make(batch=2, observations=2, seed=17, image_size=size)
    keys = ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation")
    params = [p for p in model.object_composer.parameters() if p.requires_grad]

    def run(fused, read_boxes):
        model.fused_scene_setup = fused
        sc = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in scene.items()}
        for k in keys:
            sc[k] = sc[k].clone().requires_grad_(True)
        for p in params:
            p.grad = None
        torch.manual_seed(5)
        out = model(sc["camera_rotations"], sc["camera_translations"], sc["focals"], size, sc["object_rotation_parameters"],
                    sc["object_translation_parameters"], sc["object_style"], sc["object_deformation"], sc["object_in_scene"],
                    400, True, 0, patch_size=16, patch_stride=[2, 4], mode="scene_encodings")
        g = out["coarse"]["global"]
        loss = (g["integrated_features"] ** 2).mean() + g["opacity"].mean() + 0.1 * g["depth"].mean()
        if read_boxes:
            loss = loss + out["reconstructed_bounding_boxes"].square().mean() + out["projected_axes"].square().mean() + \
                out["reconstructed_3d_bounding_boxes"].mean()
        loss.backward()
        torch.cuda.synchronize()
        fields = {k: out["coarse"]["global"][k].detach().clone() for k in ("integrated_features", "opacity", "depth")}
        fields.update({k: out[k].detach().clone() for k in ("reconstructed_bounding_boxes", "projected_axes", "reconstructed_3d_bounding_boxes")})
        return fields, {k: sc[k].grad.clone() for k in keys}, [p.grad.clone() for p in params]
    for read_boxes in (False, True):
        want = run(False, read_boxes)
        got = run(True, read_boxes)
        for k in want[0]:
            assert torch.equal(torch.nan_to_num(got[0][k]), torch.nan_to_num(want[0][k])), (read_boxes, k)
        for k in keys:
            a, b = got[1][k], want[1][k]
            assert a.shape == b.shape and float(b.abs().max()) > 0, k
            # (the renderer accumulates d style / d deformation / d w2o over the samples with atomics: equal up to the order of
            # those sums; with read_boxes the projections' gradients are added - tensor ops on both sides)
            assert torch.allclose(a, b, rtol=1e-4 if read_boxes else 1e-5, atol=2e-6 * float(b.abs().max())), (read_boxes, k, float((a - b).abs().max()))
        for a, b in zip(got[2], want[2]):      # (the AdaIN scale / bias gradients are atomic sums over the tiles: equal up to their order)
            assert torch.allclose(a, b, rtol=1e-5, atol=2e-6 * float(b.abs().max()) + 1e-12), float((a - b).abs().max())
    model.fused_scene_setup = True


def test_automatic_frame_replay():
    """``EnvironmentModel.frame_replay`` ("clone" by DEFAULT): the UNCHANGED evaluation calls (forward_from_scene_encoding with the
    strided grids - what the reference's autoencoder subclasses issue - and forward_from_observations) are recorded the second time a
    shape is seen and replayed: results bit-identical to the eager call for the renderer-only mode ("clone": copies that survive
    the next call; "alias": the recording's static tensors), within the encoders' own run-to-run noise for the observation mode;
    training-mode, perturbed and differentiable calls are never replayed; a weight update re-records."""
    from playableenvironments_amd.frame_graph import OBSERVATION_KEYS, SCENE_KEYS
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.minecraft_config(encoders=True), **small)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda().eval()
    assert model.frame_replay == "clone"               # what a caller that changes nothing gets
    size = (96, 128)
    scenes = [{k: v.cuda() for k, v in synthetic.minecraft_scene(seed=s, image_size=size).items() if torch.is_tensor(v)} for s in (5, 6)]

    def call(sc, **kw):
        with torch.no_grad():
            return model.forward_from_scene_encoding(*[sc[k] for k in SCENE_KEYS[:3]], size, *[sc[k] for k in SCENE_KEYS[3:]], 0, False,
                                                     1200, patch_stride=[4, 8], **kw)

    def recorded():
        return [k for k, e in model._replays.items() if e[1] not in (None, False)]
    model.frame_replay = None
    eager = [call(sc) for sc in scenes]
    assert model._replays == {}
    model.frame_replay = "clone"
    first = call(scenes[0])                       # first sight of the shape: runs eagerly (and warms the recording up)
    assert len(model._replays) == 1 and recorded() == []
    second = call(scenes[1])                      # second sight: recorded, replayed, copied out
    assert len(recorded()) == 1
    third = call(scenes[0])                       # replayed
    _same_results(first, eager[0], "first (eager) call")
    _same_results(second, eager[1], "recording call")
    _same_results(third, eager[0], "replayed call")
    _same_results(second, eager[1], "a copy survives the next call")
    assert third["coarse"]["global"]["integrated_features"].data_ptr() != second["coarse"]["global"]["integrated_features"].data_ptr()
    # "alias": the recording's own tensors, overwritten by the next call of the same shape
    model.frame_replay = "alias"
    a = call(scenes[0])
    kept = a["coarse"]["global"]["integrated_features"]
    assert torch.equal(kept, eager[0]["coarse"]["global"]["integrated_features"])
    b = call(scenes[1])
    assert b["coarse"]["global"]["integrated_features"] is kept and torch.equal(kept, eager[1]["coarse"]["global"]["integrated_features"])
    model.frame_replay = "clone"
    # another option set is another recording; perturbed / training / differentiable / pixel-sampling calls run eagerly
    call(scenes[0], canonical_pose=True)
    assert len(model._replays) == 2 and len(recorded()) == 1
    with torch.no_grad():
        for _ in range(2):
            model.forward_from_scene_encoding(*[scenes[0][k] for k in SCENE_KEYS[:3]], size, *[scenes[0][k] for k in SCENE_KEYS[3:]], 0, True,
                                              patch_stride=[4, 8])
            model.forward_from_scene_encoding(*[scenes[0][k] for k in SCENE_KEYS[:3]], size, *[scenes[0][k] for k in SCENE_KEYS[3:]], 50, False)
    model.object_composer.train()                  # (the composer alone in training mode: batch statistics - never a recording)
    with torch.no_grad():
        for _ in range(2):
            model.forward_from_scene_encoding(*[scenes[0][k] for k in SCENE_KEYS[:3]], size, *[scenes[0][k] for k in SCENE_KEYS[3:]], 0, False,
                                              patch_stride=[4, 8])
    model.object_composer.eval()
    assert len(model._replays) == 2 and len(recorded()) == 1
    # a weight update invalidates the recordings: the stale one is replaced (not another slot evicted), results follow the weights
    with torch.no_grad():
        next(model.object_composer.parameters()).add_(1e-3)
    model.frame_replay = None
    want = call(scenes[1])
    model.frame_replay = "clone"
    for i in range(3):                             # eager (re-seen), recording, replay
        got = call(scenes[1])
        _same_results(got, want, f"after a weight update, call {i}")
    assert len(model._replays) == 2 and len(recorded()) == 1
    assert not torch.equal(got["coarse"]["global"]["integrated_features"], eager[1]["coarse"]["global"]["integrated_features"])
    # switches the recording baked in are part of its signature
    model.focal_length_multiplier = model.focal_length_multiplier * 1.25
    model.frame_replay = None
    want = call(scenes[0])
    model.frame_replay = "clone"
    for i in range(3):
        _same_results(call(scenes[0]), want, f"focal_length_multiplier changed, call {i}")
    # the observation-driven evaluation call
    batches = [{k: v.cuda() for k, v in synthetic.observation_batch(synthetic.minecraft_scene(batch=2, seed=s, image_size=size),
                                                                    boxes_seed=s).items()} for s in (3, 4)]
    model.frame_replay = None
    with torch.no_grad():
        plain = [model.forward_from_observations(*[b[k] for k in OBSERVATION_KEYS], 0, False, 1200, patch_stride=[4, 8]) for b in batches]
    model.frame_replay = "clone"
    for round_ in range(3):
        with torch.no_grad():
            replayed = [model.forward_from_observations(*[b[k] for k in OBSERVATION_KEYS], 0, False, 1200, patch_stride=[4, 8]) for b in batches]
        torch.cuda.synchronize()
        for p_, r_ in zip(plain, replayed):
            for key in ("integrated_features", "opacity"):
                x, y = p_["coarse"]["global"][key], r_["coarse"]["global"][key]
                assert torch.allclose(x, y, rtol=1e-3, atol=1e-3), (round_, key, float((x - y).abs().max()))
            assert torch.allclose(p_["scene_encoding"]["object_style"], r_["scene_encoding"]["object_style"], rtol=1e-4, atol=1e-5)
    assert any(k[0] == "observations" for k in recorded())
    import copy
    clone = copy.deepcopy(model)               # recorded graphs stay with the original
    assert clone._replays == {}


def test_frame_replay_soak_alternating_shapes_precisions_and_state():
    """The default recording under everything an evaluation / play session does to a model between frames: 1 000 calls of the
    UNCHANGED ``forward_from_scene_encoding`` alternating three frame shapes (more shapes than recording slots at one point), the
    three precisions, ``set_step``, ``load_state_dict`` through the parent, in-place weight updates, ``object_entry_fields``, host
    synchronisations between some frames and none between others - every result dictionary bit-identical to the same call with
    ``frame_replay = None`` on a twin model (same weights and state throughout), and most calls actually replayed."""
    from playableenvironments_amd.frame_graph import SCENE_KEYS
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    torch.manual_seed(0)
    model, twin = em.EnvironmentModel(cfg), em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    twin.load_state_dict(model.state_dict())
    model, twin = model.cuda().eval(), twin.cuda().eval()
    twin.frame_replay = None
    model.frame_replay_slots = 2                    # (three shapes below: the oldest recording is dropped and made again)
    shapes = [((24, 32), [4, 8]), ((32, 48), 0), ((40, 40), [4])]
    scenes = {}
    for size, _ in shapes:
        scenes[size] = [{k: v.cuda() for k, v in synthetic.minecraft_scene(seed=s, image_size=size).items() if torch.is_tensor(v)}
                        for s in (11, 12, 13)]
    checkpoints = []
    for seed in (1, 2):
        other = em.EnvironmentModel(cfg)
        synthetic.randomize_module_state(other.object_composer, seed=seed, step=20000, alpha_bias=2.5, bender_scale=1e4)
        checkpoints.append({k: v.clone() for k, v in other.state_dict().items()})
    rng = np.random.default_rng(5)
    from playableenvironments_amd import frame_graph
    counted = {"replays": 0}
    original_replay = frame_graph.CapturedCall.replay

    def counting_replay(self, tensors):
        counted["replays"] += 1
        return original_replay(self, tensors)

    def call(m, size, stride, sc):
        with torch.no_grad():
            return m.forward_from_scene_encoding(*[sc[k] for k in SCENE_KEYS[:3]], size, *[sc[k] for k in SCENE_KEYS[3:]], 0, False,
                                                 1200, patch_stride=stride)
    shape_idx = 0
    frame_graph.CapturedCall.replay = counting_replay
    try:
        _soak_loop(model, twin, shapes, scenes, checkpoints, rng, call)
    finally:
        frame_graph.CapturedCall.replay = original_replay
    assert counted["replays"] >= 700, counted            # most frames of the session were replayed recordings
    assert not any(e[1] is False for e in model._replays.values())          # nothing failed to record
    assert twin._replays == {} and len(model._replays) <= 2


def _soak_loop(model, twin, shapes, scenes, checkpoints, rng, call):
    shape_idx = 0
    for it in range(1000):
        r = rng.random()
        if r < 0.03:
            shape_idx = int(rng.integers(len(shapes)))
        elif r < 0.05:
            precision = ("fp32", "f16x3", "f16")[int(rng.integers(3))]
            model.object_composer.precision = twin.object_composer.precision = precision
        elif r < 0.06:
            step = int(rng.integers(0, 60000))
            model.set_step(step), twin.set_step(step)
        elif r < 0.07:
            sd = checkpoints[int(rng.integers(2))]
            model.load_state_dict(sd), twin.load_state_dict(sd)
        elif r < 0.08:
            with torch.no_grad():
                delta = float(rng.normal()) * 1e-3
                for m in (model, twin):
                    next(m.object_composer.parameters()).add_(delta)
        elif r < 0.09:
            fields = (None, ("integrated_features", "opacity"), ())[int(rng.integers(3))]
            model.object_composer.object_entry_fields = twin.object_composer.object_entry_fields = fields
        size, stride = shapes[shape_idx]
        sc = scenes[size][int(rng.integers(3))]
        got = call(model, size, stride, sc)
        want = call(twin, size, stride, sc)
        if rng.random() < 0.3:
            torch.cuda.synchronize()
        a, b = dict(_flat_tensors(got)), dict(_flat_tensors(want))
        assert sorted(a) == sorted(b), it
        for k in a:
            x, y = a[k], b[k]
            if x.is_floating_point():
                x, y = torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)
            assert torch.equal(x, y), (it, k, model.object_composer.precision, size)


def test_frame_replay_refuses_recordings_with_memset_nodes():
    """The automatic evaluation-frame recording also records whatever decoder / encoder modules the caller injected.  torch's
    multi-block reductions (a large ``mean`` / ``sum``) record a MEMSET node, and on this HIP runtime the memset nodes of a replayed
    graph stop executing after a host synchronisation unless ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` is in the environment
    (frame_graph.GRAPH_RUNTIME_SWITCH).  ``pr_graph_node_census`` counts the nodes of every recording: the renderer's own frame holds
    kernels only and is replayed; a frame whose injected decoder reduces 6 M values is refused (one warning, the call stays eager
    and right, host synchronisations between frames included) - unless the switch is set, then it replays."""
    import ctypes as C
    import warnings
    from playableenvironments_amd import _lib, frame_graph
    from playableenvironments_amd.frame_graph import SCENE_KEYS
    x = torch.randn(1 << 22, device="cuda")
    plain = frame_graph.CapturedCall(lambda t: t * 2 + 1, [x], warmup=1)
    assert plain.census["memsets"] == 0 and plain.census["kernels"] == plain.census["nodes"] >= 1
    assert torch.equal(plain.replay([x]), x * 2 + 1)
    with pytest.raises(_lib.PlayRenderError):
        _lib.check(_lib.load().pr_graph_node_census(None, (C.c_int32 * 4)()), "pr_graph_node_census")
    safe = frame_graph.graph_runtime_is_safe()
    if safe:
        reduced = frame_graph.CapturedCall(lambda t: t.sum(), [x], warmup=1)
        assert reduced.census["memsets"] >= 1
    else:
        with pytest.raises(frame_graph.UnsafeRecording, match="memset"):
            frame_graph.CapturedCall(lambda t: t.sum(), [x], warmup=1)
    torch.cuda.synchronize()
    # the model: renderer-only frames replay; with a reducing decoder behind the renderer the recording is refused
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.0, bender_scale=1e4)
    model = model.cuda().eval()
    size = (64, 96)
    scenes = [{k: v.cuda() for k, v in synthetic.tennis_scene(seed=s, image_size=size).items() if torch.is_tensor(v)} for s in (5, 6, 7)]

    def call(sc):
        with torch.no_grad():
            return model.forward_from_scene_encoding(*[sc[k] for k in SCENE_KEYS[:3]], size, *[sc[k] for k in SCENE_KEYS[3:]], 0, False)
    for sc in scenes:
        call(sc)
    (entry,) = model._replays.values()
    assert entry[1] not in (None, False) and entry[1].census["memsets"] == 0 and entry[1].census["kernels"] >= 5

    class Sampler(torch.nn.Module):
        def forward(self, feats, positions):
            return feats

    class Decoder(torch.nn.Module):
        def forward(self, grid):                      # (..., rays, features) -> a global statistic: a multi-block reduction
            big = grid.reshape(-1).repeat(32)
            return grid[..., :3] - big.mean()
    model.use_image_decoder = True
    model.set_image_decoder(Decoder(), Sampler())
    model.frame_replay = None
    want = [call(sc)["coarse"]["global"]["decoded_images"] for sc in scenes]
    model.frame_replay = "clone"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for round_ in range(3):
            for sc, w in zip(scenes, want):
                got = call(sc)["coarse"]["global"]["decoded_images"]
                torch.cuda.synchronize()              # (what stops the memset nodes of a replayed graph on the unsafe runtime)
                assert torch.allclose(got, w, rtol=1e-5, atol=1e-6), (round_, float((got - w).abs().max()))
    refused = [w for w in caught if "memset" in str(w.message)]
    decoded = [e for k, e in model._replays.items() if e[1] is not None][-1]
    if safe:
        assert not refused and decoded[1] is not False and decoded[1].census["memsets"] >= 1
    else:
        assert len(refused) == 1 and decoded[1] is False          # detected once; the shape stays eager


def test_two_cameras_per_observation():
    """cameras_count = 2: the object tensors carry a singleton camera dimension that broadcasts against (..., O, C) rays
    (model/environment_model.py:1041-1158 shapes).  The reference itself only runs with one camera (its boolean-mask
    write of the empty-space alpha does not broadcast, object_composer.py:547), so the expectation is the oracle's
    render of each camera on its own."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    model = em.EnvironmentModel(cfg)
    torch.manual_seed(0)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=3.0, bender_scale=1e4)
    model.eval()
    two = synthetic.minecraft_scene(batch=1, observations=2, seed=91, image_size=(48, 64))
    object_keys = ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation",
                   "object_in_scene")

    def arguments(scene):
        return [scene[k] for k in ("camera_rotations", "camera_translations", "focals")] + [two["image_size"]] + \
               [scene[k] for k in object_keys]

    sd = {k: v.clone() for k, v in model.object_composer.state_dict().items()}
    per_camera = []
    with torch.no_grad():
        for c in range(2):      # camera c of the pair as a single-camera scene, objects of observation 0
            single = {k: two[k][:, c:c + 1] for k in ("camera_rotations", "camera_translations", "focals")}
            single.update({k: two[k][:, :1] for k in object_keys})
            per_camera.append(ro.render_from_scene_encoding(cfg, sd, *arguments(single), strides=[4, 8]))
        pair = {"camera_rotations": two["camera_rotations"].reshape(1, 1, 2, 3),
                "camera_translations": two["camera_translations"].reshape(1, 1, 2, 3), "focals": two["focals"].reshape(1, 1, 2)}
        pair.update({k: two[k][:, :1] for k in object_keys})
        model = model.cuda()
        got = model(*[a.cuda() if torch.is_tensor(a) else a for a in arguments(pair)], 0, False, 1000, patch_stride=[4, 8],
                    mode="scene_encodings")
    b = got["coarse"]["global"]["integrated_features"].cpu()
    assert b.shape == (1, 1, 2, 12 * 16 + 6 * 8, SMALL_NETS["features"])
    for c in range(2):
        a = per_camera[c]["coarse"]["global"]["integrated_features"][:, :, 0]
        bad = ((a - b[:, :, c]).abs() > ATOL + RTOL * a.abs()).any(-1).float().mean()
        assert float(bad) <= 0.005, (c, float(bad))
    assert not torch.equal(b[0, 0, 0], b[0, 0, 1])                      # the two cameras see different images
    assert tuple(got["reconstructed_bounding_boxes"].shape) == (1, 1, 2, 4, 4)
    assert tuple(got["coarse"]["object_2"]["weights"].shape) == (1, 1, 2, 240, 32)


def test_empty_ray_list():
    """R = 0 (an empty pixel list): the reference's tensor ops return empty results; so does the renderer, without a launch."""
    cfg = configs.reduced_config(configs.enable_fine(configs.tennis_config()), positions=HIER_POSITIONS, **SMALL_NETS)
    comp = build(cfg).cuda()
    o, d, n, w2o, sty, dfm, ins = [t.cuda() for t in composer_inputs(cfg, synthetic.tennis_scene(seed=5), pixels=grid_pixels(256, 256, 4))]
    with torch.no_grad():
        out = comp(o, d[..., :0, :], n, w2o, sty, dfm, ins, False)
    assert set(out) == {"coarse", "fine", "pytorch_hook"}
    lead = tuple(d.shape[:-2])
    assert tuple(out["fine"]["global"]["integrated_features"].shape) == lead + (0, SMALL_NETS["features"])
    assert tuple(out["fine"]["object_2"]["weights"].shape) == lead + (0, 32)
    assert tuple(out["coarse"]["global"]["weights"].shape) == lead + (0, 8 + 8 + 12 + 12)
    assert tuple(out["coarse"]["object_0"]["opacity"].shape) == lead + (0,)
    comp.train()
    with pytest.raises(ValueError):
        comp(o, d[..., :0, :], n, w2o, sty, dfm, ins, False)     # a differentiable call needs rays


def test_many_samples_per_ray():
    """512 coarse + 512 resampled positions per ray (a 1024-entry list per ray through placement, resampling, the
    1024-key sort and compositing) against the oracle; beyond the kernels' LDS budget the library refuses the call."""
    cfg = configs.reduced_config(configs.enable_fine(configs.tennis_single_player_config()), positions={"player_1": (512, 512)},
                                 **SMALL_NETS)
    comp = build(cfg, alpha_bias=1.0)
    inputs = composer_inputs(cfg, synthetic.single_player_scene(seed=3, image_size=(16, 16)), pixels=grid_pixels(16, 16, 6))
    want, got = run_both(cfg, comp, inputs)
    rep = compare_results(want, got, rtol=RTOL, atol=ATOL)
    bad = {k: f"{v[0]:.3e}" for k, v in rep.items() if not v[1]}
    assert not bad, bad
    assert tuple(got["fine"]["global"]["weights"].shape)[-1] == 1024
    huge = configs.reduced_config(configs.tennis_single_player_config(), positions={"player_1": (8192, 8192)}, **SMALL_NETS)
    with torch.no_grad(), pytest.raises(Exception, match="positions|samples"):
        build(huge).cuda()(*[t.cuda() for t in inputs], False)


# ---------------------------------------------------------------------------------------------------------------------
# sigma-gated feature head (PR_FLAG_GATE_HEAD)
def mixed_sigma(comp, scale=40.0):
    """Densities of both signs inside every object: the default initialisation gives |sigma| ~ 0.05 with a spread of 3e-3,
    i.e. one sign per network; scaling the sigma head's weights spreads the values across zero."""
    with torch.no_grad():
        for name, p in comp.named_parameters():
            if name.endswith("alpha_head.weight"):
                p.mul_(scale)
            elif name.endswith("alpha_head.bias"):
                p.zero_()
    return comp


GATE_CASES = {
    "tennis_hierarchical": (lambda: configs.tennis_config(hierarchical=(16, 32)), lambda: synthetic.tennis_scene(seed=5), 20),
    "tennis_two_frames": (lambda: configs.tennis_config(), lambda: synthetic.tennis_scene(batch=2, observations=2, seed=3), 14),
    "minecraft": (lambda: configs.minecraft_config(), lambda: synthetic.minecraft_scene(seed=9), 28),
    "tennis_c2_64_128": (lambda: configs.tennis_config(hierarchical=(64, 128)), lambda: synthetic.tennis_scene(seed=1234), 16),
}


@pytest.mark.parametrize("name", list(GATE_CASES))
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_sigma_gated_head_is_bit_identical(name, precision):
    """The gated head (feature head only for samples with density > 0, live rows of different tiles sharing head tiles)
    against the ungated kernel: EVERY result field bit for bit; against the oracle at the usual tolerance; and the number
    of rows sent through the head equals the number of evaluated samples with a positive (or NaN) density."""
    make_cfg, make_scene, n = GATE_CASES[name]
    cfg, scene = make_cfg(), make_scene()
    comp = mixed_sigma(build(cfg, alpha_bias=0.0, precision=precision))
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    comp = comp.cuda()
    gin = [v.cuda() for v in inputs]
    with torch.no_grad():
        comp.gate_feature_head = True
        gated = comp(*gin, False, _export=True)
        comp.gate_feature_head = False
        plain = comp(*gin, False, _export=True)
        want = ro.composer_forward(cfg, sd, *inputs, False, stable_merge=True)
    torch.cuda.synchronize()
    skipped = 0
    for ty in [t for t in ("coarse", "fine") if t in gated]:
        for entry in gated[ty]:
            if entry.startswith("_"):
                continue
            for key in ("integrated_features", "opacity", "weights", "depth", "disparity", "integrated_displacements_magnitude"):
                a, b = gated[ty][entry][key], plain[ty][entry][key]
                assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) and torch.equal(torch.isnan(a), torch.isnan(b)), (ty, entry, key)
        ex, ex_plain = gated[ty]["_samples"][0], plain[ty]["_samples"][0]
        assert torch.equal(ex["evaluated"], ex_plain["evaluated"])
        assert torch.equal(ex_plain["head_evaluated"], ex_plain["evaluated"])      # no gate: every evaluated sample
        for k in range(len(ex["sigma"])):
            sigma, slot = ex["sigma"][k], ex["slot"][k]
            live = int(((slot >= 0) & ~(sigma <= 0)).sum())
            assert int(ex["head_evaluated"][k]) == live, (ty, k)
            skipped += int(ex["evaluated"][k]) - live
    assert skipped > 0, "the case does not exercise the gate"
    # against the oracle: with the sigma head scaled 40x the per-sample weights of the hierarchical cases are ill-conditioned
    # (one ulp on the ORACLE's network weights moves its own fine weights by 2e-5 .. 8e-5, measured) - the bit-identity with
    # the ungated kernel above is the gate's test, this one only guards against gross errors
    assert_close(want, gated, rtol=1e-3, atol=5e-4)


def test_sigma_gate_is_ignored_when_noise_is_added():
    """perturb=True adds noise to the densities before the ReLU: a sample with sigma <= 0 can contribute, the gate must not
    apply (and the result still matches the oracle)."""
    cfg, scene = configs.tennis_config(), synthetic.tennis_scene(seed=11)
    comp = mixed_sigma(build(cfg, alpha_bias=0.0))
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 16))
    sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
    rec = {}
    with torch.no_grad():
        torch.manual_seed(3)
        want = ro.composer_forward(cfg, sd, *inputs, True, record_noise=rec, stable_merge=True)
        got = comp.cuda()(*[v.cuda() for v in inputs], True, _noise=rec, _export=True)
    ex = got["coarse"]["_samples"][0]
    assert torch.equal(ex["head_evaluated"], ex["evaluated"])
    assert_close(want, got)


def test_sigma_gated_head_full_frame_counts():
    """Full 256x256 frame of the headline configuration with densities of both signs: thousands of tiles per workgroup,
    pending stacks that fill and drain many times, the final flush; gated == ungated bit for bit."""
    cfg = configs.tennis_config(hierarchical=(64, 128))
    comp = mixed_sigma(build(cfg, alpha_bias=0.0)).cuda()
    gin = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=1234))]
    with torch.no_grad():
        gated = comp(*gin, False, _export=True)
        comp.gate_feature_head = False
        plain = comp(*gin, False)
    for ty in ("coarse", "fine"):
        for entry in ("global", "object_0", "object_2"):
            for key in ("integrated_features", "opacity", "depth"):
                assert torch.equal(torch.nan_to_num(gated[ty][entry][key]), torch.nan_to_num(plain[ty][entry][key])), (ty, entry, key)
        ex = gated[ty]["_samples"][0]
        head, ev = ex["head_evaluated"].sum().item(), ex["evaluated"].sum().item()
        assert 0 < head < ev


# ---------------------------------------------------------------------------------------------------------------------
# observation-driven modes with injected stand-in encoders
import ast
import glob

import numpy as np

from tests.helpers import observation_batch, stand_in_encoders
from playableenvironments_amd import ray_sampling  # noqa: E402

OBS_KEYS = ("observations", "camera_rotations", "camera_translations", "focals", "bounding_boxes", "bounding_boxes_validity",
            "global_frame_indexes", "video_frame_indexes", "video_indexes")
OBS_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "observations", "*.npz")))


def test_observation_fixtures_present():
    assert len(OBS_GOLDEN) >= 2


@pytest.mark.parametrize("path", OBS_GOLDEN, ids=[os.path.basename(p)[:-4] for p in OBS_GOLDEN])
def test_forward_from_observations_matches_reference_fixture(path):
    """EnvironmentModel.forward(mode="observations") on the HIP renderer against what the REFERENCE's
    forward_from_observations returned for the same stand-in encoders, weights and dataset tensors (fixtures recorded in
    the build container by oracle/make_golden.py): the reference's result keys, ground-truth pixels and positions exactly,
    geometry to fp32 rounding, the rendered fields at the renderer's tolerance (the two sides invert their rigid matrices
    differently - closed form vs LU, 2e-6 apart - so at most 1 % of the rays may flip an AABB decision)."""
    z = np.load(path)
    recipe = ast.literal_eval(bytes(z["recipe"]).decode())
    meta = ast.literal_eval(bytes(z["meta"]).decode())
    cfg = recipe_config(recipe)
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, meta["world"]))
    model.object_composer.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
    model = model.eval().cuda()
    args = [torch.from_numpy(z["in/" + k]).cuda() for k in OBS_KEYS]
    with torch.no_grad():
        got = model(*args, **meta["kwargs"])
    assert set(got) == {"coarse", "observations", "positions", "object_rotation_parameters", "object_translation_parameters",
                        "ray_object_distances", "reconstructed_bounding_boxes", "reconstructed_3d_bounding_boxes",
                        "projected_axes", "object_attention", "object_crops", "scene_encoding"}

    def fetch(key):
        node = got
        for part in key.split("/"):
            node = node[part]
        return node.detach().cpu().float()

    checked = 0
    for name in z.files:
        if not name.startswith("out/"):
            continue
        key = name[4:]
        want, have = torch.from_numpy(z[name]).float(), fetch(key)
        assert want.shape == have.shape, key
        if key == "observations" or key.startswith("scene_encoding/camera") or key == "scene_encoding/focals":
            assert torch.equal(want, have), key
        elif key == "positions":     # row / H, col / W: the device's fp32 division may round the last bit differently
            assert torch.allclose(want, have, rtol=0, atol=1e-7), key
        elif key.startswith("coarse/"):
            if key.endswith("weights"):
                want, have = torch.sort(want, -1)[0], torch.sort(have, -1)[0]
            bad = ~torch.isclose(want, have, rtol=1e-3, atol=1e-4, equal_nan=True)
            rays = bad.reshape(bad.shape[:4] + (-1,)).any(-1) if bad.dim() > 4 else bad
            assert float(rays.float().mean()) <= 0.01, (key, float(rays.float().mean()))
        else:
            assert torch.allclose(want, have, rtol=1e-4, atol=1e-5), (key, float((want - have).abs().max()))
        checked += 1
    assert checked >= 45
    # the object_in_scene quirk of the reference (static block repeated static_count times): static^2 + dynamic entries
    helper = model.object_id_helper
    assert got["scene_encoding"]["object_in_scene"].shape[-1] == helper.static_objects_count ** 2 + helper.dynamic_objects_count


DROPIN_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dropin", "*.npz")))


def test_dropin_fixtures_present():
    assert len(DROPIN_GOLDEN) >= 2


@pytest.mark.parametrize("path", DROPIN_GOLDEN, ids=[os.path.basename(p)[:-4] for p in DROPIN_GOLDEN])
def test_trainer_patch_decoder_inputs_match_reference_subclass(path):
    """What the REFERENCE's EnvironmentModelMultiresolutionBackpropagatedDecoder handed to its decoder on the trainer's call
    (training/trainer_multiresolution_backpropagated_decoder.py:52-53: a strided patch, two strides, 64 + 128 channels) -
    recorded in the build container by oracle/check_dropin.py from the reference's own subclass, fold / split_features_by_layer
    glue and autoencoder - against the HIP renderer's ``decoder_features`` for the same dataset tensors, weights, stand-in
    encoders and patch positions (the fixture's positions replace the random draw)."""
    z = np.load(path)
    meta = ast.literal_eval(bytes(z["meta"]).decode())
    base = configs.tennis_config() if meta["world"] == "tennis" else configs.minecraft_config()
    cfg = configs.reduced_config(base, features=192, **meta["reduce"])
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, meta["world"]))
    model.object_composer.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
    model = model.eval().cuda()
    height, width = meta["image_size"]
    positions = torch.from_numpy(z["positions"]).cuda()
    recorded = (torch.round(positions[..., 0] * height) * width + torch.round(positions[..., 1] * width)).to(torch.int64)
    # (row / H, col / W: the device's fp32 division may round the last bit differently)
    assert torch.allclose(ray_sampling.positions_from_indices(recorded, height, width), positions, rtol=0, atol=1e-7)
    model._select_pixels = lambda *a, **k: recorded
    args = [torch.from_numpy(z["in/" + k]).cuda() for k in OBS_KEYS]
    patch = meta["patch_size"]
    counts = [int(z[f"decoder_input_{i}"].shape[-3]) for i in range(len(meta["strides"]))]
    assert counts == [64, 128]
    with torch.no_grad():
        got = model(*args, samples_per_image=patch * patch, perturb=False, shuffle_style=False, patch_size=patch,
                    patch_stride=meta["strides"], align_grid=True, _decoder_features=counts)
    assert torch.allclose(got["positions"], positions, rtol=0, atol=1e-7)
    maps = got["coarse"]["global"]["decoder_features"]
    assert len(maps) == len(counts)
    # rendered fields at the renderer's tolerance; the two sides invert their rigid matrices differently (2e-6 apart), so a
    # ray in a hundred may flip a box decision
    for i, have in enumerate(maps):
        want = torch.from_numpy(z[f"decoder_input_{i}"]).cuda()
        assert tuple(have.shape) == tuple(want.shape), (have.shape, want.shape)
        bad = ~torch.isclose(want, have, rtol=1e-3, atol=1e-4)
        pixels = bad.any(-3)
        assert float(pixels.float().mean()) <= 0.02, (i, float(pixels.float().mean()), float((want - have).abs().max()))
    want = torch.from_numpy(z["integrated_features"]).cuda()
    have = got["coarse"]["global"]["integrated_features"]
    bad = (~torch.isclose(want, have, rtol=1e-3, atol=1e-4)).any(-1)
    assert float(bad.float().mean()) <= 0.02


def _observation_model(world, size, batch=2, observations=2):
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    base = configs.tennis_config() if world == "tennis" else configs.minecraft_config()
    cfg = configs.reduced_config(base, **small)
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, world))
    torch.manual_seed(0)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    scene = (synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene)(batch=batch, observations=observations,
                                                                                         seed=7, image_size=size)
    batch_tensors = {k: v.cuda() for k, v in observation_batch(scene).items()}
    return cfg, model.cuda(), batch_tensors


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_observation_mode_is_scene_encoding_mode_plus_encoders(world):
    """mode="observations" = the injected encoders + the scene-encoding render: rendering the scene encoding it returns
    through mode="scene_encodings" reproduces its composer results bit for bit (same rays, same kernels), the training
    patch carries the ground-truth pixels of its positions, and render_full_frame_from_observations folds to (H, W)."""
    size = (48, 64)
    cfg, model, b = _observation_model(world, size)
    model.eval()
    args = [b[k] for k in OBS_KEYS]
    with torch.no_grad():
        torch.manual_seed(2)
        out = model(*args, samples_per_image=0, perturb=False, patch_stride=[4, 8])
        se = out["scene_encoding"]
        K = model.object_id_helper.objects_count
        again = model(se["camera_rotations"], se["camera_translations"], se["focals"], size, se["object_rotation_parameters"],
                      se["object_translation_parameters"], se["object_style"], se["object_deformation"],
                      torch.ones_like(se["object_in_scene"][..., :K]), 0, False, patch_stride=[4, 8], mode="scene_encodings")
        for entry in ("global", "object_0", f"object_{K - 1}"):
            for key in ("integrated_features", "opacity", "depth", "weights"):
                assert torch.equal(out["coarse"][entry][key], again["coarse"][entry][key]), (entry, key)
        assert torch.equal(out["reconstructed_bounding_boxes"], again["reconstructed_bounding_boxes"])
        # training-shaped patch call (device-side random centre, noise, shuffled style; eval-mode BatchNorm so that a patch
        # that misses an object cannot trip the batch-statistics check): positions and ground-truth pixels belong together
        torch.manual_seed(3)
        patch = model(*args, samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8], shuffle_style=True)
        full = model.render_full_frame_from_observations(*args, False)
    rays = 8 * 8 + 4 * 4
    assert tuple(patch["coarse"]["global"]["integrated_features"].shape)[-2:] == (rays, 32)
    assert tuple(patch["observations"].shape)[-2:] == (rays, 3) and tuple(patch["positions"].shape)[-2:] == (rays, 2)
    rows = (patch["positions"][..., 0] * size[0]).round().long()
    cols = (patch["positions"][..., 1] * size[1]).round().long()
    obs = b["observations"]
    lead = obs.shape[:3]
    flat = obs.reshape(-1, 3, size[0] * size[1])
    picked = torch.gather(flat, 2, (rows * size[1] + cols).reshape(-1, 1, rays).expand(-1, 3, -1)).transpose(1, 2)
    assert torch.equal(picked.reshape(lead + (rays, 3)), patch["observations"])
    assert tuple(patch["ray_object_distances"].shape) == tuple(lead) + (rays, model.object_id_helper.objects_count)
    assert tuple(full["coarse"]["global"]["integrated_features"].shape) == tuple(lead) + (size[0], size[1], 32)
    assert tuple(full["observations"].shape) == tuple(lead) + (size[0], size[1], 3)
    assert torch.equal(full["observations"], obs.movedim(-3, -1))


def test_observation_mode_gradients_reach_the_encoders():
    """Differentiable call through the observation mode: the loss on the rendered features back-propagates through
    pr_render_backward into the injected encoders (poses through w2o, style, deformation) and the composer.  (Frozen
    BatchNorm statistics: a random patch that misses an object would trip train mode's batch-statistics check.)"""
    cfg, model, b = _observation_model("minecraft", (48, 64), batch=1, observations=3)
    model.eval()
    torch.manual_seed(5)
    out = model(*[b[k] for k in OBS_KEYS], samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8])
    out["coarse"]["global"]["integrated_features"].square().mean().backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert any(n.startswith("object_encoders.") and float(g.abs().sum()) > 0 for n, g in grads.items())
    assert any(n.startswith("object_parameters_encoders.") and float(g.abs().sum()) > 0 for n, g in grads.items())
    assert any(n.startswith("object_composer.") and float(g.abs().sum()) > 0 for n, g in grads.items())
    assert all(torch.isfinite(g).all() for g in grads.values())


def test_pose_matrices_kernel_matches_the_torch_path():
    """pr_pose_matrices (Euler angles -> [R t; 0 1] and its rigid inverse in one launch; the no-graph path of the cameras and
    the objects of every call) against euler_to_matrix / rigid_inverse - the torch ops pinned against the reference's
    homogeneous_rotation_translation + torch.inverse (tests/golden/host/pose_math_*.npz, check_against_reference.py)."""
    g = torch.Generator().manual_seed(9)
    rot = ((torch.rand(3, 2, 5, 3, generator=g) - 0.5) * 6.0).cuda()
    tr = (torch.randn(3, 2, 5, 3, generator=g) * 20).cuda()
    m, inv = em.pose_matrices(rot, tr)
    ref_m = em.euler_to_matrix(rot, tr)
    ref_inv = em.rigid_inverse(ref_m)
    assert m.shape == ref_m.shape == (3, 2, 5, 4, 4)
    assert float((m - ref_m).abs().max()) <= 1e-6 and float((inv - ref_inv).abs().max()) <= 2e-5     # |t| ~ 60: 1 ulp = 4e-6
    eye = torch.eye(4, device="cuda").expand_as(m)
    assert float((torch.matmul(m, inv) - eye).abs().max()) <= 2e-5
    # with a graph: the same kernel as an autograd node, its backward (pr_pose_matrices_backward) against torch.autograd
    # through euler_to_matrix / rigid_inverse; a translation broadcast over a leading dimension gets the summed gradient
    probes = [torch.randn(m.shape, generator=g).cuda() for _ in range(2)]
    tr_b = tr[:1].clone()
    grads = []
    for fn in (em.pose_matrices, lambda r, t: (lambda mm: (mm, em.rigid_inverse(mm)))(em.euler_to_matrix(r, torch.broadcast_to(t, r.shape)))):
        rot_g, tr_g = rot.clone().requires_grad_(True), tr_b.clone().requires_grad_(True)
        mg, ig = fn(rot_g, tr_g)
        assert mg.requires_grad and mg.shape == ref_m.shape
        ((mg * probes[0]).sum() + (ig * probes[1]).sum()).backward()
        grads.append((rot_g.grad.clone(), tr_g.grad.clone()))
    for a, b in zip(grads[0], grads[1]):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_projection_kernel_matches_the_torch_path(world):
    """pr_project_points (bounding boxes + box points, object axes: one launch each on the no-graph path) against the torch
    ops of compute_object_bounding_boxes / compute_object_axes_projection, which check_against_reference.py pins against
    the reference (model/environment_model.py:234-404) - including cameras that see an object partly behind them."""
    cfg = configs.tennis_config() if world == "tennis" else configs.minecraft_config()
    model = em.EnvironmentModel(cfg).cuda().eval()
    scene_fn = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
    sc = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in scene_fn(batch=2, observations=3, seed=13, image_size=(96, 128)).items()}
    rot, tr = sc["object_rotation_parameters"], sc["object_translation_parameters"].clone()
    tr[0, 0, :, -1] = sc["camera_translations"][0, 0, 0]          # one object around the camera: points on both sides of it
    w2o, o2w = model.compute_transformation_matrix_w2o_o2w(rot, tr)
    c2w, w2c = em.pose_matrices(sc["camera_rotations"], sc["camera_translations"])
    focals = sc["focals"] * cfg["data"]["focal_length_multiplier"]
    with torch.no_grad():
        boxes, points = model.compute_object_bounding_boxes(o2w, w2c, focals, 96, 128)
        axes = model.compute_object_axes_projection(o2w, w2c, focals, 96, 128)
    o2w_g = o2w.clone().requires_grad_(True)                      # the differentiable formulation: tensor ops
    ref_boxes, ref_points = model.compute_object_bounding_boxes(o2w_g, w2c, focals, 96, 128, _lazy=True)
    ref_axes = model.compute_object_axes_projection(o2w_g, w2c, focals, 96, 128, _lazy=True)
    assert ref_boxes.requires_grad and not boxes.requires_grad
    for got, want in ((boxes, ref_boxes), (points, ref_points)):
        assert got.shape == want.shape and float((got - want.detach()).abs().max()) <= 1e-5
    # (the unclamped axes of the object that sits ON the camera are a division by ~0: compared for the other objects)
    close = torch.isclose(axes, ref_axes.detach(), rtol=1e-4, atol=1e-4)
    assert axes.shape == ref_axes.shape and bool(close[..., :-1].all()) and bool(close[1:].all())
    # with a graph wanted, the call returns the kernel's values and defers the tensor ops to the backward pass: same values as
    # the no-graph call bit for bit, the gradients of the differentiable formulation exactly
    o2w_l = o2w.clone().requires_grad_(True)
    lazy_boxes, lazy_points = model.compute_object_bounding_boxes(o2w_l, w2c, focals, 96, 128)
    lazy_axes = model.compute_object_axes_projection(o2w_l, w2c, focals, 96, 128)
    assert lazy_boxes.requires_grad and torch.equal(lazy_boxes, boxes) and torch.equal(lazy_points, points)
    assert torch.equal(torch.nan_to_num(lazy_axes), torch.nan_to_num(axes))
    g = torch.Generator().manual_seed(3)
    pb, pp = torch.randn(boxes.shape, generator=g).cuda(), torch.randn(points.shape, generator=g).cuda()
    pa = torch.randn(axes.shape, generator=g).cuda() * close.float()
    ((ref_boxes * pb).sum() + (ref_points * pp).sum() + (ref_axes * pa).sum()).backward()
    ((lazy_boxes * pb).sum() + (lazy_points * pp).sum() + (lazy_axes * pa).sum()).backward()
    assert float(torch.nan_to_num(o2w_g.grad).abs().max()) > 0
    assert torch.equal(torch.nan_to_num(o2w_l.grad), torch.nan_to_num(o2w_g.grad))
    assert 0.0 < float(boxes.min()) or float(boxes.max()) <= 1.0


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_two_cameras_equal_two_single_camera_renders(world):
    """cameras_count > 1 (the reference's (..., observations, cameras, ...) layout: one scene, several cameras; the object
    tensors carry no camera dimension): every field of camera c equals the single-camera render with that camera, bit for bit -
    renderer outputs and the projected boxes / box points / axes alike."""
    cfg = configs.reduced_config(configs.tennis_config() if world == "tennis" else configs.minecraft_config(), **SMALL_NETS)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.0, bender_scale=1e4)
    model.eval().cuda()
    fn = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
    size = (64, 96)
    a = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in fn(batch=2, observations=2, seed=3, image_size=size).items()}
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in fn(batch=2, observations=2, seed=4, image_size=size).items()}

    def run(rotations, translations, focals):
        with torch.no_grad():
            return model(rotations, translations, focals, size, a["object_rotation_parameters"], a["object_translation_parameters"],
                         a["object_style"], a["object_deformation"], a["object_in_scene"], 0, False, patch_stride=[4, 8],
                         mode="scene_encodings")
    cams = [(s["camera_rotations"], s["camera_translations"], s["focals"]) for s in (a, b)]
    both = run(*[torch.cat([cams[0][i], cams[1][i]], dim=2) for i in range(3)])
    assert both["coarse"]["global"]["integrated_features"].shape[:3] == (2, 2, 2)
    for c in range(2):
        one = run(*cams[c])
        for name in [f"object_{k}" for k in range(4)] + ["global"]:
            for key in ("integrated_features", "opacity", "depth", "weights"):
                assert torch.equal(torch.nan_to_num(both["coarse"][name][key][:, :, c:c + 1]), torch.nan_to_num(one["coarse"][name][key])), (c, name, key)
        for key in ("reconstructed_bounding_boxes", "reconstructed_3d_bounding_boxes", "projected_axes"):
            assert torch.equal(both[key][:, :, c:c + 1], one[key]), (c, key)


def test_camera_rays_are_differentiable_for_learnable_cameras():
    """camera_rays with a graph (c2w / focals require gradients): the HIP kernel's values, and a backward pass equal to
    torch.autograd through the closed form d_cam = ((col - W/2)/f, -(row - H/2)/f, -1), d = R d_cam, o = t, normal = -R[:, 2]."""
    g = torch.Generator().manual_seed(3)
    c2w = em.euler_to_matrix(torch.rand(2, 3, 3, generator=g) - 0.5, torch.randn(2, 3, 3, generator=g)).cuda()
    focals = (torch.rand(2, 3, generator=g) * 100 + 150).cuda()
    height, width = 40, 56
    idx = torch.randint(0, height * width, (2, 3, 37), generator=g).cuda()
    rows, cols = ray_sampling.split_indices(idx, width)
    probes = [torch.randn(s, generator=g).cuda() for s in ((2, 3, 3), (2, 3, 37, 3), (2, 3, 3))]
    a_c2w, a_f = c2w.clone().requires_grad_(True), focals.clone().requires_grad_(True)
    outs = em.camera_rays(a_c2w, a_f, height, width, rows, cols)
    with torch.no_grad():
        plain = em.camera_rays(c2w, focals, height, width, rows, cols)
    assert all(torch.equal(x, y) for x, y in zip(outs, plain))
    sum((o * p).sum() for o, p in zip(outs, probes)).backward()
    b_c2w, b_f = c2w.clone().requires_grad_(True), focals.clone().requires_grad_(True)
    f = b_f.unsqueeze(-1)
    d_cam = torch.stack([(cols.float() - width / 2) / f, -(rows.float() - height / 2) / f, -torch.ones_like(cols.float())], -1)
    ref = (b_c2w[..., :3, 3], torch.einsum("...ij,...rj->...ri", b_c2w[..., :3, :3], d_cam), -b_c2w[..., :3, 2])
    assert all(torch.allclose(x, y, rtol=1e-5, atol=1e-5) for x, y in zip(outs, ref))
    sum((o * p).sum() for o, p in zip(ref, probes)).backward()
    assert torch.allclose(a_c2w.grad, b_c2w.grad, rtol=1e-4, atol=1e-4 * float(b_c2w.grad.abs().max()))
    assert torch.allclose(a_f.grad, b_f.grad, rtol=1e-4, atol=1e-4 * float(b_f.grad.abs().max()))


def test_learnable_camera_offsets_receive_gradients_through_the_render():
    """enable_camera_parameters_offsets (model/layers/camera_parameters_storage.py): in training mode the observation-driven
    forward adds the per-frame offsets to the measured cameras, the rays are generated with a graph, and the loss on the
    rendered features reaches the offset table through pr_render_backward's ray gradients - for the (frame, camera) entries
    the batch used and for those only; in evaluation mode the offsets are zero and nothing requires the ray gradients."""
    size = (48, 64)
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.tennis_config(), **small)
    cfg["model"]["enable_camera_parameters_offsets"] = True
    cfg["model"]["camera_parameters_memory_size"] = 16
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, "tennis"))
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda()
    b = {k: v.cuda() for k, v in observation_batch(synthetic.tennis_scene(batch=2, observations=2, seed=7, image_size=size)).items()}
    b["global_frame_indexes"] = torch.tensor([[3, 4], [9, 3]], device="cuda")
    with torch.no_grad():
        model.camera_parameters_offsets.table.normal_(0, 1e-3)
    args = [b[k] for k in OBS_KEYS]
    model.eval()
    model.camera_parameters_offsets.train()          # offsets on; BatchNorm keeps its running statistics (small random patches)
    torch.manual_seed(5)
    out = model(*args, samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8])
    out["coarse"]["global"]["integrated_features"].square().mean().backward()
    grad = model.camera_parameters_offsets.table.grad
    assert grad is not None and torch.isfinite(grad).all()
    used = sorted(set(b["global_frame_indexes"].flatten().tolist()))
    touched = sorted(torch.nonzero(grad.abs().sum(-1) > 0).flatten().tolist())
    assert touched == used, (touched, used)
    assert all(float(grad[used][:, c].abs().max()) > 0 for c in range(7))       # rotation, translation and focal offsets
    # the offsets change the render (translations scaled by 10, focals by 1000)
    with torch.no_grad():
        torch.manual_seed(5)
        base = model(*args, samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8])
        model.camera_parameters_offsets.eval()
        torch.manual_seed(5)
        plain = model(*args, samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8])
    # (the render without a graph builds its pose matrices with pr_pose_matrices, the one with a graph with torch ops: ulps apart)
    assert torch.allclose(base["coarse"]["global"]["integrated_features"], out["coarse"]["global"]["integrated_features"].detach(),
                          rtol=1e-3, atol=1e-3)
    assert not torch.equal(base["coarse"]["global"]["integrated_features"], plain["coarse"]["global"]["integrated_features"])


def test_image_decoder_hook():
    """config["model"]["image_decoder"] (compute_decoded_image, environment_model.py:708-741): the decoder CNN and the grid
    sampler are injected modules; the result gains coarse.global.decoded_images in the observation and the scene-encoding mode;
    without the modules the call says what to inject, and fine models raise like the reference.  (Values against the
    reference's own method: check_against_reference.py.)"""
    cfg, model, b = _observation_model("tennis", (48, 64))
    model.eval()
    model.use_image_decoder = True
    args = [b[k] for k in OBS_KEYS]
    with torch.no_grad(), pytest.raises(RuntimeError, match="inject"):
        model(*args, samples_per_image=0, perturb=False, patch_stride=[4, 8])

    class Sampler(torch.nn.Module):
        def forward(self, feats, positions):
            return torch.cat([feats, positions], dim=-1).mean(dim=-2)

    class Decoder(torch.nn.Module):
        def forward(self, grid):
            return (grid * 2.0).unsqueeze(-1).unsqueeze(-1).expand(list(grid.shape) + [2, 3])
    model.set_image_decoder(Decoder(), Sampler())
    with torch.no_grad():
        out = model(*args, samples_per_image=0, perturb=False, patch_stride=[4, 8])
        se = out["scene_encoding"]
        again = model(se["camera_rotations"], se["camera_translations"], se["focals"], (48, 64), se["object_rotation_parameters"],
                      se["object_translation_parameters"], se["object_style"], se["object_deformation"], se["object_in_scene"],
                      0, False, patch_stride=[4, 8], mode="scene_encodings")
    feats, pos = out["coarse"]["global"]["integrated_features"], out["positions"]
    want = Decoder()(Sampler()(feats, pos))
    assert torch.equal(out["coarse"]["global"]["decoded_images"], want)
    assert torch.equal(again["coarse"]["global"]["decoded_images"], want)
    with pytest.raises(Exception, match="fine features"):
        model.compute_decoded_image({"coarse": out["coarse"], "fine": out["coarse"]}, pos)


def test_missing_encoders_raise():
    cfg = configs.tennis_config()
    model = em.EnvironmentModel(cfg).eval().cuda()
    b = {k: v.cuda() for k, v in observation_batch(synthetic.tennis_scene(image_size=(48, 64))).items()}
    with pytest.raises(RuntimeError, match="inject"):
        model(*[b[k] for k in OBS_KEYS], 0, False)


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_consistency_forwards(world):
    """forward_pose_consistency / forward_keypoint_consistency: the reference's result structure, and expected positions equal
    to ObjectComposer.forward_expected_positions on the rays the forwards sample (same seed)."""
    size = (48, 64)
    cfg, model, b = _observation_model(world, size, batch=2, observations=3)
    model.eval()
    with torch.no_grad():
        se = model(*[b[k] for k in OBS_KEYS], mode="observations_scene_encoding_only")
        lead = list(b["observations"].shape[:3])
        g = torch.Generator().manual_seed(1)
        flow = ((torch.rand(lead + [2, size[0], size[1]], generator=g) - 0.5) * 0.05).cuda()
        keypoints = torch.rand(lead + [17, 3, 2], generator=g).cuda()
        common = [b[k] for k in OBS_KEYS[1:]] + [se["object_style"], se["object_deformation"],
                                                 se["object_rotation_parameters"], se["object_translation_parameters"]]
        torch.manual_seed(4)
        pose = model(flow, *common, 30, False, mode="pose_consistency")
        torch.manual_seed(4)
        pose_again = model.forward_pose_consistency(flow, *common, 30, False)
        torch.manual_seed(6)
        kp = model(b["observations"], *common, keypoints, b["bounding_boxes_validity"], 20, False, mode="keypoint_consistency")
    assert set(pose) == {"coarse", "pytorch_backward_hook"} and set(pose["coarse"]) == {"dynamic_object_0", "dynamic_object_1"}
    for name, (previous, following) in pose["coarse"].items():
        assert tuple(previous[0].shape) == (2, 2, 1, 30, 3) and tuple(following[0].shape) == (2, 2, 1, 30, 3)
        assert tuple(previous[1].shape) == (2, 2, 1, 30)
        assert torch.equal(previous[0], pose_again["coarse"][name][0][0])
        assert torch.isfinite(previous[0]).all() and torch.isfinite(following[0]).all()
    assert pose["pytorch_backward_hook"] is pose["coarse"]["dynamic_object_0"][0]
    for name, (positions, confidence, opacity, sampled) in kp["coarse"].items():
        assert tuple(positions.shape) == (2, 3, 1, 20, 3) and tuple(confidence.shape) == (2, 3, 1, 20)
        assert tuple(opacity.shape) == (2, 3, 1, 20) and tuple(sampled.shape) == (2, 3, 1, 20, 2)
        box = model.object_composer.object_models_coarse[2].bounding_box
        hit = opacity > 1e-3                    # expected positions of rays that hit the object lie inside its box
        lo = torch.as_tensor([r[0] for r in cfg["model"]["object_models"][2]["bounding_box"]], device=positions.device) - 1e-3
        hi = torch.as_tensor([r[1] for r in cfg["model"]["object_models"][2]["bounding_box"]], device=positions.device) + 1e-3
        inside = ((positions >= lo) & (positions <= hi)).all(-1)
        assert bool(inside[hit].all())


from tests.test_cpu import CONSISTENCY_GOLDEN  # noqa: E402


@pytest.mark.parametrize("path", CONSISTENCY_GOLDEN, ids=[os.path.basename(p)[:-4] for p in CONSISTENCY_GOLDEN])
def test_consistency_forwards_match_reference_fixture(path, monkeypatch):
    """forward_pose_consistency / forward_keypoint_consistency on the HIP renderer against what the REFERENCE's methods returned
    (tests/golden/consistency, recorded by oracle/make_golden.py) for the same dataset tensors, optical flow, keypoints, weights
    and - replayed from the fixture - the same randomly drawn pixels: expected positions, opacities, confidences and sampled
    positions value for value (the renderer's tolerance; positions are opacity-weighted means inside a box of a few units)."""
    from tests.helpers import run_consistency_fixture
    assert len(CONSISTENCY_GOLDEN) >= 2
    z = np.load(path)
    meta = ast.literal_eval(bytes(z["meta"]).decode())
    cfg = recipe_config(ast.literal_eval(bytes(z["recipe"]).decode()))
    model = em.EnvironmentModel(cfg, *stand_in_encoders(cfg, meta["world"]))
    model.object_composer.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
    pairs = run_consistency_fixture(z, model.eval().cuda(), "cuda", monkeypatch)
    hit = 0
    for k, (want, got) in pairs.items():
        assert want.shape == got.shape, k
        if k.endswith(("confidence", "sampled")):
            assert torch.equal(want, got), k
        else:
            assert torch.allclose(want, got, rtol=1e-4, atol=2e-5), (k, float((want - got).abs().max()))
        if k.endswith("opacity"):
            hit += int((want > 1e-3).sum())
    assert len(pairs) == 16 and hit > 50          # the drawn rays do meet the players


# ---------------------------------------------------------------------------------------------------------------------
# multi-rank paths on the single GPU of the test box: RCCL with one rank, gloo with two ranks on one
# device (RCCL refuses two ranks on the same GPU); the 8-GPU runs are the driver's
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tennis_model_and_args(batch, size=(32, 48)):
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.tennis_config(), **small)
    torch.manual_seed(0)        # BEFORE the construction: every rank of a multi-process test must draw the same weights
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.0, bender_scale=1e4)
    scene = synthetic.tennis_scene(batch=batch, observations=1, seed=33, image_size=size)
    args = [scene[k].cuda() for k in ("camera_rotations", "camera_translations", "focals")] + [size] + \
           [scene[k].cuda() for k in ("object_rotation_parameters", "object_translation_parameters", "object_style",
                                      "object_deformation", "object_in_scene")]
    return model.eval().cuda(), args


@pytest.mark.parametrize("mode", ["adam", "adam_l2", "adamw", "maximize", "capturable"])
def test_arena_adam_matches_torch_adam(mode):
    """parallel.ArenaAdam (pr_adam_step: one full-grid launch per flat tensor) against torch.optim.Adam / AdamW on the same
    parameters and gradients over several steps: parameters and both moment buffers to fp32 round-off (the formulas and their
    order are torch's; torch's own fused kernel contracts multiply-adds, so bit identity is not the yardstick), odd sizes and
    unaligned views included; optimiser state dictionaries move between the two."""
    from playableenvironments_amd import parallel
    torch.manual_seed(3)
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    if mode == "adam_l2":
        kw["weight_decay"] = 0.05
    if mode == "maximize":
        kw["maximize"] = True
    theirs_cls = torch.optim.AdamW if mode == "adamw" else torch.optim.Adam
    if mode == "adamw":
        kw["weight_decay"] = 0.05
    sizes = [1 << 20, 4099, 3]
    storage = torch.randn(sum(sizes) + 1, device="cuda")
    base, at = [], 1                                            # (views at an odd offset: not 16-byte aligned)
    for n in sizes:
        base.append(storage[at:at + n])
        at += n
    mine_p = [torch.nn.Parameter(b.clone()) for b in base]
    theirs_p = [torch.nn.Parameter(b.clone()) for b in base]
    mine_kw = dict(kw, decoupled_weight_decay=(mode == "adamw"), capturable=(mode == "capturable"))
    mine = parallel.ArenaAdam(mine_p, **mine_kw)
    theirs = theirs_cls(theirs_p, **kw)
    for step in range(6):
        for a, b in zip(mine_p, theirs_p):
            g = torch.randn_like(a) * (10.0 ** (step - 3))
            a.grad, b.grad = g.clone(), g.clone()
        versions = [p._version for p in mine_p]
        mine.step()
        theirs.step()
        assert all(p._version > v for p, v in zip(mine_p, versions))          # the raw-pointer update is visible to autograd
        if step == 2:        # the state of one optimiser continues in the other
            state = mine.state_dict()
            assert sorted(state["state"][0]) == ["exp_avg", "exp_avg_sq", "step"]
            other = parallel.ArenaAdam(mine_p, **mine_kw)
            other.load_state_dict(theirs.state_dict())
            for k in ("exp_avg", "exp_avg_sq"):
                x, y = other.state[mine_p[0]][k], mine.state[mine_p[0]][k]
                assert torch.allclose(x, y, rtol=1e-5, atol=1e-6 * float(y.abs().max())), (k, float((x - y).abs().max()))
            assert float(other.state[mine_p[0]]["step"]) == float(mine.state[mine_p[0]]["step"]) == 3.0
    torch.cuda.synchronize()
    for a, b in zip(mine_p, theirs_p):
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-5, atol=1e-5), float((a.detach() - b.detach()).abs().max())
        for k in ("exp_avg", "exp_avg_sq"):
            x, y = mine.state[a][k], theirs.state[b][k]
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-6 * float(y.abs().max())), (k, float((x - y).abs().max()))
    assert float(mine.state[mine_p[0]]["step"]) == 6.0 and mine.state[mine_p[0]]["step"].is_cuda == (mode == "capturable")
    with pytest.raises(RuntimeError, match="contiguous fp32 device tensors"):
        bad = torch.nn.Parameter(torch.zeros(4))
        bad.grad = torch.zeros(4)
        parallel.ArenaAdam([bad]).step()


@pytest.mark.parametrize("optimizer", ["torch_fused_separate", "torch_fused_arena", "arena_adam", "torch_foreach"])
def test_optimizer_steps_reach_the_next_render(optimizer):
    """Training with the optimisers a trainer may build, ``torch.optim.Adam(fused=True)`` included: torch's fused optimisers update
    the parameter storages WITHOUT moving the tensors' version counters (tools/perf/dbg_version_counters.py: values change,
    ``_version`` stays) - the composer's packed MFMA weight copies and the recorded evaluation frames therefore also key on
    ``weights_epoch`` (moved by every backward pass that produced parameter gradients, and again by the step of the optimiser that
    holds the parameters: every other iteration renders between ``backward()`` and ``step()``).  After every step the training render,
    an evaluation render and a RECORDED evaluation frame must show the new weights: equal to a freshly built composer holding
    the same state."""
    from playableenvironments_amd import parallel
    from playableenvironments_amd.frame_graph import SCENE_KEYS
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.0, bender_scale=1e4)
    model = model.cuda()
    comp = model.object_composer
    size = (32, 48)
    sc = {k: v.cuda() for k, v in synthetic.tennis_scene(batch=1, seed=9, image_size=size).items() if torch.is_tensor(v)}
    args = [sc[k] for k in SCENE_KEYS[:3]] + [size] + [sc[k] for k in SCENE_KEYS[3:]]
    arena = parallel.flatten_parameters(comp) if "arena" in optimizer else None
    params = list(comp.parameters())
    if optimizer == "torch_fused_separate":
        opt = torch.optim.Adam(params, lr=1e-2, fused=True)
    elif optimizer == "torch_fused_arena":
        opt = torch.optim.Adam([arena], lr=1e-2, fused=True)
    elif optimizer == "arena_adam":
        opt = parallel.ArenaAdam([arena], lr=1e-2)
    else:
        opt = torch.optim.Adam(params, lr=1e-2, foreach=True)

    def fresh_eval():
        other = em.EnvironmentModel(cfg).cuda().eval()
        other.load_state_dict(model.state_dict())
        other.frame_replay = None
        with torch.no_grad():
            return other(*args, 0, False, 0, patch_stride=[4, 8], mode="scene_encodings")["coarse"]["global"]["integrated_features"]
    previous = None
    for step in range(4):
        model.train()
        opt.zero_grad(set_to_none=True)
        out = model(*args, 0, False, 0, patch_stride=[4, 8], mode="scene_encodings")
        out["coarse"]["global"]["integrated_features"].square().mean().backward()
        if arena is not None:
            parallel.flat_gradient(arena, comp)
        if step % 2 == 1:
            # a validation / logging render BETWEEN backward() and the optimiser step packs the pre-step weights: the epoch has to
            # move again once the optimiser that owns the parameters has stepped (torch's optimiser post-step hook)
            model.eval()
            with torch.no_grad():
                before = model(*args, 0, False, 0, patch_stride=[4, 8], mode="scene_encodings")["coarse"]["global"]["integrated_features"]
            assert torch.equal(before, fresh_eval()), (optimizer, step)      # (pre-step weights, this iteration's BatchNorm statistics)
        opt.step()
        model.eval()
        with torch.no_grad():        # (the default frame_replay: eager, recording, replay, replay ... across the optimiser steps)
            got = model(*args, 0, False, 0, patch_stride=[4, 8], mode="scene_encodings")["coarse"]["global"]["integrated_features"]
            again = model(*args, 0, False, 0, patch_stride=[4, 8], mode="scene_encodings")["coarse"]["global"]["integrated_features"]
        want = fresh_eval()
        assert torch.equal(got, want) and torch.equal(again, want), (optimizer, step, float((got - want).abs().max()))
        assert previous is None or not torch.equal(want, previous), (optimizer, step)       # the step did change the render
        previous = want


def test_data_parallel_wrapper_matches_the_plain_call():
    """``nn.DataParallel(model)`` - how the reference wraps its model unconditionally (train.py:61; called as ``self.model(...)`` in
    training/trainer.py:148,630): ``device_ids=[0]`` (pass-through) and ``device_ids=[0, 0]`` (TWO replicas on this box's one GPU:
    scatter along the batch, ``replicate`` with its shallow ``__dict__`` copies, two threads, gather) against the unwrapped call -
    evaluation results bit for bit, and a train-mode forward + backward against the per-chunk calls the replicas stand for
    (train-mode BatchNorm normalises per replica, like the reference's).  Every replica has its own pointer structs, packed weights
    and workspace (``ObjectComposer._replicate_for_data_parallel``); the original's caches are untouched."""
    model, args = _tennis_model_and_args(batch=2)
    kw = dict(patch_stride=[4, 8], mode="scene_encodings")
    model.frame_replay = None
    with torch.no_grad():
        plain = model(*args, 0, False, 1200, **kw)
    comp = model.object_composer
    for ids in ([0], [0, 0]):
        wrapped = torch.nn.DataParallel(model, device_ids=ids)
        structs_before, workspace_before, packed_before = dict(comp._structs), comp._workspace, dict(comp._packed)
        with torch.no_grad():
            got = wrapped(*args, 0, False, 1200, **kw)
        torch.cuda.synchronize()
        if len(ids) > 1:      # the replicas used their OWN pointer structs / packed weights / workspace: the original's are untouched
            assert comp._workspace is workspace_before and comp._structs == structs_before and comp._packed == packed_before
        a, b = dict(_flat_tensors(got)), dict(_flat_tensors(plain))
        assert sorted(k for k in a if k != "pytorch_hook") == sorted(k for k in b if k != "pytorch_hook")
        for k in b:
            if k == "pytorch_hook":          # (the reference's dummy tensor: one per replica after the gather)
                continue
            x, y = torch.nan_to_num(a[k].float(), nan=-7.0), torch.nan_to_num(b[k].float(), nan=-7.0)
            assert x.shape == y.shape and torch.equal(x, y), (ids, k, float((x - y).abs().max()))
    # training: forward + backward through the wrapper = the sum over the replicas' chunks (one frame each)
    model.train()
    params = [p for p in comp.parameters() if p.requires_grad]

    def loss_of(out):
        g = out["coarse"]["global"]
        return (g["integrated_features"] ** 2).mean() + g["opacity"].mean() + 0.1 * g["depth"].mean()

    def half(i):
        return [a[i:i + 1] if torch.is_tensor(a) else a for a in args]
    for p in params:
        p.grad = None
    per_chunk = [model(*half(i), 0, False, 1200, **kw) for i in range(2)]
    sum(loss_of(o) for o in per_chunk).backward()
    want = [p.grad.clone() for p in params]
    want_features = torch.cat([o["coarse"]["global"]["integrated_features"] for o in per_chunk], dim=0).detach()
    for p in params:
        p.grad = None
    wrapped = torch.nn.DataParallel(model, device_ids=[0, 0])
    out = wrapped(*args, 0, False, 1200, **kw)
    assert torch.equal(out["coarse"]["global"]["integrated_features"].detach(), want_features)
    # (a mean over both frames = half the sum of the per-frame means)
    g = out["coarse"]["global"]
    loss = 2.0 * ((g["integrated_features"] ** 2).mean() + g["opacity"].mean() + 0.1 * g["depth"].mean())
    loss.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for p, w in zip(params, want):
        assert p.grad is not None
        scale = float(w.abs().max()) + 1e-12
        worst = max(worst, float((p.grad - w).abs().max()) / scale)
    assert worst < 1e-5, worst
    model.eval()
    # ... and the optimiser step that follows reaches the ORIGINAL's renders: the replicas (which ran the backward pass) report their
    # parameter gradients to the module they were copied from - torch's fused Adam moves no version counter, and the reference's
    # evaluators render through ``model.module`` outside the wrapper (evaluation/evaluator.py:58)
    epoch = comp.weights_epoch
    with torch.no_grad():
        before = model(*args, 0, False, 1200, **kw)["coarse"]["global"]["integrated_features"].clone()
    opt = torch.optim.Adam(params, lr=1e-2, fused=True)
    model.train()
    for p in params:
        p.grad = None
    out = wrapped(*args, 0, False, 1200, **kw)
    (out["coarse"]["global"]["integrated_features"] ** 2).mean().backward()
    assert comp.weights_epoch > epoch
    opt.step()
    model.eval()
    fresh = em.EnvironmentModel(model.config).cuda().eval()
    fresh.load_state_dict(model.state_dict())
    fresh.frame_replay = None
    with torch.no_grad():
        after = model(*args, 0, False, 1200, **kw)["coarse"]["global"]["integrated_features"]
        want_after = fresh(*args, 0, False, 1200, **kw)["coarse"]["global"]["integrated_features"]
    assert not torch.equal(after, before) and torch.equal(after, want_after), float((after - want_after).abs().max())


def test_backward_passes_of_two_threads_on_their_own_streams():
    """Two host threads, each with its own stream and its own model (same weights), run training steps at the same time: the backward
    pass forks into the library's second stream, which is keyed by (device, CALLER stream) - one shared per device would interleave the
    two threads' forks and joins (and be pulled into a thread-local graph capture of either).  Every thread's gradients equal the ones
    the same call produces alone (up to the order of the kernels' atomic sums)."""
    import threading
    models, argss = [], []
    for _ in range(2):
        model, args = _tennis_model_and_args(batch=2)
        model.frame_replay = None
        models.append(model.train())
        argss.append(args)
    kw = dict(patch_stride=[4, 8], mode="scene_encodings")

    def step(model, args):
        for p in model.parameters():
            p.grad = None
        out = model(*args, 0, False, 1200, **kw)
        g = out["coarse"]["global"]
        ((g["integrated_features"] ** 2).mean() + g["opacity"].mean() + 0.1 * g["depth"].mean()).backward()
        return [p.grad.clone() for p in model.object_composer.parameters() if p.grad is not None]
    alone = step(models[0], argss[0])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    results, errors = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(6):
                    results[i] = step(models[i], argss[i])
                streams[i].synchronize()
        except Exception as error:      # noqa: BLE001
            errors.append(error)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        assert len(results[i]) == len(alone)
        worst = max(float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12) for a, b in zip(results[i], alone))
        assert worst < 1e-5, (i, worst)


def test_replaced_parameters_buffers_and_modules_are_noticed():
    """The composer caches parameter lists and raw-pointer structs of its module tree; the tree's classes (modules.Tracked) report
    every re-registration, without process-wide hooks: a replaced ``nn.Parameter`` object, a replaced BatchNorm buffer,
    ``load_state_dict(assign=True)``, a submodule swapped for a module of a FOREIGN class (a stock ``nn.Linear``: the tree is then
    walked on every call) - each time the next render equals a freshly built composer holding the same state."""
    from playableenvironments_amd import modules
    assert not torch.nn.modules.module._global_parameter_registration_hooks      # nothing process-wide is installed
    assert not torch.nn.modules.module._global_buffer_registration_hooks and not torch.nn.modules.module._global_module_registration_hooks
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=20000, alpha_bias=2.0, bender_scale=1e4)
    comp = comp.cuda().eval()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=3), pixels=grid_pixels(256, 256, 24))]

    def fresh_render():
        other = ObjectComposer(cfg).cuda().eval()
        other.load_state_dict(comp.state_dict())
        with torch.no_grad():
            return other(*inputs, False)["coarse"]["global"]["integrated_features"]

    def render():
        with torch.no_grad():
            return comp(*inputs, False)["coarse"]["global"]["integrated_features"]
    base = render()
    assert torch.equal(base, fresh_render())
    nerf = comp.object_models_coarse[2].nerf_model
    epoch = modules.REGISTRATION_EPOCH[0]
    torch.nn.Linear(3, 3), torch.nn.BatchNorm1d(3)          # other people's modules do not move the epoch
    assert modules.REGISTRATION_EPOCH[0] == epoch
    # (1) a new Parameter OBJECT under an existing name
    nerf.backbone_layers[1].weight = torch.nn.Parameter(nerf.backbone_layers[1].weight.detach() * 1.5)
    assert modules.REGISTRATION_EPOCH[0] > epoch
    one = render()
    assert not torch.equal(one, base) and torch.equal(one, fresh_render())
    # (2) a replaced BatchNorm buffer
    bn = nerf.features_head[1].ada_in.normalization
    bn.running_var = bn.running_var.detach() * 2.0 + 0.1
    two = render()
    assert not torch.equal(two, one) and torch.equal(two, fresh_render())
    # (3) load_state_dict(assign=True): every tensor object replaced
    sd = {k: (v.detach().clone() * (1.01 if v.is_floating_point() and "weight" in k else 1)) for k, v in comp.state_dict().items()}
    comp.load_state_dict(sd, assign=True)
    three = render()
    assert not torch.equal(three, two) and torch.equal(three, fresh_render())
    # (4) a submodule of a foreign class: no caching of that tree, results still follow its tensors
    stock = torch.nn.Linear(nerf.alpha_head.in_features, 1).cuda()
    with torch.no_grad():
        stock.weight.copy_(nerf.alpha_head.weight * 0.5), stock.bias.copy_(nerf.alpha_head.bias)
    nerf.alpha_head = stock
    assert not comp._tree_is_tracked(comp.object_models_coarse[2]) and comp._tree_is_tracked(comp.object_models_coarse[0])
    four = render()
    assert not torch.equal(four, three) and torch.equal(four, fresh_render())
    stock.weight = torch.nn.Parameter(stock.weight.detach() * 3.0)        # (a registration this package cannot see)
    five = render()
    assert not torch.equal(five, four) and torch.equal(five, fresh_render())
    # (5) values written through .data (its own version counter: the parameter's does not move, nothing observable changes) are the
    # one update the composer cannot see - ObjectComposer.weights_changed() is the documented call for them
    comp.object_models_coarse[0].nerf_model.backbone_layers[0].weight.data.mul_(1.25)
    comp.weights_changed()
    six = render()
    assert not torch.equal(six, five) and torch.equal(six, fresh_render())


def test_render_sharded_and_async_gather_rccl_world1():
    """torch.distributed backend "nccl" (= RCCL) with a single rank: render_sharded degenerates to the plain render and the
    pipelined feature gather returns the rank's own maps - the code path of the multi-GPU runs, through RCCL."""
    import torch.distributed as dist
    from playableenvironments_amd.parallel import AsyncFeatureGather, gather_ray_shards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        model, args = _tennis_model_and_args(batch=2)
        with torch.no_grad():
            whole = model(*args, 0, False, mode="scene_encodings")
        for shard in ("frames", "rays", "tiles", "auto"):
            out = model.render_sharded(*args, False, shard=shard)
            assert torch.equal(out["coarse"]["global"]["integrated_features"], whole["coarse"]["global"]["integrated_features"])
        feats = whole["coarse"]["global"]["integrated_features"]
        stacked = torch.empty_like(feats)
        dist.all_gather_into_tensor(stacked, feats.contiguous())           # a real RCCL collective on the device
        assert torch.equal(stacked, feats)
        gather = AsyncFeatureGather(depth=1)
        gather.submit(feats)
        gather.submit(feats * 2)
        done = gather.drain()
        assert len(done) == 2 and torch.equal(done[1], feats * 2)
        assert torch.equal(gather_ray_shards(feats, feats.size(0), 0, dst=None), feats)
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def _rank_worker(rank, world, port, results):
    import torch.distributed as dist
    from playableenvironments_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        cases = ((3, "frames"), (1, "rays"), (2, "auto"), (1, "tiles"), (1, "auto"), (2, "tiles")) if world == 2 else \
                ((8, "frames"), (11, "frames"), (1, "tiles"), (1, "rays"), (3, "auto"))
        for batch, shard in cases:
            model, args = _tennis_model_and_args(batch=batch)
            with torch.no_grad():
                whole = model(*args, 0, False, patch_stride=[4, 8], mode="scene_encodings")
            out = model.render_sharded(*args, False, patch_stride=[4, 8], shard=shard, fields=("integrated_features", "opacity"))
            for field in ("integrated_features", "opacity"):
                ok = ok and torch.equal(out["coarse"]["global"][field], whole["coarse"]["global"][field])
            # the evaluator's flow: only the consumer rank receives the maps
            only = model.render_sharded(*args, False, patch_stride=[4, 8], shard=shard, fields=("integrated_features",), dst=0)
            ok = ok and ((only is None) if rank else torch.equal(only["coarse"]["global"]["integrated_features"],
                                                                  whole["coarse"]["global"]["integrated_features"]))
        # data-parallel training step: every rank differentiates its own frame, the flat gradient buffer is reduced by a collective
        # started inside backward() (parallel.OverlappedGradientAllReduce); result = the mean of the ranks' single-rank gradients
        model, _ = _tennis_model_and_args(batch=1)
        comp = model.object_composer.train()
        overlap = parallel.OverlappedGradientAllReduce(comp)
        cfg = comp.config

        def gradients(frame, reduce):
            scene = synthetic.tennis_scene(seed=900 + frame, image_size=(32, 48))
            inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(32, 48, 10))]
            for q in comp.parameters():
                q.grad = None
            state = {k: v.clone() for k, v in comp.state_dict().items()}
            out = comp(*inputs, False)
            out["coarse"]["global"]["integrated_features"].square().mean().backward()
            started = overlap.finish() if reduce else len(overlap.pending)
            if not reduce:
                for work, _ in overlap.pending:      # (single-rank reference passes: complete and discard the collective)
                    work.wait()
                overlap.pending = []
            comp.load_state_dict(state)              # (train-mode BatchNorm moved the running statistics)
            return torch.cat([q.grad.reshape(-1) for q in comp.parameters()]).clone(), started
        mine, started = gradients(rank, reduce=True)
        ok = ok and started == 1
        overlap.remove()
        singles = [gradients(r, reduce=False)[0] for r in range(world)]
        want = torch.stack(singles).mean(0)
        scale = float(want.abs().max())
        ok = ok and float((mine - want).abs().max()) <= 2e-5 * scale
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_render_sharded_ranks_on_one_gpu_gloo(world):
    """Two / eight ranks (gloo) sharing the box's GPU - what the driver's 8-GPU run does with one GPU each: frame shards (ragged:
    3 frames over 2 ranks, 11 over 8), contiguous ray shards and 8 x 8-pixel tiles dealt round robin (a single frame's default),
    gathered on every rank and on rank 0 only - the HIP renderer's assembled feature maps equal the unsharded render bit for bit
    on every rank; and a data-parallel step whose gradient all-reduce starts inside backward() returns the mean of the ranks'
    single-rank gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert all(results[r] for r in range(world)), dict(results)


def _run_bench(argv, env, tmp_path):
    """bench.py as the driver runs it: stdout must hold exactly ONE line, compact json under bench.LINE_BUDGET bytes with the contract
    keys; returns (line, full record read back from --full-json)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full_path = os.path.join(str(tmp_path), "bench_full.json")
    # the ranks share this box's ONE GPU with the test process: hand back what its allocator caches (after ~200 tests: most of the
    # 288 GB - eight ranks of ~25 GB each then ran out of memory in their decoder leg)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv + ["--full-json", full_path], env=env,
                          capture_output=True, text=True, timeout=1500)
    if proc.returncode != 0:       # the first error of any rank, not only the launcher's summary at the end
        import re
        first = re.search(r"Traceback|what\(\)|HIP error|Error:|terminate called", proc.stderr)
        at = first.start() if first else max(0, len(proc.stderr) - 3000)
        raise AssertionError(f"bench.py exited with {proc.returncode}\n--- first error ---\n{proc.stderr[at:at + 3000]}\n--- tail ---\n{proc.stderr[-1500:]}")
    out_lines = [line for line in proc.stdout.splitlines() if line.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), proc.stdout[-2000:]      # nothing but the line on stdout
    assert len(out_lines[0].encode()) < 4096, len(out_lines[0])
    line = json.loads(out_lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "distributed", "full"):
        assert key in line, key
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["frac"] > 0 and "workload" in line["config"]
    with open(full_path) as f:
        full = json.load(f)
    assert full["value"] == line["value"] and full["ms_per_step"] == line["ms_per_step"]
    return line, full


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_self_launches_ranks(tmp_path, ranks):
    """`python bench.py --gpus N` (no launcher): bench.py starts torch.distributed.run itself; with the PR_BENCH_DEVICE /
    PR_BENCH_BACKEND knobs all N ranks share this box's GPU over gloo.  N = 8 is the driver's scaling run in miniature: one
    compact JSON line (< 4 KB with eight rank_devices entries), both collectives of the feature exchange, the identical-frame
    secondary, per-rank times - and every N-rank leg done within five minutes even with the eight ranks queueing on ONE GPU."""
    import time
    env = dict(os.environ, PR_BENCH_DEVICE="0", PR_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # every rank's training legs keep ~28 GB (capacity-sized forward / backward workspaces of a 3 x 2880-ray step + the decoder leg): eight of
    # them need most of the GPU they share HERE with this test process.  When the process's own live tensors leave less than that, the run
    # skips the training legs (they are covered by the 2-rank run and by the 8-rank data-parallel test) instead of running out of memory.
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    light = free < ranks * 32e9
    t0 = time.perf_counter()
    line, result = _run_bench(["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--image", "64", "--no-cpu-baseline",
                               "--no-split-precision"] + (["--no-train-step"] if light else []), env, tmp_path)
    elapsed = time.perf_counter() - t0
    assert elapsed < 300.0, elapsed
    assert line["n_gpus"] == ranks and line["distributed"]["world_size"] == ranks and line["distributed"]["backend"] == "gloo"
    assert len(line["distributed"]["rank_devices"]) == ranks
    assert line["value"] > 0 and line["steps"] == 2
    assert line["feature_gather"]["all_gather_ms"] > 0 and line["feature_gather"]["gather_dst0_GB_per_s"] > 0
    assert line["summary"]["identical_frames_mrays"] > 0
    assert len(result["distinct_frames"]["shipped_p72"]["per_rank_ms"]) == ranks
    if not light:
        assert line["summary"]["train_step_ms"] > 0
        assert result["train_step"]["parallelism"].startswith(f"data parallel x{ranks}")
    # the exchange on its own, both collectives, and the identical-frame secondary beside the per-rank frames of the headline
    gather = result["feature_gather"]
    assert gather["world_size"] == ranks and gather["bytes_per_rank"] == 64 * 64 * 192 * 4
    assert gather["all_gather"]["ms"] > 0 and gather["gather_dst0"]["receiving_ranks"] == 1
    assert result["identical_frames"]["value"] > 0 and "seed 1234 + r" in result["config"]["workload"]
    assert len(result["library_sha256"]) == 64


def test_bench_collectives_through_rccl_with_one_rank(tmp_path):
    """PR_BENCH_FORCE_DIST=1: bench.py initialises the process group with backend "nccl" (= RCCL) for its single rank and runs
    every collective of the multi-rank legs (barriers, the max over ranks, all_gather_into_tensor and gather of the feature
    map, the identical-frame secondary) - the code path of `--gpus 8`, which only the driver can run with 8 devices."""
    env = dict(os.environ, PR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    line, result = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--image", "64", "--no-cpu-baseline", "--no-split-precision",
                               "--no-minecraft", "--no-distinct-frames", "--no-shard-balance"], env, tmp_path)
    assert line["distributed"]["backend"] == "nccl" and line["distributed"]["world_size"] == 1
    assert line["distributed"]["nccl_version"][0].isdigit(), line["distributed"]
    assert result["feature_gather"]["all_gather"]["ms"] > 0 and result["feature_gather"]["gather_dst0"]["ms"] > 0
    assert result["identical_frames"]["value"] > 0 and result["train_step"]["value"] > 0
    assert result["train_step_with_decoder"]["maps_route"]["ms_per_step"] > 0


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_psnr_against_oracle_headline_config(precision):
    """BASELINE.json's "PSNR vs reference": the reference's formula (evaluation/metrics/psnr.py:10-34,
    -10 log10(mse + 1e-8)) on fine.global.integrated_features of the headline configuration (bench weights and scene),
    rescaled to [0, 1] with the oracle's range - >= 60 dB for the exact kernel AND for the split-precision kernel (SURVEY's
    40 dB bound for 16-bit inputs would be too lax for a kernel that passes the fp32 tolerance)."""
    cfg = configs.tennis_config(hierarchical=(64, 128))
    torch.manual_seed(0)
    comp = ObjectComposer(cfg)
    synthetic.randomize_module_state(comp, seed=0, step=60000, alpha_bias=0.0, bender_scale=1e4)
    comp.precision = precision
    comp.eval()
    inputs = composer_inputs(cfg, synthetic.tennis_scene(seed=1234), pixels=grid_pixels(256, 256, 20))
    sd = {k: v.detach().clone() for k, v in comp.state_dict().items()}
    with torch.no_grad():
        want = ro.composer_forward(cfg, sd, *inputs, False, stable_merge=True)
        got = comp.cuda()(*[v.cuda() for v in inputs], False)
    for ty in ("coarse", "fine"):
        a = want[ty]["global"]["integrated_features"]
        b = got[ty]["global"]["integrated_features"].cpu()
        lo, hi = float(a.min()), float(a.max())
        value = ro.psnr((a - lo) / (hi - lo), (b - lo) / (hi - lo))
        assert value >= 60.0, (ty, value)


# ---------------------------------------------------------------------------------------------------------------------
# producers of the renderer's inputs: roi_pool kernel, encoders, Batch
def _random_boxes(g, count, images, height, width):
    centre = torch.rand((count, 2), generator=g) * torch.tensor([width, height])
    half = torch.rand((count, 2), generator=g) * torch.tensor([width, height]) * 0.4
    boxes = torch.cat([centre - half, centre + half], dim=-1)                       # some leave the image on every side
    boxes[0] = torch.tensor([-20.0, -20.0, -5.0, -5.0])                            # entirely outside: every bin empty
    boxes[1] = torch.tensor([3.2, 4.4, 3.3, 4.45])                                 # a single pixel
    boxes[2] = torch.tensor([0.0, 0.0, width - 1.0, height - 1.0])                 # the whole image
    index = torch.randint(0, images, (count, 1), generator=g).float()
    return torch.cat([index, boxes], dim=-1)


@pytest.mark.parametrize("shape,out_size", [((3, 3, 36, 64), (8, 8)), ((2, 5, 72, 128), (8, 32)), ((1, 1, 9, 7), (3, 2))])
def test_roi_pool_kernel_matches_oracle(shape, out_size):
    """pr_roi_pool_forward / _backward against the CPU restatement of torchvision.ops.roi_pool: values and argmax positions
    bit for bit (max pooling copies elements), input gradients up to the summation order of the atomic scatter."""
    from oracle import roi_pool_oracle as rp
    from playableenvironments_amd import encoders
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(shape, generator=g)
    x[0, 0, :3, :3] = 1.5                                                            # ties: the first maximum wins
    boxes = _random_boxes(g, 24, shape[0], shape[2], shape[3])
    want, arg = rp.roi_pool(x, boxes, out_size)
    xg = x.cuda().requires_grad_(True)
    got = encoders.roi_pool(xg, boxes.cuda(), out_size)
    assert torch.equal(got.detach().cpu(), want)
    grad_out = torch.randn(want.shape, generator=g)
    got.backward(grad_out.cuda())
    want_grad = rp.roi_pool_backward(grad_out, arg, boxes, shape)
    assert torch.allclose(xg.grad.cpu(), want_grad, rtol=1e-5, atol=1e-5)
    assert encoders.roi_pool(xg, boxes[:0].cuda(), out_size).shape == (0, shape[1]) + tuple(out_size)
    with pytest.raises(RuntimeError, match="device tensors"):
        encoders.roi_pool(x, boxes, out_size)


@pytest.mark.parametrize("world", ["tennis", "minecraft"])
def test_encoders_on_gpu_match_cpu_modules(world):
    """The encoder modules on the GPU (HIP roi_pool + MIOpen convolutions) against the same modules on the CPU with the
    crop served by the roi_pool restatement - the configuration check_against_reference.py pins to the reference's modules."""
    from oracle import roi_pool_oracle as rp
    from playableenvironments_amd import encoders
    from playableenvironments_amd.environment_model import euler_to_matrix, rigid_inverse
    cfg = configs.tennis_config(encoders=True) if world == "tennis" else configs.minecraft_config(encoders=True)
    for entry in cfg["model"]["object_encoders"] + cfg["model"]["object_parameters_encoder"]:
        if "input_size" in entry:      # small crops: the CPU side loops over every bin in Python
            entry["input_size"] = [32, 64] if entry["architecture"].endswith("v5") else [32, 32]
    scene = (synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene)(batch=2, observations=2, seed=13, image_size=(72, 128))
    b = observation_batch(scene)
    w2c = rigid_inverse(euler_to_matrix(b["camera_rotations"], b["camera_translations"]))
    focals = b["focals"] * cfg["data"]["focal_length_multiplier"]
    torch.manual_seed(1)
    enc, par = encoders.create_encoders(cfg)
    original = encoders.roi_pool
    for m, module in enumerate(enc + par):
        module.eval()
        is_encoder = m < len(enc)
        entry = (cfg["model"]["object_encoders"] if is_encoder else cfg["model"]["object_parameters_encoder"])[m % len(enc)]
        if is_encoder:
            args = [b["observations"], b["bounding_boxes"][..., 0], b["camera_rotations"], b["camera_translations"],
                    b["global_frame_indexes"], b["video_frame_indexes"], b["video_indexes"]]
        elif "static" in entry["architecture"]:
            args = [b["observations"]]
        else:
            n = entry["objects_count"]
            args = [b["observations"], w2c, b["camera_rotations"], focals, b["bounding_boxes"][..., :n], b["bounding_boxes_validity"][..., :n]]
        encoders.roi_pool = lambda x, boxes, size, spatial_scale=1.0: rp.roi_pool(x.detach(), boxes.detach(), size, spatial_scale)[0]
        try:
            with torch.no_grad():
                want = module(*args)
        finally:
            encoders.roi_pool = original
        with torch.no_grad():
            got = module.cuda()(*[a.cuda() for a in args])
        for a, c in zip(want, got):
            assert a.shape == c.shape
            scale = 1.0 + float(a.detach().abs().max()) if a.numel() else 1.0
            assert float((a.detach() - c.cpu()).abs().max()) <= 2e-4 * scale, (entry["architecture"], float((a - c.cpu()).abs().max()))


def test_native_observation_pipeline_end_to_end():
    """The reference's whole Phase-2 forward on this package alone: Batch (one pinned arena, one asynchronous copy) ->
    EnvironmentModel built from a full configuration (its own encoders: HIP roi_pool crops + PyTorch-ROCm ResNets, pose
    estimators) -> HIP renderer; a training-shaped differentiable call sends gradients into the encoders' convolutions."""
    from playableenvironments_amd import batching
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.minecraft_config(encoders=True), **small)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda().eval()
    scene = synthetic.minecraft_scene(batch=2, observations=2, seed=17, image_size=(96, 128))
    b = observation_batch(scene)
    batch = batching.batch_from_tensors(b["observations"], b["camera_rotations"], b["camera_translations"], b["focals"], b["bounding_boxes"],
                                        b["bounding_boxes_validity"], b["global_frame_indexes"], b["video_frame_indexes"], b["video_indexes"])
    batch.pin_memory()
    assert batch.observations.is_pinned()
    args = batch.observation_mode_arguments()                    # one arena copy on the side stream
    assert all(a.is_cuda for a in args) and torch.equal(args[0].cpu(), b["observations"]) and args[5].dtype == torch.bool
    with torch.no_grad():
        out = model(*args, samples_per_image=0, perturb=False, patch_stride=[4, 8])
        full = model.render_full_frame_from_observations(*args, False)
    rays = 24 * 32 + 12 * 16
    assert tuple(out["coarse"]["global"]["integrated_features"].shape) == (2, 2, 1, rays, 32)
    assert torch.isfinite(out["coarse"]["global"]["integrated_features"]).all()
    assert tuple(out["scene_encoding"]["object_style"].shape) == (2, 2, 32, 4)
    assert tuple(out["object_crops"][2].shape) == (2, 2, 1, 3, 64, 64) and tuple(out["object_attention"][3].shape) == (2, 2, 1, 1, 32, 32)
    assert tuple(full["coarse"]["global"]["opacity"].shape) == (2, 2, 1, 96, 128)
    # the players stand where their boxes' feet hit the ground plane (y = 0)
    assert float(out["object_translation_parameters"][..., 1, 2:].abs().max()) == 0.0
    torch.manual_seed(1)
    grad_out = model(*args, samples_per_image=10, perturb=True, patch_size=8, patch_stride=[4, 8])
    grad_out["coarse"]["global"]["integrated_features"].square().mean().backward()
    touched = [n for n, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().sum()) > 0]
    assert any(n.startswith("object_encoders.2.conv1") for n in touched)
    assert any(n.startswith("object_encoders.0.final_backbone") for n in touched)
    assert any(n.startswith("object_parameters_encoders.2.rotation_head") for n in touched)
    assert any(n.startswith("object_composer.object_models_coarse.2.nerf_model") for n in touched)


# ---------------------------------------------------------------------------------------------------------------------
# renderer -> decoder wire format emitted by the compositing kernel
@pytest.mark.parametrize("mode", ["full_frame", "training_patch"])
def test_decoder_layout_emission_matches_wire_format(mode):
    """``decoder_features`` written by k_composite (channels-first map per stride, the stride's own channel slice) against
    the reference's glue restated in wire_format.py (fold / split by stride, split_features_by_layer, permute), applied to
    the ray-major integrated features of the SAME call: bit for bit; and the gradient that arrives through the maps equals
    the gradient that arrives through the ray-major tensor."""
    from playableenvironments_amd import wire_format as wf
    small = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)
    cfg = configs.reduced_config(configs.minecraft_config(), **small)
    torch.manual_seed(0)
    model = em.EnvironmentModel(cfg)
    synthetic.randomize_module_state(model.object_composer, seed=0, step=20000, alpha_bias=2.5, bender_scale=1e4)
    model = model.cuda().eval()
    size = (48, 64)
    scene = synthetic.minecraft_scene(batch=2, observations=2, seed=23, image_size=size)
    sc = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in scene.items()}
    args = [sc[k] for k in ("camera_rotations", "camera_translations", "focals")] + [size] + \
           [sc[k] for k in ("object_rotation_parameters", "object_translation_parameters", "object_style", "object_deformation",
                            "object_in_scene")]
    counts = [8, 24]
    if mode == "full_frame":
        kw = dict(samples_per_image=0, perturb=False, patch_stride=[4, 8])
    else:
        kw = dict(samples_per_image=10, perturb=False, patch_size=8, patch_stride=[4, 8])
    for a in args:
        if torch.is_tensor(a) and a.is_floating_point():
            a.requires_grad_(False)
    style = args[6].clone().requires_grad_(True)
    args[6] = style
    torch.manual_seed(3)
    out = model(*args[:9], kw["samples_per_image"], kw["perturb"], patch_size=kw.get("patch_size", 0), patch_stride=kw["patch_stride"],
                _decoder_features=counts, mode="scene_encodings")
    feats = out["coarse"]["global"]["integrated_features"]                      # (2, 2, 1, R, 32)
    maps = out["coarse"]["global"]["decoder_features"]
    if mode == "full_frame":
        folded = wf.fold_strided_grid_samples(feats, [4, 8], size, dim=3)        # [(.., h, w, F)] per stride
    else:
        folded = [wf.strided_patch_ray_samples_to_patch(t) for t in wf.split_strided_patch_ray_samples(feats, 8, [4, 8])]
    # channels-first slice of stride i (the patch helper already returns (..., C, p, p); the grid fold returns (..., h, w, C))
    chw = (lambda f, b, c: f[..., b:b + c].movedim(-1, -3)) if mode == "full_frame" else (lambda f, b, c: f[..., b:b + c, :, :])
    begin = 0
    for i, (m, f) in enumerate(zip(maps, folded)):
        want = chw(f, begin, counts[i])
        assert tuple(m.shape) == tuple(want.shape), (m.shape, want.shape)
        assert torch.equal(m, want), i
        begin += counts[i]
    # gradients: a loss on the maps == the same loss on the corresponding slices of the ray-major features
    g = torch.Generator().manual_seed(1)
    probes = [torch.randn(m.shape, generator=g).cuda() for m in maps]
    sum((m * p).sum() for m, p in zip(maps, probes)).backward()
    via_maps = style.grad.clone()
    style.grad = None
    torch.manual_seed(3)
    again = model(*args[:9], kw["samples_per_image"], kw["perturb"], patch_size=kw.get("patch_size", 0), patch_stride=kw["patch_stride"],
                  mode="scene_encodings")
    feats = again["coarse"]["global"]["integrated_features"]
    if mode == "full_frame":
        folded = wf.fold_strided_grid_samples(feats, [4, 8], size, dim=3)
    else:
        folded = [wf.strided_patch_ray_samples_to_patch(t) for t in wf.split_strided_patch_ray_samples(feats, 8, [4, 8])]
    begin, loss = 0, 0.0
    for i, f in enumerate(folded):
        loss = loss + (chw(f, begin, counts[i]) * probes[i]).sum()
        begin += counts[i]
    loss.backward()
    assert float(via_maps.abs().max()) > 0
    # (style gradients are accumulated with atomics: equal up to the summation order)
    assert torch.allclose(via_maps, style.grad, rtol=1e-4, atol=1e-6 * float(style.grad.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------
# in-kernel noise (PR_FLAG_DEVICE_NOISE): Philox4x32-10 streams instead of materialised torch.rand / torch.randn tensors
def _noise_fill(seed, kind, ty, obj, shape):
    import ctypes as C
    from playableenvironments_amd import _lib
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().pr_noise_fill(C.c_uint64(seed), kind, ty, obj, out.numel(), out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "pr_noise_fill")
    return out


def test_generated_noise_distributions():
    """The generator's streams: uniform and normal moments, a Kolmogorov-Smirnov bound, no correlation between neighbouring
    elements or between streams, reproducibility, and dependence on the seed."""
    n = 1 << 20
    u = _noise_fill(1234, 0, 0, 0, (n,)).double().cpu()
    z = _noise_fill(1234, 3, 0, 2, (n,)).double().cpu()
    z2 = _noise_fill(1234, 3, 1, 2, (n,)).double().cpu()          # another stream
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    assert abs(float(z.mean())) < 4e-3 and abs(float(z.var()) - 1.0) < 6e-3
    assert abs(float((z ** 3).mean())) < 2e-2 and abs(float((z ** 4).mean()) - 3.0) < 6e-2
    srt, _ = torch.sort(u)
    ks = float((srt - torch.arange(1, n + 1, dtype=torch.float64) / n).abs().max())
    assert ks < 1.95 / n ** 0.5                                     # 0.1 % level
    phi = 0.5 * (1 + torch.erf(torch.sort(z)[0] / 2 ** 0.5))
    assert float((phi - torch.arange(1, n + 1, dtype=torch.float64) / n).abs().max()) < 1.95 / n ** 0.5
    corr = lambda a, b: float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
    assert abs(corr(u[:-1], u[1:])) < 5e-3 and abs(corr(z[:-1], z[1:])) < 5e-3 and abs(corr(z, z2)) < 5e-3
    assert torch.equal(z, _noise_fill(1234, 3, 0, 2, (n,)).double().cpu())
    assert not torch.equal(z, _noise_fill(1235, 3, 0, 2, (n,)).double().cpu())


@pytest.mark.parametrize("name", ["tennis_hierarchical", "minecraft"])
def test_generated_noise_render_equals_explicit_replay(name):
    """A perturbed render with in-kernel noise == the same render with the explicit-noise path fed the tensors pr_noise_fill
    writes for that seed (bit for bit), and therefore == the oracle replaying them (fp32 tolerance): the in-kernel generator
    sits behind the same arithmetic the parity tests cover."""
    make_cfg, make_scene, n, bias = CASES[name]
    cfg, scene = make_cfg(), make_scene()
    comp = build(cfg, alpha_bias=bias).cuda()
    inputs = composer_inputs(cfg, scene, pixels=grid_pixels(scene["image_size"][0], scene["image_size"][1], n))
    gin = [v.cuda() for v in inputs]
    with torch.no_grad():
        torch.manual_seed(77)
        generated = comp(*gin, True)
        seed = comp.last_noise_seed
        again_seed = None
        torch.manual_seed(77)
        again = comp(*gin, True)
        again_seed = comp.last_noise_seed
    assert seed == again_seed
    lay = ro.ObjectLayout(cfg)
    K = lay.objects_count
    N, R = 1, gin[1].size(-2)
    for v in gin[1].shape[:-2]:
        N *= v
    pc = [cfg["model"]["object_models"][lay.model_of_object[k]]["positions_count_coarse"] for k in range(K)]
    pf = [cfg["model"]["object_models"][lay.model_of_object[k]]["positions_count_fine"] for k in range(K)]
    fine = "fine" in generated
    explicit = {}
    for k in range(K):
        explicit[f"jitter_{k}"] = _noise_fill(seed, 0, 0, k, (N, R, pc[k]))
        explicit[f"alpha_{k}"] = _noise_fill(seed, 1, 0, k, (N, R, pc[k]))
        if fine:
            explicit[f"pdf_{k}"] = _noise_fill(seed, 2, 0, k, (N, R, pf[k]))
        explicit[f"int_coarse_{k}"] = _noise_fill(seed, 3, 0, k, (N, R, pc[k]))
        if fine:
            explicit[f"int_fine_{k}"] = _noise_fill(seed, 3, 1, k, (N, R, pc[k] + pf[k]))
    explicit["int_coarse_global"] = _noise_fill(seed, 4, 0, 0, (N, R, sum(pc)))
    if fine:
        explicit["int_fine_global"] = _noise_fill(seed, 4, 1, 0, (N, R, sum(pc) + sum(pf)))
    with torch.no_grad():
        replay = comp(*gin, True, _noise=explicit)
        lead = list(gin[1].shape[:-2])
        cpu_noise = {k: v.cpu().reshape(lead + list(v.shape[1:])) for k, v in explicit.items()}
        sd = {k: v.detach().cpu().clone() for k, v in comp.state_dict().items()}
        want = ro.composer_forward(cfg, sd, *inputs, True, noise=cpu_noise, stable_merge=True)
    for ty in [t for t in ("coarse", "fine") if t in generated]:
        for entry in generated[ty]:
            for key in ("integrated_features", "opacity", "depth", "weights"):
                a, b, c = generated[ty][entry][key], replay[ty][entry][key], again[ty][entry][key]
                assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), (ty, entry, key)
                assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(c)), (ty, entry, key)
    assert_close(want, generated, rtol=1e-3, atol=1e-4) if fine else assert_close(want, generated)
    with torch.no_grad():                              # another seed, another image
        other = comp(*gin, True)
    assert comp.last_noise_seed != seed
    assert not torch.equal(other["coarse"]["global"]["opacity"], generated["coarse"]["global"]["opacity"])


def test_generated_noise_training_gradients_match_explicit_replay():
    """Training step with in-kernel noise (jitter, density noise, Hutchinson probes regenerated by pr_render_backward) against
    the same step with the generator's values replayed as explicit tensors: identical results and identical gradients."""
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    scene = synthetic.minecraft_scene(batch=2, seed=19)
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 12))]
    results = []
    seed = None
    for mode in ("device", "explicit"):
        comp = build(cfg, alpha_bias=3.0).cuda().train()
        o, d, n, w2o, sty, dfm, ins = [v.clone() for v in inputs]
        sty.requires_grad_(True)
        w2o.requires_grad_(True)
        lay = ro.ObjectLayout(cfg)
        K, N, R = lay.objects_count, 2, d.size(-2)
        pc = [cfg["model"]["object_models"][lay.model_of_object[k]]["positions_count_coarse"] for k in range(K)]
        if mode == "device":
            torch.manual_seed(5)
            out = comp(o, d, n, w2o, sty, dfm, ins, True)
            seed = comp.last_noise_seed
        else:
            explicit = {}
            for k in range(K):
                explicit[f"jitter_{k}"] = _noise_fill(seed, 0, 0, k, (N, R, pc[k]))
                explicit[f"alpha_{k}"] = _noise_fill(seed, 1, 0, k, (N, R, pc[k]))
                explicit[f"int_coarse_{k}"] = _noise_fill(seed, 3, 0, k, (N, R, pc[k]))
                explicit[f"div_coarse_{k}"] = _noise_fill(seed, 5, 0, k, (N, R, pc[k], 3))
            explicit["int_coarse_global"] = _noise_fill(seed, 4, 0, 0, (N, R, sum(pc)))
            out = comp(o, d, n, w2o, sty, dfm, ins, True, _noise=explicit)
        g = out["coarse"]["global"]
        # (the divergence term exercises the probes regenerated inside the second-order pass of pr_render_backward)
        (g["integrated_features"].square().mean() + g["opacity"].mean() + g["depth"].mean() * 0.01 +
         g["integrated_divergence"].mean() * 10 + out["coarse"]["object_2"]["integrated_divergence"].mean()).backward()
        grads = {name: p.grad.clone() for name, p in comp.named_parameters() if p.grad is not None}
        results.append((out, sty.grad.clone(), w2o.grad.clone(), grads))
    (a, sa, wa, ga), (b, sb, wb, gb) = results
    for key in ("integrated_features", "opacity", "depth", "integrated_divergence", "integrated_displacements_magnitude"):
        assert torch.equal(a["coarse"]["global"][key], b["coarse"]["global"][key]), key
    assert float(a["coarse"]["global"]["integrated_divergence"].detach().abs().max()) > 0
    # (the pose / style gradients are accumulated with atomics: equal up to the summation order, run to run)
    close = lambda x, y: torch.allclose(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()))
    assert close(sa, sb) and close(wa, wb)
    assert set(ga) == set(gb) and all(close(ga[k], gb[k]) for k in ga)


def test_autograd_node_guards():
    """What torch's own saved-tensor machinery would catch for a torch graph: a second backward through a freed renderer call
    and an in-place parameter update between forward and backward raise clear errors (pr_render_backward reads the
    parameter storages and the forward workspace in place); a render at the other precision between forward and backward
    (which repacks the weights) does not disturb the pending backward."""
    cfg = configs.reduced_config(configs.tennis_config(), **SMALL_NETS)
    comp = build(cfg).cuda().train()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=3), pixels=grid_pixels(256, 256, 10))]
    inputs[4] = inputs[4].clone().requires_grad_(True)

    def render():
        torch.manual_seed(1)
        return comp(*inputs, False)["coarse"]["global"]["integrated_features"].square().mean()

    loss = render()
    loss.backward()
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # retain_graph keeps torch's own nodes alive: the second pass reaches the renderer's node, which has freed its workspace
    loss = render()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="renderer call a second time"):
        loss.backward()
    loss = render()
    with torch.no_grad():
        next(comp.parameters()).mul_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        loss.backward()
    # reference gradient, then the same with an evaluation render at the other precision squeezed in between
    comp.zero_grad()
    inputs[4].grad = None
    render().backward()
    want = inputs[4].grad.clone()
    inputs[4].grad = None
    loss = render()
    comp.eval()
    comp.precision = "f16x3"
    with torch.no_grad():
        comp(*[t.detach() for t in inputs], False)
    comp.precision = "fp32"
    comp.train()
    loss.backward()
    assert torch.allclose(inputs[4].grad, want, rtol=1e-4, atol=1e-6 * float(want.abs().max()))


def test_workspace_budget_follows_free_memory(monkeypatch):
    """The scratch budget of a call is capped by what the device can still provide (free memory as the driver reports it +
    torch's cached blocks + the module's own workspace): with little memory reported free a full-frame render is split along
    the rays (exactly) instead of asking for the full scratch."""
    cfg = configs.tennis_config()
    comp = build(cfg).cuda()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=1234))]
    with torch.no_grad():
        want = comp(*inputs, False)["coarse"]["global"]["integrated_features"].clone()
    full = comp._workspace.numel()
    comp._workspace = None
    comp._budget_ok = 0          # (the size a device query has granted before: no second query for it on the hot path)
    torch.cuda.empty_cache()
    total = torch.cuda.mem_get_info()[1]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (256 << 20, total))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda device=None: 0)
    assert comp._workspace_budget(inputs[1].device, full) <= 256 << 20
    with torch.no_grad():
        got = comp(*inputs, False)["coarse"]["global"]["integrated_features"]
    assert comp._workspace.numel() < full // 4                   # the call was split into ray chunks
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


def test_generated_noise_is_independent_of_ray_chunking():
    """A perturbed render split along the rays (workspace budget) draws the noise of the UNSPLIT tensors (noise_ray_offset /
    noise_total_rays): identical to the unsplit render for the same seed, also for a multi-frame call."""
    cfg = configs.tennis_config(hierarchical=(16, 32))
    comp = build(cfg).cuda()
    scene = synthetic.tennis_scene(batch=2, seed=7)
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 48))]
    with torch.no_grad():
        torch.manual_seed(11)
        whole = comp(*inputs, True)
        seed = comp.last_noise_seed
        comp.max_workspace_bytes = 192 << 20           # forces several ray chunks
        comp._workspace = None
        torch.manual_seed(11)
        chunked = comp(*inputs, True)
    assert comp.last_noise_seed == seed and comp._workspace.numel() <= 192 << 20
    for ty in ("coarse", "fine"):
        for entry in ("global", "object_0", "object_3"):
            for key in ("integrated_features", "opacity", "depth", "weights"):
                assert torch.equal(torch.nan_to_num(whole[ty][entry][key]), torch.nan_to_num(chunked[ty][entry][key])), (ty, entry, key)


# ---------------------------------------------------------------------------------------------------------------------
# a fixed-seed slice of the randomized sweep (tests/gpu_fuzz.py: random network shapes, sample counts, frames, flags, absent
# objects, both precisions; forward fields and every gradient against the oracle)
#: plain "ok" cases of the seed-0 slices as recorded on the MI355X box (round 5: see profiles/r05_sweep_*.log) minus two cases for
#: run-to-run differences of atomically accumulated sums; and the most cases per slice whose FORWARD fields the harness may settle
#: (float64 arbitration / divergence kink: tests.test_gpu.SETTLEMENTS - logged with their numbers, never silent)
SWEEP_RECORDED_OK = {"forward": 40, "backward": 17, "backward_f16x3": 17}      # (forward: 38 plain + 2 arbitrated in round 5)
SWEEP_MAX_SETTLED = {"forward": 0, "backward": 1, "backward_f16x3": 1}       # (recorded: none in any slice)


@pytest.mark.parametrize("sweep,cases", [("forward", 40), ("backward", 20), ("backward_f16x3", 20)])
def test_randomized_sweep_slice(sweep, cases, capsys, monkeypatch):
    import random
    from tests import gpu_fuzz
    name = sweep
    run = gpu_fuzz.forward_sweep if sweep == "forward" else gpu_fuzz.backward_sweep
    if sweep == "backward_f16x3":        # the same backward slice on the split-precision training kernels (precision="f16x3")
        monkeypatch.setenv("PR_FUZZ_PRECISION", "f16x3")
    drain_settlements()
    failures = run(cases, random.Random(0))
    report = capsys.readouterr().out
    if os.environ.get("PR_SWEEP_REPORT_DIR"):            # (how profiles/r05_sweep_*.log were recorded)
        with open(os.path.join(os.environ["PR_SWEEP_REPORT_DIR"], f"sweep_{name}_{cases}_seed0.log"), "w") as f:
            f.write(report)
    assert failures == 0, report[-4000:]
    plain = report.count("ok case")
    settled = report.count("ok (forward ") + report.count("divergence kink")
    single_pass = report.count("ok (arbitrated, single pass)")
    isolated = report.count("ok (arbitrated, isolated rays)")
    classified = (report.count("ok (arbitrated) case") + single_pass + isolated + report.count("ill-conditioned") + report.count("noise kink") +
                  settled + report.count("skipped"))
    assert single_pass == 0, report[-3000:]          # (recorded: none in this slice; 1 in the 1 340 forward cases of the round's sweeps)
    assert isolated == 0, report[-3000:]             # (recorded: none in this slice; 1 in the 80 LARGE forward cases of rounds 5 - 6)
    assert plain + classified == cases, report[-2000:]
    # the harness classifies its own excesses: a regression that turned every case "ill-conditioned" must not pass.  Floor = the
    # plain-ok count recorded for this slice - 2; forward fields settled by arbitration / as a kink are capped per slice
    assert plain >= SWEEP_RECORDED_OK[name] - 2, (plain, SWEEP_RECORDED_OK[name], report[-3000:])
    assert settled <= SWEEP_MAX_SETTLED[name], (settled, report[-3000:])


def test_object_entry_fields_extension():
    """``ObjectComposer.object_entry_fields`` (evaluation extension): the per-object entries carry the requested fields only, the
    global entry and the requested fields are bit-identical to the full render; differentiable calls ignore the switch."""
    cfg = configs.tennis_config(hierarchical=(16, 32))
    comp = build(cfg, alpha_bias=2.0).cuda().eval()
    inputs = [v.cuda() for v in composer_inputs(cfg, synthetic.tennis_scene(seed=5), pixels=grid_pixels(256, 256, 24))]
    with torch.no_grad():
        full = comp(*inputs, False)
        comp.object_entry_fields = ("opacity", "depth")
        some = comp(*inputs, False)
        comp.object_entry_fields = ()
        none = comp(*inputs, False)
    for ty in ("coarse", "fine"):
        for key, value in full[ty]["global"].items():
            assert torch.equal(torch.nan_to_num(value), torch.nan_to_num(some[ty]["global"][key])), (ty, key)
            assert torch.equal(torch.nan_to_num(value), torch.nan_to_num(none[ty]["global"][key])), (ty, key)
        for k in range(4):
            assert set(some[ty][f"object_{k}"]) == {"opacity", "depth", "extra_outputs"}
            assert set(none[ty][f"object_{k}"]) == {"extra_outputs"}
            for key in ("opacity", "depth"):
                assert torch.equal(full[ty][f"object_{k}"][key], some[ty][f"object_{k}"][key]), (ty, k, key)
    out = comp(*inputs, False)            # gradients enabled: a differentiable call keeps the full schema
    assert "integrated_features" in out["coarse"]["object_0"] and out["coarse"]["global"]["integrated_features"].requires_grad
    comp.object_entry_fields = ("colour",)
    with torch.no_grad(), pytest.raises(ValueError, match="unknown field"):
        comp(*inputs, False)


def test_parameter_arena_trains_like_separate_tensors():
    """parallel.flatten_parameters on the HIP composer: the parameters become views of one arena, the packed-weight cache follows
    the arena's in-place updates (shared version counter), and two SGD steps through pr_render_backward move the parameters as
    they move with separate tensors (the kernels' atomically accumulated gradients differ by rounding from run to run)."""
    from playableenvironments_amd import parallel
    cfg = configs.reduced_config(configs.minecraft_config(), **SMALL_NETS)
    scene = synthetic.minecraft_scene(batch=2, seed=21)
    inputs = [v.cuda() for v in composer_inputs(cfg, scene, pixels=grid_pixels(256, 256, 12))]
    models = []
    for flatten in (False, True):
        comp = build(cfg, alpha_bias=3.0).cuda().train()
        names = [n for n, _ in comp.named_parameters()]
        arena = parallel.flatten_parameters(comp) if flatten else None
        assert [n for n, _ in comp.named_parameters()] == names
        opt = torch.optim.SGD([arena] if flatten else list(comp.parameters()), lr=1e-3)
        losses = []
        for step in range(3):
            opt.zero_grad(set_to_none=True)      # (arena: the documented recipe - flat_gradient has emptied the views' gradients)
            torch.manual_seed(100 + step)
            out = comp(*inputs, True)
            loss = out["coarse"]["global"]["integrated_features"].square().mean()
            loss.backward()
            if flatten:
                shared = next(comp.parameters()).grad.data_ptr()
                parallel.flat_gradient(arena, comp)
                assert arena.grad.data_ptr() == shared                                      # the shared buffer, no copy
                assert all(q.grad is None for q in comp.parameters())                       # ... now owned by the arena
            opt.step()
            losses.append(float(loss))
        models.append((comp, losses))
    (a, la), (b, lb) = models
    assert la[0] == lb[0] and all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(la, lb)), (la, lb)
    assert la[2] != la[0]                      # the updates reach the renderer (packed weights rebuilt)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        scale = float(p.abs().max())
        assert float((p - q).abs().max()) <= 1e-5 * scale + 1e-8, n
