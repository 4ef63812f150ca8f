"""Generates tests/golden/*.npz from the ORIGINAL reference (build container only).

Each fixture holds, for a reduced-size model (so that the files stay small): the config recipe,
the seven ObjectComposer.forward inputs, the reference's state_dict, every output field the
reference produced, and - for perturbed cases - the noise tensors in the order the reference drew
them (recorded by the oracle, which consumes torch's generator in exactly the reference's order;
the script asserts that oracle and reference agree bitwise before writing).

Fixtures are DATA (inputs and expected outputs), not reference source.  Usage:
    python -m oracle.make_golden
"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import refshim  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from playableenvironments_amd import configs, synthetic  # noqa: E402
from tests.helpers import composer_inputs, grid_pixels  # noqa: E402

OUT = os.path.join("tests", "golden")


def recipe_config(recipe: dict) -> dict:
    """Rebuilds the config of a fixture from its recipe (also used by the tests)."""
    base = {"tennis": configs.tennis_config, "minecraft": configs.minecraft_config,
            "single": configs.tennis_single_player_config}[recipe["base"]]()
    if recipe.get("fine"):
        base = configs.enable_fine(base)
    return configs.reduced_config(base, positions=recipe.get("positions"), **recipe.get("reduce", {}))


def flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if k in ("pytorch_hook", "extra_outputs"):
            continue
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v.detach().cpu().numpy()
    return out


def make(name, recipe, scene, pixels, perturb=False, seed=0, alpha_bias=2.0, step=20000):
    cfg = recipe_config(recipe)
    torch.manual_seed(seed)
    ref = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(ref, seed=seed, step=step, alpha_bias=alpha_bias, bender_scale=1e4)
    ref.eval()
    inputs = composer_inputs(cfg, scene, pixels=pixels)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        torch.manual_seed(seed + 1)
        out_ref = ref(*[v.clone() for v in inputs], perturb)
        torch.manual_seed(seed + 1)
        rec = {}
        out_or = ro.composer_forward(cfg, sd, *inputs, perturb, record_noise=rec)
    fr, fo = flatten(out_ref), flatten(out_or)
    for k in fr:
        assert np.array_equal(fr[k], fo[k], equal_nan=True), f"{name}: oracle != reference on {k}"
    data = {"out/" + k: v for k, v in fr.items()}
    for i, v in enumerate(inputs):
        data[f"in/{i}"] = v.numpy()
    for k, v in sd.items():
        data["sd/" + k] = v.numpy()
    for k, v in rec.items():
        if v is not None:
            data["noise/" + k] = v.numpy()
    data["recipe"] = np.frombuffer(repr(recipe).encode(), dtype=np.uint8)
    data["perturb"] = np.array(int(perturb))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(fr)} output fields")


GRAD_FIELDS = ("integrated_features", "opacity", "depth", "integrated_displacements_magnitude")


def make_gradients(name, recipe, scene, pixels, perturb=True, seed=0, alpha_bias=2.0, step=20000):
    """Train-mode fixture: reference forward + backward of a fixed random linear functional of every differentiable
    result field; stores the functional's coefficients, the forward results and d loss / d (every parameter, w2o,
    style, deformation) as the REFERENCE's autograd computed them."""
    cfg = recipe_config(recipe)
    torch.manual_seed(seed)
    ref = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(ref, seed=seed, step=step, alpha_bias=alpha_bias, bender_scale=1e4)
    ref.train()
    inputs = composer_inputs(cfg, scene, pixels=pixels)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    leaf = [inputs[i].clone().requires_grad_(True) for i in (3, 4, 5)]
    args = list(inputs[:3]) + leaf + [inputs[6]]
    torch.manual_seed(seed + 1)
    out_ref = ref(*args, perturb)
    torch.manual_seed(seed + 1)
    rec = {}
    with torch.enable_grad():
        out_or = ro.composer_forward(cfg, {k: v.clone() for k, v in sd.items()}, *[a.detach() for a in args[:3]],
                                     *[t.detach().clone().requires_grad_(True) for t in leaf], inputs[6], perturb,
                                     training=True, record_noise=rec)
    fr, fo = flatten(out_ref), flatten(out_or)
    for k in fr:
        assert np.array_equal(fr[k], fo[k], equal_nan=True), f"{name}: oracle != reference on {k}"
    gen = torch.Generator().manual_seed(seed + 2)
    data = {}
    loss = 0.0
    for entry in sorted(out_ref["coarse"].keys()):
        for key in GRAD_FIELDS:
            t = out_ref["coarse"][entry][key]
            w = torch.randn(t.shape, generator=gen)
            data[f"probe/{entry}/{key}"] = w.numpy()
            loss = loss + (t * w).sum()
    loss.backward()
    for k, p in ref.named_parameters():
        data["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for label, t in zip(("w2o", "style", "deformation"), leaf):
        data["grad/" + label] = t.grad.numpy()
    data.update({"out/" + k: v for k, v in fr.items()})
    for i, v in enumerate(inputs):
        data[f"in/{i}"] = v.numpy()
    for k, v in sd.items():
        data["sd/" + k] = v.numpy()
    for k, v in rec.items():
        if v is not None and not k.startswith("div_"):
            data["noise/" + k] = v.detach().numpy()
    data["recipe"] = np.frombuffer(repr(recipe).encode(), dtype=np.uint8)
    data["perturb"] = np.array(int(perturb))
    os.makedirs(os.path.join(OUT, "grads"), exist_ok=True)
    path = os.path.join(OUT, "grads", name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {sum(k.startswith('grad/') for k in data)} gradient tensors")


def make_expected_positions_gradients(name, recipe, scene, pixels, object_id, perturb, seed=0, alpha_bias=2.0, step=20000):
    """Train-mode fixture of ObjectComposer.forward_expected_positions: reference forward + backward of a fixed random
    linear functional of (expected positions, opacity); stores inputs, state_dict, replay noise (recorded by the oracle,
    which draws in the reference's order - asserted bitwise), outputs and the REFERENCE's gradients."""
    cfg = recipe_config(recipe)
    torch.manual_seed(seed)
    ref = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(ref, seed=seed, step=step, alpha_bias=alpha_bias, bender_scale=1e4)
    ref.train()
    o, d, n, w2o, sty, dfm, ins = composer_inputs(cfg, scene, pixels=pixels)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    leaf = [t[..., object_id].clone().requires_grad_(True) for t in (w2o, sty, dfm)]
    torch.manual_seed(seed + 1)
    out_ref = ref.forward_expected_positions(o, d, n, *leaf, ins[..., object_id], object_id, perturb)
    torch.manual_seed(seed + 1)
    rec = {}
    with torch.no_grad():
        out_or = ro.expected_positions_forward(cfg, {k: v.clone() for k, v in sd.items()}, o, d, n, *[t.detach() for t in leaf],
                                               ins[..., object_id], object_id, perturb, training=True, record_noise=rec)
    gen = torch.Generator().manual_seed(seed + 2)
    data, loss = {}, 0.0
    for ty in out_ref:
        for i, (a, b) in enumerate(zip(out_ref[ty], out_or[ty])):
            assert torch.equal(a.detach(), b), f"{name}: oracle != reference on {ty}[{i}]"
            w = torch.randn(a.shape, generator=gen)
            data[f"probe/{ty}/{i}"] = w.numpy()
            data[f"out/{ty}/{i}"] = a.detach().numpy()
            loss = loss + (a * w).sum()
    loss.backward()
    for k, p in ref.named_parameters():
        data["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for label, t in zip(("w2o", "style", "deformation"), leaf):
        data["grad/" + label] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    for i, v in enumerate((o, d, n, w2o, sty, dfm, ins)):
        data[f"in/{i}"] = v.numpy()
    for k, v in sd.items():
        data["sd/" + k] = v.numpy()
    for k, v in rec.items():
        if v is not None:
            data["noise/" + k] = v.detach().numpy()
    data["recipe"] = np.frombuffer(repr(recipe).encode(), dtype=np.uint8)
    data["perturb"] = np.array(int(perturb))
    data["object_id"] = np.array(int(object_id))
    os.makedirs(os.path.join(OUT, "expected_positions"), exist_ok=True)
    path = os.path.join(OUT, "expected_positions", name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


REDUCE = dict(width=64, layers=4, skip=2, features=32, octaves=4, bender_width=32, bender_layers=3, bender_skip=1,
              bender_octaves=3)
ODD = dict(width=48, layers=3, skip=1, features=16, octaves=3, bender_width=16, bender_layers=3, bender_skip=2,
           bender_octaves=2)


C2_POSITIONS = {"background": (64, 128), "background_backplate": (64, 128), "player_1": (64, 128), "player_2": (64, 128)}


def make_observation_mode(name, world, recipe, scene, alpha_bias, mode_kwargs):
    """Fixture of EnvironmentModel.forward_from_observations: the REFERENCE's method (its EnvironmentModel assembled around
    stand-in encoders, tests/helpers.py) on synthetic dataset tensors; stores the inputs, the composer's state_dict and the
    reference's result tensors.  The encoders are rebuilt from their seeds by the test."""
    from oracle.check_against_reference import OBS_KEYS, build_reference_environment_model
    from tests.helpers import observation_batch, stand_in_encoders
    cfg = recipe_config(recipe)
    torch.manual_seed(0)
    composer = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(composer, seed=0, step=20000, alpha_bias=alpha_bias, bender_scale=1e4)
    ref = build_reference_environment_model(cfg, composer.eval(), *stand_in_encoders(cfg, world)).eval()
    batch = observation_batch(scene)
    with torch.no_grad():
        out = ref(*[batch[k].clone() for k in OBS_KEYS], **mode_kwargs)
    data = {"in/" + k: batch[k].numpy() for k in OBS_KEYS}
    for k, v in composer.state_dict().items():
        data["sd/" + k] = v.numpy()

    def put(prefix, value):
        if isinstance(value, dict):
            for k, v in value.items():
                if k not in ("extra_outputs", "object_attention", "object_crops"):
                    put(f"{prefix}{k}/", v)
        elif torch.is_tensor(value):
            data["out/" + prefix[:-1]] = value.detach().numpy()
    put("", {k: v for k, v in out.items() if k not in ("object_attention", "object_crops")})
    data["recipe"] = np.frombuffer(repr(recipe).encode(), dtype=np.uint8)
    data["meta"] = np.frombuffer(repr({"world": world, "image_size": list(scene["image_size"]), "kwargs": mode_kwargs}).encode(),
                                 dtype=np.uint8)
    os.makedirs(os.path.join(OUT, "observations"), exist_ok=True)
    path = os.path.join(OUT, "observations", name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {sum(k.startswith('out/') for k in data)} output tensors")


def make_consistency(name, world, recipe, scene, alpha_bias):
    """Fixture of EnvironmentModel.forward_pose_consistency / forward_keypoint_consistency (model/environment_model.py:1197-1505):
    the REFERENCE's methods on synthetic dataset tensors, optical flow and keypoints.  Both draw random pixels inside
    RayHelper.sample_rays_at_object / sample_rays_at_keypoints: what those calls returned is recorded too (``draw/...``), so that
    the test can replay the same pixels through the product (whose own random stream is the device's) and compare the expected
    positions, opacities and confidences value for value."""
    from oracle.check_against_reference import OBS_KEYS, build_reference_environment_model
    from tests.helpers import observation_batch, stand_in_encoders
    from utils.lib_3d.ray_helper import RayHelper
    cfg = recipe_config(recipe)
    torch.manual_seed(0)
    composer = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(composer, seed=0, step=20000, alpha_bias=alpha_bias, bender_scale=1e4)
    ref = build_reference_environment_model(cfg, composer.eval(), *stand_in_encoders(cfg, world)).eval()
    batch = observation_batch(scene)
    size = scene["image_size"]
    lead = list(batch["observations"].shape[:3])
    g = torch.Generator().manual_seed(1)
    flow = (torch.rand(lead + [2, size[0], size[1]], generator=g) - 0.5) * 0.05
    keypoints = torch.rand(lead + [17, 3, 2], generator=g)
    with torch.no_grad():
        se = ref(*[batch[k].clone() for k in OBS_KEYS], mode="observations_scene_encoding_only")
    common = [batch[k] for k in OBS_KEYS[1:]] + [se["object_style"], se["object_deformation"],
                                                 se["object_rotation_parameters"], se["object_translation_parameters"]]
    data = {"in/" + k: batch[k].numpy() for k in OBS_KEYS}
    data.update({"in/optical_flow": flow.numpy(), "in/keypoints": keypoints.numpy()})
    for k in ("object_style", "object_deformation", "object_rotation_parameters", "object_translation_parameters"):
        data["se/" + k] = se[k].numpy()
    for k, v in composer.state_dict().items():
        data["sd/" + k] = v.numpy()
    draws = {"object": [], "keypoints": []}
    originals = (RayHelper.sample_rays_at_object, RayHelper.sample_rays_at_keypoints)

    def spy(kind, fn):
        def wrapped(*a, **kw):
            out = fn(*a, **kw)
            draws[kind].append([t.detach().clone() for t in out])
            return out
        return staticmethod(wrapped)
    RayHelper.sample_rays_at_object = spy("object", originals[0])
    RayHelper.sample_rays_at_keypoints = spy("keypoints", originals[1])
    try:
        torch.manual_seed(13)
        with torch.no_grad():
            pose = ref(flow.clone(), *[x.clone() for x in common], 30, False, mode="pose_consistency")
            kp = ref(batch["observations"].clone(), *[x.clone() for x in common], keypoints.clone(),
                     batch["bounding_boxes_validity"].clone(), 20, False, mode="keypoint_consistency")
    finally:
        RayHelper.sample_rays_at_object, RayHelper.sample_rays_at_keypoints = (staticmethod(f) for f in originals)
    for kind, calls in draws.items():
        for i, triple in enumerate(calls):
            for j, t in enumerate(triple):
                data[f"draw/{kind}/{i}/{j}"] = t.numpy()
    for name_, (previous, following) in pose["coarse"].items():
        for tag, (positions, opacity) in (("previous", previous), ("following", following)):
            data[f"pose/{name_}/{tag}/positions"] = positions.numpy()
            data[f"pose/{name_}/{tag}/opacity"] = opacity.numpy()
    for name_, (positions, confidence, opacity, sampled) in kp["coarse"].items():
        for tag, t in (("positions", positions), ("confidence", confidence), ("opacity", opacity), ("sampled", sampled)):
            data[f"keypoint/{name_}/{tag}"] = t.numpy()
    data["recipe"] = np.frombuffer(repr(recipe).encode(), dtype=np.uint8)
    data["meta"] = np.frombuffer(repr({"world": world, "image_size": list(size), "pose_samples": 30, "keypoint_samples": 20}).encode(),
                                 dtype=np.uint8)
    os.makedirs(os.path.join(OUT, "consistency"), exist_ok=True)
    path = os.path.join(OUT, "consistency", name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(draws['object'])} + {len(draws['keypoints'])} recorded draws, "
          f"{sum(k.startswith(('pose/', 'keypoint/')) for k in data)} output tensors")


def main():
    refshim.install()
    if len(sys.argv) > 1 and sys.argv[1] == "consistency":
        make_consistency("tennis", "tennis", {"base": "tennis", "reduce": REDUCE},
                         synthetic.tennis_scene(batch=2, observations=3, seed=7, image_size=(48, 64)), 2.0)
        make_consistency("minecraft", "minecraft", {"base": "minecraft", "reduce": REDUCE},
                         synthetic.minecraft_scene(batch=1, observations=3, seed=8, image_size=(48, 64)), 3.0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "observations":
        make_observation_mode("tennis_strided_grid", "tennis", {"base": "tennis", "reduce": REDUCE},
                              synthetic.tennis_scene(batch=2, observations=2, seed=3, image_size=(48, 64)), 2.0,
                              dict(samples_per_image=0, perturb=False, patch_stride=[4, 8]))
        make_observation_mode("minecraft_all_pixels", "minecraft", {"base": "minecraft", "reduce": REDUCE},
                              synthetic.minecraft_scene(batch=1, observations=3, seed=4, image_size=(24, 32)), 3.0,
                              dict(samples_per_image=0, perturb=False))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c2":
        # BASELINE.json configs[1] sample counts (64 coarse + 128 resampled per object, 256 / 768 merged entries per ray)
        # on reduced network widths: the fixture of the headline configuration's list lengths
        make("tennis_small_c2_hier_eval", {"base": "tennis", "reduce": REDUCE, "fine": True, "positions": C2_POSITIONS},
             synthetic.tennis_scene(seed=28), grid_pixels(256, 256, 8))
        return
    make("tennis_small_eval", {"base": "tennis", "reduce": REDUCE}, synthetic.tennis_scene(seed=21),
         grid_pixels(256, 256, 16))
    make("tennis_odd_widths_eval", {"base": "tennis", "reduce": ODD}, synthetic.tennis_scene(seed=22, batch=2),
         grid_pixels(256, 256, 10))
    make("minecraft_small_eval", {"base": "minecraft", "reduce": REDUCE}, synthetic.minecraft_scene(seed=23),
         grid_pixels(256, 256, 16), alpha_bias=3.0)
    make("tennis_small_hier_eval", {"base": "tennis", "reduce": REDUCE, "fine": True,
                                    "positions": {"background": (8, 12), "background_backplate": (8, 12),
                                                  "player_1": (12, 20), "player_2": (12, 20)}},
         synthetic.tennis_scene(seed=24), grid_pixels(256, 256, 12))
    make("tennis_small_perturb", {"base": "tennis", "reduce": REDUCE}, synthetic.tennis_scene(seed=25),
         grid_pixels(256, 256, 12), perturb=True)
    make("minecraft_small_hier_perturb", {"base": "minecraft", "reduce": REDUCE, "fine": True,
                                          "positions": {"background": (8, 8), "skybox": (3, 2), "player_1": (12, 12)}},
         synthetic.minecraft_scene(seed=26), grid_pixels(256, 256, 12), perturb=True, alpha_bias=3.0)
    make_gradients("tennis_small_train", {"base": "tennis", "reduce": REDUCE}, synthetic.tennis_scene(seed=31),
                   grid_pixels(256, 256, 12))
    make_gradients("minecraft_small_train", {"base": "minecraft", "reduce": REDUCE}, synthetic.minecraft_scene(seed=32),
                   grid_pixels(256, 256, 12), alpha_bias=3.0)
    make_gradients("tennis_small_train_two_frames", {"base": "tennis", "reduce": REDUCE},
                   synthetic.tennis_scene(seed=33, batch=2), grid_pixels(256, 256, 8), perturb=False)
    make_expected_positions_gradients("tennis_player_train", {"base": "tennis", "reduce": REDUCE}, synthetic.tennis_scene(seed=34),
                                      grid_pixels(256, 256, 12), 2, perturb=True)
    make_expected_positions_gradients("minecraft_player_train", {"base": "minecraft", "reduce": REDUCE},
                                      synthetic.minecraft_scene(seed=35), grid_pixels(256, 256, 12), 3, perturb=False, alpha_bias=3.0)
    from oracle.check_against_reference import pose_math_case
    assert pose_math_case(write_to=os.path.join(OUT, "host", "pose_math_minecraft.npz"))
    make("single_player_eval", {"base": "single", "reduce": REDUCE, "positions": {"player_1": (16, 16)}},
         synthetic.single_player_scene(seed=27, image_size=(16, 16)), None)


if __name__ == "__main__":
    main()
