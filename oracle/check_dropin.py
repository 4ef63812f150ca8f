"""Row b2 of the scope table: the drop-in EXECUTED with the reference's own callers (build container only; TEST INFRASTRUCTURE).

BASELINE.json's north_star says the renderer "drops into training/trainer.py and the reconstructed-dataset evaluators unchanged".
The callers are the reference's model subclasses

    EnvironmentModelBackpropagatedAutoencoder -> ...MultiresolutionBackpropagatedAutoencoder -> ...MultiresolutionBackpropagatedDecoder
    (model/environment_model_backpropagated_autoencoder.py:17-402, ..._multiresolution_backpropagated_autoencoder.py:14-220,
     ..._multiresolution_backpropagated_decoder.py:11-108)

which derive from ``model.environment_model.EnvironmentModel`` and are what ``training/trainer_multiresolution_backpropagated_decoder.py:52-53``
(``model(observations, ..., patch_size=..., patch_stride=..., align_grid=...)``) and ``evaluation/reconstructed_dataset_creator.py:121``
(``model.module.render_full_frame_from_observations(...)``) call.  Here the three subclass modules are loaded a SECOND time with
``model.environment_model.EnvironmentModel`` replaced by ``playableenvironments_amd.environment_model.EnvironmentModel``: the
reference's unmodified subclass code then runs on the product's base class (its constructor, its forward_from_observations /
forward_from_scene_encoding, fold helpers, attribute names).  Both instances get the same composer weights, the same stand-in
encoders (tests/helpers.py), the reference's own CNN autoencoder with the same weights, and the same dataset tensors; every tensor
of every result dictionary - the decoder's output included - is compared.  The composer behind the product's host logic is the
CPU oracle (there is no GPU in the build container); the GPU suite re-renders the recorded decoder input maps on the HIP path.

    python oracle/check_dropin.py [write]        (``write``: also records tests/golden/dropin/*.npz)
"""
import copy
import importlib
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import refshim  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from playableenvironments_amd import synthetic  # noqa: E402

SUBCLASS_MODULES = ("model.environment_model_backpropagated_autoencoder",
                    "model.environment_model_multiresolution_backpropagated_autoencoder",
                    "model.environment_model_multiresolution_backpropagated_decoder")
OUT = os.path.join(ROOT, "tests", "golden", "dropin")
REDUCE = dict(width=64, layers=4, skip=2, octaves=4, bender_width=32, bender_layers=3, bender_skip=1, bender_octaves=3)


def load_subclasses():
    """(reference subclass, the same source on the product's base class)."""
    refshim.install()
    import model.environment_model as ref_env_module
    from playableenvironments_amd import environment_model as em
    unswapped = [importlib.import_module(n) for n in SUBCLASS_MODULES]
    reference_base = ref_env_module.EnvironmentModel
    saved = {n: sys.modules[n] for n in SUBCLASS_MODULES}
    swapped = []
    ref_env_module.EnvironmentModel = em.EnvironmentModel
    try:
        for n, original in zip(SUBCLASS_MODULES, unswapped):
            spec = importlib.util.spec_from_file_location(n, original.__file__)
            module = importlib.util.module_from_spec(spec)
            sys.modules[n] = module          # the next module of the chain imports its parent class from here
            spec.loader.exec_module(module)
            swapped.append(module)
    finally:
        ref_env_module.EnvironmentModel = reference_base
        sys.modules.update(saved)
    name = "EnvironmentModelMultiresolutionBackpropagatedDecoder"
    ref_cls, new_cls = getattr(unswapped[2], name), getattr(swapped[2], name)
    assert reference_base in ref_cls.__mro__ and em.EnvironmentModel not in ref_cls.__mro__
    assert em.EnvironmentModel in new_cls.__mro__ and reference_base not in new_cls.__mro__
    # the feature drawer writes images through cv2 (absent here): results do not depend on it
    for m in unswapped + swapped:
        if hasattr(m, "AutoencoderFeaturesDrawer"):
            m.AutoencoderFeaturesDrawer.draw_features = staticmethod(lambda *a, **k: None)
    return ref_cls, new_cls


def dropin_config(world: str, reduce: bool):
    """The reference's shipped YAML with the defaults utils/configuration.py derives (:146-158), an untrained autoencoder, and -
    for the committed fixture - reduced renderer networks (the feature count stays 192: the decoder consumes 64 + 128)."""
    cfg = refshim.load_reference_config(world)
    ae = cfg["model"]["autoencoder"]
    ae["weights_filename"] = "untrained_model"
    ae.setdefault("also_freeze_bn", True)
    stride, factors = 1, []
    for count in ae["downsampling_layers_count"]:
        stride *= 2 ** count
        factors.append(stride)
    ae.setdefault("downsample_factor", factors)
    # (utils/configuration.py fills the logging directories at run time; the evaluator path creates one per rendered frame)
    cfg["logging"]["output_images_directory"] = tempfile.mkdtemp(prefix="dropin_check_")
    if reduce:
        for o in cfg["model"]["object_models"]:
            n = o["nerf_model"]
            n["layers_width"], n["backbone_layers_count"], n["skip_layer_idx"] = REDUCE["width"], REDUCE["layers"], REDUCE["skip"]
            n["position_encoder"]["octaves"] = REDUCE["octaves"]
            b = o["ray_bender_model"]
            if b["architecture"].endswith("positional_ray_bender_model"):
                b["layers_width"], b["layers_count"], b["skip_layer_idx"] = REDUCE["bender_width"], REDUCE["bender_layers"], REDUCE["bender_skip"]
                b["position_encoder"]["octaves"] = REDUCE["bender_octaves"]
    return cfg


class _GraphOracleComposer(torch.nn.Module):
    """The oracle behind the product's host logic, WITH autograd to the composer's parameters (the training-shaped iteration)."""

    def __init__(self, config, product_composer):
        super().__init__()
        self.cfg = config
        self.inner = product_composer

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.inner, name)

    def forward(self, ray_origins, ray_directions, focal_normals, w2o, style, deformation, object_in_scene, perturb,
                video_indexes=None, canonical_pose=False):
        sd = dict(self.inner.state_dict(keep_vars=True))
        return ro.composer_forward(self.cfg, sd, ray_origins, ray_directions, focal_normals, w2o, style, deformation,
                                   object_in_scene, perturb, canonical_pose=canonical_pose, training=self.inner.training)


def build_pair(world, cfg, seed=0, alpha_bias=2.0):
    from oracle.check_against_reference import _cpu_camera_rays, build_reference_environment_model
    from playableenvironments_amd import environment_model as em
    from tests.helpers import stand_in_encoders
    ref_cls, new_cls = load_subclasses()
    torch.manual_seed(seed)
    ref_composer = refshim.build_reference_composer(copy.deepcopy(cfg))
    synthetic.randomize_module_state(ref_composer, seed=seed, step=20000, alpha_bias=alpha_bias, bender_scale=1e4)
    ae_cfg = cfg["model"]["autoencoder"]
    torch.manual_seed(seed + 1)
    autoencoder = importlib.import_module(ae_cfg["architecture"]).model(ae_cfg).eval()
    # reference side: assembled attribute by attribute (its constructor would build the torchvision encoders)
    ref = build_reference_environment_model(cfg, ref_composer, *stand_in_encoders(cfg, world))
    ref.__class__ = ref_cls
    ref.autoencoder_model = autoencoder
    ref.autoencoder_bottleneck_transform = lambda x: x
    ref.is_autoencoder_frozen = False
    ref.strides = ae_cfg["downsample_factor"]
    # product side: the reference's subclass constructor on the product's base class, from the configuration alone
    mine = new_cls(cfg)
    mine.set_encoders(*stand_in_encoders(cfg, world))
    mine.object_composer.load_state_dict(ref_composer.state_dict(), strict=True)
    mine.autoencoder_model.load_state_dict(autoencoder.state_dict(), strict=True)
    mine.object_composer = _GraphOracleComposer(cfg, mine.object_composer)
    em.camera_rays = _cpu_camera_rays
    return ref, mine


class _Float32Draws:
    """Inside the float64 arbitration run the random draws (perturbation noise, patch positions) have to be the float32 run's:
    torch.rand / torch.randn under a float64 default dtype consume the generator differently.  Draws without an explicit dtype
    are made in float32 and widened."""
    NAMES = ("rand", "randn", "rand_like", "randn_like")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        self.multinomial = torch.multinomial
        # (the weighted pixel / patch samplers: a float64 probability tensor walks the generator differently)
        torch.multinomial = lambda weights, *a, _fn=self.multinomial, **k: _fn(weights.float(), *a, **k)
        for n, fn in self.saved.items():
            def draw(*a, _fn=fn, _like=n.endswith("_like"), **k):
                wanted = k.pop("dtype", None)
                if wanted is not None and wanted != torch.float64:
                    return _fn(*a, dtype=wanted, **k)
                if _like:
                    return _fn(a[0].float(), *a[1:], **k).to(a[0].dtype)
                return _fn(*a, dtype=torch.float32, **k).to(torch.float64 if wanted is not None else torch.get_default_dtype())
            setattr(torch, n, draw)
        return self

    def __exit__(self, *exc):
        torch.multinomial = self.multinomial
        for n, fn in self.saved.items():
            setattr(torch, n, fn)


def compare(want, got, path="", report=None):
    from oracle.check_against_reference import _compare_nested
    return _compare_nested(want, got, path, report=report)


def run_world(world, reduce, image_size, patch, write):
    from oracle.check_against_reference import OBS_KEYS
    from playableenvironments_amd import environment_model as em
    from tests.helpers import observation_batch
    original_camera_rays = em.camera_rays
    ok = True
    try:
        cfg = dropin_config(world, reduce)
        make_scene = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
        scene = make_scene(batch=1, observations=2, seed=7, image_size=image_size)
        ref, mine = build_pair(world, cfg, alpha_bias=2.0 if world == "tennis" else 3.0)
        batch = observation_batch(scene)
        args = [batch[k] for k in OBS_KEYS]
        strides = cfg["model"]["autoencoder"]["downsample_factor"]
        tag = f"{world}{' reduced' if reduce else ' shipped sizes'}"

        # ---- the trainer's call (training/trainer_multiresolution_backpropagated_decoder.py:52-53), evaluation arithmetic
        seen = {}
        for side, model in (("ref", ref), ("mine", mine)):
            decoder = model.autoencoder_model.forward_decoder

            def spy(features, _decoder=decoder, _side=side):
                seen[_side] = [f.detach().clone() for f in features]
                return _decoder(features)
            model.autoencoder_model.forward_decoder = spy
        kw = dict(samples_per_image=patch * patch, perturb=False, shuffle_style=False, patch_size=patch, patch_stride=strides,
                  align_grid=True)
        ref.eval(), mine.eval()
        torch.manual_seed(21)
        with torch.no_grad():
            want = ref(*[a.clone() for a in args], **kw)
        torch.manual_seed(21)
        with torch.no_grad():
            got = mine(*[a.clone() for a in args], **kw)
        rep = compare(want, got)
        rep.update(compare(seen["ref"], seen["mine"], "decoder_inputs."))
        bad = {k: v[0] for k, v in rep.items() if not v[1]}
        has = "reconstructed_observations" in got["coarse"]["global"] and "splitted_positions" in got
        print(f"[drop-in, {tag}: trainer call, patch {patch} @ strides {strides}] fields={len(rep)} decoder output present={has} "
              f"worst|diff|={max(v[0] for v in rep.values()):.2e} failing={bad}")
        ok &= not bad and has
        if write:
            data = {"in/" + k: batch[k].numpy() for k in OBS_KEYS}
            for k, v in ref.object_composer.state_dict().items():
                data["sd/" + k] = v.numpy()
            data["positions"] = want["positions"].numpy()
            for i, f in enumerate(seen["ref"]):
                data[f"decoder_input_{i}"] = f.numpy()
            data["integrated_features"] = want["coarse"]["global"]["integrated_features"].numpy()
            data["meta"] = np.frombuffer(repr({"world": world, "image_size": list(image_size), "patch_size": patch,
                                               "strides": list(strides), "reduce": REDUCE, "alpha_bias": 2.0 if world == "tennis" else 3.0}).encode(),
                                         dtype=np.uint8)
            os.makedirs(OUT, exist_ok=True)
            path = os.path.join(OUT, f"{world}_trainer_patch.npz")
            np.savez_compressed(path, **data)
            print(f"  wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")

        # ---- the evaluator's call (evaluation/reconstructed_dataset_creator.py:121) and its scene-encoding twin
        with torch.no_grad():
            full_a = ref.render_full_frame_from_observations(*[a.clone() for a in args], perturb=False, upsample_factor=1.0)
            full_b = mine.render_full_frame_from_observations(*[a.clone() for a in args], perturb=False, upsample_factor=1.0)
        rep = compare(full_a, full_b)
        bad = {k: v[0] for k, v in rep.items() if not v[1]}
        shape = tuple(full_b["coarse"]["global"]["reconstructed_observations"].shape)
        print(f"[drop-in, {tag}: render_full_frame_from_observations] fields={len(rep)} reconstructed_observations {shape} "
              f"worst|diff|={max(v[0] for v in rep.values()):.2e} failing={bad}")
        ok &= not bad
        se = full_a["scene_encoding"]
        se_args = [se["camera_rotations"], se["camera_translations"], se["focals"], tuple(image_size), se["object_rotation_parameters"],
                   se["object_translation_parameters"], se["object_style"], se["object_deformation"], se["object_in_scene"], False]
        with torch.no_grad():
            enc_a = ref.render_full_frame_from_scene_encoding(*se_args)
            enc_b = mine.render_full_frame_from_scene_encoding(*se_args)
        rep = compare(enc_a, enc_b)
        bad = {k: v[0] for k, v in rep.items() if not v[1]}
        print(f"[drop-in, {tag}: render_full_frame_from_scene_encoding] fields={len(rep)} worst|diff|={max(v[0] for v in rep.values()):.2e} failing={bad}")
        ok &= not bad

        # ---- one training-shaped iteration: train mode, perturbation, a reconstruction loss on the decoder output + the
        # trainer's opacity / bounding-box style terms, backward through decoder, wire format, renderer and encoders' outputs
        def iteration(side, model, seed):
            model.train()
            model.autoencoder_model.eval()
            for p in model.parameters():
                p.grad = None
            torch.manual_seed(seed)
            out = model(*[a.clone() for a in args], **dict(kw, perturb=True))
            target = torch.linspace(0, 1, out["coarse"]["global"]["reconstructed_observations"].numel()).reshape(
                out["coarse"]["global"]["reconstructed_observations"].shape)
            loss = (out["coarse"]["global"]["reconstructed_observations"] - target).square().mean()
            loss = loss + 0.1 * out["coarse"]["global"]["integrated_displacements_magnitude"].mean()
            loss = loss + 0.05 * sum(out["coarse"][f"object_{k}"]["opacity"].mean() for k in range(2))
            loss.backward()
            composer = model.object_composer.inner if side == "mine" else model.object_composer
            return {"loss": loss.detach(),
                    "composer": {n: p.grad.clone() for n, p in composer.named_parameters() if p.grad is not None},
                    "decoder": {n: p.grad.clone() for n, p in model.autoencoder_model.named_parameters() if p.grad is not None},
                    "running": {n: b.clone() for n, b in composer.named_buffers() if "running" in n}}
        # (a random patch that misses an object leaves its BatchNorm with <= 1 sample: torch raises - on both sides, for the same
        # seed; the trainer would skip such a batch.  Take the first seed whose patch sees every object.)
        grads = {}
        states = {side: copy.deepcopy(model.state_dict()) for side, model in (("ref", ref), ("mine", mine))}
        for seed in range(31, 60):
            try:
                grads["ref"] = iteration("ref", ref, seed)
            except ValueError:
                ref.load_state_dict(states["ref"])
                try:
                    iteration("mine", mine, seed)
                    print(f"  seed {seed}: the reference raises for a starved BatchNorm, the swapped model does not")
                    ok = False
                except ValueError:
                    mine.load_state_dict(states["mine"])
                continue
            grads["mine"] = iteration("mine", mine, seed)
            break
        # The two sides differ by the last bits of their pose matrices (closed-form rigid inverse vs LU, 2e-6) and the shipped 8 x 256
        # networks amplify that in single tensors (ReLU / box decisions flip).  Which side is off is decided
        # in FLOAT64: the same iteration - same seed, same draws - through the product's host logic with the oracle behind it and
        # every module widened is the exact result; per tensor the product's gradient may be as far from it as the reference's
        # is (x 4, + 1e-6 of the group's largest gradient), not farther.  The relative L2 figures are printed beside it.
        from tests.helpers import oracle_in_float64, to_double
        mine.load_state_dict(states["mine"])
        mine.double()
        saved_args = args
        try:
            with oracle_in_float64(), _Float32Draws():
                args = to_double([a.clone() for a in saved_args])
                grads["exact"] = iteration("mine", mine, seed)
        finally:
            args = saved_args
            mine.float()
            mine.load_state_dict(states["mine"])
        worst, l2, farther = 0.0, {}, {}
        err = {"ref": 0.0, "mine": 0.0}
        ill, apart = {}, {}          # tensors where the REFERENCE itself is > 1 % of the group's scale from float64: named, and held to the
                                     # direct yardstick instead (an arbiter both sides are far from arbitrates nothing)
        for group in ("composer", "decoder", "running"):
            assert set(grads["ref"][group]) == set(grads["mine"][group]) == set(grads["exact"][group]) and grads["ref"][group], group
            scale = max(float(t.abs().max()) for t in grads["exact"][group].values())
            num = den = 0.0
            for n in grads["ref"][group]:
                a, b, e = grads["ref"][group][n].double(), grads["mine"][group][n].double(), grads["exact"][group][n].double()
                worst = max(worst, float((a - b).abs().max()) / max(float(a.abs().max()), 1e-2 * scale, 1e-12))
                num += float((a - b).square().sum())
                den += float(a.square().sum())
                err_ref, err_mine = float((a - e).abs().max()), float((b - e).abs().max())
                err["ref"], err["mine"] = max(err["ref"], err_ref / max(scale, 1e-300)), max(err["mine"], err_mine / max(scale, 1e-300))
                if err_mine > 4.0 * err_ref + 1e-6 * scale:
                    farther[f"{group}.{n}"] = (err_mine, err_ref)
                if err_ref > 1e-2 * scale:
                    between = float((a - b).abs().max())
                    ill[f"{group}.{n}"] = (err_ref / scale, err_mine / scale, between / scale)
                    if between > 0.25 * err_ref:         # the two fp32 sides must sit on the same side of whatever float64 decides differently
                        apart[f"{group}.{n}"] = (between / scale, err_ref / scale)
            l2[group] = (num / max(den, 1e-300)) ** 0.5
        same_loss = abs(float(grads["ref"]["loss"]) - float(grads["mine"]["loss"])) <= 1e-5 * abs(float(grads["ref"]["loss"]))
        print(f"[drop-in, {tag}: training-shaped iteration] loss {float(grads['ref']['loss']):.6f} / {float(grads['mine']['loss']):.6f} "
              f"(float64: {float(grads['exact']['loss']):.6f}), "
              f"{len(grads['ref']['composer'])} renderer + {len(grads['ref']['decoder'])} decoder gradients, BatchNorm running statistics: "
              f"relative L2 difference renderer {l2['composer']:.2e}, decoder {l2['decoder']:.2e}, statistics {l2['running']:.2e}; "
              f"worst tensor {worst:.2e}")
        print(f"    float64 arbitration: worst error relative to the group's largest gradient - reference {err['ref']:.2e}, swapped model "
              f"{err['mine']:.2e}; tensors where the swapped model is farther than 4 x the reference: {farther}")
        if ill:
            print("    ill-conditioned in fp32 by the reference's own distance to float64 (> 1e-2 of the group's largest gradient), held to the direct "
                  "comparison of the two fp32 sides instead: " +
                  "; ".join(f"{k}: reference {v[0]:.1e}, swapped {v[1]:.1e}, |reference - swapped| {v[2]:.1e}" for k, v in ill.items()))
        ok &= same_loss and not farther and not apart
    finally:
        em.camera_rays = original_camera_rays
    return ok


def main(write=False):
    refshim.install()
    ok = True
    # the committed fixtures: reduced renderer networks
    ok &= run_world("tennis", True, (64, 96), 8, write)
    ok &= run_world("minecraft", True, (64, 96), 8, write)
    # the shipped network sizes, trainer call + evaluator call on a small frame
    ok &= run_world("minecraft", False, (48, 64), 4, False)
    ok &= run_world("tennis", False, (48, 64), 4, False)
    print("DROP-IN " + ("OK" if ok else "FAILED"))
    return ok


if __name__ == "__main__":
    sys.exit(0 if main(write=len(sys.argv) > 1 and sys.argv[1] == "write") else 1)
