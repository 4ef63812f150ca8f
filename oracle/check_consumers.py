"""The reference's own CONSUMER code executed on the swapped base class (build container only; TEST INFRASTRUCTURE).

``oracle/check_dropin.py`` replays the model calls the reference's callers make and reads the result keys by hand.  Here the
callers themselves run - unmodified reference code, imported from /root/reference - once around the reference's model and once
around the same subclass on ``playableenvironments_amd.environment_model.EnvironmentModel``:

* ``TrainerMultiresolutionBackpropagatedDecoder`` (training/trainer_multiresolution_backpropagated_decoder.py:23-160 over
  training/trainer_backpropagated_autoencoder.py and training/trainer.py): its constructor (``get_optimizer`` ->
  ``model.get_autoencoder_parameters / get_main_parameters / get_object_encoder_parameters / get_camera_offsets_parameters``,
  the loss objects, the data loader over a stub dataset) and ``compute_losses`` on a synthetic ``Batch`` - the loss code that
  consumes ``splitted_positions``, ``object_attention``, ``object_crops``, ``reconstructed_bounding_boxes[..., static:]``,
  ``splitted_integrated_features``, every per-object ``opacity`` ... - with the perceptual loss switched off (its VGG needs
  torchvision, absent here).  Every ``loss_info`` entry and the gradient of the total loss with respect to every parameter are
  compared; which side is off where they differ by more than round-off is decided in float64 (check_dropin's arbitration).
* ``PlayableEnvironmentModel.initialize_interactive_generation`` (model/playable_environment_model.py:222-290): the play loop's
  first frame - ``mode="observations_scene_encoding_only"`` then ``render_full_frame_from_scene_encoding(...,
  samples_per_image_batching=1200)`` and ``get_reconstructed_observations_from_render_results`` - called unbound on a holder
  object (the animation networks of the playable model are outside the renderer's path).

The composer behind the swapped model is the CPU oracle (no GPU in the container), as in check_dropin.py.

    python oracle/check_consumers.py [write]      (``write``: records tests/golden/consumers/<world>_loss_info.npz)
"""
import copy
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import check_dropin, refshim  # noqa: E402
from playableenvironments_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "consumers")


class _Wrapped(torch.nn.Module):
    """What nn.DataParallel is to the trainer: ``model(...)`` and ``model.module``."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class _Batch:
    """dataset/batching.py Batch as compute_losses reads it: to_tuple (12 tensors), no ground-truth object poses."""

    def __init__(self, tensors):
        self.tensors = tensors

    def to_tuple(self):
        t = self.tensors
        lead = t["observations"].shape[:2]
        zeros = torch.zeros(tuple(lead) + (1,))
        return (t["observations"], zeros, zeros, zeros.bool(), t["camera_rotations"], t["camera_translations"], t["focals"],
                t["bounding_boxes"], t["bounding_boxes_validity"], t["global_frame_indexes"], t["video_frame_indexes"], t["video_indexes"])

    def has_object_poses(self):
        return False


class _Logger:
    def print(self, *a, **k):
        pass


def build_trainer(cfg, model):
    from training.trainer_multiresolution_backpropagated_decoder import TrainerMultiresolutionBackpropagatedDecoder
    trainer = TrainerMultiresolutionBackpropagatedDecoder(cfg, model, dataset=list(range(8)), logger=_Logger())
    # (step 0 writes a debug image through cv2; behind frozen_autoencoder_steps the decoder trains too)
    trainer.global_step = max(1000, int(cfg["training"]["frozen_autoencoder_steps"])) + 1
    trainer.save_reconstructed_observation = lambda *a, **k: None
    return trainer


def run_world(world, write):
    from oracle.check_against_reference import OBS_KEYS
    from playableenvironments_amd import environment_model as em
    from tests.helpers import observation_batch, oracle_in_float64, to_double
    original_camera_rays = em.camera_rays
    ok = True
    try:
        cfg = check_dropin.dropin_config(world, reduce=True)
        # the defaults the reference derives at start-up, by its own code (utils/configuration.py:30-242)
        import tempfile
        from utils.configuration import Configuration
        images_directory = cfg["logging"]["output_images_directory"]
        cfg["data"]["data_root"] = tempfile.mkdtemp(prefix="consumers_data_")
        holder = Configuration.__new__(Configuration)
        holder.config = cfg
        holder.check_config()
        cfg["logging"]["output_images_directory"] = images_directory
        cfg["training"].setdefault("camera_parameters_learning_rate", 0.0)
        tr = cfg["training"]
        tr["loss_weights"]["perceptual_loss_lambda"] = 0.0          # VGG features: torchvision is absent here
        tr["patch_size"], tr["samples_per_image"] = 8, 64
        tr["batching"]["num_workers"] = 0
        # every renderer-side loss term takes part (the shipped weights switch most of them off)
        for name in ("displacements_magnitude_loss_lambda", "divergence_loss_lambda", "opacity_loss_lambda", "sharpness_loss_lambda",
                     "attention_loss_lambda", "bounding_box_loss_lambda"):
            tr["loss_weights"][name] = max(float(tr["loss_weights"].get(name, 0.0)), 0.1)
        image_size = (64, 96)
        make_scene = synthetic.tennis_scene if world == "tennis" else synthetic.minecraft_scene
        scene = make_scene(batch=1, observations=2, seed=7, image_size=image_size)
        ref, mine = check_dropin.build_pair(world, cfg, alpha_bias=2.0 if world == "tennis" else 3.0)
        # (the reference side is assembled attribute by attribute: give it the offsets table its constructor would build)
        from model.layers.camera_parameters_storage import CameraParametersStorage
        ref.camera_parameters_offsets = CameraParametersStorage(cfg["model"]["camera_parameters_memory_size"],
                                                                len(cfg["training"]["batching"]["allowed_cameras"]))
        tensors = observation_batch(scene)
        tensors["observations"] = tensors["observations"] * 2 - 1           # the dataset's range
        batch = _Batch(tensors)

        # ---- the trainer: constructor + compute_losses ---------------------------------------------------------------
        trainers = {side: build_trainer(cfg, model) for side, model in (("ref", ref), ("mine", mine))}
        groups = {side: [len(g["params"]) for g in t.optimizer.param_groups] for side, t in trainers.items()}
        print(f"[consumers, {world}] Trainer.__init__ on both classes: optimiser parameter groups {groups['ref']} / {groups['mine']}")
        ok &= groups["ref"] == groups["mine"]
        states = {side: copy.deepcopy(model.state_dict()) for side, model in (("ref", ref), ("mine", mine))}

        def iteration(side, model, seed, b=batch):
            model.train()
            for p in model.parameters():
                p.grad = None
            torch.manual_seed(seed)
            total, info, _, encodings = trainers[side].compute_losses(_Wrapped(model), b)
            total.backward()
            composer = model.object_composer.inner if side == "mine" else model.object_composer
            grads = {"composer." + n: p.grad.clone() for n, p in composer.named_parameters() if p.grad is not None}
            grads.update({"decoder." + n: p.grad.clone() for n, p in model.autoencoder_model.named_parameters() if p.grad is not None})
            return total.detach(), info, grads, encodings

        def float64_iteration(seed):
            mine.load_state_dict(states["mine"])
            mine.double()
            try:
                with oracle_in_float64(), check_dropin._Float32Draws():
                    return iteration("mine", mine, seed, _Batch(to_double(tensors)))
            finally:
                mine.float()
                mine.load_state_dict(states["mine"])

        # The seed of the compared iteration: (a) no object's BatchNorm is left with one sample (both sides raise: next seed);
        # (b) the float64 run of the SAME iteration makes the same discrete choices - the weighted patch sampler compares pixel
        # centres with box edges, and in float64 a box edge can fall on the other side of a pixel: the patch moves, the float64
        # gradients belong to another patch and "no farther from float64 than the reference" would compare both fp32 sides with
        # something unrelated (seen on minecraft with seed 41: both sides 1.02 of the largest gradient away).  Such a seed is
        # SKIPPED BY NAME, not passed.
        out, exact = {}, None
        seed = None
        skipped = []
        for candidate in range(41, 70):
            for side, model in (("ref", ref), ("mine", mine)):
                model.load_state_dict(states[side])
            try:
                out["ref"] = iteration("ref", ref, candidate)
            except ValueError:
                skipped.append((candidate, "an object's train-mode BatchNorm saw one sample"))
                continue
            out["mine"] = iteration("mine", mine, candidate)
            exact = float64_iteration(candidate)
            if abs(float(exact[0]) - float(out["mine"][0])) > 1e-4 * abs(float(out["mine"][0])):
                skipped.append((candidate, f"the float64 run sampled another patch (loss {float(exact[0]):.6f} vs {float(out['mine'][0]):.6f})"))
                continue
            seed = candidate
            break
        assert seed is not None, skipped
        print(f"[consumers, {world}] iteration seed {seed}" + ("".join(f"; seed {c} skipped: {why}" for c, why in skipped)))
        info_ref, info_mine = out["ref"][1], out["mine"][1]
        keys_same = sorted(info_ref) == sorted(info_mine)
        worst_info = max(abs(float(info_ref[k]) - float(info_mine[k])) / max(abs(float(info_ref[k])), 1e-3) for k in info_ref) if keys_same else float("inf")
        print(f"[consumers, {world}] compute_losses: {len(info_ref)} loss_info entries, same keys: {keys_same}, worst relative difference "
              f"{worst_info:.2e}; total loss {float(out['ref'][0]):.6f} / {float(out['mine'][0]):.6f}")
        ok &= keys_same and worst_info < 2e-3
        # gradients of the total loss: float64 arbitration, as in check_dropin.py (a float64 run that made the SAME discrete choices)
        same_sets = set(out["ref"][2]) == set(out["mine"][2]) == set(exact[2])
        farther, err = {}, {"ref": 0.0, "mine": 0.0}
        if same_sets:
            for prefix in ("composer.", "decoder."):
                names = [n for n in exact[2] if n.startswith(prefix)]
                if not names:
                    continue
                scale = max(float(exact[2][n].abs().max()) for n in names)
                for n in names:
                    a, b, e = out["ref"][2][n].double(), out["mine"][2][n].double(), exact[2][n].double()
                    err_ref, err_mine = float((a - e).abs().max()), float((b - e).abs().max())
                    err["ref"], err["mine"] = max(err["ref"], err_ref / scale), max(err["mine"], err_mine / scale)
                    if err_mine > 4.0 * err_ref + 1e-6 * scale:
                        farther[n] = (err_mine, err_ref)
        worst_name = max(exact[2], key=lambda n: float((out["mine"][2][n].double() - exact[2][n].double()).abs().max())) if same_sets else None
        print(f"    (float64 total loss {float(exact[0]):.6f}; tensor farthest from it: {worst_name})")
        print(f"[consumers, {world}] gradient of the trainer's total loss: {len(out['ref'][2])} tensors, same set: {same_sets}; float64 "
              f"arbitration: worst error relative to the group's largest gradient - reference {err['ref']:.2e}, swapped model "
              f"{err['mine']:.2e}; farther than 4 x the reference: {farther}")
        ok &= same_sets and not farther
        # the direct yardstick stands beside the arbitration: relative L2 difference of all gradients of a group between the two fp32 sides
        same_choices = abs(float(exact[0]) - float(out["mine"][0])) <= 1e-4 * abs(float(out["mine"][0]))
        ok &= same_choices and max(err.values()) < 5e-2          # (an arbiter 100 % away from both sides arbitrates nothing)
        l2 = {}
        for prefix in ("composer.", "decoder."):
            names = [n for n in out["ref"][2] if n.startswith(prefix)]
            if names:
                num = sum(float((out["ref"][2][n].double() - out["mine"][2][n].double()).square().sum()) for n in names)
                den = sum(float(out["ref"][2][n].double().square().sum()) for n in names)
                l2[prefix[:-1]] = (num / max(den, 1e-300)) ** 0.5
        print(f"    float64 run made the same discrete choices (same loss): {same_choices}; relative L2 difference reference vs swapped model: "
              + ", ".join(f"{k} {v:.2e}" for k, v in l2.items()))
        ok &= max(l2.values()) < 2e-3
        if write:
            os.makedirs(OUT, exist_ok=True)
            path = os.path.join(OUT, f"{world}_loss_info.npz")
            data = {"info/" + k: np.float64(v) for k, v in info_ref.items()}
            data["total_loss"] = np.float64(out["ref"][0])
            data["meta"] = np.frombuffer(repr({"world": world, "seed": seed, "image_size": list(image_size), "source":
                                               "TrainerMultiresolutionBackpropagatedDecoder.compute_losses of the reference on its own model "
                                               "(oracle/check_consumers.py write)"}).encode(), dtype=np.uint8)
            np.savez_compressed(path, **data)
            print(f"  wrote {path}")

        # ---- the play loop's first frame ---------------------------------------------------------------------------
        from model.playable_environment_model import PlayableEnvironmentModel
        ref.load_state_dict(states["ref"]), mine.load_state_dict(states["mine"])
        ref.eval(), mine.eval()
        frames = {}
        for side, model in (("ref", ref), ("mine", mine)):
            holder = types.SimpleNamespace(
                object_animation_models=[], object_id_helper=types.SimpleNamespace(dynamic_objects_count=0), environment_model=model,
                get_reconstructed_observations_from_render_results=lambda r: PlayableEnvironmentModel.get_reconstructed_observations_from_render_results(None, r))
            frames[side] = PlayableEnvironmentModel.initialize_interactive_generation(
                holder, *[tensors[k].clone() for k in OBS_KEYS], batch_idx=0, observation_idx=1)
        image_a, enc_a = frames["ref"]
        image_b, enc_b = frames["mine"]
        diff = float((image_a - image_b).abs().max())
        enc_same = sorted(enc_a) == sorted(enc_b) and all(
            torch.allclose(enc_a[k].float(), enc_b[k].float(), rtol=1e-4, atol=1e-5) for k in enc_a if torch.is_tensor(enc_a[k]))
        print(f"[consumers, {world}] PlayableEnvironmentModel.initialize_interactive_generation: frame {tuple(image_b.shape)} "
              f"max|diff| {diff:.2e}, scene encoding keys / values equal: {enc_same}")
        ok &= diff < 1e-4 and enc_same and tuple(image_a.shape) == tuple(image_b.shape) == (image_size[0], image_size[1], 3)
    finally:
        em.camera_rays = original_camera_rays
    return ok


def main(write=False):
    refshim.install()
    torch.cuda.synchronize = lambda *a, **k: None       # utils/torch_time_meter.py synchronises around every timed section
    ok = True
    for world in ("tennis", "minecraft"):
        ok &= run_world(world, write)
    print("CONSUMERS " + ("OK" if ok else "FAILED"))
    return ok


if __name__ == "__main__":
    sys.exit(0 if main(write=len(sys.argv) > 1 and sys.argv[1] == "write") else 1)
