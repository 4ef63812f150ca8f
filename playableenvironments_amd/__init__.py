"""playableenvironments_amd - MI355X-native volumetric renderer for Playable Environments.

Public surface (mirrors the reference's modules for the renderer hot path only):
  ObjectComposer            drop-in for model/object_composer.py:ObjectComposer (forward, forward_expected_positions,
                            autograd through pr_render_backward)
  EnvironmentModel          model/environment_model.py:EnvironmentModel: every forward mode (scene encodings, observations,
                            scene-encoding-only, pose / keypoint consistency), render_full_frame_*, render_sharded
  encoders                  the object encoders / pose estimators that produce the renderer's inputs, with their
                            region-of-interest crop on a HIP kernel (roi_pool); batching: the pinned one-copy Batch path
  FrameGraph                a whole evaluation frame captured and replayed as one HIP graph
  ray_sampling, wire_format pixel / ray samplers and the renderer <-> decoder tensor glue of the reference
  parallel                  frame shards, overlapped feature all-gather, gradient all-reduce (torch.distributed / RCCL)
  configs / synthetic       shipped renderer configurations and seeded synthetic scenes
"""
from . import batching, configs, encoders, parallel, ray_sampling, synthetic, wire_format  # noqa: F401
from .environment_model import EnvironmentModel  # noqa: F401
from .frame_graph import FrameGraph  # noqa: F401
from .object_composer import ObjectComposer, ObjectIDsHelper  # noqa: F401

__all__ = ["ObjectComposer", "ObjectIDsHelper", "EnvironmentModel", "FrameGraph", "configs", "synthetic", "parallel",
           "ray_sampling", "wire_format", "encoders", "batching"]
