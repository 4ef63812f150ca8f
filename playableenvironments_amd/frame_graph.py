"""A rendered frame as a HIP graph (torch.cuda.CUDAGraph): for the interactive playback loop of a playable environment.

``EnvironmentModel.forward(mode="scene_encodings")`` under ``torch.no_grad()`` is a fixed sequence of kernel launches
for a fixed image size and object layout: pose math (a few dozen small torch kernels), ``pr_camera_rays``, the renderer's
placement / compaction / MLP / compositing launches.  The library never allocates or synchronises and takes its
stream from the caller, so the whole frame captures as it is; replaying it costs ~0.04 ms of host time per frame instead
of ~3.5 ms of Python and launch calls, which matters once a frame is a few milliseconds of GPU work (minecraft at
256x256: 10.8 ms exact fp32, 4.8 ms split precision) or the host is busy with the rest of the application.

The graph holds raw device pointers.  What it captured must stay alive and in place: the input buffers (owned here),
the weights (re-capture after an optimiser step or ``load_state_dict``), the composer's workspace (kept alive here).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

SCENE_KEYS = ("camera_rotations", "camera_translations", "focals", "object_rotation_parameters",
              "object_translation_parameters", "object_style", "object_deformation", "object_in_scene")


class FrameGraph:
    """Captures ``model(..., mode="scene_encodings")`` once and replays it for new scene encodings of the same shapes.

    >>> graph = FrameGraph(model, example_scene, image_size=(256, 256))      # model.eval(), tensors on the GPU
    >>> results = graph.render(next_scene)                                   # dict of STATIC output tensors
    The result tensors are overwritten by the next ``render``; clone what must survive."""

    def __init__(self, model, scene: Dict[str, torch.Tensor], image_size: Tuple[int, int], perturb: bool = False,
                 patch_stride=0, upsample_factor: float = 1.0, canonical_pose: bool = False, warmup: int = 2):
        if model.training:
            raise ValueError("FrameGraph replays an evaluation render: call model.eval() first (train-mode BatchNorm updates "
                             "buffers and draws noise on every call)")
        if perturb:
            raise ValueError("perturb=True draws fresh noise on every call, which a captured graph cannot do")
        self.model = model
        self.inputs = {k: scene[k].detach().clone() for k in SCENE_KEYS}
        device = self.inputs["camera_rotations"].device
        if device.type != "cuda":
            raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
        self._call = lambda: model(*[self.inputs[k] for k in SCENE_KEYS[:3]], image_size, *[self.inputs[k] for k in SCENE_KEYS[3:]],
                                   0, False, patch_stride=patch_stride, upsample_factor=upsample_factor,
                                   canonical_pose=canonical_pose, mode="scene_encodings")
        # warm-up on a side stream (packs the weights, sizes the workspace, fills the host-side caches), then capture
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                self._call()
        torch.cuda.current_stream(device).wait_stream(side)
        composer = model.object_composer
        self._workspace = composer._workspace      # the graph writes through this pointer: keep it alive
        self._weights_version = self._signature()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.results = self._call()
        # the graph also reads the composer's packed weight buffers through raw pointers: hold them, so that a repack
        # (another precision, a differentiable call) can drop them from the composer's cache without freeing them
        self._packed = [entry[1] for entry in composer._packed.values()]
        self._linspace = list(composer._linspace.values())
        if composer._workspace is not self._workspace:
            raise RuntimeError("the composer re-allocated its workspace during the capture")

    def _signature(self):
        """Everything the captured launches baked in besides the input buffers: parameter storages and values, the
        arithmetic precision (selects the kernel and the packed layout), the sigma gate, and the annealing step of the ray benders
        (their octave weights are kernel arguments)."""
        composer = self.model.object_composer
        # state_epoch counts set_step / load_state_dict / .to() calls (reading the step buffers back would synchronise)
        return (tuple((p.data_ptr(), p._version) for p in composer.parameters()), composer.precision,
                bool(composer.gate_feature_head), composer.state_epoch)

    def render(self, scene: Dict[str, torch.Tensor]) -> Dict:
        """Copies the scene encoding into the captured input buffers and replays the frame."""
        if self._signature() != self._weights_version:
            raise RuntimeError("the composer's parameters, precision or annealing step changed since the frame was captured: "
                               "build a new FrameGraph")
        for k in SCENE_KEYS:
            src = scene[k]
            if src.shape != self.inputs[k].shape:
                raise ValueError(f"{k}: shape {tuple(src.shape)} differs from the captured {tuple(self.inputs[k].shape)}")
            self.inputs[k].copy_(src, non_blocking=True)
        self.graph.replay()
        return self.results
