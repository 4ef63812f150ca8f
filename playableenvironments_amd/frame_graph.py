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

import contextlib
import gc
import os
from typing import Dict, Tuple

import torch

SCENE_KEYS = ("camera_rotations", "camera_translations", "focals", "object_rotation_parameters",
              "object_translation_parameters", "object_style", "object_deformation", "object_in_scene")

#: ROCm 7.0.2, HIP runtime with AQL packet capture (its default): replays of a recorded TRAINING step go wrong once the host has
#: synchronised between them - the graph's memset nodes stop executing (torch's multi-block reductions, ``x.mean()`` /
#: ``x.sum()``, zero their semaphores with one: the loss freezes, accumulators keep stale values, gradients explode).  Measured
#: with tools/perf/perf_train_graph.py; neither one replay in flight at a time nor waiting on events avoids it.  With the capture
#: path off the replays are correct (tests/graph_step_check.py) - and cost what eager launches cost on the device side, so
#: what a recording buys is HOST time.  The switch has to be in the environment before the process makes its first HIP call.
GRAPH_RUNTIME_SWITCH = ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


@contextlib.contextmanager
def _no_collections_while_recording():
    """Python's cyclic garbage collector can run at any allocation - also between the launches of a recording - and a
    ``torch.cuda.CUDAGraph`` that dies there (an older recording kept alive by a reference cycle) calls hipGraphDestroy inside
    the capture: "operation not permitted when stream is capturing", raised from a destructor, i.e. the process aborts (seen
    with a deleted ``FrameGraph`` still pending collection when ``EnvironmentModel.frame_replay`` recorded its next frame).
    Collect BEFORE the recording starts, keep the collector off during it."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def graph_runtime_is_safe() -> bool:
    return os.environ.get(GRAPH_RUNTIME_SWITCH[0]) == GRAPH_RUNTIME_SWITCH[1]


class UnsafeRecording(RuntimeError):
    """A recorded graph holds memset nodes and the runtime switch that keeps them alive across host synchronisations is not set."""


def node_census(graph: "torch.cuda.CUDAGraph") -> Dict[str, int]:
    """Node counts of a recording made with ``torch.cuda.CUDAGraph(keep_graph=True)`` (``pr_graph_node_census``)."""
    import ctypes as C
    from . import _lib
    counts = (C.c_int32 * 4)()
    _lib.check(_lib.load().pr_graph_node_census(C.c_void_p(graph.raw_cuda_graph()), counts), "pr_graph_node_census")
    return {"nodes": counts[0], "kernels": counts[1], "memsets": counts[2], "memcpys": counts[3]}


def _checked_recording(graph: "torch.cuda.CUDAGraph", what: str) -> Dict[str, int]:
    """The renderer's own launches never record a memset (every zero fill of the library is a kernel), the torch modules around
    it may (multi-block ``sum`` / ``mean`` reductions zero their semaphores with one, ``torch.zeros`` of a large tensor): such a
    recording replays correctly only with ``GRAPH_RUNTIME_SWITCH`` in the environment (see above) - without it the recording is
    refused, which leaves the call eager (``EnvironmentModel.frame_replay``) or raises to the caller (``FrameGraph``)."""
    census = node_census(graph)
    if census["memsets"] and not graph_runtime_is_safe():
        graph.reset()
        raise UnsafeRecording(
            f"{what}: the recording holds {census['memsets']} memset node(s) (of {census['nodes']} nodes) from torch modules around the "
            f"renderer; on this HIP runtime they stop executing after a host synchronisation between replays unless "
            f"{GRAPH_RUNTIME_SWITCH[0]}={GRAPH_RUNTIME_SWITCH[1]} is in the environment before the first HIP call")
    return census


OBSERVATION_KEYS = ("observations", "camera_rotations", "camera_translations", "focals", "bounding_boxes", "bounding_boxes_validity",
                    "global_frame_indexes", "video_frame_indexes", "video_indexes")


class FrameGraph:
    """Captures an evaluation frame once and replays it for new inputs of the same shapes.

    ``mode="scene_encodings"`` (default): ``model(..., mode="scene_encodings")`` - pose math, ``pr_camera_rays``, the renderer;
    ``scene`` holds the tensors of ``SCENE_KEYS``.  ``mode="observations"``: ``model.forward_from_observations`` - the CNN
    encoders / pose estimators in front of it as well (what ``render_full_frame_from_observations`` of the reference's evaluators
    runs, evaluation/reconstructed_dataset_creator.py:121); ``scene`` holds the tensors of ``OBSERVATION_KEYS`` (``Batch.to_tuple``
    order without actions / rewards / dones).  ``patch_stride=[4, 8]`` gives the strided grids the reference's autoencoder
    subclasses render (environment_model_backpropagated_autoencoder.py:173-236); ``decoder_features=[64, 128]`` additionally
    emits the decoder's channels-first maps.

    >>> graph = FrameGraph(model, example_scene, image_size=(288, 512), patch_stride=[4, 8])     # model.eval(), tensors on the GPU
    >>> results = graph.render(next_scene)                                   # dict of STATIC output tensors
    The result tensors are overwritten by the next ``render``; clone what must survive."""

    def __init__(self, model, scene: Dict[str, torch.Tensor], image_size: Tuple[int, int] = None, perturb: bool = False,
                 patch_stride=0, upsample_factor: float = 1.0, canonical_pose: bool = False, warmup: int = 2,
                 mode: str = "scene_encodings", decoder_features=None):
        if model.training:
            raise ValueError("FrameGraph replays an evaluation render: call model.eval() first (train-mode BatchNorm updates "
                             "buffers and draws noise on every call)")
        if perturb:
            raise ValueError("perturb=True draws fresh noise on every call, which a captured graph cannot do")
        if mode not in ("scene_encodings", "observations"):
            raise ValueError(f"unknown mode {mode!r} (expected 'scene_encodings' or 'observations')")
        self.model = model
        self.mode = mode
        self.keys = SCENE_KEYS if mode == "scene_encodings" else OBSERVATION_KEYS
        self.inputs = {k: scene[k].detach().clone() for k in self.keys}
        device = self.inputs["camera_rotations"].device
        if device.type != "cuda":
            raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
        extra = {} if decoder_features is None else {"_decoder_features": list(decoder_features)}
        if mode == "scene_encodings":
            if image_size is None:
                raise ValueError("mode='scene_encodings' needs image_size")
            inputs = self.inputs       # (the closure must not hold `self`: a reference cycle would leave the graph to the cyclic collector)
            self._call = lambda: model(*[inputs[k] for k in SCENE_KEYS[:3]], image_size, *[inputs[k] for k in SCENE_KEYS[3:]],
                                       0, False, patch_stride=patch_stride, upsample_factor=upsample_factor,
                                       canonical_pose=canonical_pose, mode="scene_encodings", **extra)
        else:
            inputs = self.inputs
            self._call = lambda: model(*[inputs[k] for k in OBSERVATION_KEYS], 0, False, patch_stride=patch_stride,
                                       upsample_factor=upsample_factor, canonical_pose=canonical_pose, mode="observations", **extra)
        # warm-up on a side stream (packs the weights, sizes the workspace, fills the host-side caches, lets the convolution
        # library pick its algorithms), then capture
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        composer = model.object_composer
        was_recording = getattr(model, "_in_replay", False)
        model._in_replay = True              # (the model's own automatic recording - EnvironmentModel.frame_replay - stays out of this one)
        try:
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(max(1, warmup)):
                    self._call()
            torch.cuda.current_stream(device).wait_stream(side)
            self._workspace = composer._workspace      # the graph writes through this pointer: keep it alive
            self._weights_version = self._signature()
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            try:
                with _no_collections_while_recording(), torch.cuda.graph(self.graph), torch.no_grad():
                    self.results = self._call()
            except BaseException:
                # (a failed capture can take the process down when the half-built graph is destroyed: say why first)
                import traceback
                traceback.print_exc()
                raise
            self.census = _checked_recording(self.graph, f"FrameGraph(mode={mode!r})")
            self.graph.instantiate()
        finally:
            model._in_replay = was_recording
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
        owner = composer if self.mode == "scene_encodings" else self.model       # (observations: the encoders' weights too)
        return (tuple((p.data_ptr(), p._version) for p in owner.parameters()), composer.precision,
                bool(composer.gate_feature_head), composer.state_epoch, composer.weights_epoch,
                None if composer.object_entry_fields is None else tuple(composer.object_entry_fields))

    def render(self, scene: Dict[str, torch.Tensor]) -> Dict:
        """Copies the inputs into the captured buffers and replays the frame."""
        if self._signature() != self._weights_version:
            raise RuntimeError("the composer's parameters, precision or annealing step changed since the frame was captured: "
                               "build a new FrameGraph")
        for k in self.keys:
            src = scene[k]
            if src.shape != self.inputs[k].shape:
                raise ValueError(f"{k}: shape {tuple(src.shape)} differs from the captured {tuple(self.inputs[k].shape)}")
            self.inputs[k].copy_(src, non_blocking=True)
        self.graph.replay()
        return self.results


class CapturedCall:
    """``fn(*tensors)`` recorded once as a HIP graph and replayed for new values of the same-shaped tensors (copied into the
    recorded input buffers).  The building block of ``EnvironmentModel.frame_replay``; results are the graph's static tensors."""

    def __init__(self, fn, tensors, warmup: int = 2, what: str = "CapturedCall"):
        self.inputs = [t.detach().clone() for t in tensors]
        device = self.inputs[0].device
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                fn(*self.inputs)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        # capture_error_mode="thread_local": other threads of the process (a DataLoader's pin-memory thread, a writer) keep making
        # HIP calls while this thread records - the default "global" mode would fail THEIR calls
        with _no_collections_while_recording(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.results = fn(*self.inputs)
        self.census = _checked_recording(self.graph, what)     # (raises UnsafeRecording: the caller stays eager)
        self.graph.instantiate()

    def replay(self, tensors):
        for dst, src in zip(self.inputs, tensors):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.results


class GraphedStep:
    """A whole training iteration - renderer forward, loss, ``backward()``, optimiser step - recorded once as a HIP graph
    and replayed: ``step_fn()`` (no arguments; it reads its inputs from tensors that stay in place, e.g. a ``Batch`` arena
    the caller copies every new batch into) is run ``warmup`` times on a side stream, recorded, and every ``replay()``
    re-runs the recorded launches - the renderer's ~40 and the few hundred small torch kernels around them - with one host call.
    On ROCm 7.0.2 this buys host time only, and needs a runtime switch (``GRAPH_RUNTIME_SWITCH`` above).

    What makes the renderer recordable: the library never allocates, synchronises or reads anything back, launches on the
    caller's stream, and draws the noise of a recorded call from a seed WORD on the device (``pr_call_t.noise_seed_device``,
    filled by torch's device generator inside the graph), so that every replay perturbs differently and its backward
    pass regenerates exactly its own forward pass's noise.  The pixel samplers draw on the device too (ray_sampling.py).

    Requirements on ``step_fn``: fixed shapes; no host synchronisation (``.item()``, ``.cpu()``); an optimiser built with
    ``capturable=True``; ``optimizer.zero_grad(set_to_none=True)`` inside it (gradients then live in the graph's pool; with a
    parameter arena, ``parallel.flat_gradient`` between ``backward()`` and ``step()`` hands the views' gradients to the arena and
    clears them, so that the arena's ``zero_grad`` is enough).

    Two things a replay cannot see, handled through ``modules`` (the modules whose ``ObjectComposer`` s the step renders with):
    (1) replays update the parameters ON THE DEVICE, the Python version counters of the tensors do not move - after every
    replay the composers forget their packed weight copies (``ObjectComposer.after_graph_replay``), so that an eager
    evaluation / validation render between replays packs the current weights; (2) the ray benders' annealing weights are
    computed on the host from ``set_step`` and baked into the recorded kernel arguments - ``replay()`` raises as soon as a
    composer's ``set_step`` has moved them away from the recorded values (while the annealing schedule is still running,
    i.e. step < num_steps: record again; afterwards the weights are constant and ``set_step`` is harmless).
    A recorded training call cannot raise from its replays: the BatchNorm sample-count check of ``ObjectComposer`` is left
    to the caller (``ObjectComposer.last_normalised_samples``).  Whatever ``step_fn`` returns is returned by ``replay()``
    as the same (static) tensors, overwritten by the next replay.

    >>> opt = torch.optim.Adam(params, lr=1e-4, capturable=True, fused=True)
    >>> step = GraphedStep(lambda: train_iteration(batch_arena, model, opt))
    >>> for batch in loader: batch_arena.copy_from(batch); loss = step.replay()"""

    def __init__(self, step_fn, warmup: int = 3, device=None, modules=()):
        from .object_composer import ObjectComposer
        self.composers = []
        for module in ([modules] if isinstance(modules, torch.nn.Module) else list(modules)):
            self.composers.extend(m for m in module.modules() if isinstance(m, ObjectComposer) and m not in self.composers)
        if not graph_runtime_is_safe():
            raise RuntimeError(f"GraphedStep needs {GRAPH_RUNTIME_SWITCH[0]}={GRAPH_RUNTIME_SWITCH[1]} in the environment before the first HIP "
                               "call of the process: with the runtime's AQL packet capture, replays of a recorded training step "
                               "compute garbage once the host has synchronised between them (ROCm 7.0.2; see frame_graph.py)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):      # packs weights, sizes workspaces, creates optimiser state, raises LDS limits
                step_fn()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with _no_collections_while_recording(), torch.cuda.graph(self.graph):
            self.outputs = step_fn()
        self.replays = 0
        for c in self.composers:
            c.resolve_host_step()          # (after a checkpoint load the host learns the step from the device, once)
        self._annealing = [c.annealing_fingerprint() for c in self.composers]
        for c in self.composers:
            c.after_graph_replay()         # (the capture itself left packed copies keyed on counters the replays will not move)

    def replay(self):
        for c, recorded in zip(self.composers, self._annealing):
            if c.annealing_fingerprint() != recorded:
                raise RuntimeError("the ray benders' annealing weights changed since the step was recorded (set_step / load_state_dict): "
                                   "they are kernel arguments of the recorded launches - record the step again")
        self.graph.replay()
        self.replays += 1
        for c in self.composers:
            c.after_graph_replay()
        return self.outputs
