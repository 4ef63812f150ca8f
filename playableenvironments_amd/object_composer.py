"""Drop-in for the reference's ``ObjectComposer`` running on the HIP renderer.

Same constructor (``ObjectComposer(config)``), same ``forward`` signature and result-dict schema
as model/object_composer.py:18-57, :786-892 of the reference, same ``state_dict`` keys.  The body
of ``forward`` marshals raw device pointers into ``pr_render_forward`` (include/playrender.h); all
arithmetic runs in hand-written gfx950 kernels.  There is no CPU / PyTorch fallback: without the
built extension or without a GPU tensor the call raises.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
import warnings
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .modules import OBJECT_MODEL_CLASSES, REGISTRATION_EPOCH as _REGISTRATION_EPOCH, ModuleList, RayBendingStyleNerfModel, Tracked, \
    tree_is_tracked

# (_REGISTRATION_EPOCH: moved by this package's own module classes whenever one of them (re-)registers a parameter, buffer or
# submodule - modules.Tracked; the composer's cached parameter lists and model structs hold Python objects and raw pointers, so
# they are rebuilt when its module tree may have changed under them.  No process-wide registration hooks.)

ENTRY_KEYS = ("integrated_features", "opacity", "weights", "depth", "disparity",
              "integrated_displacements_magnitude", "integrated_divergence")

# Composers whose last backward pass produced parameter gradients and whose optimiser has not stepped since.  torch's FUSED
# optimisers update the storages without moving the tensors' version counters, so the packed MFMA copies (and recorded frames) key on
# ``weights_epoch`` as well - and that has to move AFTER the step: a validation / logging render between ``backward()`` and
# ``optimizer.step()`` would otherwise pack the pre-step weights under the key the post-step forward looks up.  The hook below is
# torch's optimiser post-step hook, installed by the first such backward pass (not on import); it only touches composers in this
# set, and only when the optimiser that stepped holds (views of) their parameters' storages.
_AWAITING_STEP: "weakref.WeakSet" = weakref.WeakSet()
_AWAITING_LOCK = threading.Lock()      # (backward passes run on autograd's threads, optimisers step on the caller's)
_STEP_HOOK = []


def _after_optimizer_step(optimizer, args, kwargs):
    if not _AWAITING_STEP:
        return
    try:
        storages = {p.untyped_storage().data_ptr() for group in optimizer.param_groups for p in group["params"] if torch.is_tensor(p)}
    except Exception:              # (a tensor without a plain storage among the optimiser's: assume it may own ours)
        storages = None
    with _AWAITING_LOCK:
        waiting = list(_AWAITING_STEP)
    for composer in waiting:
        if storages is None or not storages.isdisjoint(composer._parameter_storages()):
            composer.weights_epoch += 1
            with _AWAITING_LOCK:
                _AWAITING_STEP.discard(composer)


def _watch_optimizer_steps(composer):
    with _AWAITING_LOCK:
        _AWAITING_STEP.add(composer)
        if not _STEP_HOOK:
            from torch.optim.optimizer import register_optimizer_step_post_hook
            _STEP_HOOK.append(register_optimizer_step_post_hook(_after_optimizer_step))


class ObjectIDsHelper:
    """Object instance <-> model index bookkeeping (model/utils/object_ids_helper.py:4-153): static
    models first; model m owns ``object_parameters_encoder[m].objects_count`` consecutive instances."""

    def __init__(self, config):
        self.config = config
        model = config["model"]
        self.static_object_models_count = model["static_object_models"]
        self.object_models_count = len(model["object_models"])
        self.dynamic_object_models_count = self.object_models_count - self.static_object_models_count
        self._counts = [int(e["objects_count"]) for e in model["object_parameters_encoder"]]
        self.model_idx_by_object_idx_map = {}
        self.first_object_idx_by_model_idx_map = {}
        idx = 0
        for m in range(self.object_models_count):
            self.first_object_idx_by_model_idx_map[m] = idx
            for _ in range(self._counts[m]):
                self.model_idx_by_object_idx_map[idx] = m
                idx += 1
        self.objects_count = idx
        self.static_objects_count = sum(self._counts[: self.static_object_models_count])
        self.dynamic_objects_count = self.objects_count - self.static_objects_count

    def is_static(self, model_idx: int) -> bool:
        return model_idx < self.static_object_models_count

    def is_dynamic(self, model_idx: int) -> bool:
        return not self.is_static(model_idx)

    def objects_count_by_model_idx(self, model_idx: int) -> int:
        return self._counts[model_idx]

    def model_idx_by_object_idx(self, object_idx: int) -> int:
        return self.model_idx_by_object_idx_map[object_idx]

    def object_idx_by_dynamic_object_idx(self, dynamic_object_idx: int) -> int:
        object_idx = dynamic_object_idx + self.static_objects_count
        if object_idx >= self.objects_count:
            raise Exception(f"The provided object id {dynamic_object_idx} is out of range")
        return object_idx

    def dynamic_object_idx_by_object_idx(self, object_idx: int) -> int:
        if object_idx < self.static_objects_count:
            raise Exception(f"The provided object id {object_idx} does not correspond to a dynamic object")
        return object_idx - self.static_objects_count

    def dynamic_object_idx_range_by_model_idx(self, model_idx: int):
        """[begin, end) of the dynamic-object ids a dynamic model owns (object_ids_helper.py:137-153)."""
        if not self.is_dynamic(model_idx):
            raise Exception(f"Model id {model_idx} does not refer to a dynamic object")
        first = self.dynamic_object_idx_by_object_idx(self.first_object_idx_by_model_idx_map[model_idx])
        return first, first + self.objects_count_by_model_idx(model_idx)


#: limits of the kernels behind the C ABI (include/playrender.h: PR_MAX_OBJECTS / PR_MAX_LAYERS / PR_MAX_OCTAVES; csrc/pr_common.h:
#: MAX_WIDTH = 256 padded columns per layer, MAX_ENC = 128 padded input columns) - every shipped configuration fits
MAX_LAYER_WIDTH = 256
MAX_ENCODING_WIDTH = 128


def validate_config_limits(config) -> None:
    """Raises ``ValueError`` naming the configuration key when the renderer's kernels cannot run the configuration - at
    construction, not at the first render.  What the reference accepts and this renderer refuses: more than PR_MAX_OBJECTS
    object instances, layers wider than 256, more than PR_MAX_LAYERS layers / PR_MAX_OCTAVES octaves, encodings wider than 128
    columns, ``append_original: False`` (model/nerf_models/adain_style_nerf_model.py:24-45, model/positional_encoder.py:41-65)."""
    def pad(v):
        return (int(v) + 31) // 32 * 32

    model = config["model"]
    instances = sum(int(e["objects_count"]) for e in model["object_parameters_encoder"])
    if not 1 <= instances <= _lib.PR_MAX_OBJECTS:
        raise ValueError(f"config['model']['object_parameters_encoder'][*]['objects_count'] add up to {instances} object instances: the HIP "
                         f"renderer composes 1..{_lib.PR_MAX_OBJECTS} (PR_MAX_OBJECTS, include/playrender.h)")
    features = set()
    for i, m in enumerate(model["object_models"]):
        where = f"config['model']['object_models'][{i}]"
        n, b = m["nerf_model"], m["ray_bender_model"]
        skybox = n["architecture"].endswith("skybox_adain_style_nerf_model_v3")
        din = 6 if skybox else 3
        pe = n["position_encoder"]
        if not pe["append_original"]:
            raise ValueError(f"{where}['nerf_model']['position_encoder']['append_original'] = False is not supported by the HIP renderer "
                             "(the kernels' encodings always start with the raw input)")
        if not 0 <= pe["octaves"] <= _lib.PR_MAX_OCTAVES or pad(din * (1 + 2 * pe["octaves"])) > MAX_ENCODING_WIDTH:
            raise ValueError(f"{where}['nerf_model']['position_encoder']['octaves'] = {pe['octaves']}: at most {_lib.PR_MAX_OCTAVES} octaves "
                             f"(PR_MAX_OCTAVES) and an encoding of at most {MAX_ENCODING_WIDTH} columns ({din} x (1 + 2 x octaves))")
        w = n["layers_width"]
        if w < 2 or pad(w) > MAX_LAYER_WIDTH:
            raise ValueError(f"{where}['nerf_model']['layers_width'] = {w}: the HIP renderer's MLP kernels hold layers of 2..{MAX_LAYER_WIDTH} "
                             "units (MAX_WIDTH, csrc/pr_common.h)")
        f = n["output_features"]
        if f < 1 or pad(f) > MAX_LAYER_WIDTH:
            raise ValueError(f"{where}['nerf_model']['output_features'] = {f}: 1..{MAX_LAYER_WIDTH}")
        features.add(f)
        if not 2 <= n["backbone_layers_count"] <= _lib.PR_MAX_LAYERS:
            raise ValueError(f"{where}['nerf_model']['backbone_layers_count'] = {n['backbone_layers_count']}: 2..{_lib.PR_MAX_LAYERS} "
                             "(PR_MAX_LAYERS, include/playrender.h)")
        if not 1 <= n["skip_layer_idx"] < n["backbone_layers_count"]:
            raise ValueError(f"{where}['nerf_model']['skip_layer_idx'] = {n['skip_layer_idx']}: 1..backbone_layers_count - 1")
        if b["architecture"].endswith("positional_ray_bender_model"):
            if skybox:
                raise ValueError(f"{where}: a positional ray bender on the skybox model is not supported")
            bpe = b["position_encoder"]
            if not bpe["append_original"]:
                raise ValueError(f"{where}['ray_bender_model']['position_encoder']['append_original'] = False is not supported by the HIP "
                                 "renderer")
            width_in = 3 * (1 + 2 * bpe["octaves"]) + m["deformation_features"]
            if not 0 <= bpe["octaves"] <= _lib.PR_MAX_OCTAVES or pad(width_in) > MAX_ENCODING_WIDTH:
                raise ValueError(f"{where}['ray_bender_model']['position_encoder']['octaves'] = {bpe['octaves']} with deformation_features = "
                                 f"{m['deformation_features']}: at most {_lib.PR_MAX_OCTAVES} octaves and {MAX_ENCODING_WIDTH} input columns "
                                 "(3 x (1 + 2 x octaves) + deformation_features)")
            if b["layers_width"] < 1 or pad(b["layers_width"]) > MAX_LAYER_WIDTH:
                raise ValueError(f"{where}['ray_bender_model']['layers_width'] = {b['layers_width']}: 1..{MAX_LAYER_WIDTH}")
            if not 2 <= b["layers_count"] <= _lib.PR_MAX_LAYERS:
                raise ValueError(f"{where}['ray_bender_model']['layers_count'] = {b['layers_count']}: 2..{_lib.PR_MAX_LAYERS}")
            if not 1 <= b["skip_layer_idx"] < b["layers_count"]:
                raise ValueError(f"{where}['ray_bender_model']['skip_layer_idx'] = {b['skip_layer_idx']}: 1..layers_count - 1")
    if len(features) > 1:
        raise ValueError(f"config['model']['object_models'][*]['nerf_model']['output_features'] differ ({sorted(features)}): the composer "
                         "merges the objects' features per ray, as the reference's does")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _linear(layer: Optional[nn.Linear]) -> _lib.Linear:
    s = _lib.Linear()
    if layer is not None:
        s.weight = layer.weight.data_ptr()
        s.bias = layer.bias.data_ptr() if layer.bias is not None else None
        s.out_features, s.in_features = layer.weight.shape
    return s


GRAD_KEYS = ("integrated_features", "opacity", "depth", "integrated_displacements_magnitude", "weights", "integrated_divergence")


class _RenderFunction(torch.autograd.Function):
    """autograd node of one renderer call: forward = pr_render_forward with PR_FLAG_SAVE_FOR_BACKWARD,
    backward = pr_render_backward.  Inputs after the bookkeeping arguments: ray_origins, ray_directions (differentiated only
    when they carry a graph: learnable camera parameters), transformation_matrix_w2o, style, deformation, then every
    trainable parameter of the composer."""

    @staticmethod
    def forward(ctx, composer, kwargs, holder, ray_origins, ray_directions, w2o, style, deformation, *params):
        results, state = composer._render(**kwargs, _save=True)
        ctx.composer, ctx.state, ctx.params = composer, state, params
        ctx.ray_shapes = (ray_origins.shape, ray_directions.shape)
        ctx.ray_grads = ray_origins.requires_grad or ray_directions.requires_grad
        # the backward pass reads the parameter storages in place: an in-place update between forward and backward
        # (an optimiser step, a GAN-style second network update) must raise, as torch's saved-tensor check would
        ctx.versions = tuple(p._version for p in composer._parameter_list())
        ctx.shapes = (w2o.shape, style.shape, deformation.shape)
        # inputs in the RENDERER's layouts (EnvironmentModel's fused scene set-up: w2o (N, K, 3, 4), style (N, K, S), deformation
        # (N, K, D)): their gradients are what pr_render_backward writes - no permuting copies on either side
        ctx.prepared = kwargs.get("_prepared") is not None
        ctx.set_materialize_grads(False)      # outputs the loss does not read arrive as None, not as zero tensors
        holder["results"], holder["types"] = results, state["types"]
        flat, skip = [], []
        for ty in state["types"]:
            for name in [f"object_{k}" for k in range(state["K"])] + ["global"]:
                for key in ENTRY_KEYS:
                    t = results[ty][name][key]
                    flat.append(t)
                    if key not in GRAD_KEYS:
                        skip.append(t)
        ctx.layout = state.get("layout")
        if ctx.layout is not None:
            for ty in state["types"]:
                flat.extend(results[ty]["global"]["decoder_features"])
        ctx.exported = bool(kwargs.get("_export"))
        if ctx.exported:
            # per-sample exports as differentiable outputs: depths and displacement vectors of every object
            for ty in state["types"]:
                samples = results[ty]["_samples"][0]       # differentiable calls are never split along the rays
                for k in range(state["K"]):
                    flat.append(samples["t"][k])
                    flat.append(samples["delta"][k])
        ctx.mark_non_differentiable(*skip)
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grad_outputs):
        if ctx.state is None:
            raise RuntimeError("Trying to backward through the renderer call a second time: its saved activations (the forward "
                               "workspace) have already been freed.  Render again, or sum the losses before calling backward().")
        # the library launches on the caller's stream: that stream's device has to be the current one
        with torch.cuda.device(ctx.state["workspace"].device):
            return _RenderFunction._backward(ctx, *grad_outputs)

    @staticmethod
    def _backward(ctx, *grad_outputs):
        st, composer = ctx.state, ctx.composer
        if tuple(p._version for p in composer._parameter_list()) != ctx.versions:
            raise RuntimeError("one of the composer's parameters was modified in place between the renderer's forward and "
                               "backward calls (pr_render_backward reads the parameter storages): call backward() before "
                               "the optimiser step")
        N, R, K, S, D, F = st["N"], st["R"], st["K"], st["S"], st["D"], st["F"]
        lib = _lib.load()
        dev = st["workspace"].device
        f32 = dict(dtype=torch.float32, device=dev)
        keep = []
        ogs = {ty: _lib.OutputGrads() for ty in ("coarse", "fine")}
        i = 0
        for ty in st["types"]:
            og = ogs[ty]
            for k in range(K + 1):
                entry = og.object[k] if k < K else og.global_
                for key in ENTRY_KEYS:
                    g = grad_outputs[i]
                    i += 1
                    if key in GRAD_KEYS and g is not None:
                        shape = {"integrated_features": (N, R, F), "weights": (N, R, -1)}.get(key, (N, R))
                        g = g.to(torch.float32).reshape(shape).contiguous()
                        keep.append(g)
                        setattr(entry, key, g.data_ptr())
                        if key == "integrated_divergence":      # the double backward of the Hutchinson estimate
                            st["call"].flags |= _lib.PR_FLAG_DIVERGENCE_GRAD
        if ctx.layout is not None:
            # gradients of the decoder-layout maps: the same numbers as global.integrated_features, channels-first per ray group
            groups = len(ctx.layout["rays"])
            for ty in st["types"]:
                extra = None
                ray0 = ch0 = 0
                for gi in range(groups):
                    g = grad_outputs[i]
                    i += 1
                    rays_i, channels_i = ctx.layout["rays"][gi], ctx.layout["channels"][gi]
                    if g is not None:
                        if extra is None:
                            extra = torch.zeros((N, R, F), **f32)
                        extra[:, ray0:ray0 + rays_i, ch0:ch0 + channels_i] = g.to(torch.float32).reshape(N, channels_i, rays_i).transpose(1, 2)
                    ray0 += rays_i
                    ch0 += channels_i
                if extra is not None:
                    entry = ogs[ty].global_
                    if entry.integrated_features:
                        base = next(t for t in keep if t.data_ptr() == entry.integrated_features)
                        extra = extra + base
                    keep.append(extra)
                    entry.integrated_features = extra.data_ptr()
        if ctx.exported:
            for ty in st["types"]:
                for k in range(K):
                    for field in ("sample_t", "sample_delta"):
                        g = grad_outputs[i]
                        i += 1
                        if g is not None:
                            g = g.to(torch.float32).contiguous()
                            keep.append(g)
                            getattr(ogs[ty], field)[k] = g.data_ptr()
        # every parameter gradient is a view of ONE zero-initialised buffer (a single fill instead of one per tensor;
        # parallel.allreduce_gradients reduces such a buffer in place, without flattening copies)
        # ONE zero fill for the parameter gradients and the three input-gradient buffers behind them
        total = sum(p.numel() for p in ctx.params)
        total_padded = (total + 63) // 64 * 64
        both = torch.zeros(total_padded + N * K * (12 + S + D), **f32)
        flat = both[:total]
        grads, offset = {}, 0
        for p in ctx.params:
            grads[id(p)] = flat[offset:offset + p.numel()].view(p.shape)
            offset += p.numel()
        ig = _lib.InputGrads()
        small = both[total_padded:]
        d_w2o = small[:N * K * 12].view(N, K, 3, 4)
        d_style = small[N * K * 12:N * K * (12 + S)].view(N, K, S)
        d_def = small[N * K * (12 + S):].view(N, K, D)
        ig.w2o, ig.style, ig.deformation = d_w2o.data_ptr(), d_style.data_ptr(), d_def.data_ptr()
        d_ray_o = d_ray_d = None
        if ctx.ray_grads:
            d_ray_o, d_ray_d = torch.zeros((N, 3), **f32), torch.zeros((N, R, 3), **f32)
            ig.ray_origins, ig.ray_directions = d_ray_o.data_ptr(), d_ray_d.data_ptr()
        for k in range(K):
            ig.model[k] = composer._model_grad_struct(st["models"][k], grads)
            if "fine" in st["types"]:
                ig.model_fine[k] = composer._model_grad_struct(st["models_fine"][k], grads)
        size = C.c_size_t()
        _lib.check(lib.pr_backward_workspace_size(C.byref(st["call"]), st["objs"], C.byref(size)), "pr_backward_workspace_size")
        scratch = torch.empty(size.value, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.pr_render_backward(C.byref(st["call"]), st["objs"], C.byref(ogs["coarse"]),
                                          C.byref(ogs["fine"]) if "fine" in st["types"] else None, C.byref(ig),
                                          st["workspace"].data_ptr(),
                                          st["workspace"].numel(), scratch.data_ptr(), scratch.numel(), stream),
                   "pr_render_backward")
        for hook in composer.gradient_hooks:
            hook(flat)
        # parameter gradients exist now: an optimiser step is about to change the weights, and torch's FUSED optimisers
        # (torch.optim.Adam(fused=True): torch._fused_adam_) update the storages WITHOUT moving the tensors' version counters
        # (measured, tools/perf/dbg_version_counters.py) - the packed MFMA copies and recorded frames cannot key on them alone.
        # The epoch moves here (an update written by hand, p.data.add_(...), is seen by the next forward pass) AND after the step of
        # the optimiser that owns the parameters (a render between backward() and step() packs the pre-step weights)
        if ctx.params:
            owner = composer
            if getattr(composer, "_is_replica", False):          # (an nn.DataParallel replica lives for one call: its original counts)
                ref = composer.__dict__.get("_replica_of")
                owner = ref() if ref is not None else None
            if owner is not None:
                owner.weights_epoch += 1
                _watch_optimizer_steps(owner)
        if ctx.prepared:
            ctx.state = None   # releases the forward workspace
            if ctx.ray_grads:
                d_ray_o, d_ray_d = d_ray_o.reshape(ctx.ray_shapes[0]), d_ray_d.reshape(ctx.ray_shapes[1])
            return (None, None, None, d_ray_o, d_ray_d, d_w2o, d_style, d_def) + tuple(grads[id(p)] for p in ctx.params)
        lead = st["lead"]
        w_shape, s_shape, d_shape = ctx.shapes
        g_w2o = torch.zeros((N, 4, 4, K), **f32)
        g_w2o[:, :3] = d_w2o.permute(0, 2, 3, 1)
        g_w2o = g_w2o.reshape(lead + [4, 4, K]).sum_to_size(w_shape)
        g_style = d_style.permute(0, 2, 1).reshape(lead + [S, K]).sum_to_size(s_shape)
        g_def = d_def.permute(0, 2, 1).reshape(lead + [D, K]).sum_to_size(d_shape)
        ctx.state = None   # releases the forward workspace
        if ctx.ray_grads:
            d_ray_o, d_ray_d = d_ray_o.reshape(ctx.ray_shapes[0]), d_ray_d.reshape(ctx.ray_shapes[1])
        return (None, None, None, d_ray_o, d_ray_d, g_w2o, g_style, g_def) + tuple(grads[id(p)] for p in ctx.params)


class ObjectComposer(Tracked, nn.Module):

    #: upper limit for the per-call scratch (MLP feature rows dominate); larger calls are split along the ray dimension,
    #: which is exact because rays are independent.  The effective budget of a call is the smaller of this and 80 % of
    #: the device memory that is free (or held by this module's own workspace / torch's cache) at call time.
    max_workspace_bytes = 96 << 30

    def __init__(self, config):
        super().__init__()
        self.config = config
        validate_config_limits(config)
        self.register_load_state_dict_post_hook(ObjectComposer._after_load_state_dict)
        self.object_models_coarse = ModuleList(self.create_object_models(fine=False))
        self.object_models_fine = ModuleList(self.create_object_models(fine=True))
        self.apply_activation = self.config["model"]["apply_activation"]
        if self.object_models_coarse[0].model_config["nerf_model"]["output_features"] != 3 and self.apply_activation:
            raise Exception("The application of activations to the nerf output is requested, but the model seem not "
                            "to output colors directly. Please make sure this is the behavior you desire")
        self.object_id_helper = ObjectIDsHelper(self.config)
        #: bumped by set_step / load_state_dict / .to(): captured frame graphs compare it (frame_graph.FrameGraph)
        self.state_epoch = 0
        #: bumped by every backward pass that produced parameter gradients (an optimiser step follows; torch's fused optimisers do
        #: not move the parameters' version counters): the packed weights are re-made and recorded frames dropped when it moved
        self.weights_epoch = 0
        self._packed: Dict[tuple, tuple] = {}
        self._param_lists: Dict[int, list] = {}      # id(module) -> list(module.parameters())
        self._structs: Dict[tuple, tuple] = {}       # (id(model), positions) -> (key, pr_object_model_t)
        self._tracked: Dict[int, bool] = {}          # id(module) -> its tree holds this package's tracked classes only
        self._registration_epoch = _REGISTRATION_EPOCH[0]
        self._budget_ok = 0                          # largest workspace size a device query has granted
        self._annealing: Dict[int, tuple] = {}
        self._host_step: Optional[int] = 0          # the step last given to set_step (None: the buffers were loaded / moved)
        self._linspace: Dict[tuple, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self.use_naive_mlp = False  # debugging switch (PR_FLAG_NAIVE_MLP)
        #: "fp32": exact fp32 matrix-core arithmetic (default).  "f16x3" (split precision): evaluation renders compute every product
        #: as three fp16 MFMAs with fp32 accumulation (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi with x = hi + lo in fp16, ~22 significant
        #: bits).  Differentiable / training calls keep the fp32 forward PIPELINE (train-mode BatchNorm phases, fp32 saved activations,
        #: fp32 head phases) and run, where a split kernel exists (PR_FLAG_SPLIT_BACKWARD): phase 1 of the forward pass and the backward
        #: CHAINS (dX) on fp16 pairs (weights x 2^8, tiles scaled by a power of two: all scalings exact), the WEIGHT GRADIENTS (dW) on
        #: fp16 pairs of 16-row half slabs (gradient rows x alpha, activation rows x C / alpha, powers of two: exact; bf16 triples until
        #: round 5), fp32 accumulation everywhere - gradients pass the fp32
        #: path's tests, the float64 arbitration at shipped sizes included.  A train-mode call WITHOUT gradients (no_grad) runs exact
        #: fp32.  The first differentiable call at a non-fp32 precision says so once (a model switched to "f16x3" for evaluation and
        #: trained afterwards changes its training numerics at round-off level).
        #: "f16" (throughput tier, interactive play): evaluation renders keep the a_hi*w_hi product only - plain fp16 operands, fp32
        #: accumulation, one MFMA per step; ~1e-3 relative error on the rendered features (>= 40 dB PSNR against the oracle), so
        #: NOT a parity configuration.  Shares the packed weights with "f16x3"; training / differentiable calls behave as "f16x3".
        self.precision = "fp32"
        self._noted_split_training = False
        #: sigma-gated feature head (PR_FLAG_GATE_HEAD): evaluation renders skip the feature head of samples whose raw density
        #: is <= 0 - their compositing weight is exactly 0, so the results are bit-identical.  Ignored (by the library) for
        #: perturbed, training and differentiable calls.
        self.gate_feature_head = True
        #: device tensors (K,) per model type: samples that entered the BatchNorm batch statistics of the last training call
        self.last_normalised_samples: Dict[str, torch.Tensor] = {}
        #: where the random draws of perturbed / training calls come from when no explicit ``_noise`` is replayed:
        #: "device" - generated inside the kernels that consume them (PR_FLAG_DEVICE_NOISE: Philox4x32-10 keyed by a per-call
        #: seed, regenerated by the backward pass; nothing of size (N, R, P) is materialised); "torch" - torch.rand / torch.randn
        #: tensors as in round 1.  The per-call seed is drawn from torch's CPU generator (torch.manual_seed makes runs repeatable).
        self.noise_source = "device"
        #: where the seed of a call's generated noise lives: "host" = drawn from torch's CPU generator and handed over by value
        #: (no device work), "device" = one word drawn by torch's DEVICE generator that the kernels read when they run - what a
        #: call recorded into a HIP graph needs to draw fresh noise on every replay (chosen automatically while a stream is
        #: being captured; see frame_graph.GraphedStep)
        self.noise_seed_source = "host"
        self.last_noise_seed = None      # int ("host") or a one-element int64 device tensor ("device")
        #: Train-mode BatchNorm raises when an object call normalises exactly ONE sample (torch.nn.functional.batch_norm does,
        #: the reference does not guard it; an EMPTY batch passes: running statistics untouched, num_batches_tracked + 1).
        #: The sample counts live on the device: "eager" (default, the reference's behaviour) reads them back before
        #: ``forward`` returns - one host synchronisation per training call; "deferred" copies them to pinned memory
        #: asynchronously and raises the same ValueError at the NEXT call of the composer, which lets the host run ahead of
        #: the device.
        self.batchnorm_check = "eager"
        self._pending_bn_check: Optional[tuple] = None
        #: Extension for evaluation renders whose consumer reads the merged ("global") entry only (the evaluators, play.py, the
        #: decoder): ``None`` (default) = the reference's result schema, every field of every ``object_k`` entry;  a tuple of field
        #: names = only those fields are computed and returned for the per-object entries (``()``: none - the entries then hold
        #: ``extra_outputs`` only).  On a shipped 256x256 tennis frame the per-object maps are 4/5 of the compositing kernel's
        #: 344 MB of output.  Ignored (everything is produced) by differentiable / training calls.
        self.object_entry_fields: Optional[Tuple[str, ...]] = None
        #: callables invoked by the autograd node of a differentiable call with the flat fp32 buffer that every parameter gradient
        #: of the call is a view of, right after ``pr_render_backward`` is enqueued (parallel.OverlappedGradientAllReduce starts the
        #: gradient all-reduce from here, so that it overlaps the rest of ``backward()``)
        self.gradient_hooks: List = []

    def _raise_pending_batchnorm_check(self):
        if torch.cuda.is_current_stream_capturing():
            return     # (no event waits while a graph is recorded: the check stays pending for the next eager call)
        pending, self._pending_bn_check = self._pending_bn_check, None
        if pending is None:
            return
        host, event, types, count = pending
        event.synchronize()
        counts = host.tolist()
        for i, ty in enumerate(types):
            cur = counts[i * count:(i + 1) * count]
            if any(c == 1 for c in cur):      # exactly one: torch.nn.functional.batch_norm raises; an empty batch passes
                raise ValueError(f"Expected more than 1 value per channel when training, got {cur} evaluated "
                                 f"samples per object ({ty} pass of the previous composer call)")

    # ------------------------------------------------------------------ construction
    def create_object_models(self, fine: bool) -> List[Optional[nn.Module]]:
        models = []
        for cfg in self.config["model"]["object_models"]:
            if fine and "use_fine" in cfg and cfg["use_fine"] == False:  # noqa: E712  (reference semantics)
                models.append(None)
            else:
                try:
                    cls = OBJECT_MODEL_CLASSES[cfg["architecture"]]
                except KeyError:
                    raise Exception(f"object model architecture {cfg['architecture']} is not supported")
                models.append(cls(self.config, cfg))
        return models

    def set_step(self, current_step: int):
        for m in self.object_models_coarse:
            m.set_step(current_step)
        for m in self.object_models_fine:
            if m is not None:
                m.set_step(current_step)
        # (the step buffers are filled in place: their version counters move, which the host copies of the octave weights key on;
        # dropping them here as well costs one small read-back at the next render and keeps the rule simple)
        self._annealing.clear()
        self._host_step = int(current_step)
        self.state_epoch += 1

    def annealing_fingerprint(self):
        """The ray benders' octave weights as a function of the step last handed to ``set_step``, computed on the host (no device
        read): what a recorded call baked into its kernel arguments (``pr_object_model_t.bender_octave_weights``).  ``None`` when
        the step buffers were last written by something else (``load_state_dict``, ``.to()``)."""
        if self._host_step is None:
            return None
        out = []
        for m in list(self.object_models_coarse) + [m for m in self.object_models_fine if m is not None]:
            if m.ray_bender.has_weights:
                enc = m.ray_bender.positional_encoder
                alpha = self._host_step * enc.octaves_count / enc.num_steps
                out.append(tuple((1 - math.cos(math.pi * min(1.0, max(0.0, alpha - k)))) / 2 for k in range(enc.octaves_count)))
        return tuple(out)

    def resolve_host_step(self) -> Optional[int]:
        """The annealing step, read back from the device ONCE when the host does not know it (after ``load_state_dict`` / ``.to()``:
        the buffers hold the checkpoint's step).  One synchronisation; ``GraphedStep`` calls it when it records, so that a resumed
        run's first ``set_step(step)`` - the step the checkpoint already holds - does not look like a change of the baked weights."""
        if self._host_step is None:
            for m in list(self.object_models_coarse) + [m for m in self.object_models_fine if m is not None]:
                if m.ray_bender.has_weights:
                    self._host_step = int(m.ray_bender.positional_encoder.current_step.item())
                    break
        return self._host_step

    def after_graph_replay(self):
        """A replayed HIP graph (frame_graph.GraphedStep) updates parameter storages on the device without moving the Python
        version counters the packed-weight cache keys on: forget the packed copies, so that the next eager render packs the
        weights it finds (the graph's own packed buffers live in its memory pool and are re-filled by every replay)."""
        self._packed.clear()
        self.state_epoch += 1

    def weights_changed(self):
        """Call after writing parameter VALUES behind autograd's back - ``p.data.mul_(...)`` / ``p.data = ...`` style updates (a hand-written
        EMA or optimiser; ``.data`` has its own version counter, the parameter's does not move) when no backward pass of this composer
        preceded them: the packed MFMA copies and recorded evaluation frames are re-made from the current values.  In-place ops under
        ``torch.no_grad()``, ``load_state_dict``, ``.to()`` and every ``torch.optim`` optimiser (fused ones included) need no call."""
        self.weights_epoch += 1

    def _parameter_storages(self) -> set:
        """Base addresses of the storages the composer's parameters live in (views of a ``parallel.flatten_parameters`` arena share
        the arena's): what ``_after_optimizer_step`` intersects with the stepping optimiser's tensors."""
        return {p.untyped_storage().data_ptr() for p in self._parameter_list()}

    def _parameter_list(self, module=None) -> list:
        """``list(module.parameters())`` (module = None: the composer), cached: walking the module tree costs ~0.3 ms per call and
        a training step asks five times.  Dropped with the other storage-derived caches (``_apply``, ``load_state_dict``) and when a
        module of this package re-registered anything; a tree holding a module of a foreign class (whose registrations this package
        cannot see) is walked on every call instead, and so is an ``nn.DataParallel`` replica - whose parameters are the broadcast
        copies ``replicate`` files under ``_former_parameters`` (they are no ``nn.Parameter`` s: the gradients flow back to the
        originals through the broadcast's autograd node)."""
        root = self if module is None else module
        if getattr(root, "_is_replica", False):
            seen, out = set(), []
            for m in root.modules():
                for p in getattr(m, "_former_parameters", {}).values():
                    if id(p) not in seen:
                        seen.add(id(p))
                        out.append(p)
            return out
        self._sync_registrations()
        key = id(root)
        cached = self._param_lists.get(key)
        if cached is None:
            cached = list(root.parameters())
            if self._tree_is_tracked(root):
                self._param_lists[key] = cached
        return cached

    def _sync_registrations(self):
        if self._registration_epoch != _REGISTRATION_EPOCH[0]:
            # a module of this package (re-)registered a parameter / buffer / submodule since the lists were built: they may hold
            # replaced objects
            self._registration_epoch = _REGISTRATION_EPOCH[0]
            self._param_lists.clear()
            self._structs.clear()
            self._tracked.clear()

    def _tree_is_tracked(self, root) -> bool:
        """Cached per registration epoch: every module under ``root`` is one of this package's tracked classes."""
        key = id(root)
        self._sync_registrations()
        known = self._tracked.get(key)
        if known is None:
            known = self._tracked[key] = tree_is_tracked(root)
        return known

    def _replicate_for_data_parallel(self):
        """``nn.DataParallel`` (the reference wraps its model unconditionally, train.py:61): every replica gets its OWN caches -
        packed weights, pointer structs, workspace, pending checks - on its own device; the shallow ``__dict__`` copy of
        ``nn.Module._replicate_for_data_parallel`` would otherwise share dictionaries that hold device-0 pointers between the
        replicas' threads.  A replica lives for one call: it packs the broadcast weights it is given and never caches lists."""
        replica = super()._replicate_for_data_parallel()
        fresh = dict(gradient_hooks=[], _packed={}, _param_lists={}, _structs={}, _tracked={}, _annealing={}, _linspace={}, _workspace=None,
                     _budget_ok=0, _pending_bn_check=None, last_normalised_samples={}, last_noise_seed=None)
        replica.__dict__.update(fresh)
        # (the replica's backward pass reports parameter gradients to the ORIGINAL: its packed copies / recorded frames are what an
        # evaluation render outside the wrapper - model.module.render_full_frame_* - reads after the optimiser step)
        replica.__dict__["_replica_of"] = weakref.ref(self.__dict__["_replica_of"]() if "_replica_of" in self.__dict__ else self)
        return replica

    def __getstate__(self):
        # copy.deepcopy / pickle (EMA helpers, swa_utils.AveragedModel): the caches hold ctypes structures with raw pointers
        # (not picklable) and device scratch that a copy must not share
        state = dict(self.__dict__)
        state.update(gradient_hooks=[], _packed={}, _param_lists={}, _structs={}, _tracked={}, _annealing={}, _linspace={}, _workspace=None, _budget_ok=0,
                     _pending_bn_check=None, last_normalised_samples={}, last_noise_seed=None, _host_step=None)
        state.pop("_replica_of", None)
        return state

    def _drop_device_caches(self):
        """Everything derived from parameter / buffer storages or tied to a device."""
        self._param_lists.clear()
        self._structs.clear()
        self._tracked.clear()
        self._budget_ok = 0
        self._packed.clear()
        self._annealing.clear()
        self._linspace.clear()
        self._workspace = None
        self.state_epoch += 1

    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .float(): new storages, possibly at recycled addresses
        out = super()._apply(fn, *args, **kwargs)
        self._drop_device_caches()
        return out

    def _after_load_state_dict(self, *_):
        # a checkpoint was loaded - through this module or through a parent (EnvironmentModel.load_state_dict, the usual way): the
        # step buffers now hold the checkpoint's value, unknown to the host until set_step; storage-derived caches are stale
        self._drop_device_caches()
        self._host_step = None

    # ------------------------------------------------------------------ marshalling
    def _model_struct(self, model: RayBendingStyleNerfModel, positions: int) -> _lib.ObjectModel:
        """The model's pr_object_model_t (raw parameter / buffer pointers, shapes, octave weights), cached per (model, positions):
        rebuilt when the storages may have moved (state_epoch), the annealing step changed, a parameter's storage is not where it
        was (``p.data = ...``), or any module re-registered a parameter / buffer since (``_parameter_list`` drops the structs then:
        ``module.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)``, replaced BatchNorm buffers)."""
        params = self._parameter_list(model)
        step = model.ray_bender.positional_encoder.current_step if model.ray_bender.has_weights else None
        key = (self.state_epoch, tuple(p.data_ptr() for p in params),
               None if step is None else (step.data_ptr(), step._version))
        if getattr(model, "_is_replica", False) or not self._tree_is_tracked(model):
            return self._build_model_struct(model, positions)     # (buffers of a foreign / per-call tree may move unnoticed)
        cached = self._structs.get((id(model), positions))
        if cached is not None and cached[0] == key:
            return cached[1]
        s = self._build_model_struct(model, positions)
        self._structs[(id(model), positions)] = (key, s)
        return s

    def _build_model_struct(self, model: RayBendingStyleNerfModel, positions: int) -> _lib.ObjectModel:
        cfg = model.model_config
        nerf, bender = model.nerf_model, model.ray_bender
        s = _lib.ObjectModel()
        s.kind = nerf.kind
        s.has_bender = 1 if bender.has_weights else 0
        s.positions = positions
        s.style_features = cfg["style_features"]
        s.deformation_features = cfg["deformation_features"]
        s.output_features = nerf.output_features
        s.layers_width = nerf.layers_width
        s.backbone_count = nerf.backbone_layers_count
        s.skip_layer_idx = nerf.skip_layer_idx
        s.octaves = nerf.octaves
        box = [float(v) for row in cfg["bounding_box"] for v in row]
        for i in range(6):
            s.bbox[i] = box[i]
        s.empty_space_alpha = float(cfg["empty_space_alpha"])
        s.z_near_min = float(cfg["z_near_min"])
        s.z_far_max = float(cfg["z_far_max"])
        head = nerf.features_head
        s.bn_eps = float(head[1].ada_in.normalization.eps)
        for i, layer in enumerate(nerf.backbone_layers):
            s.backbone[i] = _linear(layer)
        s.alpha_head = _linear(nerf.alpha_head if nerf.kind == 0 else None)
        s.head0 = _linear(head[0])
        s.affine1 = _linear(head[1].affine_transform)
        s.bn1_mean = head[1].ada_in.normalization.running_mean.data_ptr()
        s.bn1_var = head[1].ada_in.normalization.running_var.data_ptr()
        s.bn1_batches = head[1].ada_in.normalization.num_batches_tracked.data_ptr()
        s.head3 = _linear(head[3])
        s.affine4 = _linear(head[4].affine_transform)
        s.bn4_mean = head[4].ada_in.normalization.running_mean.data_ptr()
        s.bn4_var = head[4].ada_in.normalization.running_var.data_ptr()
        s.bn4_batches = head[4].ada_in.normalization.num_batches_tracked.data_ptr()
        s.head6 = _linear(head[6])
        if bender.has_weights:
            s.bender_width = bender.layers_width
            s.bender_count = bender.layers_count
            s.bender_skip = bender.skip_layer_idx
            s.bender_octaves = bender.positional_encoder.octaves_count
            for i, v in enumerate(self._annealing_weights(bender.positional_encoder)):
                s.bender_octave_weights[i] = v
            for i, layer in enumerate(bender.backbone_layers):
                s.bender[i] = _linear(layer)
            s.bender_out = _linear(bender.output_head)
        return s

    def _annealing_weights(self, encoder) -> list:
        """Host copy of the bender's octave weights, refreshed only when ``current_step`` changed (reading the
        buffer back is a device sync, which must not happen on every render call)."""
        step = encoder.current_step
        key = (id(encoder), step.data_ptr(), step._version)
        cached = self._annealing.get(id(encoder))
        if cached is None or cached[0] != key:
            cached = (key, encoder.annealing_weights().detach().cpu().tolist())
            self._annealing[id(encoder)] = cached
        return cached[1]

    def _precision_code(self, differentiable: bool = False) -> int:
        if self.precision not in ("fp32", "f16x3", "f16"):
            raise ValueError(f"unknown precision {self.precision!r} (expected 'fp32', 'f16x3' or 'f16')")
        if self.precision != "fp32" and (self.training or differentiable):
            # the forward pass of training / differentiable calls runs the exact fp32 kernels (train-mode BatchNorm phases, saved
            # activations) on fp32-packed weights; "f16x3" selects the split-precision products of those kernels (_render)
            return _lib.PR_PRECISION_FP32
        return {"fp32": _lib.PR_PRECISION_FP32, "f16x3": _lib.PR_PRECISION_F16X3, "f16": _lib.PR_PRECISION_F16}[self.precision]

    def _packed_weights(self, model: RayBendingStyleNerfModel, struct: _lib.ObjectModel, stream: int,
                        differentiable: bool = False) -> torch.Tensor:
        """MFMA-fragment-ordered copy of the model's weights, rebuilt whenever a parameter changed."""
        return self._packed_weights_many([(model, struct)], stream, differentiable)[0]

    def _packed_weights_many(self, pairs, stream: int, differentiable: bool = False) -> List[torch.Tensor]:
        """The packed copies of several (model, struct) pairs; those that are stale - a parameter's storage or version moved, or a
        backward pass has produced gradients since (``weights_epoch``) - are re-made by ONE ``pr_pack_models`` call (a training step
        re-packs every model after every optimiser step)."""
        precision = self._precision_code(differentiable)
        if precision == _lib.PR_PRECISION_F16:
            precision = _lib.PR_PRECISION_F16X3      # the same fp16 (hi, lo) fragments; the kernel skips the lo halves
        out, todo = [], {}
        lib = None
        for model, struct in pairs:
            params = self._parameter_list(model)
            key = (self.weights_epoch,) + tuple((p.data_ptr(), p._version) for p in params)
            slot = (id(model), precision)   # one buffer per layout: a render at the other precision never evicts this one
            cached = self._packed.get(slot)
            if cached is not None and cached[0] == key:
                out.append(cached[1])
                continue
            if slot in todo:                # (object instances that share a model)
                out.append(todo[slot][1])
                continue
            lib = lib or _lib.load()
            size = C.c_size_t()
            _lib.check(lib.pr_packed_size(C.byref(struct), C.byref(size)), "pr_packed_size")
            buf = torch.empty(size.value, dtype=torch.uint8, device=params[0].device)
            todo[slot] = (key, buf, struct, size.value)
            out.append(buf)
        if todo:
            n = len(todo)
            entries = list(todo.values())
            models = (C.POINTER(_lib.ObjectModel) * n)(*[C.pointer(e[2]) for e in entries])
            precisions = (C.c_int32 * n)(*([precision] * n))
            buffers = (C.c_void_p * n)(*[e[1].data_ptr() for e in entries])
            sizes = (C.c_size_t * n)(*[e[3] for e in entries])
            _lib.check(lib.pr_pack_models(n, models, precisions, buffers, sizes, stream), "pr_pack_models")
            for slot, (key, buf, _, _) in todo.items():
                self._packed[slot] = (key, buf)
        return out

    def _workspace_budget(self, dev, need: int) -> int:
        """Scratch bytes a call may use: ``max_workspace_bytes``, and - when the call needs more than the workspace this
        module already holds - at most 80 % of what the device can still provide (free memory + torch's cached blocks +
        the workspace that would be released first)."""
        cap = int(self.max_workspace_bytes)
        held = self._workspace.numel() if (self._workspace is not None and self._workspace.device == dev) else 0
        if need <= min(cap, max(held, self._budget_ok)):
            return cap          # fits what is already allocated / what a device query has granted before: no query on the hot path
        if torch.cuda.is_current_stream_capturing():
            return cap          # (no device queries while a graph is recorded: the warm-up calls have sized the call already)
        free, _ = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        budget = max(1 << 20, min(cap, int(0.8 * (free + cached + held))))   # the result tensors need room too
        if need <= budget:
            self._budget_ok = need       # (a training loop asks for the same size every step; reset with the other caches)
        return budget

    def _hook_tensor(self, device) -> torch.Tensor:
        """The reference's ``pytorch_hook`` entry (object_composer.py:890: a dummy tensor that gives ``nn.DataParallel`` something to
        gather): one zero tensor per device, made once - a fill launch per call is 1 % of a small frame.  Shared between the result
        dictionaries of this composer: read-only by contract."""
        key = ("hook", str(device))
        if key not in self._linspace:
            self._linspace[key] = torch.zeros((1, 1, 1, 1, 1, 1, 1, 1, 1), device=device)
        return self._linspace[key]

    def _linspace_for(self, count: int, device) -> torch.Tensor:
        key = (count, str(device))
        if key not in self._linspace:
            # evaluated by torch so that it is bit-identical to the reference's torch.linspace
            self._linspace[key] = torch.linspace(0.0, 1.0, count, device=device)
        return self._linspace[key]

    # ------------------------------------------------------------------ forward
    def forward(self, ray_origins: torch.Tensor, ray_directions: torch.Tensor, focal_normals: torch.Tensor,
                transformation_matrix_w2o: torch.Tensor, style: torch.Tensor, deformation: torch.Tensor,
                object_in_scene: torch.Tensor, perturb: bool, video_indexes: torch.Tensor = None,
                canonical_pose: bool = False, _noise: Optional[dict] = None, _export: bool = False,
                _decoder_layout: Optional[dict] = None, _prepared: Optional[dict] = None) -> Dict:
        """See model/object_composer.py:786-811 for the argument and result documentation.

        ray_origins (..., 3); ray_directions (..., R, 3); focal_normals (..., 3) [unused by the
        renderer, as in the reference]; transformation_matrix_w2o (..., 4, 4, K); style (..., S, K);
        deformation (..., D, K); object_in_scene (..., K).  ``_noise`` (extension, optional) replays
        explicit noise tensors keyed as in oracle/render_oracle.py; ``_export`` adds per-sample state;
        ``_decoder_layout`` = {"rays": [R_0, R_1, ...], "width": [w_0, ...], "channels": [c_0, ...]} additionally emits
        ``global.integrated_features`` as the channels-first maps the reference's CNN decoder consumes (the rays are the
        concatenation of row-major grids of R_i = h_i * w_i rays; grid i owns the next c_i feature channels):
        ``results[type]["global"]["decoder_features"]`` = [(..., c_i, h_i, w_i)], written by the compositing kernel.

        Autograd: with gradients enabled the call is differentiable (training mode: through the batch statistics of the
        BatchNorm layers; eval mode: with the running statistics as constants, e.g. test-time optimisation) with
        respect to the parameters, ``style``, ``deformation`` and ``transformation_matrix_w2o`` through
        ``integrated_features``, ``opacity``, ``depth``, ``integrated_displacements_magnitude`` and
        ``integrated_divergence`` of every entry (pr_render_backward).  ``integrated_divergence`` carries the Hutchinson
        estimate of the reference (object_composer.py:582-601) in training mode; its gradient (the reference's double
        backward) is one more pass over the ray bender with the probe tangents in place of the activations.  ``ray_origins`` /
        ``ray_directions`` are differentiated when they carry a graph (learnable camera parameters; dataset inputs otherwise).
        Not differentiated: ``focal_normals`` (unused by the reference's composer too) and ``disparity`` (no consumer)."""
        if ray_directions.is_cuda:
            # the library launches on the caller's stream: that stream's device has to be the current one
            with torch.cuda.device(ray_directions.device):
                return self._forward(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style, deformation,
                                     object_in_scene, perturb, canonical_pose, _noise, _export, _decoder_layout, _prepared)
        raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")

    def _forward(self, ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style, deformation,
                 object_in_scene, perturb, canonical_pose, _noise, _export, _decoder_layout=None, _prepared=None) -> Dict:
        K = self.object_id_helper.objects_count
        self._raise_pending_batchnorm_check()
        if _prepared is not None:
            # EnvironmentModel's fused scene set-up (pr_scene_setup) has produced the renderer's inputs in the renderer's layouts
            if not ray_directions.is_cuda:
                raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
            params = [p for p in self._parameter_list() if p.requires_grad] if torch.is_grad_enabled() else []
            wants_grad = torch.is_grad_enabled() and (bool(params) or any(_prepared[k].requires_grad for k in ("w2o", "style", "deformation"))
                                                      or ray_origins.requires_grad or ray_directions.requires_grad)
            if not wants_grad:
                return self._render(ray_origins, ray_directions, focal_normals, None, style, deformation, object_in_scene, perturb,
                                    canonical_pose, _noise, _export, False, None, _decoder_layout, _prepared=_prepared)[0]
            # a training call: the same autograd node; its w2o / style / deformation inputs ARE the prepared tensors, and their
            # gradients come back in those layouts (EnvironmentModel's set-up node takes them to the scene tensors in one launch)
            kwargs = dict(ray_origins=ray_origins, ray_directions=ray_directions, focal_normals=focal_normals,
                          transformation_matrix_w2o=None, style=style, deformation=deformation, object_in_scene=object_in_scene,
                          perturb=perturb, canonical_pose=canonical_pose, _noise=_noise, _export=_export, _decoder_layout=_decoder_layout,
                          _prepared=_prepared)
            return self._render_with_graph(kwargs, K, params)
        if transformation_matrix_w2o.size(-1) != K:
            raise Exception(f"Transformation matrix must specifies transformations for"
                            f"({transformation_matrix_w2o.size(-1)}) objects instead of ({K})")
        if not ray_directions.is_cuda:
            raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
        args = (ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style, deformation, object_in_scene,
                perturb, canonical_pose, _noise, _export, False, None, _decoder_layout)
        params = [p for p in self._parameter_list() if p.requires_grad] if torch.is_grad_enabled() else []
        wants_grad = torch.is_grad_enabled() and (bool(params) or style.requires_grad or deformation.requires_grad or
                                                  transformation_matrix_w2o.requires_grad or ray_origins.requires_grad or
                                                  ray_directions.requires_grad)
        if not wants_grad:
            return self._render(*args)[0]
        kwargs = dict(ray_origins=ray_origins, ray_directions=ray_directions, focal_normals=focal_normals,
                      transformation_matrix_w2o=transformation_matrix_w2o, style=style, deformation=deformation,
                      object_in_scene=object_in_scene, perturb=perturb, canonical_pose=canonical_pose, _noise=_noise,
                      _export=_export, _decoder_layout=_decoder_layout)
        return self._render_with_graph(kwargs, K, params)

    def _render_with_graph(self, kwargs: dict, K: int, params) -> Dict:
        """One differentiable renderer call: the tensors of the result dictionary are the outputs of the autograd node."""
        holder = {}
        prepared = kwargs.get("_prepared")
        if prepared is not None:
            flat = _RenderFunction.apply(self, kwargs, holder, kwargs["ray_origins"], kwargs["ray_directions"], prepared["w2o"],
                                         prepared["style"], prepared["deformation"], *params)
        else:
            flat = _RenderFunction.apply(self, kwargs, holder, kwargs["ray_origins"], kwargs["ray_directions"],
                                         kwargs["transformation_matrix_w2o"], kwargs["style"], kwargs["deformation"], *params)
        results = holder["results"]
        i = 0
        for ty in holder["types"]:
            for name in [f"object_{k}" for k in range(K)] + ["global"]:
                for key in ENTRY_KEYS:
                    results[ty][name][key] = flat[i]
                    i += 1
        if kwargs.get("_decoder_layout") is not None:
            for ty in holder["types"]:
                maps = results[ty]["global"]["decoder_features"]
                for j in range(len(maps)):
                    maps[j] = flat[i]
                    i += 1
        if kwargs.get("_export"):
            for ty in holder["types"]:
                samples = results[ty]["_samples"][0]
                for k in range(K):
                    samples["t"][k], samples["delta"][k] = flat[i], flat[i + 1]
                    i += 2
        return results

    def _render(self, ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style, deformation,
                object_in_scene, perturb, canonical_pose=False, _noise=None, _export=False, _save=False, _object_ids=None,
                _decoder_layout=None, _retry=False, _prepared=None):
        """The renderer call proper.  Returns (results, state); ``state`` (only with ``_save``) keeps what
        pr_render_backward needs: the call structures, their tensors and the forward workspace.
        ``_object_ids``: render only these object instances (the tensors then carry ``len(_object_ids)`` objects);
        used by forward_expected_positions."""
        helper = self.object_id_helper
        ids = list(range(helper.objects_count)) if _object_ids is None else list(_object_ids)
        K = len(ids)

        lead = list(ray_directions.shape[:-2])
        R = ray_directions.size(-2)
        N = int(math.prod(lead)) if lead else 1
        dev = ray_directions.device
        f32 = dict(dtype=torch.float32, device=dev)

        dirs = ray_directions.detach().to(torch.float32).reshape(N, R, 3).contiguous()
        origins = torch.broadcast_to(ray_origins.detach().to(torch.float32), lead + [3]).reshape(N, 3).contiguous()
        if _prepared is not None:
            if _prepared["frames"] != N:
                raise RuntimeError("prepared renderer inputs do not match the call")
            w2o, sty, dfm, present = (_prepared["w2o"].detach(), _prepared["style"].detach(), _prepared["deformation"].detach(),
                                      _prepared["present"])
            S, D = _prepared["S"], _prepared["D"]
        else:
            w2o = torch.broadcast_to(transformation_matrix_w2o.detach().to(torch.float32), lead + [4, 4, K])
            w2o = w2o.reshape(N, 4, 4, K).permute(0, 3, 1, 2)[:, :, :3, :].contiguous()          # (N, K, 3, 4)
            S = style.size(-2)
            D = deformation.size(-2)
            sty = torch.broadcast_to(style.detach().to(torch.float32), lead + [S, K]).reshape(N, S, K).permute(0, 2, 1).contiguous()
            dfm = torch.broadcast_to(deformation.detach().to(torch.float32), lead + [D, K]).reshape(N, D, K).permute(0, 2, 1).contiguous()
            if _object_ids is None and object_in_scene.size(-1) > K:
                # the reference indexes object_in_scene[..., object_idx] (object_composer.py:823-830): entries beyond the K
                # objects are never read (forward_from_observations hands over such a tensor, environment_model.py:990-992)
                object_in_scene = object_in_scene[..., :K]
            present = torch.broadcast_to(object_in_scene, lead + [K]).reshape(N, K).to(torch.uint8).contiguous()

        models_c = [self.object_models_coarse[helper.model_idx_by_object_idx(k)] for k in ids]
        models_f = [self.object_models_fine[helper.model_idx_by_object_idx(k)] for k in ids]
        # the reference iterates the result types of object 0 (object_composer.py:851)
        use_fine = models_f[0] is not None
        if use_fine and any(m is None for m in models_f):
            raise KeyError("fine")  # what the reference does when only some objects have a fine model
        pc = [m.model_config["positions_count_coarse"] for m in models_c]
        pf = [m.model_config["positions_count_fine"] for m in models_c]

        if N * R == 0:
            # no rays (an empty batch or an empty pixel list): the result dictionary with empty tensors, no launch
            if _save:
                raise ValueError("a differentiable renderer call needs at least one ray")
            F = models_c[0].nerf_model.output_features
            results: Dict = {}
            for ty in ["coarse"] + (["fine"] if use_fine else []):
                counts = pc if ty == "coarse" else [a + b for a, b in zip(pc, pf)]
                results[ty] = {}
                for k in range(K + 1):
                    P = counts[k] if k < K else sum(counts)
                    entry = {"integrated_features": torch.empty(lead + [R, F], **f32), "weights": torch.empty(lead + [R, P], **f32)}
                    for key in ("opacity", "depth", "disparity", "integrated_displacements_magnitude", "integrated_divergence"):
                        entry[key] = torch.empty(lead + [R], **f32)
                    if k < K:
                        entry["extra_outputs"] = {}
                    results[ty][f"object_{k}" if k < K else "global"] = entry
            results["pytorch_hook"] = self._hook_tensor(dev)
            return results, None

        stream = torch.cuda.current_stream(dev).cuda_stream
        lib = _lib.load()
        keep = []  # tensors that must outlive the enqueue
        objs = (_lib.Object * K)()
        packed_keep = []   # the packed buffers this call reads (kept alive by the autograd state of differentiable calls)
        pairs = []
        for k in range(K):
            for attr in ("style_features", "deformation_features"):
                want = S if attr == "style_features" else D
                if models_c[k].model_config[attr] != want:
                    raise Exception(f"object {k}: {attr} is {models_c[k].model_config[attr]} but the tensor has {want}")
            objs[k].coarse = self._model_struct(models_c[k], pc[k])
            pairs.append((models_c[k], objs[k].coarse))
            if use_fine:
                objs[k].fine = self._model_struct(models_f[k], pc[k] + pf[k])
                pairs.append((models_f[k], objs[k].fine))
        packed_keep = self._packed_weights_many(pairs, stream, _save)      # (stale copies re-made in one launch)
        for k in range(K):
            objs[k].packed_coarse = packed_keep[k * (2 if use_fine else 1)].data_ptr()
            if use_fine:
                objs[k].packed_fine = packed_keep[2 * k + 1].data_ptr()

        flags = 0
        if perturb:
            flags |= _lib.PR_FLAG_PERTURB
        if canonical_pose:
            flags |= _lib.PR_FLAG_CANONICAL_POSE
        if self.config["model"]["fix_object_overlaps"] and _object_ids is None:
            flags |= _lib.PR_FLAG_FIX_OVERLAPS
        if self.apply_activation:
            flags |= _lib.PR_FLAG_SIGMOID_FEATURES
        if self.use_naive_mlp:
            flags |= _lib.PR_FLAG_NAIVE_MLP
        if self.training:
            # BatchNorm1d of the AdaIN layers in training mode: batch statistics per object call,
            # running statistics updated in place (per replica, like the reference under DataParallel)
            flags |= _lib.PR_FLAG_TRAIN_BN
        if _save:
            flags |= _lib.PR_FLAG_SAVE_FOR_BACKWARD
            if self.precision in ("f16x3", "f16"):
                # phase 1 of the forward pass, the backward chains and the weight gradients on fp16 pairs of power-of-two scaled operands
                flags |= _lib.PR_FLAG_SPLIT_BACKWARD
                if not self._noted_split_training:
                    self._noted_split_training = True
                    warnings.warn(f"ObjectComposer.precision = {self.precision!r} applies to this differentiable call too: phase 1 of the "
                                  "forward pass, the backward chains and the weight gradients run on fp16 pairs of power-of-two scaled operands "
                                  "(fp32 accumulation; gradients within the fp32 path's tolerances).  Set precision = 'fp32' for exact-fp32 "
                                  "training.", UserWarning, stacklevel=2)
        if self.gate_feature_head:
            flags |= _lib.PR_FLAG_GATE_HEAD      # honoured by the library for unperturbed evaluation calls only

        # ---- noise -----------------------------------------------------------------------------
        types = ["coarse"] + (["fine"] if use_fine else [])
        object_fields = None
        if self.object_entry_fields is not None and not _save and not self.training:
            object_fields = tuple(self.object_entry_fields)
            unknown = [f for f in object_fields if f not in ENTRY_KEYS]
            if unknown:
                raise ValueError(f"object_entry_fields: unknown field(s) {unknown}; the fields are {ENTRY_KEYS}")
        ptot = {"coarse": pc, "fine": [a + b for a, b in zip(pc, pf)]}
        noise: Dict[str, torch.Tensor] = {}
        if self.noise_source not in ("device", "torch"):
            raise ValueError(f"unknown noise_source {self.noise_source!r} (expected 'device' or 'torch')")
        needs_noise = perturb or (_save and self.training)
        device_noise = needs_noise and _noise is None and self.noise_source == "device"
        noise_seed = 0
        seed_word = None
        capturing = torch.cuda.is_current_stream_capturing()
        if device_noise:
            flags |= _lib.PR_FLAG_DEVICE_NOISE
            if self.noise_seed_source not in ("host", "device"):
                raise ValueError(f"unknown noise_seed_source {self.noise_seed_source!r} (expected 'host' or 'device')")
            if self.noise_seed_source == "device" or capturing:
                # one word per call (the backward pass of THIS call regenerates the forward's values from it, whatever other
                # calls run in between), drawn by the device generator: part of the graph when the call is captured
                seed_word = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev)
                self.last_noise_seed = seed_word
            else:
                noise_seed = int(torch.empty((), dtype=torch.int64).random_())      # CPU generator: no device synchronisation
                self.last_noise_seed = noise_seed

        def get(name, shape, normal):
            if device_noise:
                return                     # generated in the kernels
            if _noise is not None and _noise.get(name) is not None:
                t = _noise[name].to(**f32).reshape(shape).contiguous()
            else:
                t = (torch.randn if normal else torch.rand)(shape, **f32)
            noise[name] = t
        if _save and self.training:
            # Hutchinson probes of compute_approximate_divergence (object_composer.py:597): drawn whenever the
            # reference trains with a graph (zeros in eval mode), independent of `perturb`; only objects with a ray
            # bender have a non-zero displacement field
            for k in range(K):
                if models_c[k].ray_bender.has_weights:
                    get(f"div_coarse_{k}", (N, R, pc[k], 3), True)
                if use_fine and models_f[k].ray_bender.has_weights:
                    get(f"div_fine_{k}", (N, R, pc[k] + pf[k], 3), True)
        if perturb:
            for k in range(K):
                get(f"jitter_{k}", (N, R, pc[k]), False)
                get(f"alpha_{k}", (N, R, pc[k]), True)
                if use_fine:
                    get(f"pdf_{k}", (N, R, pf[k]), False)
            for ty in types:
                for k in range(K):
                    get(f"int_{ty}_{k}", (N, R, ptot[ty][k]), True)
                get(f"int_{ty}_global", (N, R, sum(ptot[ty])), True)

        # ---- ray chunking against the workspace budget -----------------------------------------
        def build_call(r0: int, r1: int):
            call = _lib.Call()
            call.frames, call.rays, call.objects = N, r1 - r0, K
            call.static_objects = helper.static_objects_count if _object_ids is None else 0
            call.use_fine = 1 if use_fine else 0
            call.flags = flags
            call.noise_seed = noise_seed
            if seed_word is not None:
                call.noise_seed_device = seed_word.data_ptr()
                keep.append(seed_word)
            call.noise_ray_offset, call.noise_total_rays = r0, R
            call.precision = self._precision_code(_save)
            d = dirs if (r0 == 0 and r1 == R) else dirs[:, r0:r1].contiguous()
            keep.append(d)
            call.ray_origins, call.ray_directions = origins.data_ptr(), d.data_ptr()
            call.w2o, call.style, call.deformation = w2o.data_ptr(), sty.data_ptr(), dfm.data_ptr()
            call.object_in_scene = present.data_ptr()
            for k in range(K):
                call.linspace_coarse[k] = self._linspace_for(pc[k], dev).data_ptr()
                if use_fine:
                    call.linspace_fine[k] = self._linspace_for(pf[k], dev).data_ptr()
                    call.positions_fine[k] = pf[k]

            def sl(name):
                t = noise.get(name)
                if t is None:
                    return None
                t = t if (r0 == 0 and r1 == R) else t[:, r0:r1].contiguous()
                keep.append(t)
                return t.data_ptr()
            for k in range(K):
                call.noise_coarse.jitter[k] = sl(f"jitter_{k}")
                call.noise_coarse.alpha[k] = sl(f"alpha_{k}")
                call.noise_coarse.pdf[k] = sl(f"pdf_{k}")
                call.noise_coarse.integrate[k] = sl(f"int_coarse_{k}")
                call.noise_coarse.divergence[k] = sl(f"div_coarse_{k}")
                call.noise_fine.divergence[k] = sl(f"div_fine_{k}")
                call.noise_fine.integrate[k] = sl(f"int_fine_{k}")
            call.noise_coarse.integrate_global = sl("int_coarse_global")
            call.noise_fine.integrate_global = sl("int_fine_global")
            return call

        def workspace_bytes(call) -> int:
            size = C.c_size_t()
            _lib.check(lib.pr_workspace_size(C.byref(call), objs, C.byref(size)), "pr_workspace_size")
            return size.value

        state = None
        chunk = R
        need = workspace_bytes(build_call(0, R))
        budget = self._workspace_budget(dev, need)
        if _save and need > budget:
            raise RuntimeError(f"this differentiable renderer call would keep {need / 2**30:.1f} GiB of activations for its "
                               f"backward pass (budget {budget / 2**30:.1f} GiB): render under "
                               "torch.no_grad(), or differentiate fewer rays per call (training uses ray patches)")
        if need > budget and self.training:
            raise RuntimeError(f"this training-mode renderer call needs {need / 2**30:.1f} GiB of scratch (budget "
                               f"{budget / 2**30:.1f} GiB) and cannot be split along the rays: the BatchNorm batch statistics "
                               "are taken over the whole call.  Use fewer rays per call (training uses ray patches)")
        if need > budget and R > 1:
            chunk = max(1, int(R * budget / need))
            chunk = max(256, chunk // 256 * 256) if chunk >= 256 else chunk
            # part of the scratch does not shrink with the rays (the pending stacks of the gated head, per-frame tables):
            # step down until a chunk really fits
            while chunk > 256 and workspace_bytes(build_call(0, chunk)) > budget:
                chunk = max(256, int(chunk * 0.8) // 256 * 256)
        F = models_c[0].nerf_model.output_features
        layout = None
        if _decoder_layout is not None:
            layout = {k: [int(v) for v in _decoder_layout[k]] for k in ("rays", "width", "channels")}
            groups = len(layout["rays"])
            if not (1 <= groups <= _lib.PR_MAX_DECODER_GROUPS) or len(layout["width"]) != groups or len(layout["channels"]) != groups:
                raise ValueError(f"decoder layout: 1..{_lib.PR_MAX_DECODER_GROUPS} groups with rays / width / channels each, got {_decoder_layout}")
            if sum(layout["rays"]) != R or sum(layout["channels"]) > F or any(r % w for r, w in zip(layout["rays"], layout["width"])):
                raise ValueError(f"decoder layout {layout} does not describe {R} rays with {F} feature channels")
            if chunk != R:
                raise RuntimeError("decoder-layout emission needs the whole call in one launch (the call was split along the rays "
                                   "to fit the workspace budget)")

        pieces = []
        for r0 in range(0, R, chunk):
            r1 = min(R, r0 + chunk)
            call = build_call(r0, r1)
            need = workspace_bytes(call)
            if _save:
                # the backward pass re-reads this workspace: it belongs to the autograd node, not to the module
                workspace = torch.empty(need, dtype=torch.uint8, device=dev)
            else:
                if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
                    self._workspace = None
                    try:
                        self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
                    except torch.OutOfMemoryError:
                        # the size was granted by an EARLIER device query (_budget_ok) and memory has become scarce since
                        # (decoder activations, a second model): forget the grant and take the call again through a fresh query
                        if self._budget_ok == 0 or _retry:
                            raise
                        self._budget_ok = 0
                        return self._render(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                            deformation, object_in_scene, perturb, canonical_pose, _noise, _export, _save,
                                            _object_ids, _decoder_layout, _retry=True, _prepared=_prepared)
                workspace = self._workspace
            rc = r1 - r0
            outs = {}
            structs = {}
            for ty in types:
                o = _lib.Outputs()
                res = {}
                for k in range(K + 1):
                    P = ptot[ty][k] if k < K else sum(ptot[ty])
                    shapes = {"integrated_features": (N, rc, F), "weights": (N, rc, P)}
                    wanted = ENTRY_KEYS if (k == K or object_fields is None) else object_fields
                    e = {name: torch.empty(shapes.get(name, (N, rc)), **f32) for name in ENTRY_KEYS if name in wanted}
                    entry = o.object[k] if k < K else o.global_
                    for name in e:       # (fields that are not wanted stay NULL: the compositing kernel skips them)
                        setattr(entry, name, e[name].data_ptr())
                    res[f"object_{k}" if k < K else "global"] = e
                if layout is not None:
                    begin = 0
                    maps = []
                    o.decoder.groups = len(layout["rays"])
                    for i, (rays_i, width_i, channels_i) in enumerate(zip(layout["rays"], layout["width"], layout["channels"])):
                        maps.append(torch.empty((N, channels_i, rays_i // width_i, width_i), **f32))
                        o.decoder.rays[i], o.decoder.width[i] = rays_i, width_i
                        o.decoder.channel_begin[i], o.decoder.channel_end[i] = begin, begin + channels_i
                        o.decoder.map[i] = maps[-1].data_ptr()
                        begin += channels_i
                    res["_decoder"] = maps
                if self.training:
                    res["_normalised"] = torch.zeros((K,), dtype=torch.int32, device=dev)
                    o.normalised_samples = res["_normalised"].data_ptr()
                if _export:
                    ex = {"t": [], "sigma": [], "slot": [], "delta": []}
                    for k in range(K):
                        ex["delta"].append(torch.empty((N, rc, ptot[ty][k], 3), **f32))
                        o.sample_delta[k] = ex["delta"][k].data_ptr()
                        ex["t"].append(torch.empty((N, rc, ptot[ty][k]), **f32))
                        ex["sigma"].append(torch.empty((N, rc, ptot[ty][k]), **f32))
                        ex["slot"].append(torch.empty((N, rc, ptot[ty][k]), dtype=torch.int32, device=dev))
                        o.sample_t[k] = ex["t"][k].data_ptr()
                        o.sample_sigma[k] = ex["sigma"][k].data_ptr()
                        o.sample_slot[k] = ex["slot"][k].data_ptr()
                    ex["evaluated"] = torch.zeros((K,), dtype=torch.int32, device=dev)
                    o.evaluated_samples = ex["evaluated"].data_ptr()
                    ex["head_evaluated"] = torch.zeros((K,), dtype=torch.int32, device=dev)   # samples through the feature head
                    o.head_samples = ex["head_evaluated"].data_ptr()
                    res["_samples"] = ex
                outs[ty] = res
                structs[ty] = o
            _lib.check(lib.pr_render_forward(C.byref(call), objs, C.byref(structs["coarse"]),
                                             C.byref(structs["fine"]) if use_fine else None,
                                             workspace.data_ptr(), workspace.numel(), stream),
                       "pr_render_forward")
            pieces.append(outs)
            if _save:
                state = dict(call=call, objs=objs, keep=keep + packed_keep + [origins, w2o, sty, dfm, present], workspace=workspace,
                             N=N, R=R, K=K, S=S, D=D, F=F, lead=lead, models=models_c, models_fine=models_f, types=types,
                             ptot=ptot, layout=layout)

        if self.training:
            self.last_normalised_samples = {ty: pieces[0][ty]["_normalised"] for ty in types}
        if self.training and capturing:
            pass       # a recorded call cannot raise from its replays: the counts stay in last_normalised_samples for the caller
        elif self.training and self.batchnorm_check == "deferred":
            counts = pieces[0][types[0]]["_normalised"] if len(types) == 1 else torch.cat([pieces[0][ty]["_normalised"] for ty in types])
            host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(dev))
            self._pending_bn_check = (host, event, list(types), K)
        elif self.training:
            if self.batchnorm_check != "eager":
                raise ValueError(f"unknown batchnorm_check {self.batchnorm_check!r} (expected 'eager' or 'deferred')")
            # BatchNorm1d raises for a single value per channel (torch.nn.functional.batch_norm); the reference
            # does not guard against it (model/layers/adain.py:58) - one device read-back per training call
            for ty in types:
                counts = pieces[0][ty]["_normalised"].cpu().tolist()
                if any(c == 1 for c in counts):
                    raise ValueError(f"Expected more than 1 value per channel when training, got {counts} evaluated "
                                     f"samples per object ({ty} pass)")

        # ---- result dictionary (object_composer.py:848-892 schema) -----------------------------
        results: Dict = {}
        for ty in types:
            results[ty] = {}
            for name in [f"object_{k}" for k in range(K)] + ["global"]:
                entry = {}
                for key in ENTRY_KEYS:
                    if key not in pieces[0][ty][name]:
                        continue         # (object_entry_fields: a per-object field nobody asked for)
                    parts = [p[ty][name][key] for p in pieces]
                    t = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
                    entry[key] = t.reshape(lead + list(t.shape[1:]))
                if name != "global":
                    entry["extra_outputs"] = {}
                results[ty][name] = entry
            if _export:
                results[ty]["_samples"] = [p[ty]["_samples"] for p in pieces]
            if layout is not None:
                results[ty]["global"]["decoder_features"] = [m.reshape(lead + list(m.shape[1:])) for m in pieces[0][ty]["_decoder"]]
        results["pytorch_hook"] = self._hook_tensor(dev)
        return results, (state if _save else None)

    def forward_expected_positions(self, ray_origins: torch.Tensor, ray_directions: torch.Tensor, focal_normals: torch.Tensor,
                                   transformation_matrix_w2o: torch.Tensor, style: torch.Tensor, deformation: torch.Tensor,
                                   object_in_scene: torch.Tensor, object_id: int, perturb: bool,
                                   video_indexes: torch.Tensor = None, canonical_pose: bool = False,
                                   _noise: Optional[dict] = None) -> Dict:
        """model/object_composer.py:624-722: the weight-averaged bent surface point of ONE object instance.

        transformation_matrix_w2o (..., 4, 4); style (..., S); deformation (..., D); object_in_scene (...).
        Returns {"coarse": (expected positions (..., R, 3), opacity (..., R)) [, "fine": (...)]} in the object frame.
        The renderer runs with this single object (pr_render_forward); without a graph pr_expected_positions forms the
        average, with gradients enabled (training mode) the call is differentiable with respect to the parameters, the
        style, the deformation and the pose (_expected_positions_with_graph), as the pose / keypoint consistency losses
        of the reference's trainers need.  ``_noise`` keys: jitter, alpha, pdf, alpha_fine
        (oracle/render_oracle.py:expected_positions_forward)."""
        if not ray_directions.is_cuda:
            raise RuntimeError("the HIP renderer needs device tensors (there is no CPU fallback)")
        with torch.cuda.device(ray_directions.device):
            return self._forward_expected_positions(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                                    deformation, object_in_scene, object_id, perturb, canonical_pose, _noise)

    def _forward_expected_positions(self, ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                    deformation, object_in_scene, object_id, perturb, canonical_pose, _noise) -> Dict:
        self._raise_pending_batchnorm_check()
        noise = None
        if _noise is not None:
            # one alpha draw feeds both the coarse weights and the resampler (object_composer.py:683-684, :697)
            noise = {"jitter_0": _noise.get("jitter"), "alpha_0": _noise.get("alpha"), "int_coarse_0": _noise.get("alpha"),
                     "pdf_0": _noise.get("pdf"), "int_fine_0": _noise.get("alpha_fine")}
        elif perturb:
            lead = list(ray_directions.shape[:-2])
            model = self.object_models_coarse[self.object_id_helper.model_idx_by_object_idx(object_id)]
            shape = lead + [ray_directions.size(-2), model.model_config["positions_count_coarse"]]
            shared = torch.randn(shape, dtype=torch.float32, device=ray_directions.device)
            noise = {"alpha_0": shared, "int_coarse_0": shared}
        params = [p for p in self._parameter_list() if p.requires_grad] if torch.is_grad_enabled() else []
        wants_grad = torch.is_grad_enabled() and (bool(params) or style.requires_grad or deformation.requires_grad or
                                                  transformation_matrix_w2o.requires_grad)
        if wants_grad:
            return self._expected_positions_with_graph(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o,
                                                       style, deformation, object_in_scene, object_id, perturb, canonical_pose,
                                                       noise, params)
        with torch.no_grad():
            results, _ = self._render(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o.unsqueeze(-1),
                                      style.unsqueeze(-1), deformation.unsqueeze(-1), object_in_scene.unsqueeze(-1), perturb,
                                      canonical_pose, noise, True, False, _object_ids=[object_id])
            lib = _lib.load()
            dev = ray_directions.device
            stream = torch.cuda.current_stream(dev).cuda_stream
            lead = list(ray_directions.shape[:-2])
            R = ray_directions.size(-2)
            N = int(math.prod(lead)) if lead else 1
            dirs = ray_directions.to(torch.float32).reshape(N, R, 3).contiguous()
            origins = torch.broadcast_to(ray_origins.to(torch.float32), lead + [3]).reshape(N, 3).contiguous()
            w2o = torch.broadcast_to(transformation_matrix_w2o.to(torch.float32), lead + [4, 4]).reshape(N, 1, 4, 4)[:, :, :3, :].contiguous()
            out = {}
            for ty in ("coarse", "fine"):
                if ty not in results:
                    continue
                pieces = results[ty]["_samples"]
                t = pieces[0]["t"][0] if len(pieces) == 1 else torch.cat([p["t"][0] for p in pieces], dim=1)
                delta = pieces[0]["delta"][0] if len(pieces) == 1 else torch.cat([p["delta"][0] for p in pieces], dim=1)
                entry = results[ty]["object_0"]
                weights = entry["weights"].reshape(N, R, -1).contiguous()
                expected = torch.empty((N, R, 3), dtype=torch.float32, device=dev)
                _lib.check(lib.pr_expected_positions(N, R, 1, 0, weights.size(-1), origins.data_ptr(), dirs.data_ptr(),
                                                     w2o.data_ptr(), t.contiguous().data_ptr(), weights.data_ptr(),
                                                     delta.contiguous().data_ptr(), expected.data_ptr(), stream),
                           "pr_expected_positions")
                out[ty] = (expected.reshape(lead + [R, 3]), entry["opacity"])
        return out

    # ------------------------------------------------------------------ backward marshalling
    def _model_grad_struct(self, model: RayBendingStyleNerfModel, grads: Dict[int, torch.Tensor]) -> _lib.ModelGrads:
        """Gradient buffers of one model in the layout of pr_model_grads_t (NULL where no gradient is wanted)."""
        def lg(layer) -> _lib.LinearGrad:
            g = _lib.LinearGrad()
            if layer is not None:
                w = grads.get(id(layer.weight))
                b = grads.get(id(layer.bias)) if layer.bias is not None else None
                g.weight = _ptr(w)
                g.bias = _ptr(b)
            return g
        nerf, bender = model.nerf_model, model.ray_bender
        s = _lib.ModelGrads()
        for i, layer in enumerate(nerf.backbone_layers):
            s.backbone[i] = lg(layer)
        s.alpha_head = lg(nerf.alpha_head if nerf.kind == 0 else None)
        head = nerf.features_head
        s.head0 = lg(head[0])
        s.affine1 = lg(head[1].affine_transform)
        s.head3 = lg(head[3])
        s.affine4 = lg(head[4].affine_transform)
        s.head6 = lg(head[6])
        if bender.has_weights:
            for i, layer in enumerate(bender.backbone_layers):
                s.bender[i] = lg(layer)
            s.bender_out = lg(bender.output_head)
        return s


    def _expected_positions_with_graph(self, ray_origins, ray_directions, focal_normals, transformation_matrix_w2o, style,
                                       deformation, object_in_scene, object_id, perturb, canonical_pose, noise, params) -> Dict:
        """Differentiable forward_expected_positions (the pose / keypoint consistency losses of the reference's trainers):
        the single-object render is one autograd node whose outputs include the compositing weights, the sample depths
        and the bender's displacement vectors (pr_render_backward takes their gradients); the weighted mean itself is a
        handful of torch ops on (N, R, P) tensors, which also carry the direct dependence of the object-frame ray on the
        pose."""
        kwargs = dict(ray_origins=ray_origins, ray_directions=ray_directions, focal_normals=focal_normals,
                      transformation_matrix_w2o=transformation_matrix_w2o.unsqueeze(-1), style=style.unsqueeze(-1),
                      deformation=deformation.unsqueeze(-1), object_in_scene=object_in_scene.unsqueeze(-1), perturb=perturb,
                      canonical_pose=canonical_pose, _noise=noise, _export=True, _object_ids=[object_id])
        results = self._render_with_graph(kwargs, 1, params)
        lead = list(ray_directions.shape[:-2])
        R = ray_directions.size(-2)
        m = torch.broadcast_to(transformation_matrix_w2o.to(torch.float32), lead + [4, 4])
        o = torch.broadcast_to(ray_origins.to(torch.float32), lead + [3])
        # object-frame ray (RayHelper.transform_rays, ray_helper.py:1203-1227)
        o_obj = torch.sum(o.unsqueeze(-2) * m[..., :3, :3], -1) + m[..., :3, 3]                               # (..., 3)
        d_obj = torch.sum(ray_directions.to(torch.float32).unsqueeze(-2) * m[..., :3, :3].unsqueeze(-3), -1)  # (..., R, 3)
        out = {}
        for ty in ("coarse", "fine"):
            if ty not in results:
                continue
            samples = results[ty]["_samples"][0]
            t = samples["t"][0].reshape(lead + [R, -1])
            delta = samples["delta"][0].reshape(lead + [R, -1, 3])
            entry = results[ty]["object_0"]
            weights = entry["weights"].detach()       # compute_expected_positions detaches them (object_composer.py:615)
            positions = o_obj.unsqueeze(-2).unsqueeze(-2) + d_obj.unsqueeze(-2) * t.unsqueeze(-1) + delta
            expected = (positions * weights.unsqueeze(-1)).sum(-2) / (weights.sum(-1, keepdim=True) + 1e-8)
            out[ty] = (expected, entry["opacity"])
        return out
