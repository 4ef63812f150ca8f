"""Multi-GPU sharding of the renderer: one process per GPU, torch.distributed (RCCL over xGMI).

Training is data-parallel over frames: every rank renders and differentiates
its own frames (per-rank BatchNorm batch statistics, as the reference's nn.DataParallel replicas do), then
the parameter gradients are summed with ``allreduce_gradients`` - one flat bucket per <= 64 MiB, because
ring all-reduce over point-to-point xGMI links is per-link bandwidth bound and wants few, large messages
(the composer's 11.5 MB of gradients travel as a single collective).

Rays are independent given the (tiny) scene encoding and the replicated weights, so the path
shards with no data-path collective; the only exchange is the gather of the rendered feature maps
(SURVEY.md section 8e).  The reference itself renders on a single GPU (evaluation bypasses
nn.DataParallel, evaluation/evaluator.py:58).  The helpers are backend-agnostic so that the
world_size-2 ``gloo`` tests on CPU exercise exactly the code the ``nccl`` (= RCCL) runs use.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of ``total`` work units for ``rank`` (the first
    ``total % world`` ranks get one extra unit)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_frames(tensor: torch.Tensor, rank: int, world: int, dim: int = 0) -> torch.Tensor:
    """This rank's frames of a per-frame tensor (scene encodings are sharded along the batch dim)."""
    b, e = shard_range(tensor.size(dim), rank, world)
    return tensor.narrow(dim, b, e - b)


def gather_ray_shards(local: torch.Tensor, total: int, dim: int, dst: Optional[int] = 0,
                      group=None) -> Optional[torch.Tensor]:
    """Reassembles a tensor whose ``dim`` was sharded with ``shard_range`` (ragged shards allowed).

    dst = rank that receives the full tensor (others return None); dst=None -> all_gather, every
    rank returns it.  One collective; shards are padded to the largest shard so that the same
    call works for RCCL and gloo."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    longest = max(e - b for b, e in sizes)
    if local.size(dim) != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("local shard does not match shard_range")
    pad = longest - local.size(dim)
    buf = local.contiguous()
    if pad:
        shape = list(local.shape)
        shape[dim] = pad
        buf = torch.cat([buf, buf.new_zeros(shape)], dim=dim).contiguous()
    if dst is None:
        parts: List[torch.Tensor] = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([p.narrow(dim, 0, e - b) for p, (b, e) in zip(parts, sizes)], dim=dim)


def tile_shard_lists(grids, world: int, tile: int = 8) -> List[torch.Tensor]:
    """Round-robin sharding of a frame's rays in ``tile`` x ``tile`` pixel tiles: for the pixel list of a render (``grids`` =
    [(rows, cols)] - one entry for a full frame, one per stride for the strided grids, concatenated in that order) the int64
    indices INTO that list each rank renders.  Tile t (row-major over the grid) belongs to rank t % world; within a rank the
    rays stay tile-major, so that the samples of neighbouring rays (same boxes, similar depths) stay neighbours in the sample
    lists.  Where contiguous ranges (``shard_range``) hand whole bands of the image to a rank - the sky to one, the players
    to another - interleaved tiles give every rank a sample of the whole frame: the evaluated-sample counts, which are what
    the MLP time follows, differ by a few percent instead of a multiple (bench.py ``shard_balance``)."""
    if world < 1 or tile < 1:
        raise ValueError(f"bad world/tile {world}/{tile}")
    lists: List[List[torch.Tensor]] = [[] for _ in range(world)]
    offset, first_tile = 0, 0
    for rows, cols in grids:
        r = torch.arange(rows, dtype=torch.int64)
        c = torch.arange(cols, dtype=torch.int64)
        tiles_per_row = (cols + tile - 1) // tile
        tile_of = (r // tile).unsqueeze(1) * tiles_per_row + (c // tile).unsqueeze(0)          # (rows, cols)
        flat = (r.unsqueeze(1) * cols + c.unsqueeze(0)).reshape(-1)
        tile_flat = tile_of.reshape(-1)
        order = torch.argsort(tile_flat, stable=True)                                        # tile-major, row-major inside
        owner = (tile_flat[order] + first_tile) % world
        for k in range(world):
            lists[k].append(flat[order][owner == k] + offset)
        offset += rows * cols
        first_tile += int(tile_of.max()) + 1 if rows * cols else 0
    return [torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64) for parts in lists]


def gather_indexed_shards(local: torch.Tensor, lists: List[torch.Tensor], dim: int, dst: Optional[int] = 0,
                          group=None) -> Optional[torch.Tensor]:
    """Reassembles a tensor whose ``dim`` was sharded with index lists (``tile_shard_lists``): rank r holds
    ``full.index_select(dim, lists[r])``.  ONE collective (padded to the longest shard, like ``gather_ray_shards``) and one
    scatter of the received rows to their places."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(lists) != world:
        raise ValueError(f"{len(lists)} index lists for {world} ranks")
    if local.size(dim) != lists[rank].numel():
        raise ValueError("local shard does not match its index list")
    total = sum(int(l.numel()) for l in lists)
    if world == 1:
        parts = [local]
    else:
        longest = max(int(l.numel()) for l in lists)
        buf = local.contiguous()
        pad = longest - local.size(dim)
        if pad:
            shape = list(local.shape)
            shape[dim] = pad
            buf = torch.cat([buf, buf.new_zeros(shape)], dim=dim).contiguous()
        if dst is None:
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf, group=group)
        else:
            parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
            dist.gather(buf, parts, dst=dst, group=group)
            if rank != dst:
                return None
        parts = [p.narrow(dim, 0, int(l.numel())) for p, l in zip(parts, lists)]
    shape = list(local.shape)
    shape[dim] = total
    full = local.new_empty(shape)
    full.index_copy_(dim, torch.cat(lists).to(local.device), torch.cat(parts, dim=dim))
    return full


def allreduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None, average: bool = True,
                        bucket_bytes: int = 64 << 20) -> int:
    """Sums (``average``: averages) ``.grad`` of ``parameters`` over the ranks, in place; parameters without a
    gradient contribute zeros so that every rank issues the same collectives.  Gradients are packed into flat
    buckets of at most ``bucket_bytes``; returns the number of collectives issued."""
    params = [p for p in parameters if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not params:
        return 0
    world = dist.get_world_size(group)
    shared = _shared_gradient_buffer(params)
    if shared is not None:
        # the renderer's backward hands out views of one flat buffer: reduce it where it lies
        per_bucket = max(1, bucket_bytes // 4)
        pieces = [shared[i:i + per_bucket] for i in range(0, shared.numel(), per_bucket)]
        for piece in pieces:
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
            if average:
                piece /= world
        return len(pieces)
    buckets: List[List[torch.nn.Parameter]] = [[]]
    used = 0
    for p in params:
        size = p.numel() * 4
        if buckets[-1] and used + size > bucket_bytes:
            buckets.append([])
            used = 0
        buckets[-1].append(p)
        used += size
    for bucket in buckets:
        ref = bucket[0]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
                          for p in bucket]).to(ref.device)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        offset = 0
        for p in bucket:
            n = p.numel()
            g = flat[offset:offset + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            offset += n
    return len(buckets)


class OverlappedGradientAllReduce:
    """Starts the all-reduce of the renderer's parameter gradients from INSIDE ``backward()``: ``ObjectComposer``'s autograd node
    calls its ``gradient_hooks`` with the flat buffer all its parameter gradients are views of, as soon as ``pr_render_backward`` is
    enqueued - i.e. before autograd walks on into whatever produced the renderer's inputs (the object / pose encoders' CNN backward in
    the reference's trainers, training/trainer.py).  The collective runs on the backend's stream behind the renderer's kernels and
    overlaps that tail; ``finish()`` (call it where ``allreduce_gradients`` would go, before ``optimizer.step()`` and before anything
    else reads the gradients) waits for it and averages.  One flat message: a ring over point-to-point xGMI links is per-link
    bound, few large collectives are the cheap ones.

    The in-place, in-flight reduction is only sound when autograd then ADOPTS the buffer's views as ``p.grad`` - ONE differentiable
    composer call per ``backward()`` into empty gradients (``zero_grad(set_to_none=True)``, no ``create_graph``, no tensor hooks on the
    parameters): the overlapped case.  Everything else - a second composer call in the same graph (train-mode
    ``batchified_composer_call`` makes one per ray chunk), gradient accumulation over micro-batches - is detected when the hook
    fires: such a buffer is NOT reduced in flight (autograd is about to ADD it to gradients that exist or are being reduced; the
    adds are ordered behind the collective already started), it is kept, and ``finish()`` reduces it and corrects the accumulated
    gradient by (reduced - local).  The result is the same in every case: ``p.grad`` = what ``allreduce_gradients`` after
    ``backward()`` would have left.

    >>> overlap = OverlappedGradientAllReduce(model.object_composer)
    >>> loss.backward(); overlap.finish(); allreduce_gradients(other_parameters); optimizer.step()"""

    def __init__(self, composer, group=None, average: bool = True):
        self.group, self.average = group, average
        self.pending: List[tuple] = []       # (work, flat): reduced in place, in flight since the hook fired
        self.late: List[torch.Tensor] = []   # buffers autograd accumulates unreduced: corrected in finish()
        self.launched = 0
        self.deferred = 0
        composer.gradient_hooks.append(self._launch)
        self._composer = composer

    def _parameters(self) -> List[torch.Tensor]:
        return [p for p in self._composer._parameter_list() if p.requires_grad]

    def _launch(self, flat: torch.Tensor) -> None:
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        adopts = (not self.pending and not self.late and not torch.is_grad_enabled() and
                  all(p.grad is None and not p._backward_hooks for p in self._parameters()))
        if adopts:
            self.pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))
            self.launched += 1
            return
        for work, _ in self.pending:
            work.wait()          # (the current stream waits: autograd's accumulation into the adopted buffer runs behind the collective)
        self.late.append(flat)
        self.deferred += 1

    def finish(self) -> int:
        """Waits for the collective started since the last call (the current stream waits, not the host), reduces what could not be
        reduced in flight, and averages; returns how many buffers there were."""
        done = len(self.pending) + len(self.late)
        if not done:
            return 0
        world = dist.get_world_size(self.group)
        scale = 1.0 / world if self.average else 1.0
        for work, flat in self.pending:
            work.wait()
        params = self._parameters()
        if not self.late:
            # the overlapped case: p.grad are the views autograd adopted
            (_, flat), = self.pending
            shared = _shared_gradient_buffer(params)
            if shared is None or shared.data_ptr() != flat.data_ptr() or shared.numel() != flat.numel():
                raise RuntimeError("OverlappedGradientAllReduce: the parameter gradients are not the buffer whose all-reduce was started "
                                   "inside backward() (something replaced or copied p.grad before finish()): call finish() right after "
                                   "backward(), or use parallel.allreduce_gradients")
            if self.average:
                flat *= scale
        else:
            # p.grad = [sum over ranks of the in-flight buffer, if any] + [gradients from before, already final] + the late buffers'
            # LOCAL values.  Rescale the first part, then swap every late buffer's local value for its reduced one.
            if self.pending and self.average:
                # (p.grad holds nothing from before: the in-flight collective only starts into empty gradients)
                torch._foreach_mul_([p.grad for p in params], scale)
            for flat in self.late:
                reduced = flat.clone()
                dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group)
                # correction = scale * reduced - (what autograd added: the local buffer, already rescaled above when there was an
                # in-flight part)
                reduced.mul_(scale).sub_(flat, alpha=scale if (self.pending and self.average) else 1.0)
                offset = 0
                for p in params:
                    p.grad.add_(reduced[offset:offset + p.numel()].view_as(p))
                    offset += p.numel()
                if offset != flat.numel():
                    raise RuntimeError("OverlappedGradientAllReduce: the composer's trainable parameters changed during backward()")
        self.pending, self.late = [], []
        return done

    def remove(self) -> None:
        if self._launch in self._composer.gradient_hooks:
            self._composer.gradient_hooks.remove(self._launch)


def _shared_gradient_buffer(params: List[torch.nn.Parameter]):
    """The flat fp32 tensor behind the gradients when they are consecutive contiguous views of one storage (what
    ObjectComposer's backward produces), else None."""
    first = params[0].grad
    if first is None or first.dtype != torch.float32:
        return None
    storage = first.untyped_storage().data_ptr()
    expect = first.storage_offset()
    for p in params:
        g = p.grad
        if (g is None or g.dtype != torch.float32 or g.device != first.device or not g.is_contiguous()
                or g.untyped_storage().data_ptr() != storage or g.storage_offset() != expect):
            return None
        expect += g.numel()
    total = expect - first.storage_offset()
    return torch.as_strided(first, (total,), (1,), first.storage_offset())


def flatten_parameters(module: torch.nn.Module) -> torch.nn.Parameter:
    """Re-homes the trainable parameters of ``module`` in ONE flat fp32 parameter (an "arena"): every parameter of the module
    becomes a view of it - same names, values, shapes, ``state_dict`` and checkpoint loading - and the arena is returned for the
    optimiser: ``torch.optim.Adam([arena], fused=True)`` updates 170 tensors of the renderer with ONE launch instead of a
    multi-tensor sweep (0.25 ms -> 0.02 ms per step on an MI355X).  Element-wise optimisers (SGD, Adam, AdamW ...) compute exactly
    what they compute on the separate tensors.  Between ``backward()`` and ``optimizer.step()`` call
    ``flat_gradient(arena, module)``: the renderer's backward already leaves its gradients as consecutive views of one buffer
    (which also travels as one all-reduce), so this costs nothing; it hands the gradient over to the arena and clears the views'
    ``.grad``, which is what makes ``optimizer.zero_grad()`` on the arena sufficient (gradients left on the views would be
    accumulated into by the next ``backward()``).  The views share the arena's version counter, so the renderer's packed-weight cache sees every update.
    Call it after the module sits on its device; a later ``.to()`` / ``.cuda()`` undoes the aliasing (flatten again)."""
    named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
    if not named:
        raise ValueError("flatten_parameters: the module has no trainable parameter")
    first = named[0][1]
    if any(p.dtype != torch.float32 or p.device != first.device for _, p in named):
        raise ValueError("flatten_parameters: the trainable parameters must be fp32 tensors on one device")
    arena = torch.nn.Parameter(torch.cat([p.detach().reshape(-1) for _, p in named]))
    base = arena.detach()          # (shares storage and version counter with the arena)
    views, offset = {}, 0
    for _, p in named:
        views[id(p)] = torch.nn.Parameter(base[offset:offset + p.numel()].view(p.shape), requires_grad=True)
        offset += p.numel()
    # every name of a parameter (a tensor registered under two names keeps one view)
    for name, p in list(module.named_parameters(remove_duplicate=False)):
        if id(p) not in views:
            continue
        owner = module
        *path, attr = name.split(".")
        for part in path:
            owner = getattr(owner, part)
        owner._parameters[attr] = views[id(p)]
    # (written into the _parameters dictionaries directly - modules.Tracked cannot see that: composers ANYWHERE that cached the old
    # Parameter objects, e.g. the parent model's composer when only the encoders were flattened, rebuild their lists)
    from .modules import REGISTRATION_EPOCH
    REGISTRATION_EPOCH[0] += 1
    for m in module.modules():     # caches derived from the old storages (ObjectComposer)
        drop = getattr(m, "_drop_device_caches", None)
        if callable(drop):
            drop()
    arena._flattened_names = [n for n, _ in named]
    return arena


def flat_gradient(arena: torch.nn.Parameter, module: torch.nn.Module) -> None:
    """Moves the gradients of the module's (flattened) parameters to ``arena.grad``: the buffer they already share when the
    renderer's backward produced them (no copy), else a concatenation; ``None`` gradients count as zeros.  The arena OWNS the
    gradient afterwards: every view's ``.grad`` is reset to ``None``, so that ``optimizer.zero_grad()`` - which only knows the
    arena - really clears the step's gradients and the next ``backward()`` starts from nothing instead of accumulating into
    the buffer ``arena.grad`` aliases.  Call it after ``backward()`` (and after ``allreduce_gradients``), before
    ``optimizer.step()``."""
    params = dict(module.named_parameters())
    views = [params[n] for n in arena._flattened_names]
    if all(p.grad is None for p in views):
        arena.grad = None
        return
    shared = _shared_gradient_buffer(views) if all(p.grad is not None for p in views) else None
    if shared is not None and shared.numel() == arena.numel():
        arena.grad = shared
    else:
        arena.grad = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in views])
    for p in views:
        p.grad = None


class ArenaAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` / ``AdamW`` for flat fp32 device tensors - the arena of ``flatten_parameters`` - as ONE launch of
    ``pr_adam_step`` per tensor (csrc/optim.hip: one thread per four elements; torch's fused multi-tensor kernel gives a single
    tensor one block per 65 536 elements, 0.10 ms for the 2.1 M parameters of the minecraft renderers on a 256-CU part).  Same
    hyper-parameters, same update formulas and order as ``torch.optim.Adam(amsgrad=False)`` (the reference's trainers:
    training/trainer.py:62-75); same ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so optimiser
    checkpoints move between the two.  ``capturable=True`` keeps the step count on the device (a training step recorded into a
    HIP graph: ``frame_graph.GraphedStep``); ``decoupled_weight_decay=True`` is AdamW.

    >>> arena = flatten_parameters(model.object_composer)
    >>> optimizer = ArenaAdam([arena], lr=1e-4)
    >>> loss.backward(); flat_gradient(arena, model.object_composer); optimizer.step()"""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, *, maximize: bool = False, capturable: bool = False, decoupled_weight_decay: bool = False):
        if amsgrad:
            raise ValueError("ArenaAdam: amsgrad is not supported (use torch.optim.Adam)")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"ArenaAdam: lr {lr} betas {betas} eps {eps} weight_decay {weight_decay} out of range")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize,
                        capturable=capturable, decoupled_weight_decay=decoupled_weight_decay, foreach=None, differentiable=False, fused=None)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None, *, grad_scaler=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import _lib
        lib = _lib.load()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            if torch.is_tensor(lr):
                lr = float(lr)          # (a tensor learning rate is read back: use a float with this optimiser)
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("ArenaAdam updates contiguous fp32 device tensors (parallel.flatten_parameters); use torch.optim.Adam "
                                       "for other parameters")
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("ArenaAdam: the gradient must be a dense fp32 tensor on the parameter's device")
                if not g.is_contiguous():
                    g = g.contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.zeros((), dtype=torch.float32, device=p.device if group["capturable"] else "cpu")
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = state["step"]
                on_device = step.is_cuda
                if not on_device:
                    step += 1
                with torch.cuda.device(p.device):
                    _lib.check(lib.pr_adam_step(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                                p.numel(), lr, beta1, beta2, group["eps"], group["weight_decay"],
                                                1 if group["decoupled_weight_decay"] else 0, 1 if group["maximize"] else 0,
                                                0.0 if on_device else float(step), step.data_ptr() if on_device else None, None, None,
                                                torch.cuda.current_stream(p.device).cuda_stream), "pr_adam_step")
                # (the kernel wrote through a raw pointer: tell autograd's version counter, which the renderer's packed-weight
                # cache and the saved-tensor checks key on)
                _bump_version(p)
        return loss


def _bump_version(t: torch.Tensor) -> None:
    """An in-place no-op through torch's own machinery: moves the version counter of ``t`` (and of every view that shares it)
    after a kernel of this library has updated the storage through a raw pointer.  A zero-size slice: no launch."""
    t.detach()[:0].zero_()


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Copies ``src``'s buffers (BatchNorm running statistics, annealing step) to every rank - the reference keeps
    replica 0's running statistics under nn.DataParallel; call this before checkpointing to match."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src=src, group=group)


class AsyncFeatureGather:
    """Pipelined all-gather of per-rank result tensors (the rendered feature maps of a frame shard).

    ``submit(t)`` enqueues one ``all_gather_into_tensor`` behind the kernels that produced ``t`` and returns at once: the
    collective runs on the backend's own stream and overlaps whatever the caller enqueues next (the next frame's
    rendering); at most ``depth`` collectives stay in flight, older ones are waited for first.  ``drain()`` waits for the
    rest and returns the gathered tensors in submission order - (world * t.shape[0], ...) each, rank-major."""

    def __init__(self, group=None, depth: int = 1):
        self.group = group
        self.depth = max(1, depth)
        self._in_flight: List[tuple] = []
        self._done: List[torch.Tensor] = []

    def submit(self, tensor: torch.Tensor) -> None:
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            self._done.append(tensor)
            return
        world = dist.get_world_size(self.group)
        src = tensor.contiguous()
        out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        work = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        self._in_flight.append((work, out, src))
        while len(self._in_flight) > self.depth:
            self._retire()

    def _retire(self) -> None:
        work, out, _ = self._in_flight.pop(0)
        work.wait()
        self._done.append(out)

    def drain(self, keep: bool = True) -> List[torch.Tensor]:
        while self._in_flight:
            self._retire()
        done, self._done = self._done, []
        return done if keep else []
