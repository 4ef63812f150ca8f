"""Host -> device hand-over of a training / evaluation batch (the DataLoader -> ``Batch`` path,
dataset/batching.py:163-356 of the reference).

``Batch`` keeps the reference's attribute names and ``to_tuple`` / ``to_keypoints_typle`` / ``to_object_poses_tuple``
contracts, so trainers that unpack a batch work unchanged.  What differs is how the tensors reach the GPU.  The reference
issues one blocking ``.cuda()`` per tensor from pageable memory (its ``pin_memory`` discards the pinned copies it makes, and
its optical flow never leaves the host: ``self.optical_flows.cuda()`` is not assigned); here every tensor of the batch is
packed into ONE page-locked arena (``pin_memory()`` - what a DataLoader worker with ``pin_memory=True`` calls) and moved
with ONE asynchronous copy on a side stream (``to_cuda``): the trainer's stream waits on an event, the host does not
block, and the next batch's copy overlaps the current step's kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

_TENSOR_FIELDS = ("observations", "actions", "rewards", "dones", "camera_rotation", "camera_translation", "focals",
                  "bounding_boxes", "bounding_boxes_validity", "global_frame_indexes", "video_frame_indexes", "video_indexes",
                  "optical_flows", "keypoints", "keypoints_validity", "object_rotation", "object_translation", "crop_regions")


def _aligned(offset: int, alignment: int = 256) -> int:
    return (offset + alignment - 1) // alignment * alignment


class Batch:
    """dataset/batching.py:163-356.  observations (bs, O, C, 3 * stacking, H, W); camera_rotation / camera_translation
    (bs, O, C, 3); focals (bs, O, C); bounding_boxes (bs, O, C, 4, dynamic objects); bounding_boxes_validity
    (bs, O, C, dynamic objects); *_indexes (bs, O) / (bs); optional optical_flows (bs, O, C, 2, H, W), keypoints
    (bs, O, C, 17, 3, dynamic objects) + validity, object poses, crop regions."""

    def __init__(self, observations, actions, rewards, metadata, dones, camera_rotation, camera_translation, focals,
                 bounding_boxes, bounding_boxes_validity, observations_paths, global_frame_indexes, video_frame_indexes,
                 video_indexes, videos, optical_flows=None, keypoints=None, keypoints_validity=None, object_rotation=None,
                 object_translation=None, crop_regions=None):
        self.size = actions.size(1)
        self.observations = observations
        self.actions = actions
        self.rewards = rewards
        self.metadata = metadata
        self.dones = dones
        self.camera_rotation = camera_rotation
        self.camera_translation = camera_translation
        self.focals = focals
        self.bounding_boxes = bounding_boxes
        self.bounding_boxes_validity = bounding_boxes_validity
        self.observations_paths = observations_paths
        self.global_frame_indexes = global_frame_indexes
        self.video_frame_indexes = video_frame_indexes
        self.video_indexes = video_indexes
        self.video = videos
        self.keypoints = keypoints
        self.keypoints_validity = keypoints_validity
        self.optical_flows = optical_flows
        self.object_rotation = object_rotation
        self.object_translation = object_translation
        self.crop_regions = crop_regions
        self._arena: Optional[torch.Tensor] = None      # the page-locked staging buffer (pin_memory)
        self._ready: Optional[torch.cuda.Event] = None  # the device copy has been enqueued; consumers wait on it

    # ------------------------------------------------------------------ presence checks (reference names)
    def has_keypoints(self) -> bool:
        return self.keypoints is not None

    def has_flow(self) -> bool:
        return self.optical_flows is not None

    def has_object_poses(self) -> bool:
        return self.object_rotation is not None and self.object_translation is not None

    def has_crop_regions(self) -> bool:
        return self.crop_regions is not None

    # ------------------------------------------------------------------ staging
    def _present(self) -> List[Tuple[str, torch.Tensor]]:
        return [(name, getattr(self, name)) for name in _TENSOR_FIELDS if torch.is_tensor(getattr(self, name))]

    def _layout(self) -> Tuple[Dict[str, Tuple[int, torch.Tensor]], int]:
        offsets, total = {}, 0
        for name, t in self._present():
            total = _aligned(total)
            offsets[name] = (total, t)
            total += t.numel() * t.element_size()
        return offsets, _aligned(total)

    def pin_memory(self):
        """Packs every tensor into one page-locked byte arena (the tensors become views of it).  A DataLoader built with
        ``pin_memory=True`` calls this in its pinning thread."""
        if self._arena is not None:
            return self
        offsets, total = self._layout()
        if any(t.is_cuda for _, t in offsets.values()):
            return self                                   # already on the device
        arena = torch.empty(total, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        for name, (offset, t) in offsets.items():
            nbytes = t.numel() * t.element_size()
            view = arena[offset:offset + nbytes].view(t.dtype).view(t.shape)
            view.copy_(t)
            setattr(self, name, view)
        self._arena = arena
        return self

    def to_cuda(self, device=None, stream: Optional[torch.cuda.Stream] = None):
        """Moves the batch to the GPU with one asynchronous copy of the arena on ``stream`` (default: a dedicated copy
        stream of the device); the CURRENT stream is made to wait for it, the host is not.  Idempotent."""
        if self._ready is not None or all(t.is_cuda for _, t in self._present()):
            return
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.pin_memory()
        offsets, total = self._layout()
        copy_stream = stream if stream is not None else _copy_stream(device)
        with torch.cuda.stream(copy_stream):
            staged = self._arena.to(device, non_blocking=True)
        self._ready = torch.cuda.Event()
        self._ready.record(copy_stream)
        torch.cuda.current_stream(device).wait_event(self._ready)
        staged.record_stream(torch.cuda.current_stream(device))      # the arena is consumed on the compute stream
        for name, (offset, t) in offsets.items():
            nbytes = t.numel() * t.element_size()
            setattr(self, name, staged[offset:offset + nbytes].view(t.dtype).view(t.shape))
        self._device_arena = staged

    # ------------------------------------------------------------------ tuples (reference contracts)
    def to_tuple(self, cuda: bool = True) -> Tuple:
        """(observations, actions, rewards, dones, camera_rotation, camera_translation, focals, bounding_boxes,
        bounding_boxes_validity, global_frame_indexes, video_frame_indexes, video_indexes); dataset/batching.py:252-264."""
        if cuda:
            self.to_cuda()
        return (self.observations, self.actions, self.rewards, self.dones, self.camera_rotation, self.camera_translation,
                self.focals, self.bounding_boxes, self.bounding_boxes_validity, self.global_frame_indexes,
                self.video_frame_indexes, self.video_indexes)

    def to_keypoints_typle(self, cuda: bool = True):     # (sic) the reference's spelling, dataset/batching.py:266
        if not self.has_keypoints():
            raise Exception("Keypoints were requested from the batch, but the batch has no keypoints information")
        if cuda:
            self.to_cuda()
        return self.keypoints, self.keypoints_validity

    def to_object_poses_tuple(self, cuda: bool = True):
        if not self.has_object_poses():
            raise Exception("Object poses were requested from the batch, but the batch has no object pose information")
        if cuda:
            self.to_cuda()
        return self.object_rotation, self.object_translation

    def observation_mode_arguments(self, cuda: bool = True) -> Tuple:
        """The nine leading arguments of EnvironmentModel.forward(mode="observations") in order (what the trainers pick out
        of ``to_tuple``, training/trainer_multiresolution_backpropagated_decoder.py:40-52)."""
        t = self.to_tuple(cuda)
        return (t[0], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11])


_COPY_STREAMS: Dict[int, torch.cuda.Stream] = {}


def _copy_stream(device: torch.device) -> torch.cuda.Stream:
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _COPY_STREAMS:
        _COPY_STREAMS[index] = torch.cuda.Stream(device)
    return _COPY_STREAMS[index]


def batch_from_tensors(observations, camera_rotation, camera_translation, focals, bounding_boxes, bounding_boxes_validity,
                       global_frame_indexes, video_frame_indexes, video_indexes, **optional) -> Batch:
    """A Batch from the tensors the renderer path reads; actions / rewards / dones are zero-filled (bs, O) placeholders
    (used by evaluation loops and tests that have no environment interaction data)."""
    bs, obs = observations.size(0), observations.size(1)
    return Batch(observations, torch.zeros((bs, obs), dtype=torch.int), torch.zeros((bs, obs)), None, torch.zeros((bs, obs), dtype=torch.bool),
                 camera_rotation, camera_translation, focals, bounding_boxes, bounding_boxes_validity, None, global_frame_indexes,
                 video_frame_indexes, video_indexes, None, **optional)
