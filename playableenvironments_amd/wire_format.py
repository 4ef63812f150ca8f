"""Renderer <-> decoder wire format.

The renderer returns features per RAY, channels last; the reference's CNN decoder wants channel-first grids or patches
per stride, with the 192 channels split by decoder layer (64 channels at stride 4, 128 at stride 8).  These helpers
restate the reference's glue as tensor views (no copies except where the reference transposes): they are what
``EnvironmentModelBackpropagatedAutoencoder`` / ``...MultiresolutionBackpropagatedDecoder`` do between
``batchified_composer_call`` and ``autoencoder_model.forward_decoder``.

  fold_strided_grid_samples, fold_strided_tensors      utils/lib_3d/ray_helper.py:484-531,
                                                       model/environment_model_backpropagated_autoencoder.py:128-168
  split_strided_patch_ray_samples, strided_patch_ray_samples_to_patch   utils/lib_3d/ray_helper.py:185-234
  split_features_by_layer                              model/environment_model_multiresolution_backpropagated_autoencoder.py:29-57
  decoder_patches                                      model/environment_model_multiresolution_backpropagated_decoder.py:86-104
  sample_features_at, sample_original_region_from_patch_samples        utils/lib_3d/ray_helper.py:1054-1155
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

Strides = Union[int, Sequence[int]]


def _as_list(strides: Strides) -> List[int]:
    return [int(strides)] if isinstance(strides, int) else [int(s) for s in strides]


def fold_strided_grid_samples(samples: torch.Tensor, strides: Strides, original_size: Tuple[int, int], dim: int) -> List[torch.Tensor]:
    """(..., sum_i H/s_i * W/s_i, ...) along ``dim`` -> one (..., H/s_i, W/s_i, ...) view per stride."""
    height, width = original_size
    dim = dim % samples.dim()
    out, begin = [], 0
    for stride in _as_list(strides):
        if height % stride != 0:
            raise Exception("The image height is not divisible by the stride")
        if width % stride != 0:
            raise Exception("The image width is not divisible by the stride")
        gh, gw = height // stride, width // stride
        part = samples.narrow(dim, begin, gh * gw)
        shape = list(part.shape)
        shape[dim:dim + 1] = [gh, gw]
        out.append(part.reshape(shape))
        begin += gh * gw
    return out


def fold_strided_tensors(dictionary: Dict, height: int, width: int, strides: Strides) -> Dict:
    """Every tensor of the (nested) dictionary with a dimension of size sum_i (H // s_i) * (W // s_i) is replaced by the
    list of its per-stride folds (first matching dimension, as the reference)."""
    strides = _as_list(strides)
    target = sum(height // s * width // s for s in strides)
    for key, value in dictionary.items():
        if type(value) is dict:
            dictionary[key] = fold_strided_tensors(value, height, width, strides)
        elif torch.is_tensor(value):
            for idx, size in enumerate(value.shape):
                if size == target:
                    dictionary[key] = fold_strided_grid_samples(value, strides, (height, width), dim=idx)
                    break
    return dictionary


def split_strided_patch_ray_samples(samples: torch.Tensor, patch_size: int, strides: Strides) -> List[torch.Tensor]:
    """(..., sum_i p_i^2, C) from the strided patch sampler -> [(..., p_i^2, C)], p_i = patch_size * s_0 // s_i."""
    strides = _as_list(strides)
    out, begin = [], 0
    for stride in strides:
        p = (patch_size * strides[0]) // stride
        out.append(samples[..., begin:begin + p * p, :])
        begin += p * p
    return out


def strided_patch_ray_samples_to_patch(samples: torch.Tensor) -> torch.Tensor:
    """(..., p^2, C) -> (..., C, p, p)."""
    samples = torch.transpose(samples, -1, -2)
    p = int(math.sqrt(samples.size(-1)))
    return samples.reshape(list(samples.shape[:-1]) + [p, p])


def split_features_by_layer(features: torch.Tensor, features_count_by_layer: Sequence[int], channel_order: str = "chw") -> List[torch.Tensor]:
    """Channel slices per decoder layer: (..., C, H, W) for "chw", (..., C) for "hwc"."""
    out, begin = [], 0
    for count in features_count_by_layer:
        if channel_order == "chw":
            out.append(features[..., begin:begin + count, :, :])
        elif channel_order == "hwc":
            out.append(features[..., begin:begin + count])
        else:
            raise Exception(f"Invalid channel order '{channel_order}'")
        begin += count
    return out


def decoder_patches(integrated_features: torch.Tensor, patch_size: int, strides: Strides,
                    features_count_by_layer: Sequence[int]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """What the multiresolution decoder consumes from a strided-patch render: for decoder layer i the channels of that
    layer at the rays of stride i.  Returns (splitted_integrated_features [(..., p_i^2, C_i)], patches [(..., C_i, p_i, p_i)])."""
    per_layer = split_features_by_layer(integrated_features, features_count_by_layer, channel_order="hwc")
    splitted, patches = [], []
    for i, layer_features in enumerate(per_layer):
        cur = split_strided_patch_ray_samples(layer_features, patch_size, strides)[i]
        splitted.append(cur)
        patches.append(strided_patch_ray_samples_to_patch(cur))
    return splitted, patches


def _flatten(t: torch.Tensor, keep: int):
    lead = list(t.shape[:keep])
    return t.reshape([-1] + list(t.shape[keep:])), lead


def sample_features_at(features: torch.Tensor, sampled_positions: torch.Tensor, mode: str = "bilinear", correct_range: bool = True,
                       original_image_size: Tuple[int, int] = None) -> torch.Tensor:
    """(..., C, H, W) sampled at (..., n, 2) normalised (row, col) positions -> (..., n, C) (grid_sample, align_corners)."""
    flat, lead = _flatten(features, -3)
    pos, _ = _flatten(sampled_positions, -2)
    if correct_range:
        size = torch.tensor(original_image_size, dtype=features.dtype, device=features.device)
        pos = pos * (size / (size - 1 + 1e-8))
    pos = pos[..., [1, 0]].unsqueeze(-2)
    pos = (pos - 0.5) * 2
    out = F.grid_sample(flat, pos, mode=mode, align_corners=True).squeeze(-1).permute([0, 2, 1])
    return out.reshape(lead + list(out.shape[1:]))


def sample_original_region_from_patch_samples(observations: torch.Tensor, sampled_positions: torch.Tensor, stride: int) -> torch.Tensor:
    """The image region (..., C, p * stride, p * stride) covered by a p x p patch of samples taken at ``stride``."""
    height, width = observations.size(-2), observations.size(-1)
    p = int(math.sqrt(sampled_positions.size(-2)))
    side = p * stride
    size = torch.tensor((height, width), dtype=observations.dtype, device=observations.device)
    flat, lead = _flatten(observations, -3)
    pos, _ = _flatten(sampled_positions, -2)
    pos = (pos * size).round()
    pos = (pos / stride).long() * stride
    top_left = pos[:, 0]
    axis = torch.arange(side, dtype=observations.dtype, device=observations.device)
    rows, cols = torch.meshgrid(axis, axis, indexing="ij")
    grid = torch.stack([rows, cols], dim=-1) + top_left.unsqueeze(1).unsqueeze(1)
    grid = (grid / (size - 1) - 0.5) * 2
    grid = grid[..., [1, 0]]
    out = F.grid_sample(flat, grid, mode="nearest", align_corners=True)
    return out.reshape(lead + list(out.shape[1:]))
