"""Shared helpers of the test-suite (tests may use the oracle; the product never does)."""
import torch

from oracle import render_oracle as ro


def grid_pixels(h, w, n):
    r = torch.linspace(0, h - 1, n).long()
    c = torch.linspace(0, w - 1, n).long()
    rr, cc = torch.meshgrid(r, c, indexing="ij")
    return rr.reshape(-1), cc.reshape(-1)


def composer_inputs(config, scene, strides=None, pixels=None):
    """Scene encoding -> the seven tensors ObjectComposer.forward takes (CPU, via the oracle's ray set-up)."""
    rows = cols = None
    if strides:
        rows, cols = ro.strided_grid_pixels(scene["image_size"][0], scene["image_size"][1], strides)
    if pixels is not None:
        rows, cols = pixels
    o, d, n = ro.world_rays_from_cameras(config, scene["camera_rotations"], scene["camera_translations"],
                                         scene["focals"], scene["image_size"], rows, cols)
    w2o, _ = ro.object_matrices(scene["object_rotation_parameters"], scene["object_translation_parameters"])
    return (o, d, n, w2o, scene["object_style"].unsqueeze(-3), scene["object_deformation"].unsqueeze(-3),
            scene["object_in_scene"].unsqueeze(-2))


def compare_results(want, got, rtol, atol, path="", out=None):
    """NaN-aware comparison of two composer result dicts; ``weights`` are compared after sorting
    (tie order inside equal-t groups is unspecified in the reference).  Returns {field: (maxdiff, ok)}."""
    out = out if out is not None else {}
    for k in want:
        if k in ("pytorch_hook", "extra_outputs") or k.startswith("_"):
            continue
        if isinstance(want[k], dict):
            compare_results(want[k], got[k], rtol, atol, path + k + ".", out)
            continue
        a, b = want[k].detach().cpu().float(), got[k].detach().cpu().float()
        assert a.shape == b.shape, f"{path + k}: shape {tuple(b.shape)} != {tuple(a.shape)}"
        if k == "weights":
            a, _ = torch.sort(a, dim=-1)
            b, _ = torch.sort(b, dim=-1)
        nan_ok = torch.equal(torch.isnan(a), torch.isnan(b))
        diff = torch.nan_to_num(a - b, nan=0.0, posinf=0.0, neginf=0.0).abs().max().item()
        ok = nan_ok and torch.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
        out[path + k] = (diff, ok)
    return out
