"""CPU restatement of torchvision.ops.roi_pool - TEST INFRASTRUCTURE, never imported by the product package.

The reference's object encoders and pose estimators crop the observations with ``torchvision.ops.roi_pool(observations,
boxes, input_size)`` (model/object_encoder_v4.py:121, model/object_encoder_v5.py:121, model/object_parameters_encoder_v4.py:131).
torchvision (0.9.1 in the reference's env.yml:105) is a third-party dependency that is neither part of /root/reference nor
installed in this image, so its operator cannot be run here: this file restates its published CPU kernel
(torchvision/csrc/ops/cpu/roi_pool_kernel.cpp, ``roi_pool_forward_kernel_impl`` / ``roi_pool_backward_kernel_impl``).

Pin status: PARTIALLY pinned.  For boxes that lie inside the image the operator is, by its definition, an adaptive max
pooling of the cropped window with the same bin rule (start = floor(i * size / out), end = ceil((i + 1) * size / out)):
``check_against_adaptive_max_pool`` compares this restatement with ATen's ``adaptive_max_pool2d`` (an independent
implementation) on such boxes, values and argmax positions exactly.  Boxes that leave the image (per-bin clipping, empty
bins -> 0 / argmax -1) are covered by the definition only: "parity unpinned" for that part.
"""
import math

import torch


def _c_round(v: float) -> int:
    """std::round: half away from zero."""
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def roi_pool(inputs: torch.Tensor, boxes: torch.Tensor, output_size, spatial_scale: float = 1.0):
    """inputs (N, C, H, W); boxes (K, 5) [image, x1, y1, x2, y2] -> output (K, C, ph, pw), argmax (K, C, ph, pw) int64."""
    n, channels, height, width = inputs.shape
    ph_count, pw_count = output_size
    k = boxes.size(0)
    out = torch.zeros((k, channels, ph_count, pw_count), dtype=inputs.dtype)
    arg = torch.full((k, channels, ph_count, pw_count), -1, dtype=torch.int64)
    for r in range(k):
        image = int(boxes[r, 0])
        start_w = _c_round(float(boxes[r, 1]) * spatial_scale)
        start_h = _c_round(float(boxes[r, 2]) * spatial_scale)
        end_w = _c_round(float(boxes[r, 3]) * spatial_scale)
        end_h = _c_round(float(boxes[r, 4]) * spatial_scale)
        roi_w = max(end_w - start_w + 1, 1)
        roi_h = max(end_h - start_h + 1, 1)
        # the kernel divides in fp32
        bin_h = float(torch.tensor(roi_h, dtype=torch.float32) / torch.tensor(ph_count, dtype=torch.float32))
        bin_w = float(torch.tensor(roi_w, dtype=torch.float32) / torch.tensor(pw_count, dtype=torch.float32))
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        for ph in range(ph_count):
            hstart = min(max(int(math.floor(f32(ph * f32(bin_h)))) + start_h, 0), height)
            hend = min(max(int(math.ceil(f32((ph + 1) * f32(bin_h)))) + start_h, 0), height)
            for pw in range(pw_count):
                wstart = min(max(int(math.floor(f32(pw * f32(bin_w)))) + start_w, 0), width)
                wend = min(max(int(math.ceil(f32((pw + 1) * f32(bin_w)))) + start_w, 0), width)
                if hend <= hstart or wend <= wstart or not (0 <= image < n):
                    continue
                window = inputs[image, :, hstart:hend, wstart:wend].reshape(channels, -1)
                best, where = window.max(dim=1)          # first maximum in row-major order, like the kernel's strict '>'
                out[r, :, ph, pw] = best
                rows = hstart + where // (wend - wstart)
                cols = wstart + where % (wend - wstart)
                arg[r, :, ph, pw] = rows * width + cols
    return out, arg


def roi_pool_backward(grad_output: torch.Tensor, argmax: torch.Tensor, boxes: torch.Tensor, input_shape) -> torch.Tensor:
    """grad_input[image, c].flat[argmax] += grad_output  (roi_pool_backward_kernel_impl)."""
    n, channels, height, width = input_shape
    grad = torch.zeros(input_shape, dtype=grad_output.dtype)
    flat = grad.reshape(n, channels, height * width)
    for r in range(boxes.size(0)):
        image = int(boxes[r, 0])
        for c in range(channels):
            a = argmax[r, c].reshape(-1)
            g = grad_output[r, c].reshape(-1)
            keep = a >= 0
            flat[image, c].index_add_(0, a[keep], g[keep])
    return grad


def check_against_adaptive_max_pool(seed: int = 0, cases: int = 40) -> bool:
    """The restatement against ATen's adaptive_max_pool2d on boxes inside the image (values and argmax exactly)."""
    g = torch.Generator().manual_seed(seed)
    ok = True
    for case in range(cases):
        n, c = 2, 3
        h = int(torch.randint(12, 40, (1,), generator=g))
        w = int(torch.randint(12, 40, (1,), generator=g))
        x = torch.randn((n, c, h, w), generator=g)
        out_size = (int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(1, 9, (1,), generator=g)))
        x1 = int(torch.randint(0, w - 2, (1,), generator=g))
        y1 = int(torch.randint(0, h - 2, (1,), generator=g))
        x2 = int(torch.randint(x1, w, (1,), generator=g))
        y2 = int(torch.randint(y1, h, (1,), generator=g))
        image = case % n
        jitter = (torch.rand((4,), generator=g) - 0.5) * 0.8          # rounds back to the integer corners
        box = torch.tensor([[image, x1 + jitter[0], y1 + jitter[1], x2 + jitter[2], y2 + jitter[3]]], dtype=torch.float32)
        got, arg = roi_pool(x, box, out_size)
        crop = x[image:image + 1, :, y1:y2 + 1, x1:x2 + 1]
        want, idx = torch.nn.functional.adaptive_max_pool2d(crop, out_size, return_indices=True)
        cw = x2 - x1 + 1
        want_arg = (y1 + idx // cw) * w + (x1 + idx % cw)
        ok &= torch.equal(got, want) and torch.equal(arg, want_arg)
    return bool(ok)


if __name__ == "__main__":
    print("roi_pool restatement == ATen adaptive_max_pool2d on in-image boxes:", check_against_adaptive_max_pool())
