// Region-of-interest max pooling: the crop the reference's object encoders / pose estimators take from the observations
// before their small ResNets (torchvision.ops.roi_pool at model/object_encoder_v4.py:121, model/object_encoder_v5.py,
// model/object_parameters_encoder_v4.py; torchvision 0.9.1 in the reference's env.yml).  torchvision is a third-party
// dependency that is not part of the reference tree: the arithmetic below follows its published operator
// (torchvision/csrc/ops/cpu/roi_pool_kernel.cpp): per box (batch index, x1, y1, x2, y2) in input pixels
//   start = round(coordinate * scale), size = max(end - start + 1, 1), bin = size / pooled size (float),
//   bin (i, j) covers rows [floor(i * bin_h), ceil((i + 1) * bin_h)) + start_h clipped to the image (same for columns),
//   output = max over the bin (0 for an empty bin), argmax = flat h * W + w of the first maximum (-1 for an empty bin).
// HBM-bound byte work: one thread per output element, lanes along the pooled width (neighbouring input columns).
#include "pr_common.h"

#include <float.h>

namespace pr {

struct RoiPoolParams {
    int images, channels, height, width, rois, ph, pw;
    float scale;
    const float* input;     // (N, C, H, W)
    const float* boxes;     // (K, 5)
    float* output;          // (K, C, ph, pw)
    int32_t* argmax;        // (K, C, ph, pw) or NULL
};

__global__ __launch_bounds__(256) void k_roi_pool(RoiPoolParams p) {
    const long total = (long)p.rois * p.channels * p.ph * p.pw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int pw = (int)(idx % p.pw);
        const int ph = (int)((idx / p.pw) % p.ph);
        const int c = (int)((idx / ((long)p.pw * p.ph)) % p.channels);
        const int k = (int)(idx / ((long)p.pw * p.ph * p.channels));
        const float* box = p.boxes + (size_t)k * 5;
        const int image = (int)box[0];
        const int start_w = (int)roundf(box[1] * p.scale);
        const int start_h = (int)roundf(box[2] * p.scale);
        const int end_w = (int)roundf(box[3] * p.scale);
        const int end_h = (int)roundf(box[4] * p.scale);
        const int roi_w = max(end_w - start_w + 1, 1);
        const int roi_h = max(end_h - start_h + 1, 1);
        const float bin_h = (float)roi_h / (float)p.ph;
        const float bin_w = (float)roi_w / (float)p.pw;
        int hstart = (int)floorf((float)ph * bin_h);
        int wstart = (int)floorf((float)pw * bin_w);
        int hend = (int)ceilf((float)(ph + 1) * bin_h);
        int wend = (int)ceilf((float)(pw + 1) * bin_w);
        hstart = min(max(hstart + start_h, 0), p.height);
        hend = min(max(hend + start_h, 0), p.height);
        wstart = min(max(wstart + start_w, 0), p.width);
        wend = min(max(wend + start_w, 0), p.width);
        const bool empty = (hend <= hstart) || (wend <= wstart) || image < 0 || image >= p.images;
        float best = empty ? 0.f : -FLT_MAX;
        int where = -1;
        if (!empty) {
            const float* plane = p.input + ((size_t)image * p.channels + c) * p.height * p.width;
            for (int h = hstart; h < hend; ++h)
                for (int w = wstart; w < wend; ++w) {
                    const float v = plane[h * p.width + w];
                    if (v > best) {
                        best = v;
                        where = h * p.width + w;
                    }
                }
        }
        p.output[idx] = best;
        if (p.argmax) p.argmax[idx] = where;
    }
}

struct RoiPoolBwdParams {
    int images, channels, height, width, rois, ph, pw;
    const float* boxes;
    const float* grad_output;   // (K, C, ph, pw)
    const int32_t* argmax;
    float* grad_input;          // (N, C, H, W), accumulated
};

__global__ __launch_bounds__(256) void k_roi_pool_bwd(RoiPoolBwdParams p) {
    const long total = (long)p.rois * p.channels * p.ph * p.pw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int where = p.argmax[idx];
        if (where < 0) continue;
        const int c = (int)((idx / ((long)p.pw * p.ph)) % p.channels);
        const int k = (int)(idx / ((long)p.pw * p.ph * p.channels));
        const int image = (int)p.boxes[(size_t)k * 5];
        if (image < 0 || image >= p.images) continue;
        atomicAdd(p.grad_input + ((size_t)image * p.channels + c) * p.height * p.width + where, p.grad_output[idx]);
    }
}

}  // namespace pr

extern "C" int pr_roi_pool_forward(int32_t images, int32_t channels, int32_t height, int32_t width, const float* input,
                                   int32_t rois, const float* boxes, int32_t pooled_height, int32_t pooled_width,
                                   float spatial_scale, float* output, int32_t* argmax, void* stream) {
    PR_REQUIRE(images >= 0 && channels > 0 && height > 0 && width > 0 && rois >= 0 && pooled_height > 0 && pooled_width > 0,
               "pr_roi_pool_forward: bad sizes");
    PR_REQUIRE((long)height * width < (1L << 31), "pr_roi_pool_forward: image too large");
    if (rois == 0) return PR_OK;
    PR_REQUIRE(input && boxes && output, "pr_roi_pool_forward: NULL pointer");
    pr::RoiPoolParams p{images, channels, height, width, rois, pooled_height, pooled_width, spatial_scale, input, boxes, output, argmax};
    const long total = (long)rois * channels * pooled_height * pooled_width;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pr::k_roi_pool, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_roi_pool_backward(int32_t images, int32_t channels, int32_t height, int32_t width, int32_t rois,
                                    const float* boxes, int32_t pooled_height, int32_t pooled_width, const float* grad_output,
                                    const int32_t* argmax, float* grad_input, void* stream) {
    PR_REQUIRE(images >= 0 && channels > 0 && height > 0 && width > 0 && rois >= 0 && pooled_height > 0 && pooled_width > 0,
               "pr_roi_pool_backward: bad sizes");
    if (rois == 0) return PR_OK;
    PR_REQUIRE(boxes && grad_output && argmax && grad_input, "pr_roi_pool_backward: NULL pointer");
    pr::RoiPoolBwdParams p{images, channels, height, width, rois, pooled_height, pooled_width, boxes, grad_output, argmax, grad_input};
    const long total = (long)rois * channels * pooled_height * pooled_width;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pr::k_roi_pool_bwd, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}
