// Adam / AdamW update of ONE flat fp32 tensor: the composer's parameter arena (parallel.flatten_parameters) in the training
// step of BASELINE.json configs[4].  The reference's trainers build torch.optim.Adam (training/trainer.py:62-75); torch's
// fused kernel (multi_tensor_apply, _fused_adam) deals one 512-thread block per 65 536 elements of a tensor - 32 blocks for the
// 2.1 M parameters of the minecraft renderers, 0.10 ms on a 256-CU part for 59 MB of traffic.  Here: one thread per four
// elements, 16-byte accesses, as many workgroups as the tensor has 1024-element pieces.  HBM-bound element-wise work: 7 floats
// of traffic per parameter (read param, grad, exp_avg, exp_avg_sq; write param, exp_avg, exp_avg_sq).
//
// The arithmetic is torch.optim.Adam's single-tensor formulation (torch/optim/adam.py _single_tensor_adam /
// aten/src/ATen/native/cuda/fused_adam_utils.cuh adam_math), in this order:
//   g = maximize ? -grad : grad;  L2 mode: g += weight_decay p;  decoupled (AdamW): p -= lr weight_decay p
//   m = m + (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g
//   p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with t the step count AFTER this update's increment.  t comes by value (host-side counter) or from a device float that a
// one-thread kernel increments first (an optimiser recorded into a HIP graph: frame_graph.GraphedStep).
#include "pr_common.h"

namespace pr {

struct AdamParams {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    long n;
    float lr, beta1, beta2, eps, weight_decay;
    int decoupled, maximize;
    double step;                 // used when step_device == NULL
    const float* step_device;    // the step count after the increment (k_adam_step_increment ran before)
    const float* grad_scale;     // optional: the gradients are divided by *grad_scale (torch.amp.GradScaler)
    const float* found_inf;      // optional: a non-zero value skips the update (GradScaler)
};

__global__ void k_adam_step_increment(float* step, const float* found_inf) {
    if (found_inf == nullptr || *found_inf == 0.f) *step += 1.0f;
}

__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, const AdamParams& q, float step_size, float bc2_sqrt) {
    if (q.maximize) g = -g;
    if (q.weight_decay != 0.f) {
        if (q.decoupled) p -= q.lr * q.weight_decay * p;
        else g += q.weight_decay * p;
    }
    m = m + (1.0f - q.beta1) * (g - m);
    v = q.beta2 * v + (1.0f - q.beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + q.eps;
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void k_adam_arena(AdamParams q) {
    if (q.found_inf != nullptr && *q.found_inf != 0.f) return;
    // the bias corrections: once per workgroup (two double-precision pow calls), not once per thread
    __shared__ float corrections[2];
    if (threadIdx.x == 0) {
        const double t = q.step_device ? (double)*q.step_device : q.step;
        corrections[0] = (float)(1.0 - pow((double)q.beta1, t));
        corrections[1] = sqrtf((float)(1.0 - pow((double)q.beta2, t)));
    }
    __syncthreads();
    const float bc1 = corrections[0], bc2_sqrt = corrections[1];
    const float step_size = q.lr / bc1;
    const float inv_scale = q.grad_scale ? 1.0f / *q.grad_scale : 1.0f;
    const long n4 = q.n >> 2;
    const bool aligned = ((((uintptr_t)q.param) | ((uintptr_t)q.grad) | ((uintptr_t)q.exp_avg) | ((uintptr_t)q.exp_avg_sq)) & 15) == 0;
    const long stride = (long)gridDim.x * 256;
    if (aligned) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 p = reinterpret_cast<float4*>(q.param)[i];
            float4 g = reinterpret_cast<const float4*>(q.grad)[i];
            float4 m = reinterpret_cast<float4*>(q.exp_avg)[i];
            float4 v = reinterpret_cast<float4*>(q.exp_avg_sq)[i];
            if (q.grad_scale) { g.x *= inv_scale; g.y *= inv_scale; g.z *= inv_scale; g.w *= inv_scale; }
            adam_element(p.x, g.x, m.x, v.x, q, step_size, bc2_sqrt);
            adam_element(p.y, g.y, m.y, v.y, q, step_size, bc2_sqrt);
            adam_element(p.z, g.z, m.z, v.z, q, step_size, bc2_sqrt);
            adam_element(p.w, g.w, m.w, v.w, q, step_size, bc2_sqrt);
            reinterpret_cast<float4*>(q.param)[i] = p;
            reinterpret_cast<float4*>(q.exp_avg)[i] = m;
            reinterpret_cast<float4*>(q.exp_avg_sq)[i] = v;
        }
    }
    // the tail (n % 4 elements), or everything when a pointer is not 16-byte aligned
    for (long i = (aligned ? n4 * 4 : 0) + (long)blockIdx.x * 256 + threadIdx.x; i < q.n; i += stride) {
        float p = q.param[i], g = q.grad[i] * inv_scale, m = q.exp_avg[i], v = q.exp_avg_sq[i];
        adam_element(p, g, m, v, q, step_size, bc2_sqrt);
        q.param[i] = p;
        q.exp_avg[i] = m;
        q.exp_avg_sq[i] = v;
    }
}

}  // namespace pr

extern "C" int pr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int32_t decoupled_weight_decay, int32_t maximize, double step,
                            float* step_device, const float* grad_scale, const float* found_inf, void* stream) {
    PR_REQUIRE(n >= 0, "pr_adam_step: n = %ld", (long)n);
    if (n == 0) return PR_OK;
    PR_REQUIRE(param && grad && exp_avg && exp_avg_sq, "pr_adam_step: NULL pointer");
    PR_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f,
               "pr_adam_step: lr %g betas (%g, %g) eps %g weight_decay %g out of range", lr, beta1, beta2, eps, weight_decay);
    PR_REQUIRE(step_device != nullptr || step >= 1.0, "pr_adam_step: step %g (the count AFTER this update, >= 1)", step);
    hipStream_t s = (hipStream_t)stream;
    if (step_device) {
        hipLaunchKernelGGL(pr::k_adam_step_increment, dim3(1), dim3(1), 0, s, step_device, found_inf);
        PR_LAUNCH_CHECK();
    }
    pr::AdamParams q{param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, decoupled_weight_decay, maximize,
                     step, step_device, grad_scale, found_inf};
    long blocks = ((n >> 2) + 1023) / 1024;       // four float4 per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pr::k_adam_arena, dim3((unsigned)blocks), dim3(256), 0, s, q);
    PR_LAUNCH_CHECK();
    return PR_OK;
}
