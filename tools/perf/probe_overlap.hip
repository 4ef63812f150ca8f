// Microbenchmark (GPU box): does VALU work issue while the matrix pipe executes?  One wave per SIMD or two; a loop of bf16
// v_mfma_f32_32x32x16 (32 cycles each) with K independent VALU operations of a given kind between two MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/perf/probe_overlap.hip -o build/probe_overlap && build/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int K, bool MFMA>
__global__ __launch_bounds__(512) void k_probe(int iterations, float* sink) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(1.0f - threadIdx.x * 1e-3f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    unsigned u[4] = {1u, 2u, 3u, 4u};
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MFMA) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) {          // v_add_f32 (independent chains)
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]));
                } else if (KIND == 1) {   // v_cvt_pk_bf16_f32
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[k & 3]) : "v"(v[k & 7]), "v"(v[(k + 3) & 7]));
                } else if (KIND == 2) {   // v_and_b32 / v_lshlrev
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[k & 3]) : "v"(u[(k + 1) & 3]));
                } else {                  // v_sub_f32
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[k & 7]) : "v"(v[(k + 2) & 7]), "v"(v[(k + 5) & 7]));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += (float)u[i];
    if (s == 123.456f) sink[0] = s;
}

template <int KIND, int K, bool MFMA>
static void run(const char* name, int waves, float* sink, int cus) {
    const int iterations = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<KIND, K, MFMA>), dim3(cus), dim3(64 * waves), 0, 0, 16, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_probe<KIND, K, MFMA>), dim3(cus), dim3(64 * waves), 0, 0, iterations, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per (MFMA + K VALU) group at 2.4 GHz nominal
    const double groups = (double)iterations * 4;
    printf("%-34s waves/CU %d  K=%2d  %s  %.3f ms  %.1f ns per group = %.1f cycles @2.4GHz\n", name, waves, K, MFMA ? "mfma+valu" : "valu only", ms,
           ms * 1e6 / groups, ms * 1e6 / groups * 2.4);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* sink;
    hipMalloc(&sink, 4);
    for (int waves : {4, 8}) {
        run<0, 0, true>("mfma only", waves, sink, cus);
        run<0, 6, true>("v_add_f32 x6", waves, sink, cus);
        run<0, 6, false>("v_add_f32 x6", waves, sink, cus);
        run<1, 6, true>("v_cvt_pk_bf16_f32 x6", waves, sink, cus);
        run<1, 6, false>("v_cvt_pk_bf16_f32 x6", waves, sink, cus);
        run<2, 6, true>("v_lshlrev_b32 x6", waves, sink, cus);
        run<3, 6, true>("v_sub_f32 x6", waves, sink, cus);
        run<0, 12, true>("v_add_f32 x12", waves, sink, cus);
        run<0, 3, true>("v_add_f32 x3", waves, sink, cus);
    }
    return 0;
}
