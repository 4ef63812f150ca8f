// Probe (GPU box): is  r = x - float(bf16 term)  through v_dot2_f32_bf16 (term pair . {-1, 0} + x, ONE instruction per value) bit for bit
// the shift / mask + v_sub_f32 it would replace in the bf16-triple split (gemm.hip: bf16_split_pair)?  Counts mismatches of the three packed
// terms over random values of every magnitude class.  (A DOT result needs three wait states in front of a VALU instruction that reads it on
// gfx940+, and hipcc does not see DOT instructions inside inline asm: without the s_nop the second stage read the register's OLD content.)
//   hipcc --offload-arch=gfx950 -O3 tools/perf/probe_dot2.hip -o build/probes/probe_dot2 && build/probes/probe_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_old(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const f32x2 v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {sub_f32(x0, __uint_as_float(p1 << 16)), sub_f32(x1, __uint_as_float(p1 & 0xffff0000u))};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 q = {sub_f32(r[0], __uint_as_float(p2 << 16)), sub_f32(r[1], __uint_as_float(p2 & 0xffff0000u))};
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}
__device__ __forceinline__ float minus_lo(unsigned pair, float x) {      // x - pair.lo
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3\n\ts_nop 3" : "=v"(r) : "v"(pair), "v"(0x0000bf80u), "v"(x));
    return r;
}
__device__ __forceinline__ float minus_hi(unsigned pair, float x) {      // x - pair.hi
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3\n\ts_nop 3" : "=v"(r) : "v"(pair), "v"(0xbf800000u), "v"(x));
    return r;
}
__device__ __forceinline__ void split_new(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const f32x2 v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {minus_lo(p1, x0), minus_hi(p1, x1)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 q = {minus_lo(p2, r[0]), minus_hi(p2, r[1])};
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

__global__ void k(const float* x, int n, unsigned* mismatch, unsigned* first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a1, a2, a3, b1, b2, b3;
    split_old(x[2 * i], x[2 * i + 1], a1, a2, a3);
    split_new(x[2 * i], x[2 * i + 1], b1, b2, b3);
    if (a1 != b1 || a2 != b2 || a3 != b3) {
        const unsigned at = atomicAdd(mismatch, 1u);
        if (at < 8) {
            first[at * 8 + 0] = __float_as_uint(x[2 * i]); first[at * 8 + 1] = __float_as_uint(x[2 * i + 1]);
            first[at * 8 + 2] = a1; first[at * 8 + 3] = a2; first[at * 8 + 4] = a3;
            first[at * 8 + 5] = b1; first[at * 8 + 6] = b2; first[at * 8 + 7] = b3;
        }
    }
}

int main() {
    struct Class { const char* name; int emin, emax; };
    const Class classes[] = {{"normal 2^-20 .. 2^20", 107, 147}, {"tiny 2^-126 .. 2^-100", 1, 27}, {"denormal", 0, 0},
                             {"huge 2^100 .. 2^127", 227, 254}, {"gradients 2^-60 .. 2^-10", 67, 117}};
    const int n = 1 << 22;
    std::vector<float> h(n);
    float* d;
    unsigned *dm, *df;
    hipMalloc(&d, n * 4); hipMalloc(&dm, 4); hipMalloc(&df, 64 * 4);
    srand(1);
    for (const Class& c : classes) {
        for (int i = 0; i < n; ++i) {
            const unsigned mant = ((unsigned)rand() << 8 ^ (unsigned)rand()) & 0x7fffffu;
            const unsigned e = c.emin + (c.emax > c.emin ? rand() % (c.emax - c.emin + 1) : 0);
            const unsigned bits = ((unsigned)(rand() & 1) << 31) | (e << 23) | mant;
            memcpy(&h[i], &bits, 4);
        }
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipMemset(dm, 0, 4);
        k<<<n / 2 / 256, 256>>>(d, n, dm, df);
        unsigned m = 0, f[64];
        hipMemcpy(&m, dm, 4, hipMemcpyDeviceToHost);
        hipMemcpy(f, df, 64 * 4, hipMemcpyDeviceToHost);
        printf("%-28s %u of %d pairs differ\n", c.name, m, n / 2);
        for (unsigned q = 0; q < (m < 3 ? m : 3); ++q)
            printf("    x = %08x %08x  old %08x %08x %08x  dot2 %08x %08x %08x\n", f[q * 8], f[q * 8 + 1], f[q * 8 + 2], f[q * 8 + 3],
                   f[q * 8 + 4], f[q * 8 + 5], f[q * 8 + 6], f[q * 8 + 7]);
    }
    return 0;
}
