// fp32 matrix-core GEMMs of the backward pass (exact fp32 arithmetic, v_mfma_f32_32x32x2_f32).
//
//   gemm_nn :  C[M x N] (+)= A[M x K] . B[K x N]   (dX = dY . W ; M = evaluated samples, read from the device)
//   gemm_tn :  C[Ni x Nj] += sum_m A[m][i] B[m][j]  (dW = dY^T . X, bias gradient = column sums of dY)
//
// Both stage 16-deep operand slabs in LDS with the reduction index as the slow dimension (row stride
// 160 floats: the two K-halves of a wavefront land 32 banks apart), four waves per workgroup, each wave
// a 64 x 64 block of the 128 x 128 output tile (2 x 2 MFMA accumulators).  gemm_tn splits the sample
// dimension over `splits` workgroups per output tile and reduces the partial tiles in a second, fixed-order
// pass, so weight gradients are bit-reproducible run to run.
#include "pr_common.h"

namespace pr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PR_MFMA32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

constexpr int GT = 128;        // output tile edge
constexpr int GK = 16;         // reduction slab depth
constexpr int GLD = 160;       // LDS row stride (floats)

struct GemmSmem {
    float A[GK * GLD];
    float B[GK * GLD];
};

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// one 16-deep slab: every wave multiplies its 64 x 64 block
__device__ __forceinline__ void slab_mfma(const GemmSmem& S, f32x16 (&acc)[2][2], int wr, int wc, int r, int half) {
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
        const float a0 = S.A[(kk + half) * GLD + wr * 64 + r];
        const float a1 = S.A[(kk + half) * GLD + wr * 64 + 32 + r];
        const float b0 = S.B[(kk + half) * GLD + wc * 64 + r];
        const float b1 = S.B[(kk + half) * GLD + wc * 64 + 32 + r];
        PR_MFMA32(acc[0][0], a0, b0);
        PR_MFMA32(acc[0][1], a0, b1);
        PR_MFMA32(acc[1][0], a1, b0);
        PR_MFMA32(acc[1][1], a1, b1);
    }
}

__global__ __launch_bounds__(256, 2) void k_gemm_nn(GemmNN p) {
    __shared__ GemmSmem S;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int n0 = blockIdx.y * GT;
    for (int row0 = blockIdx.x * GT; row0 < M; row0 += gridDim.x * GT) {
        f32x16 acc[2][2];
        zero_acc(acc);
        for (int k0 = 0; k0 < p.k; k0 += GK) {
            {   // A slab: 128 rows x 16 k, row-major source -> k-major LDS
                const int row = tid >> 1, kofs = (tid & 1) * 8;
                float v[8];
                if (row0 + row < M) {
                    const float4* src = reinterpret_cast<const float4*>(p.A + (size_t)(row0 + row) * p.lda + k0 + kofs);
                    const float4 x = src[0], y = src[1];
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) S.A[(kofs + i) * GLD + row] = v[i];
            }
            {   // B slab: 16 k x 128 n from the row-major weight (unaligned leading dimension)
                const int kk = tid >> 4, nofs = (tid & 15) * 8;
                const float* src = p.B + (size_t)(k0 + kk) * p.ldb + n0 + nofs;
#pragma unroll
                for (int i = 0; i < 8; ++i) S.B[kk * GLD + nofs + i] = (n0 + nofs + i < p.n) ? src[i] : 0.f;
            }
            __syncthreads();
            slab_mfma(S, acc, wr, wc, r, half);
            __syncthreads();
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int col = n0 + wc * 64 + cb * 32 + r;
                if (col >= p.n) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0 + wr * 64 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    if (row >= M) continue;
                    float v = acc[rb][cb][i];
                    float* dst = p.C + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *dst;
                    if (p.mask && !(p.mask[(size_t)row * p.ldm + col] > 0.f)) v = 0.f;
                    *dst = v;
                }
            }
    }
}

int launch_gemm_nn(const GemmNN& p, int max_rows, hipStream_t s) {
    PR_REQUIRE(p.k % GK == 0 && (p.lda & 3) == 0, "gemm_nn: K %d / lda %d not aligned", p.k, p.lda);
    if (max_rows <= 0 || p.n <= 0) return PR_OK;
    int row_tiles = (max_rows + GT - 1) / GT;
    if (row_tiles > 1024) row_tiles = 1024;
    hipLaunchKernelGGL(k_gemm_nn, dim3(row_tiles, (p.n + GT - 1) / GT), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// dW: reduction over the samples, split across workgroups
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_gemm_tn(GemmTN p) {
    __shared__ GemmSmem S;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int split = blockIdx.y;
    int chunk = (M + p.splits - 1) / p.splits;
    chunk = (chunk + GK - 1) / GK * GK;
    const int m_begin = split * chunk;
    const int m_end = (m_begin + chunk < M) ? m_begin + chunk : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    float bsum = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += GK) {
        const int kk = tid >> 4, ofs = (tid & 15) * 8;
        const bool live = m0 + kk < m_end;
        {
            const float* src = p.A + (size_t)(m0 + kk) * p.lda + i0 + ofs;
#pragma unroll
            for (int i = 0; i < 8; ++i) S.A[kk * GLD + ofs + i] = (live && i0 + ofs + i < p.ni) ? src[i] : 0.f;
        }
        {
            const float* src = p.B + (size_t)(m0 + kk) * p.ldb + j0 + ofs;
#pragma unroll
            for (int i = 0; i < 8; ++i) S.B[kk * GLD + ofs + i] = (live && j0 + ofs + i < p.nj) ? src[i] : 0.f;
        }
        __syncthreads();
        if (p.bias_partial && tj == 0 && tid < GT) {
#pragma unroll
            for (int q = 0; q < GK; ++q) bsum += S.A[q * GLD + tid];
        }
        slab_mfma(S, acc, wr, wc, r, half);
        __syncthreads();
    }
    // partial tile -> P[split][i][j] (padded to whole tiles)
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb][cb][i];
            }
        }
    if (p.bias_partial && tj == 0 && tid < GT) p.bias_partial[(size_t)split * rows_p + i0 + tid] = bsum;
}

// C[i][j] += sum_s P[s][i][j] in split order; bias[i] += sum_s PB[s][i]
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(GemmTN p) {
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)p.ni * p.nj) {
        const int i = (int)(idx / p.nj), j = (int)(idx - (long)i * p.nj);
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.partial[((size_t)s * rows_p + i) * ldp + j];
        p.C[(size_t)i * p.ldc + j] += v;
    }
    if (p.bias_partial && p.bias && idx < p.ni) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.bias_partial[(size_t)s * rows_p + idx];
        p.bias[idx] += v;
    }
}

size_t gemm_tn_scratch_floats(int splits) {
    // largest gradient: 256 x 384 (skip layer) partial tiles + bias partials
    return (size_t)splits * (256 * 384 + 256);
}

int launch_gemm_tn(const GemmTN& p, hipStream_t s) {
    PR_REQUIRE(p.ni <= 256 && p.nj <= 384, "gemm_tn: %d x %d exceeds the partial buffer", p.ni, p.nj);
    PR_REQUIRE(p.splits >= 1 && p.partial, "gemm_tn: no partial buffer");
    const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
    hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, p.splits), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    const long n = (long)p.ni * p.nj;
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
