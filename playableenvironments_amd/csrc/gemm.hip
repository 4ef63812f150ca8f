// fp32 matrix-core GEMMs of the backward pass (exact fp32 arithmetic, v_mfma_f32_32x32x2_f32).
//
//   gemm_nn :  C[M x N] (+)= A[M x K] . B[K x N]   (dX = dY . W ; M = evaluated samples, read from the device)
//   gemm_tn :  C[Ni x Nj] += sum_m A[m][i] B[m][j]  (dW = dY^T . X, bias gradient = column sums of dY)
//
// Both stage 32-deep operand slabs in LDS (the next slab is fetched into registers while the current one is multiplied) with the reduction index as the slow dimension (row stride
// 160 floats: the two K-halves of a wavefront land 32 banks apart), four waves per workgroup, each wave
// a 64 x 64 block of the 128 x 128 output tile (2 x 2 MFMA accumulators).  gemm_tn splits the sample
// dimension over `splits` workgroups per output tile and reduces the partial tiles in a second, fixed-order
// pass, so weight gradients are bit-reproducible run to run.
#include "pr_common.h"

namespace pr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PR_MFMA32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

#ifndef PR_GEMM_ABLATE
#define PR_GEMM_ABLATE 0       // timing builds only: 1 = no epilogue, 2 = no operand re-fetch, 4 = no MFMA
#endif
#ifndef PR_TNALL_ABLATE
#define PR_TNALL_ABLATE 0      // timing builds only (k_gemm_tn_all): 1 = no partial write-out, 2 = no operand re-fetch, 4 = no MFMA
#endif
constexpr int GT = 128;        // output tile edge
constexpr int GK = 32;         // reduction slab depth
constexpr int GLD = 160;       // LDS row stride (floats)
constexpr int TN_MIN_CHUNK = 512;   // fewest samples per split of the weight-gradient reduction

constexpr int GLA = GK + 2;    // row-major A operand (gemm_nn): 34-float rows -> conflict-free b32 fragment reads

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// The next slab travels HBM -> registers while the current one is multiplied out of LDS.  Every global
// load instruction of a wavefront covers whole contiguous segments (16-byte vectors along the contiguous
// dimension of the activations; the weight rows, whose leading dimension is arbitrary, one dword per lane).
__global__ __launch_bounds__(256, 2) void k_gemm_nn(GemmNN p) {
    __shared__ __attribute__((aligned(16))) float smem[GT * GLA + GK * GLD];
    float* SA = smem;                 // [row][k]
    float* SB = smem + GT * GLA;      // [k][n]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    // Workgroup ids are dealt to the 8 XCDs round-robin.  The column tiles of one row tile read the same A rows: they
    // get ids that differ by 8 (same XCD, dispatched one after the other), so the second read is an L2 hit.
    const int ncol_tiles = (p.n + GT - 1) / GT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int n0 = (slot % ncol_tiles) * GT;
    const int first_tile = (slot / ncol_tiles) * 8 + xcd;
    const int tile_stride = (gridDim.x / (8 * ncol_tiles)) * 8;
    const int ak4 = tid & 7, arow = tid >> 3;       // A: rows arow + 32 i, one float4 of k each
    const int bn = tid & 127, bk = tid >> 7;        // B: k rows bk + 2 i, one dword each
    for (int row0 = first_tile * GT; row0 < M; row0 += tile_stride * GT) {
        f32x16 acc[2][2];
        zero_acc(acc);
        float4 ra[4];
        float rb[16];
        auto fetch = [&](int k0) {
            const int kv = p.k_valid ? p.k_valid : p.k;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + arow + 32 * i;
                const int ka = k0 + 4 * ak4;
                ra[i] = (row < M && ka < kv)
                            ? *reinterpret_cast<const float4*>(p.A + (size_t)row * p.lda + ka)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
                // columns of A beyond k_valid are padding that nobody has to have written: 0 x garbage must stay 0
                if (ka + 3 >= kv) {
                    if (ka + 1 >= kv) ra[i].y = 0.f;
                    if (ka + 2 >= kv) ra[i].z = 0.f;
                    if (ka + 3 >= kv) ra[i].w = 0.f;
                }
            }
            const bool ncol = n0 + bn < p.n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = k0 + bk + 2 * i;
                rb[i] = (ncol && kk < kv) ? (p.b_transposed ? p.B[(size_t)(n0 + bn) * p.ldb + kk] : p.B[(size_t)kk * p.ldb + n0 + bn])
                                          : 0.f;
            }
        };
        auto stage = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* dst = SA + (arow + 32 * i) * GLA + 4 * ak4;
                *reinterpret_cast<float2*>(dst) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(ra[i].z, ra[i].w);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) SB[(bk + 2 * i) * GLD + bn] = rb[i];
        };
        fetch(0);
        stage();
        __syncthreads();
        for (int k0 = 0; k0 < p.k; k0 += GK) {
            const bool more = (k0 + GK < p.k) && !(PR_GEMM_ABLATE & 2);
            if (more) fetch(k0 + GK);
#pragma unroll
            for (int kk = 0; kk < ((PR_GEMM_ABLATE & 4) ? 2 : GK); kk += 2) {
                const float a0 = SA[(wr * 64 + r) * GLA + kk + half];
                const float a1 = SA[(wr * 64 + 32 + r) * GLA + kk + half];
                const float b0 = SB[(kk + half) * GLD + wc * 64 + r];
                const float b1 = SB[(kk + half) * GLD + wc * 64 + 32 + r];
                PR_MFMA32(acc[0][0], a0, b0);
                PR_MFMA32(acc[0][1], a0, b1);
                PR_MFMA32(acc[1][0], a1, b0);
                PR_MFMA32(acc[1][1], a1, b1);
            }
            __syncthreads();
            if (more) stage();
            __syncthreads();
        }
        // epilogue through LDS (the operand slabs are dead after the last barrier): every wave parks a 32 x 64 half
        // of its block, then moves whole 256-byte row segments with 16-byte accesses; the mask / previous values of
        // a half are all requested before the first store (C, mask and A may alias as far as the compiler knows)
        if ((PR_GEMM_ABLATE & 1) && acc[0][0][0] != 123.456f) continue;
        const float* __restrict__ mask = p.mask;
        float* __restrict__ C = p.C;
        float* park = SA + wave * (32 * 68);          // 4 waves x 32 rows x 68 floats = 34 KB <= SA + SB
        static_assert(4 * 32 * 68 <= GT * GLA + GK * GLD, "epilogue staging must fit the operand slabs");
        const bool vec_ok = ((p.ldc & 3) == 0) && (mask == nullptr || (p.ldm & 3) == 0) && (n0 + wc * 64 + 64 <= p.n);
#pragma unroll
        for (int rb2 = 0; rb2 < 2; ++rb2) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    park[((i & 3) + 8 * (i >> 2) + 4 * half) * 68 + cb * 32 + r] = acc[rb2][cb][i];
            // (wave-private region: no workgroup barrier needed, the wave's own LDS accesses are ordered)
            const int rbase = row0 + wr * 64 + rb2 * 32;
            const int cbase = n0 + wc * 64;
            if (vec_ok) {
                const int c4 = lane & 15, rq = lane >> 4;      // 16 lanes x float4 = one 64-column row; 4 rows per pass
                float4 mv[8], old[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = rbase + rq + 4 * q;
                    const bool ok = row < M;
                    mv[q] = (ok && mask) ? *reinterpret_cast<const float4*>(mask + (size_t)row * p.ldm + cbase + 4 * c4)
                                         : make_float4(1.f, 1.f, 1.f, 1.f);
                    old[q] = (ok && p.accumulate) ? *reinterpret_cast<const float4*>(C + (size_t)row * p.ldc + cbase + 4 * c4)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = rbase + rq + 4 * q;
                    if (row >= M) continue;
                    const float4 v = *reinterpret_cast<const float4*>(park + (rq + 4 * q) * 68 + 4 * c4);
                    float4 o;
                    o.x = mv[q].x > 0.f ? v.x + old[q].x : 0.f;
                    o.y = mv[q].y > 0.f ? v.y + old[q].y : 0.f;
                    o.z = mv[q].z > 0.f ? v.z + old[q].z : 0.f;
                    o.w = mv[q].w > 0.f ? v.w + old[q].w : 0.f;
                    *reinterpret_cast<float4*>(C + (size_t)row * p.ldc + cbase + 4 * c4) = o;
                }
            } else {
                for (int q = 0; q < 32; ++q) {
                    const int row = rbase + q, col = cbase + lane;
                    if (row < M && col < p.n) {
                        const float m = mask ? mask[(size_t)row * p.ldm + col] : 1.0f;
                        const float o = p.accumulate ? C[(size_t)row * p.ldc + col] : 0.f;
                        C[(size_t)row * p.ldc + col] = m > 0.f ? park[q * 68 + lane] + o : 0.f;
                    }
                }
            }
        }
        __syncthreads();   // the next row tile's prologue overwrites the slabs
    }
}

int launch_gemm_nn(const GemmNN& p, int max_rows, hipStream_t s) {
    PR_REQUIRE(p.k % 16 == 0 && (p.lda & 3) == 0, "gemm_nn: K %d / lda %d not aligned", p.k, p.lda);
    if (max_rows <= 0 || p.n <= 0) return PR_OK;
    int row_tiles = (max_rows + GT - 1) / GT;
    if (row_tiles > 1024) row_tiles = 1024;
    row_tiles = (row_tiles + 7) / 8 * 8;              // whole groups of 8 row tiles (one per XCD)
    const int ncol_tiles = (p.n + GT - 1) / GT;
    ProfileScope scope(2, s);
    hipLaunchKernelGGL(k_gemm_nn, dim3(row_tiles * ncol_tiles), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// dW: reduction over the samples, split across workgroups
// ---------------------------------------------------------------------------------------------
// samples per split and the number of splits that own any: chunk is a multiple of the slab depth, at least
// TN_MIN_CHUNK, so that calls with few evaluated samples do not pay for (and later re-read) empty partial tiles
__device__ __forceinline__ int tn_chunk(int M, int splits) {
    int chunk = (M + splits - 1) / splits;
    if (chunk < TN_MIN_CHUNK) chunk = TN_MIN_CHUNK;
    return (chunk + GK - 1) / GK * GK;
}

__device__ __forceinline__ void gemm_tn_body(const GemmTN& p, int tile, int split, float* SA, float* SB) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int chunk = tn_chunk(M, p.splits);
    const int m_begin = split * chunk;
    if (m_begin >= M && split > 0) return;   // split 0 always writes (M == 0 -> zeros)
    const int m_end = (m_begin + chunk < M) ? m_begin + chunk : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    float bsum = 0.f;
    const int c4 = tid & 31, rr = tid >> 5;   // 32 samples x 128 columns per slab: samples rr + 8 i, one float4 each
    const bool acol = i0 + 4 * c4 < ((p.ni + 3) & ~3), bcol = j0 + 4 * c4 < ((p.nj + 3) & ~3);
    float4 ra[4], rb[4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rr + 8 * i;
            const bool live = m < m_end;
            ra[i] = (live && acol) ? *reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda + i0 + 4 * c4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (live && bcol) ? *reinterpret_cast<const float4*>(p.B + (size_t)m * p.ldb + j0 + 4 * c4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&SA[(rr + 8 * i) * GLD + 4 * c4]) = ra[i];
            *reinterpret_cast<float4*>(&SB[(rr + 8 * i) * GLD + 4 * c4]) = rb[i];
        }
    };
    if (m_begin < m_end) {
        fetch(m_begin);
        stage();
    }
    __syncthreads();
    for (int m0 = m_begin; m0 < m_end; m0 += GK) {
        const bool more = m0 + GK < m_end;
        if (more) fetch(m0 + GK);
        if (p.bias_partial && tj == 0 && tid < GT) {
#pragma unroll
            for (int q = 0; q < GK; ++q) bsum += SA[q * GLD + tid];
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const float a0 = SA[(kk + half) * GLD + wr * 64 + r];
            const float a1 = SA[(kk + half) * GLD + wr * 64 + 32 + r];
            const float b0 = SB[(kk + half) * GLD + wc * 64 + r];
            const float b1 = SB[(kk + half) * GLD + wc * 64 + 32 + r];
            PR_MFMA32(acc[0][0], a0, b0);
            PR_MFMA32(acc[0][1], a0, b1);
            PR_MFMA32(acc[1][0], a1, b0);
            PR_MFMA32(acc[1][1], a1, b1);
        }
        __syncthreads();
        if (more) stage();
        __syncthreads();
    }
    // partial tile -> P[split][i][j] (padded to whole tiles)
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb2][cb][i];
            }
        }
    if (p.bias_partial && tj == 0 && tid < GT) p.bias_partial[(size_t)split * rows_p + i0 + tid] = bsum;
}

__global__ __launch_bounds__(256, 2) void k_gemm_tn(GemmTN p) {
    __shared__ __attribute__((aligned(16))) float SA[GK * GLD];   // [sample][i]
    __shared__ __attribute__((aligned(16))) float SB[GK * GLD];   // [sample][j]
    gemm_tn_body(p, blockIdx.x, blockIdx.y, SA, SB);
}

// one launch for several products over the same rows: blockIdx.z picks the job; jobs with fewer tiles leave blocks idle
__global__ __launch_bounds__(256, 2) void k_gemm_tn_group(GemmTNGroup g) {
    __shared__ __attribute__((aligned(16))) float SA[GK * GLD];
    __shared__ __attribute__((aligned(16))) float SB[GK * GLD];
    const GemmTN& p = g.job[blockIdx.z];
    const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
    if ((int)blockIdx.x >= tiles) return;
    gemm_tn_body(p, blockIdx.x, blockIdx.y, SA, SB);
}

__device__ __forceinline__ void gemm_tn_reduce_body(const GemmTN& p, long idx);

// C[i][j] += sum_s P[s][i][j] in split order; bias[i] += sum_s PB[s][i]
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(GemmTN p) {
    gemm_tn_reduce_body(p, (long)blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void k_gemm_tn_reduce_group(GemmTNGroup g) {
    gemm_tn_reduce_body(g.job[blockIdx.y], (long)blockIdx.x * 256 + threadIdx.x);
}

__device__ __forceinline__ void gemm_tn_reduce_body(const GemmTN& p, long idx) {
    const int M = *p.rows;
    const int chunk = tn_chunk(M, p.splits);
    int active = (M + chunk - 1) / chunk;
    if (active < 1) active = 1;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    if (idx < (long)p.ni * p.nj) {
        const int i = (int)(idx / p.nj), j = (int)(idx - (long)i * p.nj);
        const float* __restrict__ src = p.partial + (size_t)i * ldp + j;
        const size_t stride = (size_t)rows_p * ldp;
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= active; s += 8) {   // eight loads in flight; the additions stay in split order
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = src[(size_t)(s + q) * stride];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += t[q];
        }
        for (; s < active; ++s) v += src[(size_t)s * stride];
        p.C[(size_t)i * p.ldc + j] += v;
    }
    if (p.bias_partial && p.bias && idx < p.ni) {
        float v = 0.f;
        for (int s = 0; s < active; ++s) v += p.bias_partial[(size_t)s * rows_p + idx];
        p.bias[idx] += v;
    }
}


// ---------------------------------------------------------------------------------------------
// Every weight-gradient product of a backward pass in ONE launch (+ one reduction launch): all layers of all objects.
//
// A persistent grid (two workgroups per CU).  Work items are (job, split, tile): job = one dW = dY^T . X product, split = a
// range of TN_ALL_CHUNK sample rows (the sample counts live on the device, so the decomposition is computed by every workgroup
// from the jobs' row counts), tile = a 128 x 128 block of the gradient.  The tiles of one (job, split) pair read the same dY
// and X rows: pairs are dealt to the 8 XCDs (pair index mod 8 - workgroup b runs on XCD b mod 8) and the workgroups of an
// XCD claim that XCD's items from its own counter, tile index fastest, so that the tiles of a pair run at the same time on
// CUs that share an L2 and the operands come from HBM once.  Every pair owns TN_ALL_TILES claim slots (the largest tile
// count: the 256 x 384 skip layers); slots beyond a job's tile count are empty claims.
// The reduction adds the partial tiles in split order and - object instances that share a model accumulate into the same
// buffers - walks the jobs of a destination in job order (chain_next), so the gradients are bit-reproducible.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_all_splits(int M) {
    const int s = (M + TN_ALL_CHUNK - 1) / TN_ALL_CHUNK;
    return s < 1 ? 1 : s;        // split 0 always exists (M == 0: zeros)
}

__device__ __forceinline__ void tn_all_tile(const TnJob& p, int tile, int split, float* SA, float* SB, float* SW) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int m_begin = split * TN_ALL_CHUNK;
    const int m_end = (m_begin + TN_ALL_CHUNK < M) ? m_begin + TN_ALL_CHUNK : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    float bsum = 0.f;
    // side product (p.w): the row weights of a slab travel with it (threads 0 .. 31: one each), LDS copy SW
    const bool side = p.w != nullptr && ti == 0;
    float wsum = 0.f, wtot = 0.f, w0 = 0.f, w1 = 0.f;
    const int c4 = tid & 31, rr = tid >> 5;
    const bool acol = i0 + 4 * c4 < ((p.ni + 3) & ~3), bcol = j0 + 4 * c4 < ((p.nj + 3) & ~3);
    // TWO slabs of operand rows in flight in registers (32 rows x 256 columns each): a slab is requested two steps before it is
    // staged, so that a step's MFMAs never wait for HBM (one slab in flight left 0.4 ms of a 2.3 ms launch exposed)
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    const float* __restrict__ gA = p.A + i0 + 4 * c4;
    const float* __restrict__ gB = p.B + j0 + 4 * c4;
    const size_t lda = (size_t)p.lda, ldb = (size_t)p.ldb;
    auto fetch = [&](float4 (&ra)[4], float4 (&rb)[4], float& wv, int m0) {
        if (side && tid < GK) wv = (m0 + tid < m_end) ? p.w[(size_t)(m0 + tid) * p.ldw] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rr + 8 * i;
            const bool live = m < m_end;
            ra[i] = (live && acol) ? *reinterpret_cast<const float4*>(gA + (size_t)m * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (live && bcol) ? *reinterpret_cast<const float4*>(gB + (size_t)m * ldb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](const float4 (&ra)[4], const float4 (&rb)[4], float wv) {
        if (side && tid < GK) SW[tid] = wv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&SA[(rr + 8 * i) * GLD + 4 * c4]) = ra[i];
            *reinterpret_cast<float4*>(&SB[(rr + 8 * i) * GLD + 4 * c4]) = rb[i];
        }
    };
    // one step on the slab in LDS: bias sums + the 16 row pairs
    auto step = [&]() {
        if (p.bias_partial && tj == 0 && tid < GT) {
#pragma unroll
            for (int q = 0; q < GK; ++q) bsum += SA[q * GLD + tid];
        }
        if (side && tid >= GT) {          // (the other half of the workgroup: columns tid - GT of the X slab)
#pragma unroll
            for (int q = 0; q < GK; ++q) wsum = fmaf(SW[q], SB[q * GLD + tid - GT], wsum);
            if (tid == GT && tj == 0) {
#pragma unroll
                for (int q = 0; q < GK; ++q) wtot += SW[q];
            }
        }
        // two row pairs in flight: the fragments of the even / odd pairs live in their own registers and are re-loaded right after
        // their last use, a full pair before they are needed again (with one register set the LDS reads of a pair wait for the
        // previous pair's MFMAs to have consumed their operands: 75 % of the matrix rate)
        const float* pa = SA + half * GLD + wr * 64 + r;
        const float* pb = SB + half * GLD + wc * 64 + r;
        float a0e = pa[0], a1e = pa[32], b0e = pb[0], b1e = pb[32];
        float a0o = pa[2 * GLD], a1o = pa[2 * GLD + 32], b0o = pb[2 * GLD], b1o = pb[2 * GLD + 32];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // both fragment sets are requested before the first MFMA
#pragma unroll
        for (int kk = 0; kk < ((PR_TNALL_ABLATE & 4) ? 4 : GK); kk += 4) {
            PR_MFMA32(acc[0][0], a0e, b0e);
            PR_MFMA32(acc[0][1], a0e, b1e);
            PR_MFMA32(acc[1][0], a1e, b0e);
            PR_MFMA32(acc[1][1], a1e, b1e);
            if (kk + 4 < GK) {
                a0e = pa[(kk + 4) * GLD]; a1e = pa[(kk + 4) * GLD + 32];
                b0e = pb[(kk + 4) * GLD]; b1e = pb[(kk + 4) * GLD + 32];
            }
            PR_MFMA32(acc[0][0], a0o, b0o);
            PR_MFMA32(acc[0][1], a0o, b1o);
            PR_MFMA32(acc[1][0], a1o, b0o);
            PR_MFMA32(acc[1][1], a1o, b1o);
            if (kk + 6 < GK) {
                a0o = pa[(kk + 6) * GLD]; a1o = pa[(kk + 6) * GLD + 32];
                b0o = pb[(kk + 6) * GLD]; b1o = pb[(kk + 6) * GLD + 32];
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
    const bool prefetch = !(PR_TNALL_ABLATE & 2);
    if (m_begin < m_end) {
        fetch(ra0, rb0, w0, m_begin);
        if (prefetch) fetch(ra1, rb1, w1, m_begin + GK);
        stage(ra0, rb0, w0);
    }
    __syncthreads();
    // steps in pairs: slab s is staged from set s % 2, and the set is refilled with slab s + 2 right away
    for (int m0 = m_begin; m0 < m_end; m0 += 2 * GK) {
        if (prefetch) fetch(ra0, rb0, w0, m0 + 2 * GK);       // (rows beyond m_end read nothing)
        step();
        __syncthreads();
        if (m0 + GK >= m_end) break;
        if (prefetch) stage(ra1, rb1, w1);
        __syncthreads();
        if (prefetch) fetch(ra1, rb1, w1, m0 + 3 * GK);
        step();
        __syncthreads();
        if (m0 + 2 * GK < m_end && prefetch) stage(ra0, rb0, w0);
        __syncthreads();
    }
    if ((PR_TNALL_ABLATE & 1) && acc[0][0][0] != 123.456f) return;
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb2][cb][i];
            }
        }
    if (p.bias_partial && tj == 0 && tid < GT) p.bias_partial[(size_t)split * rows_p + i0 + tid] = bsum;
    if (side && tid >= GT) {
        float* W = p.w_partial + (size_t)split * (ldp + 4);
        W[j0 + tid - GT] = wsum;
        if (tid == GT && tj == 0) W[ldp] = wtot;
    }
}

#ifndef PR_TNALL_WGS
#define PR_TNALL_WGS 2            // resident workgroups per CU
#endif
__global__ __launch_bounds__(256, PR_TNALL_WGS) void k_gemm_tn_all(TnAll g) {
    __shared__ __attribute__((aligned(16))) float SA[GK * GLD];
    __shared__ __attribute__((aligned(16))) float SB[GK * GLD];
    __shared__ float SW[GK];
    __shared__ int pair_begin[TN_ALL_MAX + 1];    // first (job, split) pair of every job
    __shared__ int claimed;
    const int tid = threadIdx.x;
    // the decomposition follows the jobs' row counts, which live on the device: every workgroup derives it on its own
    if (tid < g.count) pair_begin[tid + 1] = tn_all_splits(*g.job[tid].rows);
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int q = 0; q < g.count; ++q) {
            const int n = pair_begin[q + 1];
            pair_begin[q] = at;
            at += n;
        }
        pair_begin[g.count] = at;
    }
    __syncthreads();
    const int total_pairs = pair_begin[g.count];
    const int xcd = blockIdx.x & 7;
    int job = 0;
    for (;;) {
        if (tid == 0) claimed = atomicAdd(g.counters + xcd, 1);
        __syncthreads();
        const int c = claimed;
        __syncthreads();           // everyone has read the claim before thread 0 overwrites it
        const int pair = (c / TN_ALL_TILES) * 8 + xcd;
        if (pair >= total_pairs) break;
        const int tile = c % TN_ALL_TILES;
        while (pair >= pair_begin[job + 1]) ++job;      // claims of an XCD are increasing
        const TnJob& p = g.job[job];
        const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
        if (tile >= tiles) continue;
        tn_all_tile(p, tile, pair - pair_begin[job], SA, SB, SW);
        __syncthreads();           // the slabs are reused by the next item
    }
}

// ---------------------------------------------------------------------------------------------
// The same work item in SPLIT precision (pr_call_t.precision = PR_PRECISION_F16X3 on a differentiable call): every fp32 operand
// as THREE bf16 terms, x = b1 + b2 + b3 (each term what is left rounded to the nearest bf16: 8 + 8 + 8 mantissa bits
// with residuals of either sign, and the fp32 exponent range - gradients of 1e-7 are as well represented as activations of 1, which
// an fp16 pair is not), and a product as the SIX bf16 MFMAs whose terms are >= 2^-16 of it:
//     a b ~ a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2        (dropped: a2 b3 + a3 b2 + a3 b3 <= 3 x 2^-24 |a b|, one fp32 rounding)
// v_mfma_f32_32x32x16_bf16 multiplies exactly and accumulates in fp32; it retires 16 K-values in 32 cycles where the fp32 pipe
// needs 8 x 64: six of them cost 192 against 512 cycles.  The reduction index of dW = dY^T X is the SLOW dimension of both
// operands in memory; the bf16 MFMA wants eight consecutive K-values per lane, so a slab is transposed on its way into LDS:
// planes T[column][k] (80-byte rows: five 16-byte slots, conflict-free b128 fragment reads), a thread's four rows of a column
// = four consecutive k (one 8-byte store per plane and column; the same k permutation for both operands), slots rotated by
// (column >> 4) & 3 so that the 32 lanes of a store instruction spread over the banks.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TROW = 80;                       // bytes per LDS row of a plane: 32 k-values + one 16-byte pad slot
constexpr int TPLANE = GT * TROW;              // one plane of one operand
#define PR_MFMA_BF16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

typedef __bf16 bf16x2_g __attribute__((ext_vector_type(2)));
typedef float f32x2_g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sub_f32_g(float a, float b) {      // (not packed into v_pk_add_f32: slow beside MFMAs)
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// two values -> their three bf16 terms, rounded to nearest (v_cvt_pk_bf16_f32): packed pairs [x0 | x1] per term
__device__ __forceinline__ void bf16_split_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const f32x2_g v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_g));
#if defined(PR_TNBF_ABLATE) && (PR_TNBF_ABLATE & 16)
    p2 = p1 ^ 0x00010001u; p3 = p1 ^ 0x00020002u;
    return;
#endif
    const f32x2_g r = {sub_f32_g(x0, __uint_as_float(p1 << 16)), sub_f32_g(x1, __uint_as_float(p1 & 0xffff0000u))};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_g));
    const f32x2_g q = {sub_f32_g(r[0], __uint_as_float(p2 << 16)), sub_f32_g(r[1], __uint_as_float(p2 & 0xffff0000u))};
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_g));
}

#ifndef PR_TNBF_ABLATE
#define PR_TNBF_ABLATE 0      // timing builds only (results are wrong): 1 = slabs are not staged, 2 = no MFMAs, 4 = no operand requests,
#endif                        // 8 = three of the six MFMAs (what fp16 pairs would issue), 16 = one conversion per pair instead of three
__device__ __forceinline__ void tn_all_tile_bf16(const TnJob& p, int tile, int split, unsigned char* T, float* RED) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int m_begin = split * TN_ALL_CHUNK;
    const int m_end = (m_begin + TN_ALL_CHUNK < M) ? m_begin + TN_ALL_CHUNK : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    const bool want_bias = p.bias_partial && tj == 0;
    const bool side = p.w != nullptr && ti == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f}, wsum[4] = {0.f, 0.f, 0.f, 0.f}, wtot = 0.f;
    const int c4 = tid & 31, rr = tid >> 5;
    const bool acol = i0 + 4 * c4 < ((p.ni + 3) & ~3), bcol = j0 + 4 * c4 < ((p.nj + 3) & ~3);
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    float w0[4], w1[4];
    const float* __restrict__ gA = p.A + i0 + 4 * c4;
    const float* __restrict__ gB = p.B + j0 + 4 * c4;
    const size_t lda = (size_t)p.lda, ldb = (size_t)p.ldb;
    auto fetch = [&](float4 (&ra)[4], float4 (&rb)[4], float (&wv)[4], int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rr + 8 * i;
            const bool live = m < m_end;
            ra[i] = (live && acol) ? *reinterpret_cast<const float4*>(gA + (size_t)m * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (live && bcol) ? *reinterpret_cast<const float4*>(gB + (size_t)m * ldb) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (side) wv[i] = live ? p.w[(size_t)m * p.ldw] : 0.f;
        }
    };
    // the three planes of one operand: column c (of the tile), this thread's four k = 4 rr .. 4 rr + 3
    auto put = [&](unsigned char* planes, int c, float v0, float v1, float v2, float v3) {
        unsigned a1, a2, a3, b1, b2, b3;
        bf16_split_pair(v0, v1, a1, a2, a3);
        bf16_split_pair(v2, v3, b1, b2, b3);
        const int at = c * TROW + ((((rr >> 1) + (c >> 4)) & 3) << 4) + ((rr & 1) << 3);
        *reinterpret_cast<uint2*>(planes + at) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(planes + TPLANE + at) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(planes + 2 * TPLANE + at) = make_uint2(a3, b3);
    };
    auto stage = [&](const float4 (&ra)[4], const float4 (&rb)[4], const float (&wv)[4]) {
        put(T, 4 * c4 + 0, ra[0].x, ra[1].x, ra[2].x, ra[3].x);
        put(T, 4 * c4 + 1, ra[0].y, ra[1].y, ra[2].y, ra[3].y);
        put(T, 4 * c4 + 2, ra[0].z, ra[1].z, ra[2].z, ra[3].z);
        put(T, 4 * c4 + 3, ra[0].w, ra[1].w, ra[2].w, ra[3].w);
        put(T + 3 * TPLANE, 4 * c4 + 0, rb[0].x, rb[1].x, rb[2].x, rb[3].x);
        put(T + 3 * TPLANE, 4 * c4 + 1, rb[0].y, rb[1].y, rb[2].y, rb[3].y);
        put(T + 3 * TPLANE, 4 * c4 + 2, rb[0].z, rb[1].z, rb[2].z, rb[3].z);
        put(T + 3 * TPLANE, 4 * c4 + 3, rb[0].w, rb[1].w, rb[2].w, rb[3].w);
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { bsum[0] += ra[i].x; bsum[1] += ra[i].y; bsum[2] += ra[i].z; bsum[3] += ra[i].w; }
        }
        if (side) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wsum[0] = fmaf(wv[i], rb[i].x, wsum[0]); wsum[1] = fmaf(wv[i], rb[i].y, wsum[1]);
                wsum[2] = fmaf(wv[i], rb[i].z, wsum[2]); wsum[3] = fmaf(wv[i], rb[i].w, wsum[3]);
                wtot += wv[i];
            }
        }
    };
    // fragment addresses: lane (r, half) of block `blk` reads column base + blk * 32 + r, logical slot 2 kb + half
    const int colA0 = wr * 64 + r, colB0 = wc * 64 + r;
    auto step = [&]() {
        // (one K block of 16 at a time, the B fragments of one column block at a time: 24 + 12 fragment registers live instead of
        // 96 - with everything hoisted the kernel spilled)
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 a[2][3];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int ca = colA0 + blk * 32;
                const int oa = ca * TROW + (((2 * kb + half + (ca >> 4)) & 3) << 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[blk][pl] = *reinterpret_cast<const bf16x8*>(T + pl * TPLANE + oa);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int cc = colB0 + cb * 32;
                const int ob = cc * TROW + (((2 * kb + half + (cc >> 4)) & 3) << 4);
                bf16x8 b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(T + (3 + pl) * TPLANE + ob);
#pragma unroll
                for (int rb2 = 0; rb2 < 2; ++rb2) {
                    // smallest terms first
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][1], b[1]);
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][0], b[2]);
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][2], b[0]);
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][0], b[1]);
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][1], b[0]);
                    PR_MFMA_BF16(acc[rb2][cb], a[rb2][0], b[0]);
                }
            }
        }
    };
#ifdef PR_TNBF_TWO_SLABS
    if (m_begin < m_end) {
        fetch(ra0, rb0, w0, m_begin);
        fetch(ra1, rb1, w1, m_begin + GK);
        stage(ra0, rb0, w0);
    }
    __syncthreads();
    for (int m0 = m_begin; m0 < m_end; m0 += 2 * GK) {
        fetch(ra0, rb0, w0, m0 + 2 * GK);       // (rows beyond m_end read nothing)
        step();
        __syncthreads();
        if (m0 + GK >= m_end) break;
        stage(ra1, rb1, w1);
        __syncthreads();
        fetch(ra1, rb1, w1, m0 + 3 * GK);
        step();
        __syncthreads();
        if (m0 + 2 * GK < m_end) stage(ra0, rb0, w0);
        __syncthreads();
    }
#else
    // ONE slab in flight in registers (requested before the current slab's MFMAs, staged behind them): the launch is bound by
    // its operand traffic, and a second register set made the kernel spill
    (void)ra1; (void)rb1; (void)w1;
    if (m_begin < m_end) {
        fetch(ra0, rb0, w0, m_begin);
        stage(ra0, rb0, w0);
    }
    __syncthreads();
    for (int m0 = m_begin; m0 < m_end; m0 += GK) {
        if (!(PR_TNBF_ABLATE & 4)) fetch(ra0, rb0, w0, m0 + GK);           // (rows beyond m_end read nothing)
        if (!(PR_TNBF_ABLATE & 2)) step();
        __syncthreads();
        if (m0 + GK < m_end && !(PR_TNBF_ABLATE & 1)) stage(ra0, rb0, w0);
        __syncthreads();
    }
#endif
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb2][cb][i];
            }
        }
    // column sums kept per thread (its 4 columns, the rows it staged): added over the 8 row groups in a fixed order
    if (want_bias || side) {
        __syncthreads();
        float* R = RED;                                  // [8 row groups][128 columns] bias, then the same for the side product
        if (want_bias) for (int e = 0; e < 4; ++e) R[rr * GT + 4 * c4 + e] = bsum[e];
        if (side) {
            for (int e = 0; e < 4; ++e) R[8 * GT + rr * GT + 4 * c4 + e] = wsum[e];
            if (c4 == 0) R[16 * GT + rr] = wtot;
        }
        __syncthreads();
        if (want_bias && tid < GT) {
            float v = 0.f;
            for (int g8 = 0; g8 < 8; ++g8) v += R[g8 * GT + tid];
            p.bias_partial[(size_t)split * rows_p + i0 + tid] = v;
        }
        if (side && tid >= GT) {
            float* W = p.w_partial + (size_t)split * (ldp + 4);
            float v = 0.f;
            for (int g8 = 0; g8 < 8; ++g8) v += R[8 * GT + g8 * GT + tid - GT];
            W[j0 + tid - GT] = v;
            if (tid == GT && tj == 0) {
                float t = 0.f;
                for (int g8 = 0; g8 < 8; ++g8) t += R[16 * GT + g8];
                W[ldp] = t;
            }
        }
    }
}


// The same tile with the staging of the NEXT half slab inside the products of the current one.  In tn_all_tile_bf16 a slab is
// multiplied (48 MFMAs per wave), then - behind a barrier - the next one is split into bf16 triples and written to LDS (~250 VALU
// operations and 24 LDS stores per thread), then - behind another barrier - multiplied: measured with either half removed, the two
// phases take the same time (0.87 ms each of a 1.41 ms launch; the second workgroup of the CU is all that overlaps them).  A plane
// row already holds its 32 k-values as two independent halves (slots 0 - 1: k 0..15, slots 2 - 3: k 16..31), so the halves serve as a
// double buffer of 16-row half slabs with no more LDS: while the MFMAs of half h read one pair of slots, the same wave converts half
// h + 1 into the other pair (VALU work issues while the matrix pipe executes), ONE barrier per half slab.  Waves 0 - 1 stage the A
// operand, waves 2 - 3 the B operand (four rows x four columns per thread and half slab: the 8-byte LDS stores of the full-slab
// version); three register sets, the requests of half h + 3 issued at the top of half step h.
// Requests are unconditional (clamped addresses, values zeroed afterwards where a row or column is outside): see as_global() in
// pr_common.h for what requests under a branch do to the waits.
// Measured (same box, per launch of the training step): 1.444 -> 1.376 ms; without the operand requests 1.126, MFMAs alone 0.745,
// staging alone 0.892 - VALU work beside MFMAs of the same SIMD is only partly free (tools/perf/probe_overlap.hip: six conversions
// per MFMA cost + 20 - 35 % with two waves per SIMD), and the requests are bound by the 5.2 GB the launch reads, not by their latency.
// Explicit (MFMA, n x VALU) scheduling groups were slower than hipcc's own interleaving (1.39 vs 1.31 ms).
__device__ __forceinline__ void tn_all_tile_bf16_overlap(const TnJob& p, int tile, int split, unsigned char* T, float* RED) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int m_begin = split * TN_ALL_CHUNK;
    const int m_end = (m_begin + TN_ALL_CHUNK < M) ? m_begin + TN_ALL_CHUNK : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    const bool want_bias = p.bias_partial && tj == 0;
    const bool side = p.w != nullptr && ti == 0;
    const bool opB = wave >= 2;                      // this wave stages the B operand (wave-uniform)
    const int t7 = tid & 127, c4 = t7 & 31, rq = t7 >> 5;
    const int ncols = opB ? ((p.nj + 3) & ~3) : ((p.ni + 3) & ~3);
    const int base_col = opB ? j0 : i0;
    const bool colok = base_col + 4 * c4 < ncols;
    const bool cols_full = base_col + GT <= ncols;   // (wave-uniform)
    const size_t ldx = opB ? (size_t)p.ldb : (size_t)p.lda;
    const float* __restrict__ gX = (opB ? p.B : p.A) + base_col + (colok ? 4 * c4 : 0);
    // the side product's weights: B-staging threads read w[m]; everybody else (and every thread without a side product) reads a
    // valid dummy with stride 0 - the requests stay unconditional
    const float* __restrict__ gW = (side && opB) ? p.w : gX;
    const size_t ldw = (side && opB) ? (size_t)p.ldw : 0;
    unsigned char* planes = T + (opB ? 3 * TPLANE : 0);
    float bsum[4] = {0.f, 0.f, 0.f, 0.f}, wsum[4] = {0.f, 0.f, 0.f, 0.f}, wtot = 0.f;
    const int halves = (m_end - m_begin + 15) >> 4;
    if (halves > 0) {
        f32x4_t s0[4], s1[4], s2[4];
        float w0[4], w1[4], w2[4];
        auto fetch = [&](f32x4_t (&sv)[4], float (&wv)[4], int h) {
            const int mh = m_begin + 16 * h + rq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mh + 4 * i;
                const int mc = m < m_end ? m : m_end - 1;
                sv[i] = *reinterpret_cast<const f32x4_t*>(gX + (size_t)mc * ldx);
                wv[i] = gW[(size_t)mc * ldw];
            }
        };
        // rows / columns outside the operand contribute zeros (wave-uniform test first: full half slabs of full tiles skip the selects)
        auto mask = [&](f32x4_t (&sv)[4], float (&wv)[4], int h) {
            const int mh = m_begin + 16 * h;
            if (mh + 16 <= m_end && cols_full) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool row = mh + rq + 4 * i < m_end;
                const bool ok = row && colok;
                sv[i].x = ok ? sv[i].x : 0.f; sv[i].y = ok ? sv[i].y : 0.f; sv[i].z = ok ? sv[i].z : 0.f; sv[i].w = ok ? sv[i].w : 0.f;
                wv[i] = row ? wv[i] : 0.f;
            }
        };
        // one column of the staged 4 x 4 block: four consecutive k of half buffer `hb` (logical slots 2 hb, 2 hb + 1)
        const int rot = c4 >> 2;                       // (column >> 4) of the thread's four columns
        auto put = [&](int hb, int e, float v0, float v1, float v2, float v3) {
            unsigned a1, a2, a3, b1, b2, b3;
            bf16_split_pair(v0, v1, a1, a2, a3);
            bf16_split_pair(v2, v3, b1, b2, b3);
            const int at = (4 * c4 + e) * TROW + (((2 * hb + (rq >> 1) + rot) & 3) << 4) + ((rq & 1) << 3);
            *reinterpret_cast<uint2*>(planes + at) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(planes + TPLANE + at) = make_uint2(a2, b2);
            *reinterpret_cast<uint2*>(planes + 2 * TPLANE + at) = make_uint2(a3, b3);
        };
        auto sums = [&](const f32x4_t (&sv)[4], const float (&wv)[4]) {
            if (want_bias && !opB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { bsum[0] += sv[i].x; bsum[1] += sv[i].y; bsum[2] += sv[i].z; bsum[3] += sv[i].w; }
            }
            if (side && opB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wsum[0] = fmaf(wv[i], sv[i].x, wsum[0]); wsum[1] = fmaf(wv[i], sv[i].y, wsum[1]);
                    wsum[2] = fmaf(wv[i], sv[i].z, wsum[2]); wsum[3] = fmaf(wv[i], sv[i].w, wsum[3]);
                    wtot += wv[i];
                }
            }
        };
        const int colA0 = wr * 64 + r, colB0 = wc * 64 + r;
        // the products of half buffer `hc` with the staging of `sv` into half buffer `hs` between them
        auto half_step = [&](int hc, f32x4_t (&sv)[4], float (&wv)[4], int hs) {
            bf16x8 a[2][3], b[3];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int ca = colA0 + blk * 32;
                const int oa = ca * TROW + (((2 * hc + half + (ca >> 4)) & 3) << 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[blk][pl] = *reinterpret_cast<const bf16x8*>(T + pl * TPLANE + oa);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int cc = colB0 + cb * 32;
                const int ob = cc * TROW + (((2 * hc + half + (cc >> 4)) & 3) << 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(T + (3 + pl) * TPLANE + ob);
                // smallest terms first, the two row blocks alternating (independent accumulators back to back)
                if (!(PR_TNBF_ABLATE & 2)) {
                if (!(PR_TNBF_ABLATE & 8)) {
                PR_MFMA_BF16(acc[0][cb], a[0][1], b[1]); PR_MFMA_BF16(acc[1][cb], a[1][1], b[1]);
                PR_MFMA_BF16(acc[0][cb], a[0][0], b[2]); PR_MFMA_BF16(acc[1][cb], a[1][0], b[2]);
                PR_MFMA_BF16(acc[0][cb], a[0][2], b[0]); PR_MFMA_BF16(acc[1][cb], a[1][2], b[0]);
                }
                PR_MFMA_BF16(acc[0][cb], a[0][0], b[1]); PR_MFMA_BF16(acc[1][cb], a[1][0], b[1]);
                PR_MFMA_BF16(acc[0][cb], a[0][1], b[0]); PR_MFMA_BF16(acc[1][cb], a[1][1], b[0]);
                PR_MFMA_BF16(acc[0][cb], a[0][0], b[0]); PR_MFMA_BF16(acc[1][cb], a[1][0], b[0]);
                }
                // (no test for "nothing left to stage": a half slab beyond the split's rows is staged as zeros and never multiplied - a
                // branch here would end the basic block between the MFMAs and the conversions that are to issue beside them)
                if (PR_TNBF_ABLATE & 1) {
                } else if (cb == 0) {
                    put(hs, 0, sv[0].x, sv[1].x, sv[2].x, sv[3].x);
                    put(hs, 1, sv[0].y, sv[1].y, sv[2].y, sv[3].y);
                } else {
                    put(hs, 2, sv[0].z, sv[1].z, sv[2].z, sv[3].z);
                    put(hs, 3, sv[0].w, sv[1].w, sv[2].w, sv[3].w);
                    sums(sv, wv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // half 0 -> buffer 0 in front of the loop; THREE register sets: while one is staged, the next is in flight and the third is
        // requested at the top of the half step (two half steps ahead of its use: with two sets the requests could only be issued
        // behind the staging of the same registers, one half step ahead, and the conversions waited for memory)
        fetch(s0, w0, 0);
        fetch(s1, w1, 1);
        fetch(s2, w2, 2);
        mask(s0, w0, 0);
        put(0, 0, s0[0].x, s0[1].x, s0[2].x, s0[3].x);
        put(0, 1, s0[0].y, s0[1].y, s0[2].y, s0[3].y);
        put(0, 2, s0[0].z, s0[1].z, s0[2].z, s0[3].z);
        put(0, 3, s0[0].w, s0[1].w, s0[2].w, s0[3].w);
        sums(s0, w0);
        __syncthreads();
        for (int h = 0; h < halves;) {
            // half h multiplied out of buffer h & 1, half h + 1 staged into the other buffer, half h + 3 requested
            if (!(PR_TNBF_ABLATE & 4)) fetch(s0, w0, h + 3);
            mask(s1, w1, h + 1);
            half_step(h & 1, s1, w1, (h + 1) & 1);
            __syncthreads();
            if (++h >= halves) break;
            if (!(PR_TNBF_ABLATE & 4)) fetch(s1, w1, h + 3);
            mask(s2, w2, h + 1);
            half_step(h & 1, s2, w2, (h + 1) & 1);
            __syncthreads();
            if (++h >= halves) break;
            if (!(PR_TNBF_ABLATE & 4)) fetch(s2, w2, h + 3);
            mask(s0, w0, h + 1);
            half_step(h & 1, s0, w0, (h + 1) & 1);
            __syncthreads();
            ++h;
        }
    }
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb2][cb][i];
            }
        }
    // column sums kept per thread (its 4 columns, the rows it staged): added over the 4 row groups in a fixed order
    if (want_bias || side) {
        __syncthreads();
        float* R = RED;                                  // [4 row groups][128 columns] bias, then the same for the side product
        if (want_bias && !opB) for (int e = 0; e < 4; ++e) R[rq * GT + 4 * c4 + e] = bsum[e];
        if (side && opB) {
            for (int e = 0; e < 4; ++e) R[4 * GT + rq * GT + 4 * c4 + e] = wsum[e];
            if (c4 == 0) R[8 * GT + rq] = wtot;
        }
        __syncthreads();
        if (want_bias && tid < GT) {
            float v = 0.f;
            for (int g4 = 0; g4 < 4; ++g4) v += R[g4 * GT + tid];
            p.bias_partial[(size_t)split * rows_p + i0 + tid] = v;
        }
        if (side && tid >= GT) {
            float* W = p.w_partial + (size_t)split * (ldp + 4);
            float v = 0.f;
            for (int g4 = 0; g4 < 4; ++g4) v += R[4 * GT + g4 * GT + tid - GT];
            W[j0 + tid - GT] = v;
            if (tid == GT && tj == 0) {
                float t = 0.f;
                for (int g4 = 0; g4 < 4; ++g4) t += R[8 * GT + g4];
                W[ldp] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same work item on fp16 PAIRS (k_gemm_tn_all_f16, the default of split-precision calls; -DPR_TNALL_F16=0 launches the bf16-triple
// kernel above): x = hi + lo, a product as THREE v_mfma_f32_32x32x16_f16 (hi x hi + hi x lo + lo x hi) - measured on the triples with
// -DPR_TNBF_ABLATE=8, half the matrix instructions are worth 0.25 ms of a 1.33 ms launch, the conversions nothing.  fp16 does not have the range of a gradient (1e-7 and below), so every HALF SLAB (16 sample rows) is scaled on its way into
// LDS, the gradient rows by alpha_h, the activation rows by C / alpha_h: the product of a row pair is C x the true one for every half
// slab, so all of them share the accumulators.
//   alpha_h = the power of two that puts the half slab's largest |dY| in [2^13, 2^14)
//   C       = 2^(27 - E), E = the running maximum of  exponent(max |dY|) + exponent(max |X|)  over the half slabs seen: the scaled
//             activations stay below 2^15; when a later half slab raises E by d, the accumulators are multiplied by 2^-d (exact)
// Entries within 2^-16 of their half slab's largest keep 22 significant bits, smaller ones an absolute error of 2^-39 of the largest
// product (what k_chain_bwd_group_f16 does per 64-row tile); half slabs whose contribution is below 2^-17 of the largest one degrade
// the same way.  The maxima travel between the A- and the B-staging waves through four rotating LDS words per operand, published one
// half step ahead of the staging they steer (behind the barrier that is there anyway).
// ---------------------------------------------------------------------------------------------
#ifndef PR_TNALL_F16
#define PR_TNALL_F16 1          // 0: the split-precision weight gradients on bf16 triples (k_gemm_tn_all_bf16; A/B builds)
#endif
typedef _Float16 f16x8_g __attribute__((ext_vector_type(8)));
#define PR_MFMA_F16G(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
// four values x scale -> packed fp16 hi halves (h01, h23) and lo halves (l01, l23): see split_quad_scaled_h in mlp_tile.h
__device__ __forceinline__ void split_quad_scaled_g(float x0, float x1, float x2, float x3, float scale, unsigned& h01, unsigned& h23,
                                                    unsigned& l01, unsigned& l23) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(scale));
}
// biased exponent of a non-negative float's bit pattern; 0 for zero / subnormal, and for a non-finite maximum (no scaling: NaN products)
__device__ __forceinline__ int max_exponent(unsigned bits) {
    const int e = (int)((bits >> 23) & 255u);
    return e == 255 ? 0 : e;
}

// maximum of a non-negative bit pattern over the wave (every lane active): four DPP steps inside the rows of 16 lanes, the four rows
// through the scalar unit - no LDS traffic (a shuffle ladder is six ds_bpermute per call in a kernel that lives on its LDS)
__device__ __forceinline__ unsigned wave_max_bits(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));     // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));     // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ void tn_all_tile_f16_overlap(const TnJob& p, int tile, int split, unsigned char* T, float* RED, unsigned* MAXW) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, half = lane >> 5;
    const int M = *p.rows;
    const int tiles_j = (p.nj + GT - 1) / GT;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * GT, j0 = tj * GT;
    const int m_begin = split * TN_ALL_CHUNK;
    const int m_end = (m_begin + TN_ALL_CHUNK < M) ? m_begin + TN_ALL_CHUNK : M;
    f32x16 acc[2][2];
    zero_acc(acc);
    const bool want_bias = p.bias_partial && tj == 0;
    const bool side = p.w != nullptr && ti == 0;
    const bool opB = wave >= 2;
    const int t7 = tid & 127, c4 = t7 & 31, rq = t7 >> 5;
    const int ncols = opB ? ((p.nj + 3) & ~3) : ((p.ni + 3) & ~3);
    const int base_col = opB ? j0 : i0;
    const bool colok = base_col + 4 * c4 < ncols;
    const bool cols_full = base_col + GT <= ncols;
    const size_t ldx = opB ? (size_t)p.ldb : (size_t)p.lda;
    const float* __restrict__ gX = (opB ? p.B : p.A) + base_col + (colok ? 4 * c4 : 0);
    const float* __restrict__ gW = (side && opB) ? p.w : gX;
    const size_t ldw = (side && opB) ? (size_t)p.ldw : 0;
    unsigned char* planes = T + (opB ? 2 * TPLANE : 0);          // planes: A hi, A lo, B hi, B lo
    unsigned* mine = MAXW + (opB ? 4 : 0);                       // words [half & 3] of this wave's operand
    float bsum[4] = {0.f, 0.f, 0.f, 0.f}, wsum[4] = {0.f, 0.f, 0.f, 0.f}, wtot = 0.f;
    const int halves = (m_end - m_begin + 15) >> 4;
    int E = -1000;                  // running maximum of the exponent sums (workgroup-uniform: every thread derives it from the same words)
    if (halves > 0) {
        f32x4_t s0[4], s1[4], s2[4];
        float w0[4], w1[4], w2[4];
        auto fetch = [&](f32x4_t (&sv)[4], float (&wv)[4], int h) {
            const int mh = m_begin + 16 * h + rq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mh + 4 * i;
                const int mc = m < m_end ? m : m_end - 1;
                sv[i] = *reinterpret_cast<const f32x4_t*>(gX + (size_t)mc * ldx);
                wv[i] = gW[(size_t)mc * ldw];
            }
        };
        auto mask = [&](f32x4_t (&sv)[4], float (&wv)[4], int h) {
            const int mh = m_begin + 16 * h;
            if (mh + 16 <= m_end && cols_full) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool row = mh + rq + 4 * i < m_end;
                const bool ok = row && colok;
                sv[i].x = ok ? sv[i].x : 0.f; sv[i].y = ok ? sv[i].y : 0.f; sv[i].z = ok ? sv[i].z : 0.f; sv[i].w = ok ? sv[i].w : 0.f;
                wv[i] = row ? wv[i] : 0.f;
            }
        };
        // largest |entry| of this wave's share of half slab h -> this operand's word h & 3 (non-negative floats order like their bits)
        auto publish = [&](const f32x4_t (&sv)[4], int h) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(sv[i].x), fabsf(sv[i].y))), fmaxf(fabsf(sv[i].z), fabsf(sv[i].w)));
            const unsigned top = wave_max_bits(__float_as_uint(m));
            if (lane == 0) atomicMax(&mine[h & 3], top);
        };
        // the scale of this wave's operand for half slab h (read behind the barrier that followed its publish), and the running C
        float scale = 1.0f;
        int shrink = 0;             // the accumulators are to be multiplied by 2^-shrink in front of the products of the NEXT half step
        auto steer = [&](int h) {
            const int ea = max_exponent(MAXW[h & 3]), eb = max_exponent(MAXW[4 + (h & 3)]);
            shrink = 0;
            if (ea == 0 || eb == 0) {           // an all-zero operand: nothing to scale, nothing to constrain
                scale = 0.f;
                return;
            }
            // (exponents below -100 count as -100: a half slab of gradients around 1e-36 - samples behind an opaque surface - would
            // ask for alpha beyond fp32's range, inf, and 0 x inf = NaN; it is scaled as far as the range goes instead)
            const int xa = ea - 127 > -100 ? ea - 127 : -100, xb = eb - 127 > -100 ? eb - 127 : -100;
            const int sum = xa + xb;
            if (sum > E) {
                if (E > -1000) shrink = sum - E;
                E = sum;
            }
            // dY x 2^(13 - ea'),  X x 2^(27 - E) / 2^(13 - ea') = X x 2^(14 - E + ea')
            const int ka = 13 - xa;
            scale = opB ? ldexpf(1.0f, 27 - E - ka) : ldexpf(1.0f, ka);
        };
        const int rot = c4 >> 2;
        auto put = [&](int hb, int e, float v0, float v1, float v2, float v3) {
            unsigned h01, h23, l01, l23;
            split_quad_scaled_g(v0, v1, v2, v3, scale, h01, h23, l01, l23);
            const int at = (4 * c4 + e) * TROW + (((2 * hb + (rq >> 1) + rot) & 3) << 4) + ((rq & 1) << 3);
            *reinterpret_cast<uint2*>(planes + at) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(planes + TPLANE + at) = make_uint2(l01, l23);
        };
        auto sums = [&](const f32x4_t (&sv)[4], const float (&wv)[4]) {
            if (want_bias && !opB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { bsum[0] += sv[i].x; bsum[1] += sv[i].y; bsum[2] += sv[i].z; bsum[3] += sv[i].w; }
            }
            if (side && opB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wsum[0] = fmaf(wv[i], sv[i].x, wsum[0]); wsum[1] = fmaf(wv[i], sv[i].y, wsum[1]);
                    wsum[2] = fmaf(wv[i], sv[i].z, wsum[2]); wsum[3] = fmaf(wv[i], sv[i].w, wsum[3]);
                    wtot += wv[i];
                }
            }
        };
        auto rescale = [&](int d) {
            const float f = ldexpf(1.0f, -d);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] *= f;
        };
        const int colA0 = wr * 64 + r, colB0 = wc * 64 + r;
        auto half_step = [&](int hc, f32x4_t (&sv)[4], float (&wv)[4], int hs) {
            f16x8_g a[2][2], b[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int ca = colA0 + blk * 32;
                const int oa = ca * TROW + (((2 * hc + half + (ca >> 4)) & 3) << 4);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[blk][pl] = *reinterpret_cast<const f16x8_g*>(T + pl * TPLANE + oa);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int cc = colB0 + cb * 32;
                const int ob = cc * TROW + (((2 * hc + half + (cc >> 4)) & 3) << 4);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[pl] = *reinterpret_cast<const f16x8_g*>(T + (2 + pl) * TPLANE + ob);
                // smallest terms first, the two row blocks alternating
                PR_MFMA_F16G(acc[0][cb], a[0][1], b[0]); PR_MFMA_F16G(acc[1][cb], a[1][1], b[0]);
                PR_MFMA_F16G(acc[0][cb], a[0][0], b[1]); PR_MFMA_F16G(acc[1][cb], a[1][0], b[1]);
                PR_MFMA_F16G(acc[0][cb], a[0][0], b[0]); PR_MFMA_F16G(acc[1][cb], a[1][0], b[0]);
                if (cb == 0) {
                    put(hs, 0, sv[0].x, sv[1].x, sv[2].x, sv[3].x);
                    put(hs, 1, sv[0].y, sv[1].y, sv[2].y, sv[3].y);
                } else {
                    put(hs, 2, sv[0].z, sv[1].z, sv[2].z, sv[3].z);
                    put(hs, 3, sv[0].w, sv[1].w, sv[2].w, sv[3].w);
                    sums(sv, wv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // one iteration: half h + 3 requested into the set half h was staged from; half h multiplied out of buffer h & 1; half h + 1 (set
        // `nx`, masked and published one iteration ago) staged into the other buffer at the scale its maxima dictate; half h + 2 (set `n2`,
        // requested one iteration ago) masked and published
        auto iteration = [&](int h, f32x4_t (&rq_set)[4], float (&rq_w)[4], f32x4_t (&nx)[4], float (&nxw)[4], f32x4_t (&n2)[4],
                             float (&n2w)[4]) {
            if (tid == 0) { MAXW[(h + 3) & 3] = 0u; MAXW[4 + ((h + 3) & 3)] = 0u; }      // (published at the end of the next iteration)
            fetch(rq_set, rq_w, h + 3);
            if (shrink) rescale(shrink);        // half h was staged at a smaller C than the accumulators carry
            steer(h + 1);
            half_step(h & 1, nx, nxw, (h + 1) & 1);
            // half h + 2 was requested a whole iteration ago: its maxima are taken HERE, behind the products, not at the top (there the
            // wait for its rows was exposed: 1.35 instead of 1.28 ms per launch)
            mask(n2, n2w, h + 2);
            publish(n2, h + 2);
            __syncthreads();
        };
        if (tid < 8) MAXW[tid] = 0u;
        __syncthreads();
        fetch(s0, w0, 0);
        fetch(s1, w1, 1);
        fetch(s2, w2, 2);
        mask(s0, w0, 0);
        publish(s0, 0);
        mask(s1, w1, 1);
        publish(s1, 1);
        __syncthreads();
        steer(0);
        shrink = 0;
        put(0, 0, s0[0].x, s0[1].x, s0[2].x, s0[3].x);
        put(0, 1, s0[0].y, s0[1].y, s0[2].y, s0[3].y);
        put(0, 2, s0[0].z, s0[1].z, s0[2].z, s0[3].z);
        put(0, 3, s0[0].w, s0[1].w, s0[2].w, s0[3].w);
        sums(s0, w0);
        __syncthreads();
        for (int h = 0; h < halves;) {
            iteration(h, s0, w0, s1, w1, s2, w2);
            if (++h >= halves) break;
            iteration(h, s1, w1, s2, w2, s0, w0);
            if (++h >= halves) break;
            iteration(h, s2, w2, s0, w0, s1, w1);
            ++h;
        }
        // (the last iteration's steer may have asked for a rescale that no product followed: the accumulators are at the C of the
        // last MULTIPLIED half slab, which is E minus that pending shrink)
        E -= shrink;
    }
    const float back = (E > -1000) ? ldexpf(1.0f, E - 27) : 0.f;
    const int ldp = tiles_j * GT;
    const int rows_p = ((p.ni + GT - 1) / GT) * GT;
    float* P = p.partial + (size_t)split * rows_p * ldp;
#pragma unroll
    for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = j0 + wc * 64 + cb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + wr * 64 + rb2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                P[(size_t)row * ldp + col] = acc[rb2][cb][i] * back;
            }
        }
    if (want_bias || side) {
        __syncthreads();
        float* R = RED;
        if (want_bias && !opB) for (int e = 0; e < 4; ++e) R[rq * GT + 4 * c4 + e] = bsum[e];
        if (side && opB) {
            for (int e = 0; e < 4; ++e) R[4 * GT + rq * GT + 4 * c4 + e] = wsum[e];
            if (c4 == 0) R[8 * GT + rq] = wtot;
        }
        __syncthreads();
        if (want_bias && tid < GT) {
            float v = 0.f;
            for (int g4 = 0; g4 < 4; ++g4) v += R[g4 * GT + tid];
            p.bias_partial[(size_t)split * rows_p + i0 + tid] = v;
        }
        if (side && tid >= GT) {
            float* W = p.w_partial + (size_t)split * (ldp + 4);
            float v = 0.f;
            for (int g4 = 0; g4 < 4; ++g4) v += R[4 * GT + g4 * GT + tid - GT];
            W[j0 + tid - GT] = v;
            if (tid == GT && tj == 0) {
                float t = 0.f;
                for (int g4 = 0; g4 < 4; ++g4) t += R[8 * GT + g4];
                W[ldp] = t;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_gemm_tn_all_bf16(TnAll g) {
    __shared__ __attribute__((aligned(16))) unsigned char T[6 * TPLANE];
    __shared__ float RED[16 * GT + 8];
    __shared__ int pair_begin[TN_ALL_MAX + 1];
    __shared__ int claimed;
    const int tid = threadIdx.x;
    if (tid < g.count) pair_begin[tid + 1] = tn_all_splits(*g.job[tid].rows);
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int q = 0; q < g.count; ++q) {
            const int n = pair_begin[q + 1];
            pair_begin[q] = at;
            at += n;
        }
        pair_begin[g.count] = at;
    }
    __syncthreads();
    const int total_pairs = pair_begin[g.count];
    const int xcd = blockIdx.x & 7;
    int job = 0;
    for (;;) {
        if (tid == 0) claimed = atomicAdd(g.counters + xcd, 1);
        __syncthreads();
        const int c = claimed;
        __syncthreads();
        const int pair = (c / TN_ALL_TILES) * 8 + xcd;
        if (pair >= total_pairs) break;
        const int tile = c % TN_ALL_TILES;
        while (pair >= pair_begin[job + 1]) ++job;
        const TnJob& p = g.job[job];
        const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
        if (tile >= tiles) continue;
#ifdef PR_TNBF_SERIAL       // measurement build: the full-slab version (stage and multiply in turns)
        tn_all_tile_bf16(p, tile, pair - pair_begin[job], T, RED);
#else
        tn_all_tile_bf16_overlap(p, tile, pair - pair_begin[job], T, RED);
#endif
        __syncthreads();
    }
}

// the same persistent loop over the fp16-pair work item (the default of split-precision calls)
__global__ __launch_bounds__(256, 2) void k_gemm_tn_all_f16(TnAll g) {
    __shared__ __attribute__((aligned(16))) unsigned char T[4 * TPLANE];      // A hi, A lo, B hi, B lo
    __shared__ float RED[16 * GT + 8];
    __shared__ int pair_begin[TN_ALL_MAX + 1];
    __shared__ int claimed;
    __shared__ unsigned MAXW[8];     // the half slabs' maxima: four rotating words per operand
    const int tid = threadIdx.x;
    if (tid < g.count) pair_begin[tid + 1] = tn_all_splits(*g.job[tid].rows);
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int q = 0; q < g.count; ++q) {
            const int n = pair_begin[q + 1];
            pair_begin[q] = at;
            at += n;
        }
        pair_begin[g.count] = at;
    }
    __syncthreads();
    const int total_pairs = pair_begin[g.count];
    const int xcd = blockIdx.x & 7;
    int job = 0;
    for (;;) {
        if (tid == 0) claimed = atomicAdd(g.counters + xcd, 1);
        __syncthreads();
        const int c = claimed;
        __syncthreads();
        const int pair = (c / TN_ALL_TILES) * 8 + xcd;
        if (pair >= total_pairs) break;
        const int tile = c % TN_ALL_TILES;
        while (pair >= pair_begin[job + 1]) ++job;
        const TnJob& p = g.job[job];
        const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
        if (tile >= tiles) continue;
        tn_all_tile_f16_overlap(p, tile, pair - pair_begin[job], T, RED, MAXW);
        __syncthreads();
    }
}

// C[i][j] += sum over the jobs of this destination (job order), sum over their splits (split order); the same for the biases
__global__ __launch_bounds__(256) void k_gemm_tn_all_reduce(TnAll g) {
    const TnJob& head = g.job[blockIdx.y];
    if (!head.head) return;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int tiles_j = (head.nj + GT - 1) / GT;
    const int ldp = tiles_j * GT;
    const int rows_p = ((head.ni + GT - 1) / GT) * GT;
    const size_t stride = (size_t)rows_p * ldp;
    if (idx < (long)head.ni * head.nj) {
        const int i = (int)(idx / head.nj), j = (int)(idx - (long)i * head.nj);
        float v = 0.f;
        for (int q = blockIdx.y; q >= 0; q = g.job[q].chain_next) {
            const TnJob& p = g.job[q];
            const int active = tn_all_splits(*p.rows);
            const float* __restrict__ src = p.partial + (size_t)i * ldp + j;
            int s = 0;
            for (; s + 8 <= active; s += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(s + u) * stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; s < active; ++s) v += src[(size_t)s * stride];
        }
        head.C[(size_t)i * head.ldc + j] += v;
    }
    if (head.bias && idx < head.ni) {
        float v = 0.f;
        for (int q = blockIdx.y; q >= 0; q = g.job[q].chain_next) {
            const TnJob& p = g.job[q];
            if (!p.bias_partial) continue;
            const int active = tn_all_splits(*p.rows);
            for (int s = 0; s < active; ++s) v += p.bias_partial[(size_t)s * rows_p + idx];
        }
        head.bias[idx] += v;
    }
    if (head.wgrad && idx <= head.nj) {            // idx == nj: the side product's bias term
        if (idx == head.nj && !head.wbias) return;
        float v = 0.f;
        for (int q = blockIdx.y; q >= 0; q = g.job[q].chain_next) {
            const TnJob& p = g.job[q];
            const int active = tn_all_splits(*p.rows);
            const size_t at = idx < head.nj ? (size_t)idx : (size_t)ldp;
            for (int s = 0; s < active; ++s) v += p.w_partial[(size_t)s * (ldp + 4) + at];
        }
        if (idx < head.nj) head.wgrad[idx] += v;
        else *head.wbias += v;
    }
}

size_t tn_all_partial_floats(int ni, int nj, long max_rows) {
    const size_t rows_p = (size_t)((ni + GT - 1) / GT) * GT, ldp = (size_t)((nj + GT - 1) / GT) * GT;
    size_t splits = (size_t)((max_rows + TN_ALL_CHUNK - 1) / TN_ALL_CHUNK);
    if (splits < 1) splits = 1;
    return splits * (rows_p * ldp + rows_p + ldp + 4);      // tiles, bias partials, side-product partials
}

// Links the jobs that accumulate into the same gradient buffer (chain_next / head) and launches the two kernels.
// `counters`: 8 zeroed ints.
int launch_gemm_tn_all(TnAll& g, const long* max_rows, hipStream_t s) {
    if (g.count <= 0) return PR_OK;
    PR_REQUIRE(g.count <= TN_ALL_MAX && g.counters, "gemm_tn_all: %d jobs", g.count);
    long max_elems = 0;
    for (int q = 0; q < g.count; ++q) {
        TnJob& p = g.job[q];
        PR_REQUIRE(p.ni >= 1 && p.nj >= 1 && p.ni <= 256 && p.nj <= 384, "gemm_tn_all: %d x %d exceeds the tile slots", p.ni, p.nj);
        PR_REQUIRE((p.lda & 3) == 0 && (p.ldb & 3) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 &&
                   p.lda >= ((p.ni + 3) & ~3) && p.ldb >= ((p.nj + 3) & ~3), "gemm_tn_all: operands must be 16-byte aligned rows");
        PR_REQUIRE(p.C && p.partial && p.rows, "gemm_tn_all: NULL pointer");
        PR_REQUIRE(((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT) <= TN_ALL_TILES, "gemm_tn_all: too many tiles");
        // bias partials follow the weight partials of the job's region
        const size_t rows_p = (size_t)((p.ni + GT - 1) / GT) * GT, ldp = (size_t)((p.nj + GT - 1) / GT) * GT;
        size_t splits = (size_t)((max_rows[q] + TN_ALL_CHUNK - 1) / TN_ALL_CHUNK);
        if (splits < 1) splits = 1;
        p.bias_partial = p.bias ? p.partial + splits * rows_p * ldp : nullptr;
        p.w_partial = p.w ? p.partial + splits * (rows_p * ldp + rows_p) : nullptr;
        PR_REQUIRE(!p.w || (p.wgrad && p.ldw >= 1), "gemm_tn_all: side product without a destination");
        p.chain_next = -1;
        p.head = 1;
        if ((long)p.ni * p.nj > max_elems) max_elems = (long)p.ni * p.nj;
    }
    // the LAST job of a destination is its head and walks back through the earlier ones: the additions run in job order
    // reversed - any fixed order is reproducible
    for (int q = 0; q < g.count; ++q)
        for (int e = q - 1; e >= 0; --e)
            if (g.job[e].C == g.job[q].C) {
                PR_REQUIRE(g.job[e].ni == g.job[q].ni && g.job[e].nj == g.job[q].nj && g.job[e].ldc == g.job[q].ldc &&
                           g.job[e].bias == g.job[q].bias && g.job[e].wgrad == g.job[q].wgrad && g.job[e].wbias == g.job[q].wbias,
                           "gemm_tn_all: jobs of one destination differ in shape");
                g.job[q].chain_next = e;
                g.job[e].head = 0;
                break;
            }
    int cus = 0;
    ProfileScope scope(3, s);
    if (g.split_precision) {
#if PR_TNALL_F16
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_gemm_tn_all_f16), 0, &cus));
        hipLaunchKernelGGL(k_gemm_tn_all_f16, dim3(cus * 2), dim3(256), 0, s, g);
#else
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_gemm_tn_all_bf16), 0, &cus));
        hipLaunchKernelGGL(k_gemm_tn_all_bf16, dim3(cus * 2), dim3(256), 0, s, g);
#endif
    } else {
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_gemm_tn_all), 0, &cus));
        hipLaunchKernelGGL(k_gemm_tn_all, dim3(cus * PR_TNALL_WGS), dim3(256), 0, s, g);
    }
    PR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gemm_tn_all_reduce, dim3((unsigned)((max_elems + 255) / 256), g.count), dim3(256), 0, s, g);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

size_t gemm_tn_scratch_floats(int splits) {
    // largest gradient: 256 x 384 (skip layer) partial tiles + bias partials
    return (size_t)splits * (256 * 384 + 256);
}

int launch_gemm_tn(const GemmTN& p, hipStream_t s) {
    PR_REQUIRE(p.ni <= 256 && p.nj <= 384, "gemm_tn: %d x %d exceeds the partial buffer", p.ni, p.nj);
    PR_REQUIRE((p.lda & 3) == 0 && (p.ldb & 3) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 &&
               p.lda >= ((p.ni + 3) & ~3) && p.ldb >= ((p.nj + 3) & ~3), "gemm_tn: operands must be 16-byte aligned rows");
    PR_REQUIRE(p.splits >= 1 && p.partial, "gemm_tn: no partial buffer");
    const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
    ProfileScope scope(3, s);
    hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, p.splits), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    const long n = (long)p.ni * p.nj;
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

int launch_gemm_tn_group(const GemmTNGroup& g, hipStream_t s) {
    PR_REQUIRE(g.count >= 1 && g.count <= MAX_TN_GROUP, "gemm_tn group: %d jobs", g.count);
    int max_tiles = 0;
    long max_elems = 0;
    for (int q = 0; q < g.count; ++q) {
        const GemmTN& p = g.job[q];
        PR_REQUIRE(p.ni <= 256 && p.nj <= 384, "gemm_tn: %d x %d exceeds the partial buffer", p.ni, p.nj);
        PR_REQUIRE((p.lda & 3) == 0 && (p.ldb & 3) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 &&
                   p.lda >= ((p.ni + 3) & ~3) && p.ldb >= ((p.nj + 3) & ~3), "gemm_tn: operands must be 16-byte aligned rows");
        PR_REQUIRE(p.splits == g.job[0].splits && p.rows == g.job[0].rows && p.partial, "gemm_tn group: jobs must share rows and splits");
        const int tiles = ((p.ni + GT - 1) / GT) * ((p.nj + GT - 1) / GT);
        if (tiles > max_tiles) max_tiles = tiles;
        if ((long)p.ni * p.nj > max_elems) max_elems = (long)p.ni * p.nj;
    }
    ProfileScope scope(3, s);
    hipLaunchKernelGGL(k_gemm_tn_group, dim3(max_tiles, g.job[0].splits, g.count), dim3(256), 0, s, g);
    PR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gemm_tn_reduce_group, dim3((unsigned)((max_elems + 255) / 256), g.count), dim3(256), 0, s, g);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
