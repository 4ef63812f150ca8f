// Fused tile kernels of the training step's backward pass (the reference: torch.autograd through
// model/nerf_models/adain_style_nerf_model.py:106-145, model/layers/adain.py:39-61, positional_ray_bender_model.py:81-163).
//
// A persistent workgroup of 4 waves owns a tile of 64 evaluated samples, the forward kernel's tile (mlp_tile.h): the running
// gradient lives in LDS as X[64][260] and every product is  out = X . W^T-fragments  on the fp32 matrix cores.  The objects
// of a call share the launches (up to four job slots per launch; every tile is claimed from the job's counter), so that a
// small object neither pays launches of its own nor leaves the chip idle behind a large one.
//
//   k_head_bwd_group    phase 1: g_feat . W6 -> AdaIN / ReLU / normalisation backward of head layer 4 -> d x_hat (+ batch sums)
//                       phase 2: BatchNorm backward (batch terms from phase 1's sums) -> . W3 -> the same for head layer 1
//   k_chain_bwd_group   NeRF:   BatchNorm backward -> . W0 (+ sigma head) -> ReLU mask -> the backbone chain, layer by layer
//                       bender: output head -> ReLU mask -> the bender chain
//
// Every pre-activation gradient is written once (the left factor of that layer's weight-gradient product, k_gemm_tn_all),
// next to the recomputed post-AdaIN activations (the right factors of the head layers).
#include "pr_common.h"
#include "mlp_tile.h"

namespace pr {

struct BSmem {
    float X[TILE_M * LDX];                          // the running gradient / the product's operand
    float cst[4][MAX_WIDTH];                        // BatchNorm backward on load: mean, rstd, mean(dxh), mean(dxh xh) per channel
    float ws[MAX_WIDTH];                            // sigma head weights / rows 0..2: bender output head (3 x BWpad <= 3 x 128 ... see use)
    float gsr[TILE_M];                              // d loss / d sigma of the tile rows
    int flat[TILE_M], frame[TILE_M], flags[TILE_M];
    int uniform_frame, next_tile;
    int tile_max[2];                                // fp16-pair chains: bit patterns of the largest |entry| of the tile in X (two alternating words)
#ifdef PR_CHAIN_TIMING
    unsigned long long phase_acc[16];               // phase timing build: thread 0's clock deltas (flushed once per chain_bwd_loop)
#endif
};
static_assert(sizeof(BSmem) * MLP_BLOCKS_PER_CU <= 160 * 1024 - MLP_BLOCKS_PER_CU * 1024, "two backward tiles per CU");
static_assert(offsetof(BSmem, cst) % 16 == 0 && offsetof(BSmem, ws) % 16 == 0, "16-byte LDS accesses");

#ifndef PR_HEADB_ABLATE
#define PR_HEADB_ABLATE 0       // timing builds only (k_head_bwd_group): 1 = no raw-activation prefetch, 2 = no a_out stores, 4 = no d_out rows,
#endif                          // 8 = no MFMA, 16 = no operand load from memory, 32 = no statistics atomics
#ifndef PR_CHAINGRP_ABLATE
#define PR_CHAINGRP_ABLATE 0    // timing builds only (k_chain_bwd_group): 1 = no gradient write-out, 2 = no mask bits, 4 = no MFMA
#endif
#ifdef PR_CHAIN_TIMING
// phase timing build: thread 0 of every workgroup accumulates shader-clock deltas per phase of k_chain_bwd_group
__device__ unsigned long long g_chain_phase[16];
#define PR_CT0() unsigned long long _ct = __builtin_amdgcn_s_memtime()
#define PR_CT(idx)                                                                         \
    do {                                                                                   \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                        \
        if (threadIdx.x == 0) S.phase_acc[idx] += _n - _ct;    /* (LDS: see PR_PHASE in mlp_tile.h) */ \
        _ct = _n;                                                                          \
    } while (0)
#else
#define PR_CT0() do {} while (0)
#define PR_CT(idx) do {} while (0)
#endif
#ifdef PR_HEAD_TIMING
// phase timing build: thread 0 of every workgroup accumulates shader-clock deltas per phase of k_head_bwd_group (slots 0..7 phase 1,
// 8..15 phase 2 of the head backward)
__device__ unsigned long long g_head_phase[16];
#define PR_HT0() unsigned long long _ht = __builtin_amdgcn_s_memtime()
#define PR_HT(idx)                                                                         \
    do {                                                                                   \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                        \
        if (threadIdx.x == 0) atomicAdd(&g_head_phase[(idx) + (p.phase == 2 ? 8 : 0)], _n - _ht); \
        _ht = _n;                                                                          \
    } while (0)
#else
#define PR_HT0() do {} while (0)
#define PR_HT(idx) do {} while (0)
#endif
#define PR_ROWS_OF(i, half, rb) (PR_ACC_ROW(i) + 4 * (half) + 32 * (rb))

__device__ __forceinline__ void zero4(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = b[i] = c[i] = d[i] = 0.f;
}

// The K loop of one fragment-ordered segment over the operand tile in X (run_layer's loop, mlp_tile.h: two steps in flight,
// even / odd fragments in their own registers).  a0x: column block `wave`, a1x: column block `wave + 4`; x0 / x1: row blocks.
// ``drain``: the operand tile in X is also WRITTEN OUT to global memory while the product runs - one 16-byte chunk per thread and
// loop iteration (a tile of width Wpad has Wpad / 16 chunks per thread and the loop Wpad / 16 iterations), so that the 64 KB of a
// tile reach the memory system spread over the K loop instead of as one burst in front of it.
// every wave runs the loop (so every thread drains its chunks) when the product has at least MLP_WAVES column blocks
__device__ __forceinline__ bool drains_in_loop(int nblk) {
#ifdef PR_NO_DRAIN
    return false;
#else
    return nblk >= MLP_WAVES;
#endif
}

// DRAIN is a template parameter: tested inside the loop (a pointer compare and a branch) it ended the basic block in front of the
// scheduling groups, and hipcc issued all 32 MFMAs of an iteration first and the operand requests of the next one behind them
template <bool DRAIN>
__device__ __forceinline__ void tile_products_loop(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                                   f32x16& a11, const Drain* drain) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int cbA = wave, cbB = wave + MLP_WAVES;
    if (cbA >= nblk) return;
    const bool two = cbB < nblk;
    if (PR_CHAINGRP_ABLATE & 4) return;
    __builtin_amdgcn_s_setprio(1);
    const int kq = sg.kq;
    const float* ap = X + r * LDX + half * 4 * kq;
    const float4* wpA = reinterpret_cast<const float4*>(sg.w) + (size_t)cbA * kq * 64 + lane;
    float4 x0e = *reinterpret_cast<const float4*>(ap);
    float4 x1e = *reinterpret_cast<const float4*>(ap + 32 * LDX);
    float4 x0o = *reinterpret_cast<const float4*>(ap + 4);
    float4 x1o = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4);
    if (two) {
        // (first requests in the loop's order - even A, even B, odd A, odd B: see run_layer in mlp_tile.h)
        const float4* wpB = reinterpret_cast<const float4*>(sg.w) + (size_t)cbB * kq * 64 + lane;
        float4 wAe = wpA[0], wBe = wpB[0];
        __builtin_amdgcn_sched_barrier(0);
        float4 wAo = wpA[64], wBo = wpB[64];
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < kq; q += 2) {
            const int qe = (q + 2 < kq) ? q + 2 : q, qo = (q + 3 < kq) ? q + 3 : q + 1;
            PR_MFMA4(a00, x0e, wAe);
            PR_MFMA4(a01, x1e, wAe);
            PR_MFMA4(a10, x0e, wBe);
            PR_MFMA4(a11, x1e, wBe);
            wAe = wpA[(size_t)qe * 64];
            wBe = wpB[(size_t)qe * 64];
            x0e = *reinterpret_cast<const float4*>(ap + 4 * qe);
            x1e = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4 * qe);
            PR_MFMA4(a00, x0o, wAo);
            PR_MFMA4(a01, x1o, wAo);
            PR_MFMA4(a10, x0o, wBo);
            PR_MFMA4(a11, x1o, wBo);
            wAo = wpA[(size_t)qo * 64];
            wBo = wpB[(size_t)qo * 64];
            x0o = *reinterpret_cast<const float4*>(ap + 4 * qo);
            x1o = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4 * qo);
            if (DRAIN) drain_chunk(*drain, X, q >> 1);
            // (the drained chunk: read from LDS with the even operands, stored behind the odd MFMAs and IN FRONT of the odd requests -
            // as the last request of the iteration, the store stood between the even fragments and the top of the next iteration in
            // the request queue, and the wait for the even B fragment also waited for the odd A fragment requested a moment before)
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, DRAIN ? 3 : 2, 0);
            if (DRAIN) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    } else {
        float4 wAe = wpA[0];
        __builtin_amdgcn_sched_barrier(0);
        float4 wAo = wpA[64];
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < kq; q += 2) {
            const int qe = (q + 2 < kq) ? q + 2 : q, qo = (q + 3 < kq) ? q + 3 : q + 1;
            PR_MFMA4(a00, x0e, wAe);
            PR_MFMA4(a01, x1e, wAe);
            wAe = wpA[(size_t)qe * 64];
            x0e = *reinterpret_cast<const float4*>(ap + 4 * qe);
            x1e = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4 * qe);
            PR_MFMA4(a00, x0o, wAo);
            PR_MFMA4(a01, x1o, wAo);
            wAo = wpA[(size_t)qo * 64];
            x0o = *reinterpret_cast<const float4*>(ap + 4 * qo);
            x1o = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4 * qo);
            if (DRAIN) drain_chunk(*drain, X, q >> 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, DRAIN ? 3 : 2, 0);
            if (DRAIN) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void tile_products(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                              f32x16& a11, const Drain* drain = nullptr) {
    if (drain) tile_products_loop<true>(sg, nblk, X, a00, a01, a10, a11, drain);
    else tile_products_loop<false>(sg, nblk, X, a00, a01, a10, a11, nullptr);
}

// MODE 0: exact fp32; 1: bf16 triples; 2: fp16 pairs of the operand x `scale` (chain_bwd_loop: per-tile power-of-two scales)
template <int MODE>
__device__ __forceinline__ void tile_products_any(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                                  f32x16& a11, const Drain* drain = nullptr, float scale = 1.0f) {
    if (MODE == 2) tile_products_f16x3(sg, nblk, X, a00, a01, a10, a11, drain, scale);
    else if (MODE == 1) tile_products_bf16(sg, nblk, X, a00, a01, a10, a11, drain);
    else tile_products(sg, nblk, X, a00, a01, a10, a11, drain);
}

// (commit_tile_max / tile_scale_log2: mlp_tile.h - every producer of a gradient tile folds what it writes into one of two LDS words)

// rows of X -> rows of a (cap, ld) array, 16-byte stores; only the tile's real rows
__device__ __forceinline__ void store_tile_rows(const float* X, float* dst, int width_pad, int ld, int tile_base, int rows_valid) {
    const int w4 = width_pad >> 2;
    for (int idx = threadIdx.x; idx < TILE_M * w4; idx += MLP_THREADS) {
        const int row = idx / w4, c = (idx - row * w4) * 4;
        if (row < rows_valid) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(X + row * LDX + c);
#ifdef PR_STORE_ROWS_NT      // measurement build (no difference for the head phases' rows)
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(dst + (size_t)(tile_base + row) * ld + c));
#else
            *reinterpret_cast<f32x4_t*>(dst + (size_t)(tile_base + row) * ld + c) = v;
#endif
        }
    }
}

// The saved ReLU mask of this lane's columns for one product: the forward pass left one 64-bit word per column and tile (bit r =
// tile row r active), so a lane fetches two words - its columns of the blocks `wave` and `wave + 4` - before the K loop and has
// them when the epilogue needs them (no LDS image, no load latency behind the product).
struct ColMasks { unsigned long long a, b; };
__device__ __forceinline__ ColMasks fetch_col_masks(const unsigned char* bits_layer, int width_pad, int nblk, int tile) {
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 31;
    const unsigned long long* words = reinterpret_cast<const unsigned long long*>(bits_layer) + (size_t)tile * width_pad;
    ColMasks m;
    m.a = wave < nblk ? words[wave * 32 + r] : 0ull;
    m.b = wave + MLP_WAVES < nblk ? words[(wave + MLP_WAVES) * 32 + r] : 0ull;
    return m;
}

// ReLU backward of a product: X[row][col] = mask bit ? acc : 0  (the callers put barriers around it)
__device__ __forceinline__ void store_masked(BSmem& S, int nblk, const ColMasks& masks, const f32x16& a00, const f32x16& a01, const f32x16& a10,
                                             const f32x16& a11, int* max_slot = nullptr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    float biggest = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int cb = wave + blk * MLP_WAVES;
        if (cb >= nblk) continue;
        const int col = cb * 32 + r;
        const f32x16& lo = blk ? a10 : a00;
        const f32x16& hi = blk ? a11 : a01;
        float* x0 = S.X + (4 * half) * LDX + col;
        unsigned long long mine = (blk ? masks.b : masks.a) >> (4 * half);     // bit ro <-> tile row ro + 4 half
        if (PR_CHAINGRP_ABLATE & 2) mine = ~0ull;
        const unsigned int mlo = (unsigned int)mine, mhi = (unsigned int)(mine >> 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ro = PR_ACC_ROW(i);
            const float v0 = ((mlo >> ro) & 1u) ? lo[i] : 0.f, v1 = ((mhi >> ro) & 1u) ? hi[i] : 0.f;
            x0[ro * LDX] = v0;
            x0[(ro + 32) * LDX] = v1;
            biggest = fmaxf(biggest, fmaxf(fabsf(v0), fabsf(v1)));
        }
    }
    commit_tile_max(biggest, max_slot);
}

// a product's rows straight to global memory (the gradient of a network input: X keeps its operand)
__device__ __forceinline__ void store_global(float* gout, int ldg, int n_real, int nblk, int tile_base, int rows_valid, bool accumulate,
                                             const f32x16& a00, const f32x16& a01, const f32x16& a10, const f32x16& a11) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    for (int blk = 0; blk < 2; ++blk) {
        const int cb = wave + blk * MLP_WAVES;
        if (cb >= nblk) break;
        const int col = cb * 32 + r;
        if (col >= n_real) continue;
        const f32x16& lo = blk ? a10 : a00;
        const f32x16& hi = blk ? a11 : a01;
        float* base = gout + (size_t)(tile_base + 4 * half) * ldg + col;
        const int limit = rows_valid - 4 * half;
        float old_lo[16], old_hi[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            old_lo[i] = (accumulate && PR_ACC_ROW(i) < limit) ? base[PR_ACC_ROW(i) * ldg] : 0.f;
            old_hi[i] = (accumulate && PR_ACC_ROW(i) + 32 < limit) ? base[(PR_ACC_ROW(i) + 32) * ldg] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PR_ACC_ROW(i) < limit) base[PR_ACC_ROW(i) * ldg] = lo[i] + old_lo[i];
            if (PR_ACC_ROW(i) + 32 < limit) base[(PR_ACC_ROW(i) + 32) * ldg] = hi[i] + old_hi[i];
        }
    }
}

// per-channel constants of a BatchNorm backward -> S.cst: mean, 1 / sqrt(var + eps), mean(dxh), mean(dxh xh) (0 when frozen)
__device__ __forceinline__ void stage_bn_constants(BSmem& S, const float* mean, const float* var, const double* sums, int width,
                                                   int width_pad, const int32_t* count, float eps, int frozen) {
    const double n = (double)*count;
    for (int c = threadIdx.x; c < width_pad; c += MLP_THREADS) {
        const bool live = c < width;
        S.cst[0][c] = live ? mean[c] : 0.f;
        S.cst[1][c] = live ? 1.0f / sqrtf(var[c] + eps) : 0.f;
        S.cst[2][c] = (live && !frozen) ? (float)(sums[c] / n) : 0.f;
        S.cst[3][c] = (live && !frozen) ? (float)(sums[width_pad + c] / n) : 0.f;
    }
}

// BatchNorm (batch statistics) backward of the tile while it is loaded: dh = rstd (dxh - mean(dxh) - xh mean(dxh xh)) for the rows
// that entered the statistics, 0 otherwise -> X, and back to `d` in place (the left factor of the layer's weight gradient)
__device__ __forceinline__ void load_bn_backward(BSmem& S, float* d, const float* h, int width_pad, int tile_base, int rows_valid,
                                                 int* max_slot = nullptr) {
    // Batches of LOADS first, then the arithmetic and the stores: written as one loop (load d, load h, compute, store d back) every
    // iteration waited for its own loads - the store to `d` may alias the next iteration's load for all the compiler knows - and a
    // tile's 8 - 16 iterations cost as many round trips to memory (21 - 28 % of the head backward's time, tools/perf/perf_head_phases.py).
    constexpr int BATCH = 8;       // (a 128-wide tile in one batch: 16 float4 in flight per thread; nothing else is live here)
    const int w4 = width_pad >> 2;
    const int count = TILE_M * w4;                 // a multiple of MLP_THREADS * BATCH (width_pad is a multiple of 64 here or handled by the guard)
    float biggest = 0.f;
    for (int base = threadIdx.x; base < count; base += MLP_THREADS * BATCH) {
        float4 g[BATCH], hv[BATCH];
        bool live[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int idx = base + b * MLP_THREADS;
            const int row = idx / w4, c = (idx - row * w4) * 4;
            live[b] = idx < count && row < rows_valid && (S.flags[row] & 3) == 3;
            const size_t at = (size_t)(tile_base + (live[b] ? row : 0)) * width_pad + (live[b] ? c : 0);
            g[b] = live[b] ? *reinterpret_cast<const float4*>(d + at) : make_float4(0.f, 0.f, 0.f, 0.f);
            hv[b] = live[b] ? *reinterpret_cast<const float4*>(h + at) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int idx = base + b * MLP_THREADS;
            if (idx >= count) continue;
            const int row = idx / w4, c = (idx - row * w4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[b]) {
                const float4 mu = *reinterpret_cast<const float4*>(&S.cst[0][c]);
                const float4 rs = *reinterpret_cast<const float4*>(&S.cst[1][c]);
                const float4 m1 = *reinterpret_cast<const float4*>(&S.cst[2][c]);
                const float4 m2 = *reinterpret_cast<const float4*>(&S.cst[3][c]);
                v.x = rs.x * (g[b].x - m1.x - (hv[b].x - mu.x) * rs.x * m2.x);
                v.y = rs.y * (g[b].y - m1.y - (hv[b].y - mu.y) * rs.y * m2.y);
                v.z = rs.z * (g[b].z - m1.z - (hv[b].z - mu.z) * rs.z * m2.z);
                v.w = rs.w * (g[b].w - m1.w - (hv[b].w - mu.w) * rs.w * m2.w);
            }
            if (row < rows_valid) *reinterpret_cast<float4*>(d + (size_t)(tile_base + row) * width_pad + c) = v;
            *reinterpret_cast<float4*>(S.X + row * LDX + c) = v;
            biggest = fmaxf(fmaxf(biggest, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    commit_tile_max(biggest, max_slot);
}

// records of the tile: flat sample index, frame, row flags (0 beyond the real rows)
__device__ __forceinline__ void load_tile_records(BSmem& S, const int32_t* rec_flat, const int32_t* row_flags, int samples_per_frame,
                                                  int tile_base, int total) {
    const int tid = threadIdx.x;
    if (tid == 0) S.uniform_frame = 1;
    if (tid < TILE_M) {
        const int idx = tile_base + tid;
        const bool valid = idx < total;
        const int flat = rec_flat[valid ? idx : tile_base];
        S.flat[tid] = flat;
        S.frame[tid] = flat / samples_per_frame;
        S.flags[tid] = valid ? row_flags[idx] : 0;
    }
}

// running sums of one lane's columns across the tiles of a workgroup
struct ColumnSums {
    double s1[2], s2[2];     // sum dxh, sum dxh xh  (column blocks A / B)
    float ds[2], db[2];      // d scale, d bias of `frame`
    int frame;
};

__device__ __forceinline__ void flush_frame_sums(ColumnSums& cs, float* dscale, float* dbias, int nblk, int width) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (cs.frame >= 0 && lane < 32) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int col = (wave + blk * MLP_WAVES) * 32 + lane;
            if (wave + blk * MLP_WAVES < nblk && col < width) {
                atomicAdd(dscale + (size_t)cs.frame * MAX_WIDTH + col, cs.ds[blk]);
                atomicAdd(dbias + (size_t)cs.frame * MAX_WIDTH + col, cs.db[blk]);
            }
        }
    }
    cs.ds[0] = cs.ds[1] = cs.db[0] = cs.db[1] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// Feature-head backward, phases 1 and 2
// ---------------------------------------------------------------------------------------------
template <int SPLIT>
__device__ __forceinline__ void head_bwd_loop(const HeadBwdJob& p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BSmem& S = *reinterpret_cast<BSmem*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, half = lane >> 5;
    const int total = *p.total;
    __syncthreads();       // every wave has left the previous job's last tile
    if (tid == 0) S.next_tile = atomicAdd(p.tile_counter, 1);
    if (p.phase == 2) stage_bn_constants(S, p.mean_in, p.var_in, p.sums_in, p.width_in, p.kpad, p.stat_count, p.eps, p.frozen);
    // this lane's columns of the layer that is differentiated: statistics of its BatchNorm
    float mu[2], rstd[2];
    bool live_col[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int col = (wave + blk * MLP_WAVES) * 32 + r;
        live_col[blk] = (wave + blk * MLP_WAVES) < p.nblk && col < p.width;
        mu[blk] = live_col[blk] ? p.mean[col] : 0.f;
        rstd[blk] = live_col[blk] ? 1.0f / sqrtf(p.var[col] + p.eps) : 0.f;
    }
    ColumnSums cs;
    cs.s1[0] = cs.s1[1] = cs.s2[0] = cs.s2[1] = 0.0;
    cs.ds[0] = cs.ds[1] = cs.db[0] = cs.db[1] = 0.f;
    cs.frame = -1;
    __syncthreads();
    for (int tile = S.next_tile; tile * TILE_M < total; tile = S.next_tile) {
        const int tile_base = tile * TILE_M;
        const int rows_valid = (total - tile_base < TILE_M) ? total - tile_base : TILE_M;
        int claimed = 0;      // (claimed late, behind the tile's product: see "Tile order" in mlp.hip)
        PR_HT0();
        load_tile_records(S, p.rec_flat, p.row_flags, p.samples_per_frame, tile_base, total);
        __syncthreads();
        PR_HT(0);
        if (tid < TILE_M && tid < rows_valid && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
        // ---- the operand: the incoming gradient of the layer's output ---------------------------------
        if (p.phase == 1) {
            // feature-row gradients from the compositing backward; rows that failed the second AABB test produced zeros in
            // the forward pass: their gradient is cleared here (it also feeds the bias gradient's column sums)
            const int k4 = p.kpad >> 2;
            constexpr int BATCH = 4;          // loads first, then the LDS stores (see load_bn_backward)
            const int count = TILE_M * k4;
            for (int base = tid; base < count; base += MLP_THREADS * BATCH) {
                float4 v[BATCH];
                int kind[BATCH];              // 0: nothing to do in memory, 1: loaded, 2: a dead row whose gradient is cleared in memory
                // the row flags first, then FOUR UNCONDITIONAL requests (a row or column outside the operand reads the tile's first
                // chunk instead and drops the value): under `if (flags ...)` every request waited for its own LDS read, and hipcc -
                // which cannot count requests issued under a branch - waited for all of them in front of the fourth
                int fl[BATCH];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int idx = base + b * MLP_THREADS;
                    const int row = idx / k4;
                    fl[b] = S.flags[idx < count ? row : 0];
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int idx = base + b * MLP_THREADS;
                    const int row = idx / k4, c = (idx - row * k4) * 4;
                    const bool inside = idx < count && row < rows_valid && c < p.ld_gin;
                    kind[b] = !inside ? 0 : ((fl[b] & 3) == 3 ? 1 : 2);
                    const float* src = p.g_in + (size_t)(tile_base + (inside ? row : 0)) * p.ld_gin + (inside ? c : 0);
                    v[b] = *reinterpret_cast<const float4*>(src);
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    if (PR_HEADB_ABLATE & 16) v[b] = make_float4(1e-3f, -1e-3f, 2e-3f, 0.f);
                    if (kind[b] != 1) v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int idx = base + b * MLP_THREADS;
                    if (idx >= count) continue;
                    const int row = idx / k4, c = (idx - row * k4) * 4;
                    if (kind[b] == 1 && c + 3 >= p.k_real) {       // padding columns of the rows
                        if (c + 0 >= p.k_real) v[b].x = 0.f;
                        if (c + 1 >= p.k_real) v[b].y = 0.f;
                        if (c + 2 >= p.k_real) v[b].z = 0.f;
                        if (c + 3 >= p.k_real) v[b].w = 0.f;
                    }
                    // feature-row gradients of rows that failed the second AABB test produced zeros in the forward pass: their
                    // gradient is cleared in memory too (it also feeds the bias gradient's column sums)
                    if (kind[b] == 2) *reinterpret_cast<float4*>(p.g_in + (size_t)(tile_base + row) * p.ld_gin + c) = v[b];
                    *reinterpret_cast<float4*>(S.X + row * LDX + c) = v[b];
                }
            }
        } else if (PR_HEADB_ABLATE & 16) {
            for (int idx = tid; idx < TILE_M * (p.kpad >> 2); idx += MLP_THREADS)
                *reinterpret_cast<float4*>(S.X + (idx / (p.kpad >> 2)) * LDX + (idx % (p.kpad >> 2)) * 4) = make_float4(1e-3f, -1e-3f, 2e-3f, 0.f);
        } else {
            load_bn_backward(S, p.d_in, p.h_in, p.kpad, tile_base, rows_valid);
        }
        __syncthreads();
        PR_HT(1);
        // the raw activations the epilogue needs (this lane's column, its 32 rows per column block) are REQUESTED before the
        // product and arrive while it runs: loaded inside the epilogue, four at a time, their latency was the tile's critical path
        // (31 TFLOP/s for the two head phases)
        const unsigned long long mine = __ballot((S.flags[lane] & 3) == 3) >> (4 * half);   // bit ro <-> tile row ro + 4 half
        // (unconditional loads from clamped addresses - a lane without a live column reads the array's first row, rows beyond the
        // tile's last read the last - instead of 32 predicated loads: every predicate was an exec-mask change around one load.
        // What the values of dead rows / columns are does not matter: the epilogue selects them away.)
        auto prefetch = [&](int blk, float (&hv)[32]) {
            const int cb = wave + blk * MLP_WAVES;
            const bool live = cb < p.nblk && live_col[blk];
            const float* hb = live ? p.h + (size_t)(tile_base + 4 * half) * p.ld + cb * 32 + r : p.h + (size_t)tile_base * p.ld;
            const int ld = p.ld;
            const int last = live ? rows_valid - 1 - 4 * half : 0;      // largest row offset of this lane inside the tile (may be < 0)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int ro = PR_ACC_ROW(i & 15) + 32 * (i >> 4);
                int rc = ro < last ? ro : last;
                rc = rc > -4 * half ? rc : (live ? -4 * half : 0);
                hv[i] = (PR_HEADB_ABLATE & 1) ? 0.25f : hb[rc * ld];
            }
        };
        float hvA[32], hvB[32];
#ifndef PR_HEAD_NO_PREFETCH
        prefetch(0, hvA);
#endif
        PR_HT(2);
        f32x16 a00, a01, a10, a11;
        zero4(a00, a01, a10, a11);
        if (!(PR_HEADB_ABLATE & 8)) tile_products_any<SPLIT>(p.wt, p.nblk, S.X, a00, a01, a10, a11);
        PR_HT(3);
        if (tid == 0) claimed = atomicAdd(p.tile_counter, 1);
#ifndef PR_HEAD_NO_PREFETCH
        prefetch(1, hvB);
#endif
        __syncthreads();      // every wave has finished reading X
        PR_HT(4);
#ifdef PR_HEAD_NO_PREFETCH
        prefetch(0, hvA);
        prefetch(1, hvB);
#endif
        // ---- AdaIN + ReLU backward, normalisation backward up to the batch terms -------------------------
        //   y = h g[frame] + b[frame] (g = scale rstd), a = relu(y);  dy = (y > 0) d a;  d scale += dy xh, d bias += dy;
        //   d xh = dy scale;  the batch terms mean(d xh), mean(d xh xh) are applied by the next phase
        const bool uniform = S.uniform_frame != 0;
        if (uniform) {
            const int frame0 = S.frame[0];
            if (frame0 != cs.frame) {
                flush_frame_sums(cs, p.dscale, p.dbias, p.nblk, p.width);
                cs.frame = frame0;
            }
            const float* tab = p.table + (size_t)frame0 * p.table_stride;
            const int limit = rows_valid - 4 * half;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int cb = wave + blk * MLP_WAVES;
                if (cb >= p.nblk) continue;
                const int col = cb * 32 + r;
                const f32x16& lo = blk ? a10 : a00;
                const f32x16& hi = blk ? a11 : a01;
                const float (&hv)[32] = blk ? hvB : hvA;
                const bool live = live_col[blk];
                const float g = live ? tab[p.goff + col] : 0.f, b = live ? tab[p.boff + col] : 0.f;
                const float scale = live ? g / rstd[blk] : 0.f;
                // one base pointer per lane; the row offsets are wave-uniform multiples of the leading dimension
                float* ab = p.a_out + (size_t)(tile_base + 4 * half) * p.ld + col;
                float* xb = S.X + (4 * half) * LDX + col;
                const int ld = p.ld;
                double s1 = 0.0, s2 = 0.0;
                float ds = 0.f, db = 0.f;
                // branch-free: a row that did not enter the statistics (or a padding column) contributes exact zeros - the same sums
                // as skipping it, without 32 divergent branches (the two halves of a wave hold different rows)
                const bool full = rows_valid == TILE_M;       // every row offset of this lane is inside the tile
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int ro = PR_ACC_ROW(i & 15) + 32 * (i >> 4);
                    const float acc = (i >> 4) ? hi[i & 15] : lo[i & 15];
                    const bool on = live && ((mine >> ro) & 1ull);
                    const float y = fmaf(hv[i], g, b);
                    const bool pass = on && y > 0.f;
                    const float a = pass ? y : 0.f;
                    const float dy = pass ? acc : 0.f;
                    const float xh = on ? (hv[i] - mu[blk]) * rstd[blk] : 0.f;
                    const float dxh = dy * scale;
                    s1 += (double)dxh;
                    s2 += (double)dxh * (double)xh;
                    ds = fmaf(dy, xh, ds);
                    db += dy;
                    xb[ro * LDX] = dxh;
                    if (!(PR_HEADB_ABLATE & 2) && (full || ro < limit)) ab[ro * ld] = a;
                }
                // the two halves of the wave hold the same column: combine
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                ds += __shfl_xor(ds, 32, 64);
                db += __shfl_xor(db, 32, 64);
                cs.s1[blk] += s1;
                cs.s2[blk] += s2;
                cs.ds[blk] += ds;
                cs.db[blk] += db;
            }
        } else {
            // a tile that straddles two frames (a handful per call): the raw products go through X, one thread per column
            for (int blk = 0; blk < 2; ++blk) {
                const int cb = wave + blk * MLP_WAVES;
                if (cb >= p.nblk) break;
                const f32x16& lo = blk ? a10 : a00;
                const f32x16& hi = blk ? a11 : a01;
                float* xb = S.X + (4 * half) * LDX + cb * 32 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    xb[PR_ACC_ROW(i) * LDX] = lo[i];
                    xb[(PR_ACC_ROW(i) + 32) * LDX] = hi[i];
                }
            }
            __syncthreads();
            for (int col = tid; col < p.nblk * 32; col += MLP_THREADS) {
                const bool live = col < p.width;
                const float m = live ? p.mean[col] : 0.f, rs = live ? 1.0f / sqrtf(p.var[col] + p.eps) : 0.f;
                double s1 = 0.0, s2 = 0.0;
                for (int row = 0; row < TILE_M; ++row) {
                    float a = 0.f, dxh = 0.f;
                    if (live && (S.flags[row] & 3) == 3) {
                        const float* tab = p.table + (size_t)S.frame[row] * p.table_stride;
                        const float g = tab[p.goff + col], b = tab[p.boff + col];
                        const float hv = p.h[(size_t)(tile_base + row) * p.ld + col];
                        const float y = fmaf(hv, g, b);
                        a = y > 0.f ? y : 0.f;
                        const float dy = y > 0.f ? S.X[row * LDX + col] : 0.f;
                        const float xh = (hv - m) * rs;
                        dxh = dy * (g / rs);
                        s1 += (double)dxh;
                        s2 += (double)dxh * (double)xh;
                        if (dy != 0.f) {
                            atomicAdd(p.dscale + (size_t)S.frame[row] * MAX_WIDTH + col, dy * xh);
                            atomicAdd(p.dbias + (size_t)S.frame[row] * MAX_WIDTH + col, dy);
                        }
                    }
                    S.X[row * LDX + col] = dxh;
                    if (row < rows_valid) p.a_out[(size_t)(tile_base + row) * p.ld + col] = a;
                }
                if (live && !p.frozen && (s1 != 0.0 || s2 != 0.0)) {
                    atomicAdd(p.sums + col, s1);
                    atomicAdd(p.sums + p.nblk * 32 + col, s2);
                }
            }
        }
        PR_HT(5);
        if (tid == 0) S.next_tile = claimed;
        __syncthreads();
        PR_HT(6);
        if (!(PR_HEADB_ABLATE & 4)) store_tile_rows(S.X, p.d_out, p.nblk * 32, p.ld, tile_base, rows_valid);
        __syncthreads();      // the next tile overwrites X and the records
        PR_HT(7);
    }
    if (PR_HEADB_ABLATE & 32) return;
    flush_frame_sums(cs, p.dscale, p.dbias, p.nblk, p.width);
    if (lane < 32 && !p.frozen) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int col = (wave + blk * MLP_WAVES) * 32 + lane;
            if (live_col[blk] && (cs.s1[blk] != 0.0 || cs.s2[blk] != 0.0)) {
                atomicAdd(p.sums + col, cs.s1[blk]);
                atomicAdd(p.sums + p.nblk * 32 + col, cs.s2[blk]);
            }
        }
    }
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_head_bwd_group(HeadBwdJob j0, HeadBwdJob j1, HeadBwdJob j2, HeadBwdJob j3,
                                                                                   int count) {
    pr_stagger(2);
    head_bwd_loop<0>(j0);
    if (count > 1) head_bwd_loop<0>(j1);
    if (count > 2) head_bwd_loop<0>(j2);
    if (count > 3) head_bwd_loop<0>(j3);
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_head_bwd_group_bf16(HeadBwdJob j0, HeadBwdJob j1, HeadBwdJob j2, HeadBwdJob j3,
                                                                                        int count) {
    head_bwd_loop<1>(j0);
    if (count > 1) head_bwd_loop<1>(j1);
    if (count > 2) head_bwd_loop<1>(j2);
    if (count > 3) head_bwd_loop<1>(j3);
}

int launch_head_bwd_group(const HeadBwdJob* jobs, const long* max_rows, int count, hipStream_t s) {
    static thread_local HeadBwdJob g[MLP_GROUP_MAX];
    for (int begin = 0; begin < count; begin += MLP_GROUP_MAX) {
        const int n = count - begin < MLP_GROUP_MAX ? count - begin : MLP_GROUP_MAX;
        long max_tiles = 0;
        for (int j = 0; j < n; ++j) {
            g[j] = jobs[begin + j];
            PR_REQUIRE(g[j].tile_counter && (g[j].kpad % 16) == 0 && g[j].nblk >= 1 && g[j].nblk <= 8, "head backward: bad job");
            max_tiles += (max_rows[begin + j] + TILE_M - 1) / TILE_M;
        }
        if (max_tiles <= 0) continue;
        int cus = 0;
        const bool split = g[0].split != 0;
        for (int j = 1; j < n; ++j) PR_REQUIRE((g[j].split != 0) == split, "head backward: jobs of one launch differ in precision");
        PR_TRY(prepare_kernel(split ? reinterpret_cast<const void*>(k_head_bwd_group_bf16) : reinterpret_cast<const void*>(k_head_bwd_group),
                              (int)sizeof(BSmem), &cus));
        #ifdef PR_HEAD_RESIDENT
        const long resident = (long)cus * PR_HEAD_RESIDENT;        // measurement build: workgroups per CU of the head backward launch
#else
        const long resident = (long)cus * MLP_BLOCKS_PER_CU;
#endif
        ProfileScope scope(2, s);
        if (split)
            hipLaunchKernelGGL(k_head_bwd_group_bf16, dim3((unsigned)(max_tiles < resident ? max_tiles : resident)), dim3(MLP_THREADS),
                               sizeof(BSmem), s, g[0], g[1], g[2], g[3], n);
        else
            hipLaunchKernelGGL(k_head_bwd_group, dim3((unsigned)(max_tiles < resident ? max_tiles : resident)), dim3(MLP_THREADS), sizeof(BSmem), s,
                               g[0], g[1], g[2], g[3], n);
        PR_LAUNCH_CHECK();
    }
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward chain of a ReLU MLP with one skip concatenation, with its entry fused in
// ---------------------------------------------------------------------------------------------
template <int SPLIT>
__device__ __forceinline__ void chain_bwd_loop(const ChainBwdJob& c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BSmem& S = *reinterpret_cast<BSmem*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, half = lane >> 5;
    const int total = *c.total;
    const int nblk = c.Wpad >> 5, in_nblk = c.in_pad >> 5;
    __syncthreads();
#ifdef PR_CHAIN_TIMING
    if (tid == 0) for (int i = 0; i < 16; ++i) S.phase_acc[i] = 0;
#endif
    if (tid == 0) S.next_tile = atomicAdd(c.tile_counter, 1);
    if (c.entry == 1) {
        stage_bn_constants(S, c.mean1, c.var1, c.sums1, c.W, c.Wpad, c.stat_count, c.eps, c.frozen);
        for (int i = tid; i < c.Wpad; i += MLP_THREADS) S.ws[i] = (c.w_sigma && i < c.W) ? c.w_sigma[i] : 0.f;
    }
    __syncthreads();
    for (int tile = S.next_tile; tile * TILE_M < total; tile = S.next_tile) {
        const int tile_base = tile * TILE_M;
        const int rows_valid = (total - tile_base < TILE_M) ? total - tile_base : TILE_M;
        int claimed = 0;      // (claimed late, in front of the chain's last layer: see "Tile order" in mlp.hip)
        PR_CT0();
        load_tile_records(S, c.rec_flat, c.row_flags, c.samples_per_frame, tile_base, total);
        ColMasks masks = fetch_col_masks(c.bits + (size_t)(c.count - 1) * c.bits_stride, c.Wpad, nblk, tile);
        // fp16-pair products (SPLIT == 2): the largest |entry| of the tile in X, in one of two LDS words - `cur` names the word of the
        // tile the next product reads, its producer wrote into it, the other word is cleared while that product runs
        int cur = 0;
        if (SPLIT == 2 && tid == 0) S.tile_max[0] = S.tile_max[1] = 0;
        __syncthreads();
        f32x16 a00, a01, a10, a11;
        // the product of the tile in X with one segment, at the tile's scale
        auto scaled_product = [&](const Seg& sg, int out_blk, const Drain* drain) {
            float scale = 1.0f;
            if (SPLIT == 2) {
                const int k = tile_scale_log2(S.tile_max[cur]);
                scale = ldexpf(1.0f, k);
                if (tid == 0) S.tile_max[cur ^ 1] = 0;      // (its next writers are behind this product's barrier)
                tile_products_any<SPLIT>(sg, out_blk, S.X, a00, a01, a10, a11, drain, scale);
                const float back = ldexpf(1.0f, -k - TRAIN_SPLIT_WEIGHT_SCALE_LOG2);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    a00[i] *= back;
                    a01[i] *= back;
                    a10[i] *= back;
                    a11[i] *= back;
                }
            } else {
                tile_products_any<SPLIT>(sg, out_blk, S.X, a00, a01, a10, a11, drain);
            }
        };
        if (c.entry == 1) {
            // NeRF: normalisation backward of head layer 1, . W0, + the density's path through the sigma head, ReLU mask of
            // the backbone's last layer
            if (tid < TILE_M) {
                float gs = 0.f;
                if (tid < rows_valid && (S.flags[tid] & 3) == 3 && c.w_sigma &&
                    c.in_scene[(size_t)S.frame[tid] * c.in_scene_stride] != 0)
                    gs = c.g_sigma[S.flat[tid]];
                S.gsr[tid] = gs;
                if (tid < rows_valid && c.gsr4) *reinterpret_cast<float4*>(c.gsr4 + (size_t)(tile_base + tid) * 4) = make_float4(gs, 0.f, 0.f, 0.f);
            }
            load_bn_backward(S, c.d1, c.h1, c.Wpad, tile_base, rows_valid, SPLIT == 2 ? &S.tile_max[0] : nullptr);
            __syncthreads();
            PR_CT(0);
            zero4(a00, a01, a10, a11);
            scaled_product(c.w0t, nblk, nullptr);
            PR_CT(1);
            __syncthreads();
            PR_CT(2);
            for (int blk = 0; blk < 2; ++blk) {
                const int cb = wave + blk * MLP_WAVES;
                if (cb >= nblk) break;
                const float w = S.ws[cb * 32 + r];
                f32x16& lo = blk ? a10 : a00;
                f32x16& hi = blk ? a11 : a01;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    lo[i] = fmaf(S.gsr[PR_ROWS_OF(i, half, 0)], w, lo[i]);
                    hi[i] = fmaf(S.gsr[PR_ROWS_OF(i, half, 1)], w, hi[i]);
                }
            }
            store_masked(S, nblk, masks, a00, a01, a10, a11, SPLIT == 2 ? &S.tile_max[1] : nullptr);
            cur = 1;
        } else {
            // ray bender: G = (g_raw . W_out) masked by the last layer's ReLU; one thread per column, the rows' raw gradients via LDS
            if (tid < TILE_M) {
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tid < rows_valid) g = *reinterpret_cast<const float4*>(c.g_braw4 + (size_t)(tile_base + tid) * 4);
                *reinterpret_cast<float4*>(&S.cst[0][tid * 4]) = g;
            }
            __syncthreads();
            for (int col = tid; col < c.Wpad; col += MLP_THREADS) {
                const bool live = col < c.W;
                const float w0 = live ? c.w_out[col] : 0.f, w1 = live ? c.w_out[c.w_out_ld + col] : 0.f, w2 = live ? c.w_out[2 * c.w_out_ld + col] : 0.f;
                const unsigned long long word = reinterpret_cast<const unsigned long long*>(c.bits + (size_t)(c.count - 1) * c.bits_stride)[(size_t)tile * c.Wpad + col];
                float biggest = 0.f;
                for (int row = 0; row < TILE_M; ++row) {
                    const float4 g = *reinterpret_cast<const float4*>(&S.cst[0][row * 4]);
                    const float v = ((word >> row) & 1ull) ? fmaf(g.x, w0, fmaf(g.y, w1, g.z * w2)) : 0.f;
                    S.X[row * LDX + col] = v;
                    biggest = fmaxf(biggest, fabsf(v));
                }
                if (SPLIT == 2) atomicMax(reinterpret_cast<unsigned int*>(&S.tile_max[0]), __float_as_uint(biggest));
            }
        }
        PR_CT(3);
        __syncthreads();
        PR_CT(4);
        // G of layer count - 1 is in X: its write-out to the gradient stack (the weight-gradient launch reads it) rides on the next
        // product's K loop, which reads the same tile (tile_products' drain); products too narrow for that write it first
        Drain pending{c.gstack + (size_t)(c.count - 1) * c.g_stride + (size_t)tile_base * c.Wpad, c.Wpad, c.Wpad >> 2, rows_valid};
        auto product = [&](const Seg& sg, int out_blk) {
            zero4(a00, a01, a10, a11);
            if (pending.dst && !drains_in_loop(out_blk)) {
                store_tile_rows(S.X, pending.dst - (size_t)tile_base * c.Wpad, c.Wpad, c.Wpad, tile_base, rows_valid);
                pending.dst = nullptr;
            }
            scaled_product(sg, out_blk, pending.dst ? &pending : nullptr);
            pending.dst = nullptr;
        };
        PR_CT(5);
        bool g_in_written = false;
        // ReLU mask words of layer l's input (layer l - 1's output), requested a whole layer ahead: behind the weight fragments of a K
        // loop they would be waited for in issue order (2 KB of a cold tile: HBM latency in front of the first MFMA)
        ColMasks next = fetch_col_masks(c.bits + (size_t)(c.count - 2) * c.bits_stride, c.Wpad, nblk, tile);
        for (int l = c.count - 1; l >= 1; --l) {
            masks = next;
            if (l >= 2) next = fetch_col_masks(c.bits + (size_t)(l - 2) * c.bits_stride, c.Wpad, nblk, tile);
            if (l == c.skip) {
                product(c.in0_skip, in_nblk);
                store_global(c.g_in, c.ld_in, c.in_real, in_nblk, tile_base, rows_valid, false, a00, a01, a10, a11);
                g_in_written = true;
            }
            PR_CT(6);
            if (l == 1 && tid == 0) claimed = atomicAdd(c.tile_counter, 1);
            product(c.act_t[l], nblk);
            PR_CT(1);
            __syncthreads();
            PR_CT(2);
            store_masked(S, nblk, masks, a00, a01, a10, a11, SPLIT == 2 ? &S.tile_max[cur ^ 1] : nullptr);
            cur ^= 1;
            PR_CT(3);
            __syncthreads();
            PR_CT(4);
            if (!(PR_CHAINGRP_ABLATE & 1))
                pending = Drain{c.gstack + (size_t)(l - 1) * c.g_stride + (size_t)tile_base * c.Wpad, c.Wpad, c.Wpad >> 2, rows_valid};
        }
        PR_CT(5);
        if (tid == 0) S.next_tile = claimed;
        product(c.in0_first, in_nblk);
        store_global(c.g_in, c.ld_in, c.in_real, in_nblk, tile_base, rows_valid, g_in_written, a00, a01, a10, a11);
        PR_CT(7);
        __syncthreads();   // the next tile overwrites X, the bits and the records
        PR_CT(8);
    }
#ifdef PR_CHAIN_TIMING
    if (tid == 0) for (int i = 0; i < 16; ++i) if (S.phase_acc[i]) atomicAdd(&g_chain_phase[i], S.phase_acc[i]);
#endif
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_chain_bwd_group(ChainBwdJob j0, ChainBwdJob j1, ChainBwdJob j2, ChainBwdJob j3,
                                                                                    int count) {
    chain_bwd_loop<0>(j0);
    if (count > 1) chain_bwd_loop<0>(j1);
    if (count > 2) chain_bwd_loop<0>(j2);
    if (count > 3) chain_bwd_loop<0>(j3);
}

// split precision (PR_FLAG_SPLIT_BACKWARD): the chains' products on bf16 triples
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_chain_bwd_group_bf16(ChainBwdJob j0, ChainBwdJob j1, ChainBwdJob j2, ChainBwdJob j3,
                                                                                         int count) {
    chain_bwd_loop<1>(j0);
    if (count > 1) chain_bwd_loop<1>(j1);
    if (count > 2) chain_bwd_loop<1>(j2);
    if (count > 3) chain_bwd_loop<1>(j3);
}

// ... or (the default of PR_FLAG_SPLIT_BACKWARD) on fp16 pairs of the tile x a per-tile power of two (job.split == 2)
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_chain_bwd_group_f16(ChainBwdJob j0, ChainBwdJob j1, ChainBwdJob j2, ChainBwdJob j3,
                                                                                        int count) {
    chain_bwd_loop<2>(j0);
    if (count > 1) chain_bwd_loop<2>(j1);
    if (count > 2) chain_bwd_loop<2>(j2);
    if (count > 3) chain_bwd_loop<2>(j3);
}

#ifdef PR_HEAD_TIMING
extern "C" int pr_debug_head_phases(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_head_phase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long zero[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_head_phase), zero, sizeof(zero));
    }
    return 0;
}
#endif
#ifdef PR_CHAIN_TIMING
extern "C" int pr_debug_chain_phases(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chain_phase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long zero[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_phase), zero, sizeof(zero));
    }
    return 0;
}
#endif

int launch_chain_bwd_group(const ChainBwdJob* jobs, const long* max_rows, int count, hipStream_t s) {
    static thread_local ChainBwdJob g[MLP_GROUP_MAX];
    for (int begin = 0; begin < count; begin += MLP_GROUP_MAX) {
        const int n = count - begin < MLP_GROUP_MAX ? count - begin : MLP_GROUP_MAX;
        long max_tiles = 0;
        for (int j = 0; j < n; ++j) {
            g[j] = jobs[begin + j];
            PR_REQUIRE(g[j].tile_counter && g[j].count >= 2 && g[j].count <= PR_MAX_LAYERS && g[j].skip >= 1 && g[j].skip < g[j].count,
                       "backward chain: bad job");
            max_tiles += (max_rows[begin + j] + TILE_M - 1) / TILE_M;
        }
        if (max_tiles <= 0) continue;
        int cus = 0;
        const int split = g[0].split;
        for (int j = 1; j < n; ++j) PR_REQUIRE(g[j].split == split, "backward chain: jobs of one launch differ in precision");
        auto* const kernel = split == 2 ? k_chain_bwd_group_f16 : (split == 1 ? k_chain_bwd_group_bf16 : k_chain_bwd_group);
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(kernel), (int)sizeof(BSmem), &cus));
        const long resident = (long)cus * MLP_BLOCKS_PER_CU;
        ProfileScope scope(2, s);
        hipLaunchKernelGGL(kernel, dim3((unsigned)(max_tiles < resident ? max_tiles : resident)), dim3(MLP_THREADS), sizeof(BSmem), s,
                           g[0], g[1], g[2], g[3], n);
        PR_LAUNCH_CHECK();
    }
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Hutchinson divergence estimate (object_composer.py:582-601): forward-mode derivative of the ray bender along the probe
//   t_0 = d input / d x . e,  t_{l+1} = relu'(z_l) (W_l t_l),  div = sum_a e_a (J e)_a
// The tangent tile lives in X[:, 0:BWpad), the input tangent t_0 beside it in X[:, BWpad + 4 ...) (the skip layer needs it again).
// ---------------------------------------------------------------------------------------------
bool div_chain_supported(int BWpad, int bin_pad) { return BWpad + 4 + bin_pad <= LDX && bin_pad <= BWpad + 4; }

__device__ __forceinline__ void div_chain_loop(const DivChainJob& c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BSmem& S = *reinterpret_cast<BSmem*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = *c.total;
    const int nblk = c.BWpad >> 5;
    const int T0 = c.BWpad + 4;
    float* probe = &S.cst[0][0];           // [row][4]: the probe of the row's sample
    float size[3];
    for (int a = 0; a < 3; ++a) size[a] = c.hi[a] - c.lo[a];
    __syncthreads();
    if (tid == 0) S.next_tile = atomicAdd(c.tile_counter, 1);
    __syncthreads();
    for (int tile = S.next_tile; tile * TILE_M < total; tile = S.next_tile) {
        const int tile_base = tile * TILE_M;
        const int rows_valid = (total - tile_base < TILE_M) ? total - tile_base : TILE_M;
        int claimed = 0;      // (claimed late, in front of the chain's last layer)
        if (tid < TILE_M) {
            const bool valid = tid < rows_valid;
            const int flat = c.rec_flat[valid ? tile_base + tid : tile_base];
            S.flat[tid] = flat;
            S.flags[tid] = valid ? c.row_flags[tile_base + tid] : 0;
            const long ray = flat / c.positions;
            const int sample = flat - (int)ray * c.positions;
            for (int a = 0; a < 3; ++a) probe[tid * 4 + a] = valid ? noise_normal(c.noise, ray, c.positions * 3, sample * 3 + a) : 0.f;
        }
        // the saved bender input of the tile -> X[:, 0:bin_pad)
        const int b4 = c.bin_pad >> 2;
        for (int idx = tid; idx < TILE_M * b4; idx += MLP_THREADS) {
            const int row = idx / b4, c4 = (idx - row * b4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows_valid) v = *reinterpret_cast<const float4*>(c.bin + (size_t)(tile_base + row) * c.bin_pad + c4);
            *reinterpret_cast<float4*>(S.X + row * LDX + c4) = v;
        }
        __syncthreads();
        // t_0: element a: e_a / size_a; sin slot: 2^k cos_saved dv_a; cos slot: -2^k sin_saved dv_a; 0 beyond the encoding
        for (int idx = tid; idx < TILE_M * c.bin_pad; idx += MLP_THREADS) {
            const int row = idx / c.bin_pad, j = idx - row * c.bin_pad;
            float v = 0.f;
            if (j < c.benc) {
                const float* b = S.X + row * LDX;
                if (j < 3) {
                    v = probe[row * 4 + j] / size[j];
                } else {
                    const int k = (j - 3) / 6, rem = (j - 3) - 6 * k;
                    const float f = ldexpf(1.0f, k);
                    if (rem < 3) v = f * b[j + 3] * (probe[row * 4 + rem] / size[rem]);
                    else v = -f * b[j - 3] * (probe[row * 4 + rem - 3] / size[rem - 3]);
                }
            }
            S.X[row * LDX + T0 + j] = v;
        }
        __syncthreads();
        for (int l = 0; l < c.b_count; ++l) {
            const ColMasks masks = fetch_col_masks(c.bbits + (size_t)l * c.bbits_stride, c.BWpad, nblk, tile);
            f32x16 a00, a01, a10, a11;
            zero4(a00, a01, a10, a11);
            if (l == c.b_count - 1 && tid == 0) claimed = atomicAdd(c.tile_counter, 1);
            tile_products(c.seg0[l], nblk, l == 0 ? S.X + T0 : S.X, a00, a01, a10, a11);
            if (l == c.b_skip) tile_products(c.seg1, nblk, S.X + T0, a00, a01, a10, a11);
            __syncthreads();
            store_masked(S, nblk, masks, a00, a01, a10, a11);
            __syncthreads();
        }
        if (tid == 0) S.next_tile = claimed;
        // output head and the clamp cases, 8 threads per row
        for (int s = tid >> 3; s < TILE_M; s += MLP_THREADS / 8) {
            const int part = tid & 7;
            float tan[3] = {0.f, 0.f, 0.f};
            for (int k = part; k < c.BW; k += 8) {
                const float x = S.X[s * LDX + k];
                for (int a = 0; a < 3; ++a) tan[a] = fmaf(x, c.w_out[a * c.BWpad + k], tan[a]);
            }
            for (int a = 0; a < 3; ++a) {
                float v = tan[a];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                tan[a] = v;
            }
            if (part != 0 || !(S.flags[s] & 1)) continue;
            const int m = tile_base + s;
            float acc = 0.f;
            for (int a = 0; a < 3; ++a) {
                const float e = probe[s * 4 + a];
                const float x = c.rec_pos[(size_t)m * 3 + a];
                const float pre = c.braw[(size_t)m * 3 + a] * size[a];
                const float lob = c.lo[a] - x, hib = c.hi[a] - x;
                const float m1 = pre > lob ? pre : lob;
                float je;
                if (m1 > hib) je = -e;
                else if (pre >= lob) je = size[a] * tan[a];
                else je = -e;
                if (c.canonical) je = 0.f;
                acc = fmaf(e, je, acc);
            }
            c.div[S.flat[s]] = acc;
        }
        __syncthreads();   // the next tile overwrites X, the probes and the records
    }
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_div_chain_group(DivChainJob j0, DivChainJob j1, DivChainJob j2, DivChainJob j3,
                                                                                    int count) {
    div_chain_loop(j0);
    if (count > 1) div_chain_loop(j1);
    if (count > 2) div_chain_loop(j2);
    if (count > 3) div_chain_loop(j3);
}

int launch_div_chain_group(const DivChainJob* jobs, const long* max_rows, int count, hipStream_t s) {
    static thread_local DivChainJob g[MLP_GROUP_MAX];
    for (int begin = 0; begin < count; begin += MLP_GROUP_MAX) {
        const int n = count - begin < MLP_GROUP_MAX ? count - begin : MLP_GROUP_MAX;
        long max_tiles = 0;
        for (int j = 0; j < n; ++j) {
            g[j] = jobs[begin + j];
            PR_REQUIRE(g[j].tile_counter && div_chain_supported(g[j].BWpad, g[j].bin_pad), "divergence chain: bad job");
            max_tiles += (max_rows[begin + j] + TILE_M - 1) / TILE_M;
        }
        if (max_tiles <= 0) continue;
        int cus = 0;
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_div_chain_group), (int)sizeof(BSmem), &cus));
        const long resident = (long)cus * MLP_BLOCKS_PER_CU;
        ProfileScope scope(2, s);
        hipLaunchKernelGGL(k_div_chain_group, dim3((unsigned)(max_tiles < resident ? max_tiles : resident)), dim3(MLP_THREADS), sizeof(BSmem), s,
                           g[0], g[1], g[2], g[3], n);
        PR_LAUNCH_CHECK();
    }
    return PR_OK;
}

}  // namespace pr
