// Split-precision variant of the fused per-object MLP: fp32 emulated with three fp16 MFMAs.
//
// Every operand is carried as an fp16 pair  x = hi + lo * 2^-11  (hi = fp16(x), lo = fp16((x - hi) * 2^11);
// ~22 significant bits, the residual keeps full fp16 precision because it is rescaled).  A product sum
// becomes      sum a*w  ~  sum a_hi*w_hi  +  2^-11 * ( sum a_hi*w_lo + sum a_lo*w_hi )
// (the dropped a_lo*w_lo term is 2^-22 relative); the two sums live in separate fp32 accumulators of
// v_mfma_f32_32x32x16_f16, whose products are exact in fp32.  The matrix pipe runs fp16 at 16x the fp32
// rate, so three MFMAs per 16 K-values replace eight fp32 MFMAs: 5.3x less matrix time.
//
// Structure = csrc/mlp.hip (64-sample tile, 4 waves x two 32-column blocks, two resident tiles per CU, weights as
// fragment-ordered hi/lo pairs straight from L2, even/odd operand pipeline); activations live in LDS as
// two fp16 planes.  Selected with pr_call_t.precision = PR_PRECISION_F16X3; eval-mode only.
#include "pr_common.h"

#include <cstddef>

namespace pr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int STILE_M = 64;          // samples per workgroup tile of the split kernel (two 32-row MFMA blocks)
constexpr int STHREADS = 256;        // 4 waves; two workgroups (tiles) per CU
constexpr int SWAVES = 4;
constexpr int SBLOCKS_PER_CU = 2;
constexpr int LDH = MAX_WIDTH + 8;   // halves per activation row (528 B = 33 16-byte slots: conflict-free b128)
constexpr int LDSTAGE = 260;         // floats per row when the activation planes are reused as an fp32 staging tile

struct SmemH {
    int uniform_frame;
    int next_tile;           // the tile claimed for this workgroup's next iteration (see mlp.hip)
    int matrix_priority;     // s_setprio level of this workgroup's K loops: the CU's two resident tiles differ (cu_arrival_parity)
    int pad_[1];
    float head_w[MAX_WIDTH + 8];          // sigma head weights + bias
    _Float16 Xh[STILE_M * LDH];            // activations, hi plane
    _Float16 Xl[STILE_M * LDH];            // activations, lo plane (scaled by 2^11)
    float adain_g[MAX_WIDTH];    // scale / shift row of the layer's AdaIN table for a tile of ONE frame: requested in front of
    float adain_b[MAX_WIDTH];    // the layer's K loop, parked here behind it (the epilogue does not wait for L2)
    float pos[STILE_M * 8];
    int flat[STILE_M];
    int frame[STILE_M];
    int flags[STILE_M];       // bit 0 real sample, bit 1 passed every AABB test, bit 2 density not <= 0 (gated head)
    int dest[STILE_M];        // gated head: compact feature row of a tile row (-1: none)
    int src[STILE_M];         // gated head: pending-stack slot a tile row is exchanged with (-1: none)
};
static_assert(sizeof(_Float16) * 2 * STILE_M * LDH >= sizeof(float) * STILE_M * LDSTAGE, "staging tile must fit the activation planes");
static_assert(sizeof(SmemH) * SBLOCKS_PER_CU <= 158 * 1024, "the workgroups of one CU must fit its LDS");
static_assert(offsetof(SmemH, head_w) % 16 == 0 && offsetof(SmemH, Xh) % 16 == 0 && offsetof(SmemH, Xl) % 16 == 0,
              "16-byte LDS reads of the activation planes");

#ifndef PR_SPLIT_ABLATE
#define PR_SPLIT_ABLATE 0   // profiling builds only (results are wrong): 1 = every K step re-reads the operands of steps 0 / 1 (L1 hits: no L2 weight stream), 8 = no epilogue
#endif
#if PR_SPLIT_ABLATE & 64
// phase timing build: thread 0 of every workgroup accumulates shader-clock deltas per phase
__device__ unsigned long long g_phase_cycles[16];
#define PR_PHASE_T0() unsigned long long _pt = __builtin_amdgcn_s_memtime()
#define PR_PHASE(idx)                                                                      \
    do {                                                                                   \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                        \
        if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[idx], _n - _pt);                   \
        _pt = _n;                                                                          \
    } while (0)
#else
#define PR_PHASE_T0() do {} while (0)
#define PR_PHASE(idx) do {} while (0)
#endif
#define PR_ACC_ROW(i) (((i) & 3) + 8 * ((i) >> 2))
// a = activation fragment, w = weight fragment: D[feature][sample] += w (A operand) x a (B operand)
#define PR_MFMA16(acc, a, w) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, acc, 0, 0, 0)

__device__ __forceinline__ void split_store(_Float16* hi_plane, _Float16* lo_plane, int idx, float v) {
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);   // fp16 range; never reached by sane activations
    const _Float16 hi = (_Float16)v;
    hi_plane[idx] = hi;
    lo_plane[idx] = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ float split_load(const _Float16* hi_plane, const _Float16* lo_plane, int idx) {
    return (float)hi_plane[idx] + (float)lo_plane[idx];
}

// Epilogues on (main, correction) accumulator pairs of one 32x32 block.  The weights are the MFMA A operand and the
// activations the B operand, so the block is D[feature][sample]: lane (r, half) holds SAMPLE r and the features
// (i & 3) + 8 (i >> 2) + 4 half - four groups of four consecutive features, i.e. one 8-byte store per group and plane.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// splits four values (already inside the fp16 range) into the hi / lo planes at halves offset `idx` (8-byte aligned)
__device__ __forceinline__ void split_store4(_Float16* hi_plane, _Float16* lo_plane, int idx, f32x4 v) {
    const f16x4 hi = __builtin_convertvector(v, f16x4);
    const f32x4 back = __builtin_convertvector(hi, f32x4);
    const f16x4 lo = __builtin_convertvector(v - back, f16x4);
    *reinterpret_cast<f16x4*>(hi_plane + idx) = hi;
    *reinterpret_cast<f16x4*>(lo_plane + idx) = lo;
}

__device__ __forceinline__ f32x4 pick4(const f32x16& m, int j) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = m[4 * j + k];
    return v;
}

// idx0 = sample row * LDH + first feature of the lane
__device__ __forceinline__ void store_relu_h(const f32x16& m, SmemH& S, int idx0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 v = pick4(m, j);
        // ReLU and the fp16 range guard (never reached by sane activations) in one v_med3_f32
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k], 0.f, 65504.0f);
        split_store4(S.Xh, S.Xl, idx0 + 8 * j, v);
    }
}
// g / b point at the lane's first feature in the AdaIN table row of the sample's frame
__device__ __forceinline__ void store_adain_h(const f32x16& m, SmemH& S, int idx0, const float* g, const float* b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 8 * j);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 8 * j);
        f32x4 v = pick4(m, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = __builtin_amdgcn_fmed3f(fmaf(v[k], gg[k], bb[k]), 0.f, 65504.0f);   // ReLU + range guard
        }
        split_store4(S.Xh, S.Xl, idx0 + 8 * j, v);
    }
}
// the same with the table row parked in LDS (a tile of one frame); feat0 = the lane's first feature
__device__ __forceinline__ void store_adain_lds_h(const f32x16& m, SmemH& S, int idx0, int feat0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(S.adain_g + feat0 + 8 * j);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(S.adain_b + feat0 + 8 * j);
        f32x4 v = pick4(m, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_fmed3f(fmaf(v[k], gg[k], bb[k]), 0.f, 65504.0f);
        split_store4(S.Xh, S.Xl, idx0 + 8 * j, v);
    }
}
// stage = fp32 staging tile over the activation planes; base = &stage[sample row * LDSTAGE + first feature]
__device__ __forceinline__ void store_stage_h(const f32x16& m, float* base) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(base + 8 * j) = pick4(m, j);
}

__device__ __forceinline__ void fill_nerf_input_h(SmemH& S, const MlpParams& p);
__device__ __forceinline__ void fill_bender_input_h(SmemH& S, const MlpParams& p);

// One layer on the tile (see run_layer in mlp.hip for the geometry: wave w owns the column blocks w and w + 4 for
// both row blocks; here every block has a main and a correction accumulator).  input_kind: what a src == 1 segment
// re-computes into X[:, 0:K) - 0 NeRF input, 1 ray-bender input.
// TERMS = 3: hi x hi + hi x lo + lo x hi (precision "f16x3", fp32-grade); TERMS = 1: hi x hi only (precision "f16": plain
// fp16 operands, fp32 accumulation) - the lo fragments are then never loaded.
#define PR_SPLIT3(M, AH, AL, BH, BL) \
    PR_MFMA16(M, AH, BH);            \
    if (TERMS == 3) {                \
        PR_MFMA16(M, AH, BL);        \
        PR_MFMA16(M, AL, BH);        \
    }

template <int TERMS>
__device__ __forceinline__ void run_layer_h(const Layer& L, SmemH& S, const MlpParams& p, int input_kind) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int nblk = L.nblk;
    const int cbA = wave, cbB = wave + SWAVES;
    const bool active = cbA < nblk;
    const bool two = cbB < nblk;
    PR_PHASE_T0();
    // accumulators: [column block A / B][row block 0 / 1]; the three partial products of a step go into the same one
    f32x16 mA0, mA1, mB0, mB1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        mA0[i] = 0.f;
        mB0[i] = 0.f;
    }
    if (L.bias != nullptr && active) {
        const float* bp = L.bias + cbA * 32 + 4 * half;   // the lane's 4 x 4 features
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 8 * j);
#pragma unroll
            for (int k = 0; k < 4; ++k) mA0[4 * j + k] = b4[k];
            if (two) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(bp + 32 * SWAVES + 8 * j);
#pragma unroll
                for (int k = 0; k < 4; ++k) mB0[4 * j + k] = c4[k];
            }
        }
    }
    mA1 = mA0;
    mB1 = mB0;
#ifndef PR_SPLIT_NO_ADAIN_PREFETCH
    // AdaIN table row of a one-frame tile: requested now, parked in LDS behind the K loop (2 VGPRs across the loop)
    const bool park = L.epi == EPI_ADAIN_RELU && S.uniform_frame != 0;
    float park_g = 0.f, park_b = 0.f;
    if (park && (int)threadIdx.x < nblk * 32) {
        const float* tab = p.adain + (size_t)S.frame[0] * p.adain_stride + L.adain_off + threadIdx.x;
        park_g = tab[0];
        park_b = tab[nblk * 32];
    }
#else
    const bool park = false;
#endif
    for (int sidx = 0; sidx < L.nseg; ++sidx) {
        const Seg& sg = L.seg[sidx];
        if (sg.src == 1 && sidx > 0) {
            __syncthreads();
            if (input_kind == 1) fill_bender_input_h(S, p); else fill_nerf_input_h(S, p);
            __syncthreads();
        }
        if (!active) continue;
        // matrix work outranks the other resident tile's serial phases, and one tile's matrix work the other's
        if (S.matrix_priority) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
        const int ks = sg.kq >> 1;   // 16-wide steps, even (K padded to 32)
        const int aoff = r * LDH + 8 * half;
        const _Float16* a0h = S.Xh + aoff;
        const _Float16* a0l = S.Xl + aoff;
        const _Float16* a1h = S.Xh + aoff + 32 * LDH;
        const _Float16* a1l = S.Xl + aoff + 32 * LDH;
        // weights: per (column block, step): hi fragment (64 lanes x 16 B) then lo fragment
        const f16x8* wpA = reinterpret_cast<const f16x8*>(sg.w) + (size_t)cbA * ks * 128 + lane;
        f16x8 ah0E = *reinterpret_cast<const f16x8*>(a0h), al0E = *reinterpret_cast<const f16x8*>(a0l);
        f16x8 ah1E = *reinterpret_cast<const f16x8*>(a1h), al1E = *reinterpret_cast<const f16x8*>(a1l);
        f16x8 ah0O = *reinterpret_cast<const f16x8*>(a0h + 16), al0O = *reinterpret_cast<const f16x8*>(a0l + 16);
        f16x8 ah1O = *reinterpret_cast<const f16x8*>(a1h + 16), al1O = *reinterpret_cast<const f16x8*>(a1l + 16);
        if (two) {
            // (the first requests in the order the loop repeats them - even A, even B, odd A, odd B: with A, A, B, B in front of the
            // loop the merged request queue turned the wait for the even B fragments into vmcnt(0): see run_layer in mlp_tile.h)
            const f16x8* wpB = reinterpret_cast<const f16x8*>(sg.w) + (size_t)cbB * ks * 128 + lane;
            f16x8 bAhE = wpA[0], bAlE = wpA[64], bBhE = wpB[0], bBlE = wpB[64];
            __builtin_amdgcn_sched_barrier(0);
            f16x8 bAhO = wpA[128], bAlO = wpA[192], bBhO = wpB[128], bBlO = wpB[192];
            __builtin_amdgcn_sched_barrier(0);
            for (int s = 0; s < ks; s += 2) {
                const int se = (PR_SPLIT_ABLATE & 1) ? 0 : ((s + 2 < ks) ? s + 2 : s), so = (PR_SPLIT_ABLATE & 1) ? 1 : ((s + 3 < ks) ? s + 3 : s + 1);
                PR_SPLIT3(mA0, ah0E, al0E, bAhE, bAlE);
                PR_SPLIT3(mA1, ah1E, al1E, bAhE, bAlE);
                PR_SPLIT3(mB0, ah0E, al0E, bBhE, bBlE);
                PR_SPLIT3(mB1, ah1E, al1E, bBhE, bBlE);
                bAhE = wpA[(size_t)se * 128];
                bAlE = wpA[(size_t)se * 128 + 64];
                bBhE = wpB[(size_t)se * 128];
                bBlE = wpB[(size_t)se * 128 + 64];
                ah0E = *reinterpret_cast<const f16x8*>(a0h + 16 * se);
                al0E = *reinterpret_cast<const f16x8*>(a0l + 16 * se);
                ah1E = *reinterpret_cast<const f16x8*>(a1h + 16 * se);
                al1E = *reinterpret_cast<const f16x8*>(a1l + 16 * se);
                PR_SPLIT3(mA0, ah0O, al0O, bAhO, bAlO);
                PR_SPLIT3(mA1, ah1O, al1O, bAhO, bAlO);
                PR_SPLIT3(mB0, ah0O, al0O, bBhO, bBlO);
                PR_SPLIT3(mB1, ah1O, al1O, bBhO, bBlO);
                bAhO = wpA[(size_t)so * 128];
                bAlO = wpA[(size_t)so * 128 + 64];
                bBhO = wpB[(size_t)so * 128];
                bBlO = wpB[(size_t)so * 128 + 64];
                ah0O = *reinterpret_cast<const f16x8*>(a0h + 16 * so);
                al0O = *reinterpret_cast<const f16x8*>(a0l + 16 * so);
                ah1O = *reinterpret_cast<const f16x8*>(a1h + 16 * so);
                al1O = *reinterpret_cast<const f16x8*>(a1l + 16 * so);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TERMS, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, TERMS == 3 ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TERMS, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, TERMS == 3 ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 4 : 2, 0);
            }
        } else {
            f16x8 bAhE = wpA[0], bAlE = wpA[64];
            __builtin_amdgcn_sched_barrier(0);
            f16x8 bAhO = wpA[128], bAlO = wpA[192];
            __builtin_amdgcn_sched_barrier(0);
            for (int s = 0; s < ks; s += 2) {
                const int se = (PR_SPLIT_ABLATE & 1) ? 0 : ((s + 2 < ks) ? s + 2 : s), so = (PR_SPLIT_ABLATE & 1) ? 1 : ((s + 3 < ks) ? s + 3 : s + 1);
                PR_SPLIT3(mA0, ah0E, al0E, bAhE, bAlE);
                PR_SPLIT3(mA1, ah1E, al1E, bAhE, bAlE);
                bAhE = wpA[(size_t)se * 128];
                bAlE = wpA[(size_t)se * 128 + 64];
                ah0E = *reinterpret_cast<const f16x8*>(a0h + 16 * se);
                al0E = *reinterpret_cast<const f16x8*>(a0l + 16 * se);
                ah1E = *reinterpret_cast<const f16x8*>(a1h + 16 * se);
                al1E = *reinterpret_cast<const f16x8*>(a1l + 16 * se);
                PR_SPLIT3(mA0, ah0O, al0O, bAhO, bAlO);
                PR_SPLIT3(mA1, ah1O, al1O, bAhO, bAlO);
                bAhO = wpA[(size_t)so * 128];
                bAlO = wpA[(size_t)so * 128 + 64];
                ah0O = *reinterpret_cast<const f16x8*>(a0h + 16 * so);
                al0O = *reinterpret_cast<const f16x8*>(a0l + 16 * so);
                ah1O = *reinterpret_cast<const f16x8*>(a1h + 16 * so);
                al1O = *reinterpret_cast<const f16x8*>(a1l + 16 * so);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, TERMS == 3 ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, TERMS == 3 ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 4 : 2, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    PR_PHASE(3);
#ifndef PR_SPLIT_NO_ADAIN_PREFETCH
    if (park && (int)threadIdx.x < nblk * 32) {
        S.adain_g[threadIdx.x] = park_g;
        S.adain_b[threadIdx.x] = park_b;
    }
#endif
    __syncthreads();  // every wave has finished reading the activation planes (and the parked table row is complete)
    PR_PHASE(4);
    if (active && !((PR_SPLIT_ABLATE & 8) && L.epi != EPI_FEATURES)) {
        for (int blk = 0; blk < (two ? 2 : 1); ++blk) {
            const int feat0 = (blk ? cbB : cbA) * 32 + 4 * half;   // first feature of this lane
            const f32x16& m0 = blk ? mB0 : mA0;   // samples 0..31
            const f32x16& m1 = blk ? mB1 : mA1;   // samples 32..63
            if (L.epi == EPI_RELU) {
                store_relu_h(m0, S, r * LDH + feat0);
                store_relu_h(m1, S, (r + 32) * LDH + feat0);
            } else if (L.epi == EPI_ADAIN_RELU) {
                if (park) {
                    store_adain_lds_h(m0, S, r * LDH + feat0, feat0);
                    store_adain_lds_h(m1, S, (r + 32) * LDH + feat0, feat0);
                    continue;
                }
                const int bofs = L.nblk * 32;
                const bool uniform = S.uniform_frame != 0;
                const float* t0 = p.adain + (size_t)S.frame[uniform ? 0 : r] * p.adain_stride + L.adain_off + feat0;
                const float* t1 = p.adain + (size_t)S.frame[uniform ? 0 : r + 32] * p.adain_stride + L.adain_off + feat0;
                store_adain_h(m0, S, r * LDH + feat0, t0, t0 + bofs);
                store_adain_h(m1, S, (r + 32) * LDH + feat0, t1, t1 + bofs);
            } else {
                float* stage = reinterpret_cast<float*>(S.Xh);
                store_stage_h(m0, stage + r * LDSTAGE + feat0);
                store_stage_h(m1, stage + (r + 32) * LDSTAGE + feat0);
            }
        }
    }
    PR_PHASE(5);
    __syncthreads();
    PR_PHASE(6);
}

// (re)computes a network input into columns [0, pad) of the activation planes (see fill_encoding in mlp.hip)
__device__ __forceinline__ void fill_encoding_h(SmemH& S, const MlpParams& p, int din, int octaves, int zero_from, int pad,
                                                const float* octave_weights, bool normalise) {
    const int part = threadIdx.x & 7;
    for (int s = threadIdx.x >> 3; s < STILE_M; s += STHREADS / 8) {
        float v[6];
        for (int a = 0; a < din; ++a) {
            const float x = S.pos[s * 8 + a];
            v[a] = normalise ? __fdiv_rn(x, p.size[a]) : x;
        }
        const int row = s * LDH;
        if (part == 0)
            for (int a = 0; a < din; ++a) split_store(S.Xh, S.Xl, row + a, v[a]);
        if (part == 1)
            for (int j = zero_from; j < pad; ++j) split_store(S.Xh, S.Xl, row + j, 0.f);
        for (int k = part; k < octaves; k += 8) {
            const float f = ldexpf(1.0f, k);
            const float w = octave_weights ? octave_weights[k] : 1.0f;
            const int dst = row + din + k * 2 * din;
            for (int a = 0; a < din; ++a) {
                const float arg = __fmul_rn(f, v[a]);
                float sn = sinf(arg), cs = cosf(arg);
                if (octave_weights) {
                    sn = __fmul_rn(sn, w);
                    cs = __fmul_rn(cs, w);
                }
                split_store(S.Xh, S.Xl, dst + a, sn);
                split_store(S.Xh, S.Xl, dst + din + a, cs);
            }
        }
    }
}

__device__ __forceinline__ void fill_nerf_input_h(SmemH& S, const MlpParams& p) {
    fill_encoding_h(S, p, p.din, p.octaves, p.enc, p.enc_pad, nullptr, p.kind == 0);
}

__device__ __forceinline__ void fill_bender_input_h(SmemH& S, const MlpParams& p) {
    fill_encoding_h(S, p, 3, p.b_octaves, p.benc + p.D, p.bin_pad, p.b_weights, true);
    for (int idx = threadIdx.x; idx < STILE_M * p.D; idx += STHREADS) {
        const int s = idx / p.D, j = idx - s * p.D;
        split_store(S.Xh, S.Xl, s * LDH + p.benc + j, p.deformation[(size_t)S.frame[s] * p.deformation_stride + j]);
    }
}

// dot products of tile row `s` with `nout` (<= 3) weight rows; 8 threads per row, result valid in all 8
__device__ __forceinline__ void row_dots_h(const SmemH& S, int s, const float* w, int width, int wstride, int nout, float* out) {
    const int part = threadIdx.x & 7;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = part; k < width; k += 8) {
        const float x = split_load(S.Xh, S.Xl, s * LDH + k);
        for (int a = 0; a < nout; ++a) acc[a] = fmaf(x, w[a * wstride + k], acc[a]);
    }
    for (int a = 0; a < nout; ++a) {
        float v = acc[a];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        out[a] = v;
    }
}

// ---- sigma-gated feature head (see gated_head in mlp.hip; here a pending row is its hi plane followed by its lo plane) ----
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void write_rows_indirect_h(const SmemH& S, const MlpParams& p) {
    const int tid = threadIdx.x;
    const float* stage = reinterpret_cast<const float*>(S.Xh);
    if ((p.F & 3) == 0) {
        const int f4 = p.F >> 2;
        for (int idx = tid; idx < STILE_M * f4; idx += STHREADS) {
            const int row = idx / f4, c = (idx - row * f4) * 4;
            const int d = S.dest[row];
            if (d >= 0) {
                const float4 v = *reinterpret_cast<const float4*>(stage + row * LDSTAGE + c);
                typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                f32x4_nt nt = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(nt, reinterpret_cast<f32x4_nt*>(p.feat + (size_t)d * p.F + c));
            }
        }
    } else {
        for (int idx = tid; idx < STILE_M * p.F; idx += STHREADS) {
            const int row = idx / p.F, c = idx - row * p.F;
            const int d = S.dest[row];
            if (d >= 0) p.feat[(size_t)d * p.F + c] = stage[row * LDSTAGE + c];
        }
    }
}

template <int TERMS>
__device__ __forceinline__ void head_on_tile_h(SmemH& S, const MlpParams& p, int valid_rows) {
    for (int l = p.n_backbone; l < p.n_layers; ++l) run_layer_h<TERMS>(p.layers[l], S, p, 0);
    write_rows_indirect_h(S, p);
    if (threadIdx.x == 0 && p.head_count) atomicAdd(p.head_count, valid_rows);
    __syncthreads();
}

// rows [0, rows) x both planes between LDS and a pending stack; slot_of(row) < 0 skips the row
template <bool TO_LDS, typename SlotOf>
__device__ __forceinline__ void move_pending_rows(SmemH& S, const MlpParams& p, u32x4_t* stack, int rows, SlotOf slot_of) {
    const int chunks = p.Wpad >> 3;            // 16-byte chunks per plane row
    const int per_row = 2 * chunks;
    for (int idx = threadIdx.x; idx < rows * per_row; idx += STHREADS) {
        const int row = idx / per_row, rem = idx - row * per_row;
        const int plane = rem / chunks, c = rem - plane * chunks;
        const int slot = slot_of(row);
        if (slot < 0) continue;
        u32x4_t* lds = reinterpret_cast<u32x4_t*>((plane ? S.Xl : S.Xh) + row * LDH) + c;
        u32x4_t* glb = stack + (size_t)slot * per_row + plane * chunks + c;
        if (TO_LDS) *lds = *glb; else *glb = *lds;
    }
}

template <int TERMS>
__device__ __forceinline__ int gated_head_h(SmemH& S, const MlpParams& p, int tile_base, int pending) {
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long live = __ballot((S.flags[lane] & 4) != 0);   // the same value in every wave
    const int L = __popcll(live);
    u32x4_t* stack = reinterpret_cast<u32x4_t*>(p.pend_act + (size_t)blockIdx.x * STILE_M * p.Wpad);
    int* pmeta = p.pend_meta + (size_t)blockIdx.x * STILE_M * 2;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (L == 0) {
        __syncthreads();
        return pending;
    }
    if (pending + L >= STILE_M) {
        const int need = STILE_M - L;
        if (tid == 0) S.uniform_frame = 1;
        if (tid < STILE_M) {
            if ((live >> tid) & 1ull) {
                S.dest[tid] = tile_base + tid;
                S.src[tid] = -1;
            } else {
                const int slot = pending - need + __popcll(~live & below);
                S.src[tid] = slot;
                S.dest[tid] = pmeta[2 * slot];
                S.frame[tid] = pmeta[2 * slot + 1];
            }
        }
        __syncthreads();
        if (need) move_pending_rows<true>(S, p, stack, STILE_M, [&](int row) { return S.src[row]; });
        if (tid < STILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
        __syncthreads();
        head_on_tile_h<TERMS>(S, p, STILE_M);
        return pending - need;
    }
    if (tid < STILE_M) {
        int slot = -1;
        if ((live >> tid) & 1ull) {
            slot = pending + __popcll(live & below);
            pmeta[2 * slot] = tile_base + tid;
            pmeta[2 * slot + 1] = S.frame[tid];
        }
        S.src[tid] = slot;
    }
    __syncthreads();
    move_pending_rows<false>(S, p, stack, STILE_M, [&](int row) { return S.src[row]; });
    __syncthreads();
    return pending + L;
}

template <int TERMS>
__device__ __forceinline__ void gated_head_flush_h(SmemH& S, const MlpParams& p, int pending) {
    if (pending <= 0) return;
    const int tid = threadIdx.x;
    u32x4_t* stack = reinterpret_cast<u32x4_t*>(p.pend_act + (size_t)blockIdx.x * STILE_M * p.Wpad);
    const int* pmeta = p.pend_meta + (size_t)blockIdx.x * STILE_M * 2;
    if (tid == 0) S.uniform_frame = 1;
    if (tid < STILE_M) {
        const bool has = tid < pending;
        S.dest[tid] = has ? pmeta[2 * tid] : -1;
        S.frame[tid] = pmeta[2 * (has ? tid : 0) + 1];
    }
    __syncthreads();
    move_pending_rows<true>(S, p, stack, pending, [&](int row) { return row; });
    if (tid < STILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
    __syncthreads();
    head_on_tile_h<TERMS>(S, p, pending);
}

// GROUP: one of several objects evaluated by the same launch (k_mlp_split_group), as in mlp.hip: every tile is claimed from
// the object's counter and a workgroup that finds an object's tiles exhausted moves on to the next object.
template <bool GROUP, int TERMS>
__device__ __forceinline__ void split_tile_loop(const MlpParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemH& S = *reinterpret_cast<SmemH*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = *p.total;
    if (GROUP) {
        __syncthreads();   // every wave has left the previous object's last tile
        if (tid == 0) S.next_tile = atomicAdd(p.tile_counter, 1);
    }
    for (int i = tid; i <= p.Wpad; i += STHREADS) S.head_w[i] = p.sigma_w[i];
    __syncthreads();
    int pending = 0;   // rows on this workgroup's pending stack (sigma-gated head)
    // dynamic tile order, as in k_mlp_mfma: further tiles are claimed from a device counter, one tile ahead
#ifdef PR_MLP_STATIC_TILES
    const bool dynamic_tiles = GROUP;
#else
    const bool dynamic_tiles = GROUP || p.tile_counter != nullptr;
#endif
    for (int tile = GROUP ? S.next_tile : (int)blockIdx.x; tile * STILE_M < total; tile = S.next_tile) {
        const int tile_base = tile * STILE_M;
        PR_PHASE_T0();
        int claimed = 0;      // (claimed late, behind the backbone: see "Tile order" in mlp.hip)
        if (tid == 0) S.uniform_frame = 1;
        if (tid < STILE_M) {
            const int idx = tile_base + tid;
            const bool valid = idx < total;
            const int src = valid ? idx : tile_base;
            const int flat = p.rec_flat[src];
            const int frame = flat / p.samples_per_frame;
            S.flat[tid] = flat;
            S.frame[tid] = frame;
            S.flags[tid] = valid ? 3 : 0;
            if (p.kind == 0) {
                S.pos[tid * 8 + 0] = p.rec_pos[(size_t)src * 3 + 0];
                S.pos[tid * 8 + 1] = p.rec_pos[(size_t)src * 3 + 1];
                S.pos[tid * 8 + 2] = p.rec_pos[(size_t)src * 3 + 2];
            } else {
                const int ray = (flat - frame * p.samples_per_frame) / p.positions;
                const ObjRay rr = object_ray(p.w2o + (size_t)frame * p.w2o_stride, p.ray_origins + (size_t)frame * 3,
                                             p.ray_directions + ((size_t)frame * p.rays + ray) * 3);
                const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rr.d[0], rr.d[0]), __fmul_rn(rr.d[1], rr.d[1])),
                                                  __fmul_rn(rr.d[2], rr.d[2])));
                for (int a = 0; a < 3; ++a) {
                    S.pos[tid * 8 + a] = __fdiv_rn(rr.o[a], p.size[a]);
                    S.pos[tid * 8 + 3 + a] = __fdiv_rn(rr.d[a], nrm);
                }
            }
        }
        __syncthreads();
        if (tid < STILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
        PR_PHASE(0);

        if (p.has_bender) {
            fill_bender_input_h(S, p);
            __syncthreads();
            for (int l = 0; l < p.b_count; ++l) run_layer_h<TERMS>(p.b_layers[l], S, p, /*input_kind=*/1);
            for (int s = tid >> 3; s < STILE_M; s += STHREADS / 8) {
                float out[3];
                row_dots_h(S, s, p.b_out, p.BWpad, p.BWpad, 3, out);
                if ((tid & 7) != 0) continue;
                float d[3], bent[3];
                for (int a = 0; a < 3; ++a) {
                    const float x = S.pos[s * 8 + a];
                    float dl = __fmul_rn(out[a], p.size[a]);
                    dl = nan_max(dl, __fsub_rn(p.lo[a], x));
                    dl = nan_min(dl, __fsub_rn(p.hi[a], x));
                    if (p.canonical) dl = __fmul_rn(dl, 0.0f);
                    d[a] = dl;
                    bent[a] = __fadd_rn(x, dl);
                    S.pos[s * 8 + a] = bent[a];
                }
                if (S.flags[s] & 1) {
                    if (p.delta_dense)
                        for (int a = 0; a < 3; ++a) p.delta_dense[(size_t)S.flat[s] * 3 + a] = d[a];
                    if (p.dispmag)
                        p.dispmag[S.flat[s]] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])),
                                                               __fmul_rn(d[2], d[2])));
                    if (!in_box(bent[0], bent[1], bent[2], p.lo, p.hi)) S.flags[s] &= ~2;
                }
            }
            __syncthreads();
        }
        PR_PHASE(1);

        fill_nerf_input_h(S, p);
        __syncthreads();
        PR_PHASE(2);
        for (int l = 0; l < p.n_backbone; ++l) run_layer_h<TERMS>(p.layers[l], S, p, 0);
        PR_PHASE(15);
        if (dynamic_tiles && tid == 0) claimed = atomicAdd(p.tile_counter, 1);

        if (p.kind == 0) {
            for (int s = tid >> 3; s < STILE_M; s += STHREADS / 8) {
                float sg;
                row_dots_h(S, s, S.head_w, p.Wpad, p.Wpad, 1, &sg);
                if ((tid & 7) == 0 && (S.flags[s] & 3) == 3) {
                    const float sv = p.in_scene[(size_t)S.frame[s] * p.in_scene_stride] ? sg + S.head_w[p.Wpad] : p.empty_alpha;
                    p.sigma[S.flat[s]] = sv;
                    if (!(sv <= 0.f)) S.flags[s] |= 4;
                }
            }
        } else if (tid < STILE_M) {
            if (S.flags[tid] & 1) {
                const bool present = p.in_scene[(size_t)S.frame[tid] * p.in_scene_stride] != 0;
                p.sigma[S.flat[tid]] = present ? 10.0f : p.empty_alpha;
                if (present || !(p.empty_alpha <= 0.f)) S.flags[tid] |= 4;
            }
        }

        if (tid == 0) S.next_tile = GROUP ? claimed : (dynamic_tiles ? (int)gridDim.x + claimed : tile + (int)gridDim.x);   // read at the end of the tile
        PR_PHASE(7);
        if (p.gate) {
            __syncthreads();   // the liveness bits are complete
            pending = gated_head_h<TERMS>(S, p, tile_base, pending);
            PR_PHASE(8);
            continue;
        }
        for (int l = p.n_backbone; l < p.n_layers; ++l) run_layer_h<TERMS>(p.layers[l], S, p, 0);
        PR_PHASE(15);

        // feature rows: the last layer staged an fp32 tile over the activation planes
        {
            const float* stage = reinterpret_cast<const float*>(S.Xh);
            if ((p.F & 3) == 0) {
                const int f4 = p.F >> 2;
                for (int idx = tid; idx < STILE_M * f4; idx += STHREADS) {
                    const int row = idx / f4, c = (idx - row * f4) * 4;
                    const int fl = S.flags[row];
                    if (fl & 1) {
                        float4 v = *reinterpret_cast<const float4*>(stage + row * LDSTAGE + c);
                        if (!(fl & 2)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        // streamed once: keep the rows from evicting the weight fragments in L2
                        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                        f32x4_nt nt = {v.x, v.y, v.z, v.w};
                        __builtin_nontemporal_store(nt, reinterpret_cast<f32x4_nt*>(p.feat + (size_t)(tile_base + row) * p.F + c));
                    }
                }
            } else {
                for (int idx = tid; idx < STILE_M * p.F; idx += STHREADS) {
                    const int row = idx / p.F, c = idx - row * p.F;
                    const int fl = S.flags[row];
                    if (fl & 1) p.feat[(size_t)(tile_base + row) * p.F + c] = (fl & 2) ? stage[row * LDSTAGE + c] : 0.f;
                }
            }
        }
        __syncthreads();
        PR_PHASE(8);
    }
    if (p.gate) gated_head_flush_h<TERMS>(S, p, pending);
}

__device__ __forceinline__ void claim_matrix_priority() {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SmemH& S = *reinterpret_cast<SmemH*>(smem_raw);
#ifdef PR_EQUAL_TILE_PRIORITY
    if (threadIdx.x == 0) S.matrix_priority = 0;
#else
    if (threadIdx.x == 0) S.matrix_priority = cu_arrival_parity();
#endif
    // (published by the tile loop's first barrier)
}

__global__ __launch_bounds__(STHREADS, SBLOCKS_PER_CU) void k_mlp_split(MlpParams p) {
    claim_matrix_priority();
    split_tile_loop<false, 3>(p);
}
// one copy of the tile loop per job slot: parameters as kernel arguments at constant offsets (see k_mlp_mfma_group)
__global__ __launch_bounds__(STHREADS, SBLOCKS_PER_CU) void k_mlp_split_group(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                                int count) {
    claim_matrix_priority();
    split_tile_loop<true, 3>(j0);
    if (count > 1) split_tile_loop<true, 3>(j1);
    if (count > 2) split_tile_loop<true, 3>(j2);
    if (count > 3) split_tile_loop<true, 3>(j3);
}
// PR_PRECISION_F16: the same tile loop with the hi x hi product only (one fp16 MFMA per step instead of three)
__global__ __launch_bounds__(STHREADS, SBLOCKS_PER_CU) void k_mlp_f16(MlpParams p) {
    claim_matrix_priority();
    split_tile_loop<false, 1>(p);
}
__global__ __launch_bounds__(STHREADS, SBLOCKS_PER_CU) void k_mlp_f16_group(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                              int count) {
    claim_matrix_priority();
    split_tile_loop<true, 1>(j0);
    if (count > 1) split_tile_loop<true, 1>(j1);
    if (count > 2) split_tile_loop<true, 1>(j2);
    if (count > 3) split_tile_loop<true, 1>(j3);
}

#if PR_SPLIT_ABLATE & 64
// phase timing build: cumulative shader-clock Mcycles of thread 0 per phase, summed over the workgroups (100 MHz clock)
static void dump_phase_cycles(hipStream_t s) {
    unsigned long long now[16];
    hipStreamSynchronize(s);
    hipMemcpyFromSymbol(now, HIP_SYMBOL(g_phase_cycles), sizeof(now));
    fprintf(stderr, "[split phases, cumulative Mcycles]");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " p%d=%.2f", i, (double)now[i] * 1e-6);
    fprintf(stderr, "\n");
}
#endif

int launch_mlp_split_group(const MlpParams* host_jobs, const int* max_rows, int count, int terms, hipStream_t s) {
    auto* const kernel = terms == 1 ? k_mlp_f16_group : k_mlp_split_group;
    PR_REQUIRE(count >= 1, "grouped MLP launch: no jobs");
    static thread_local MlpGroupParams g;     // 18 KB: not on the stack
    for (int begin = 0; begin < count; begin += MLP_GROUP_MAX) {
        const int n = count - begin < MLP_GROUP_MAX ? count - begin : MLP_GROUP_MAX;
        long max_tiles = 0;
        for (int j = 0; j < n; ++j) {
            const MlpParams& q = host_jobs[begin + j];
            PR_REQUIRE(q.phase == 0 && q.tile_counter, "grouped MLP launch: evaluation launches with a tile counter only");
            PR_REQUIRE(!q.gate || (q.pend_act && q.pend_meta), "gated head: pending buffers missing");
            max_tiles += ((long)max_rows[begin + j] + STILE_M - 1) / STILE_M;
            g.jobs[j] = q;
        }
        if (max_tiles <= 0) continue;
        int cu_count = 0;
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(kernel), (int)sizeof(SmemH), &cu_count));
        int resident = cu_count * SBLOCKS_PER_CU;
        if (resident > MAX_RESIDENT_TILES) resident = MAX_RESIDENT_TILES;
        const int grid = max_tiles < resident ? (int)max_tiles : resident;
        ProfileScope scope(0, s);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(STHREADS), sizeof(SmemH), s, g.jobs[0], g.jobs[1], g.jobs[2], g.jobs[3], n);
        PR_LAUNCH_CHECK();
#if PR_SPLIT_ABLATE & 64
        dump_phase_cycles(s);
#endif
    }
    return PR_OK;
}

int launch_mlp_split(const MlpParams& p, int max_rows, int terms, hipStream_t s) {
    if (max_rows <= 0) return PR_OK;
    auto* const kernel = terms == 1 ? k_mlp_f16 : k_mlp_split;
    const int max_tiles = (max_rows + STILE_M - 1) / STILE_M;
    int cu_count = 0;
    PR_TRY(prepare_kernel(reinterpret_cast<const void*>(kernel), (int)sizeof(SmemH), &cu_count));
    int resident = cu_count * SBLOCKS_PER_CU;
    if (resident > MAX_RESIDENT_TILES) resident = MAX_RESIDENT_TILES;   // the pending stacks of the gated head are sized for this
    const int grid = max_tiles < resident ? max_tiles : resident;
    PR_REQUIRE(!p.gate || (p.pend_act && p.pend_meta), "gated head: pending buffers missing");
    ProfileScope scope(0, s);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(STHREADS), sizeof(SmemH), s, p);
    PR_LAUNCH_CHECK();
#if PR_SPLIT_ABLATE & 64
    dump_phase_cycles(s);
#endif
    return PR_OK;
}

// fp16 matrix-pipe ceiling probe: the split kernel's own issue pattern (two main and two correction
// accumulators, six MFMAs per step) with register-only operands
__global__ __launch_bounds__(512) void k_probe_mfma_f16(int iterations, float* sink, int random_operands) {
    f32x16 m0, m1, c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        m0[i] = 0.f;
        m1[i] = 1.f;
        c0[i] = 0.f;
        c1[i] = 1.f;
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    union Frag { u32x4 u; f16x8 h; };
    Frag a, b;
    unsigned int x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int k = 0; k < 4; ++k) {
        a.u[k] = 0x3C003C00u;   // 1.0, 1.0
        b.u[k] = 0x38003800u;   // 0.5, 0.5
    }
    for (int it = 0; it < iterations; ++it) {
        if (random_operands) {
            for (int k = 0; k < 4; ++k) {
                x ^= x << 13; x ^= x >> 17; x ^= x << 5;
                a.u[k] = (x & 0x83FF83FFu) | 0x38003800u;   // halves in +-[0.5, 1)
                b.u[k] = ((x * 2654435761u) & 0x83FF83FFu) | 0x38003800u;
            }
        }
        PR_MFMA16(m0, a.h, b.h);
        PR_MFMA16(m1, b.h, a.h);
        PR_MFMA16(c0, a.h, a.h);
        PR_MFMA16(c1, b.h, b.h);
        PR_MFMA16(c0, b.h, a.h);
        PR_MFMA16(c1, a.h, b.h);
        if ((it & 63) == 63) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                m0[i] *= 0.001f;
                m1[i] *= 0.001f;
                c0[i] *= 0.001f;
                c1[i] *= 0.001f;
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += m0[i] + m1[i] + c0[i] + c1[i];
    if (t == 123.456f) sink[0] = t;
}

}  // namespace pr

extern "C" int pr_probe_mfma_f16(int32_t iterations, int32_t random_operands, double* tflops, double* milliseconds, void* stream) {
    PR_REQUIRE(iterations > 0 && tflops, "pr_probe_mfma_f16: bad argument");
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    hipStream_t s = (hipStream_t)stream;
    float* sink = nullptr;
    PR_CHECK_HIP(hipMalloc(&sink, sizeof(float)));
    hipEvent_t e0, e1;
    PR_CHECK_HIP(hipEventCreate(&e0));
    PR_CHECK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(pr::k_probe_mfma_f16, dim3(cus), dim3(512), 0, s, 16, sink, random_operands);  // warm-up
    PR_CHECK_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(pr::k_probe_mfma_f16, dim3(cus), dim3(512), 0, s, iterations, sink, random_operands);
    PR_CHECK_HIP(hipEventRecord(e1, s));
    PR_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)cus * 8 * (double)iterations * 6.0 * (2.0 * 32 * 32 * 16);
    *tflops = flop / (ms * 1e-3) / 1e12;
    if (milliseconds) *milliseconds = ms;
    PR_CHECK_HIP(hipEventDestroy(e0));
    PR_CHECK_HIP(hipEventDestroy(e1));
    PR_CHECK_HIP(hipFree(sink));
    return PR_OK;
}
