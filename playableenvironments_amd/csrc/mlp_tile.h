// Tile machinery of the fused MLP kernels (mlp.hip) and of the fused backward kernels (train_bwd.hip): the LDS image of a
// 64-sample tile, one layer on the matrix cores (run_layer), the positional encodings and the tile <-> HBM row movers.
// Device code only; included by the translation units that instantiate kernels over it.
#pragma once
#include "pr_common.h"

#include <cstddef>

namespace pr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Shared memory image of one tile
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_SIGMA = MAX_WIDTH + 8;   // sigma weights (Wpad) + bias
constexpr int HEAD_BENDER = 0;              // the 3-row bender head is read from L2 (players are few)
struct Smem {
    int uniform_frame;       // every row of the tile belongs to the same frame
    int next_tile;           // the tile this workgroup claimed for its next iteration (dynamic tile order)
    int tile_max[2];         // split-precision training forward: bit patterns of the largest |entry| of the tile in X (two alternating words)
    float head_w[HEAD_SIGMA + HEAD_BENDER];
    float X[TILE_M * LDX];   // activations; columns [0, K) also hold a layer's input encoding while it is needed
    float pos[TILE_M * 8];   // object-frame position (3) / skybox input (6)
    int flat[TILE_M];
    int frame[TILE_M];
    int flags[TILE_M];       // bit 0: row holds a real sample; bit 1: it passed every AABB test; bit 2 (sigma-gated
                             // head only): its density is not <= 0, i.e. its feature row can reach a compositing sum
    int dest[TILE_M];        // gated head: compact feature row a tile row is written to (-1: none)
    int src[TILE_M];         // gated head: slot of the workgroup's pending stack a tile row is exchanged with (-1: none)
#ifdef PR_MLP_TIMING
    unsigned long long phase_acc[16];   // phase timing build: thread 0's clock deltas, flushed once per tile loop
#endif
};
static_assert(sizeof(Smem) * MLP_BLOCKS_PER_CU <= 160 * 1024 - MLP_BLOCKS_PER_CU * 1024, "the workgroups of one CU must fit its LDS");
static_assert(offsetof(Smem, head_w) % 16 == 0 && offsetof(Smem, X) % 16 == 0 && offsetof(Smem, pos) % 16 == 0,
              "16-byte LDS reads of the activation tile");


__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// positional encoding element j of input v[din]  (model/positional_encoder.py:54-64)
__device__ __forceinline__ float pe_element(const float* v, int din, int enc, int j, const float* octave_weights) {
    if (j < din) return v[j];
    if (j >= enc) return 0.f;
    const int jj = j - din;
    const int k = jj / (2 * din);
    const int rem = jj - k * 2 * din;
    const int fn = rem / din;
    const int ax = rem - fn * din;
    const float arg = __fmul_rn(ldexpf(1.0f, k), v[ax]);
#ifdef PR_FAST_TRIG_ABLATION
    float e = fn ? __cosf(arg) : __sinf(arg);
#else
    float e = fn ? cosf(arg) : sinf(arg);
#endif
    if (octave_weights) e = __fmul_rn(e, octave_weights[k]);
    return e;
}

// Epilogue stores of one 32x32 accumulator block.  Lane (r, half) holds column r and the rows
// (i & 3) + 8 (i >> 2) + 4 half, i = 0..15: constant LDS offsets from the lane's base pointer.
#define PR_ACC_ROW(i) (((i) & 3) + 8 * ((i) >> 2))

__device__ __forceinline__ void store_relu(const f32x16& acc, float* base) {
#pragma unroll
    for (int i = 0; i < 16; ++i) base[PR_ACC_ROW(i) * LDX] = acc[i] > 0.f ? acc[i] : 0.f;
}

__device__ __forceinline__ void store_adain_uniform(const f32x16& acc, float* base, float g, float b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = fmaf(acc[i], g, b);
        base[PR_ACC_ROW(i) * LDX] = v > 0.f ? v : 0.f;
    }
}

__device__ __forceinline__ void store_adain_rows(const f32x16& acc, Smem& S, const MlpParams& p, int row0, int col,
                                             int goff, int boff) {
    // rare path (tiles that straddle two frames): kept out of line and in chunks of four rows so that
    // it does not inflate the register allocation of the hot loops
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float g[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* tab = p.adain + (size_t)S.frame[row0 + PR_ACC_ROW(4 * c + i)] * p.adain_stride;
            g[i] = tab[goff];
            b[i] = tab[boff];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = fmaf(acc[4 * c + i], g[i], b[i]);
            S.X[(row0 + PR_ACC_ROW(4 * c + i)) * LDX + col] = v > 0.f ? v : 0.f;
        }
    }
}

__device__ __forceinline__ void store_plain(const f32x16& acc, float* base) {
#pragma unroll
    for (int i = 0; i < 16; ++i) base[PR_ACC_ROW(i) * LDX] = acc[i];
}

#ifdef PR_MLP_TIMING
// phase timing build: thread 0 of every workgroup accumulates shader-clock deltas per phase
__device__ unsigned long long g_mlp_phase[16];
#define PR_PHASE_T0() unsigned long long _pt = __builtin_amdgcn_s_memtime()
#define PR_PHASE(idx)                                                                      \
    do {                                                                                   \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                        \
        if (threadIdx.x == 0) S.phase_acc[idx] += _n - _pt;                                \
        _pt = _n;                                                                          \
    } while (0)
// (global atomics per phase queue in front of the wave's weight loads and distort what they time: LDS sums, one flush per loop)
#define PR_PHASE_COUNT(idx, n) do { if (threadIdx.x == 0) S.phase_acc[idx] += (n); } while (0)
#define PR_PHASE_BEGIN()                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0)                                                              \
            for (int _i = 0; _i < 16; ++_i) S.phase_acc[_i] = 0;                           \
    } while (0)
#define PR_PHASE_FLUSH()                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0)                                                              \
            for (int _i = 0; _i < 16; ++_i)                                                \
                if (S.phase_acc[_i]) atomicAdd(&g_mlp_phase[_i], S.phase_acc[_i]);         \
    } while (0)
#else
#define PR_PHASE_BEGIN() do {} while (0)
#define PR_PHASE_COUNT(idx, n) do {} while (0)
#define PR_PHASE_FLUSH() do {} while (0)
#define PR_PHASE_T0() do {} while (0)
#define PR_PHASE(idx) do {} while (0)
#endif

struct EncRegs;
__device__ __forceinline__ void fill_nerf_input(Smem& S, const MlpParams& p, EncRegs& regs, bool replay, int* max_slot = nullptr);
__device__ __forceinline__ void fill_bender_input(Smem& S, const MlpParams& p, EncRegs& regs, bool replay, int* max_slot = nullptr);

// One layer on the tile.  All threads of the workgroup call it (workgroup barriers inside).
//
// Geometry: the 64-sample tile is two 32-row MFMA blocks; wave w of the FOUR waves owns the 32-column blocks w and
// w + 4 of the layer for both row blocks - four accumulators that share two activation and two weight fragments
// per K step (16 MFMAs between operand loads).  Two such workgroups are resident per CU (LDS ~70 KB each), so the
// serial phases of one tile (record loads, encodings, epilogues, barriers) overlap the other tile's matrix work.
//
// `input_kind` says what a segment with src == 1 (the layer's input encoding) means: 0 = NeRF input, 1 = ray-bender
// input.  The first layer finds it in X already; the skip layer's second segment re-computes it into X[:, 0:K).
#define PR_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define PR_MFMA4(acc, a, b) \
    PR_MFMA(acc, a.x, b.x); \
    PR_MFMA(acc, a.y, b.y); \
    PR_MFMA(acc, a.z, b.z); \
    PR_MFMA(acc, a.w, b.w)

// BITS (training forward): the ReLU epilogue also leaves the layer's ReLU mask behind as one 64-bit word per column
// (`bits_out[col]`, bit r = tile row r active) - what the backward chain's lane that owns the column reads instead of activations
// STATS (training forward, the raw head layers): the plain-store epilogue also adds the layer's per-column sum and sum of squares
// over the tile's rows that enter the batch statistics to the lane's running sums (`stats`), straight from the accumulators
struct ColumnStats { double s1[2], s2[2]; };      // this lane's columns of the blocks `wave` / `wave + 4`
// SPLIT (training forward with PR_FLAG_SPLIT_BACKWARD): the segments are fp16-pair packings of w x 2^8, the products run on three
// fp16 MFMAs per 16 K-values on the operand tile x a per-tile power of two (tile_products_f16x3_lean; *cur_slot names the tile's
// tile_max word and is moved on to the word of the tile this layer writes) - same accumulator layout, so every epilogue is unchanged
template <bool BWD = false, bool BITS = false, bool STATS = false, bool SPLIT = false>
__device__ __forceinline__ void run_layer(const Layer& L, Smem& S, const MlpParams& p, int tile_base, int input_kind, EncRegs& enc,
                                          const BwdEpilogue* bwd = nullptr, unsigned long long* bits_out = nullptr,
                                          ColumnStats* stats = nullptr, int* cur_slot = nullptr);

// Positional encoding of every tile row into columns [0, pad) of X (model/positional_encoder.py:54-64):
//   [v, sin(2^0 v), cos(2^0 v), sin(2^1 v), ...], each block `din` wide; columns [zero_from, pad) are zeroed.
// There is no separate encoding buffer: the encoding is (re)computed into the activation tile right before a
// layer consumes it - layer 0, and once more for the second K segment of the skip layer - which keeps the
// workgroup at ~70 KB of LDS, i.e. two independent tiles per CU.
// 8 threads per row, thread `part` takes the octaves part, part + 8, ... (one sincos per axis).
// A thread's share of a network input - two tile rows, the octaves part and part + 8, up to 6 input dimensions - kept in
// registers between the first use of the input (layer 0) and its second (the skip layer's second K segment): the
// replay writes the values back into X without evaluating the sines and cosines again (VALU work of one resident
// tile is not hidden behind the other tile's matrix work; it adds to the kernel time).
constexpr int ENC_ROWS = TILE_M / (MLP_THREADS / 8);
constexpr int ENC_SLOTS = (PR_MAX_OCTAVES + 7) / 8;
constexpr int ENC_DIMS = 3;        // positions; the 6-dimensional skybox input is simply evaluated twice
// Largest |entry| of a tile a producer writes into X, for the fp16-pair products of the split-precision training kernels: every
// thread folds what it writes into a running maximum and commits it to one of two LDS words (wave reduction, one atomicMax per
// wave; non-negative floats order like their bit patterns).  The product that reads the tile turns the word into a power of two
// that puts the tile's largest entry just under 2^15 - without it, operands below 0.25 have their lo half in fp16's subnormal range
// (absolute floor 2^-25), which a train-mode BatchNorm over small activations amplifies into 1e-3 gradient errors (randomized
// backward sweep, seed 7 case 18: the skybox model).  A product clears the word it does NOT read; the next producer writes that one.
__device__ __forceinline__ void commit_tile_max(float m, int* slot) {
    if (slot == nullptr) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(m));
}
// scale = 2^k with max x 2^k in [2^14, 2^15); k = 0 for an all-zero (or non-finite) tile.  k <= 120: a tile whose largest entry is
// below 2^-106 (gradients behind an opaque surface: transmittances of 1e-35) would ask for a scale beyond fp32's range - 2^k = inf,
// 0 x inf = NaN in every product of the tile; such a tile is scaled as far as fp32 goes and keeps what precision is left
__device__ __forceinline__ int tile_scale_log2(int bits) {
    const int e = ((bits >> 23) & 255);
    const int k = 14 - (e - 127);
    return (e == 0 || e == 255) ? 0 : (k < 120 ? k : 120);
}

struct EncRegs {
    float raw[ENC_ROWS][ENC_DIMS];
    float sc[ENC_ROWS][ENC_SLOTS][2 * ENC_DIMS];
};
static_assert(ENC_ROWS == 2, "fill_encoding keeps two rows per thread");

__device__ __forceinline__ void fill_encoding(Smem& S, const MlpParams& p, int din, int octaves, int zero_from, int pad,
                                              const float* octave_weights, bool normalise, EncRegs& regs, bool replay, float& biggest) {
    const int part = threadIdx.x & 7;
    if (din != ENC_DIMS) {
        for (int s = threadIdx.x >> 3; s < TILE_M; s += MLP_THREADS / 8) {
            float v[6];
            for (int a = 0; a < din; ++a) {
                const float x = S.pos[s * 8 + a];
                v[a] = normalise ? __fdiv_rn(x, p.size[a]) : x;
            }
            float* row = S.X + s * LDX;
            if (part == 0)
                for (int a = 0; a < din; ++a) {
                    row[a] = v[a];
                    biggest = fmaxf(biggest, fabsf(v[a]));
                }
            if (part == 1)
                for (int q = zero_from; q < pad; ++q) row[q] = 0.f;
            for (int k = part; k < octaves; k += 8) {
                const float f = ldexpf(1.0f, k);
                const float w = octave_weights ? octave_weights[k] : 1.0f;
                float* dst = row + din + k * 2 * din;
                for (int a = 0; a < din; ++a) {
                    const float arg = __fmul_rn(f, v[a]);
                    float sn = sinf(arg), cs = cosf(arg);
                    if (octave_weights) {
                        sn = __fmul_rn(sn, w);
                        cs = __fmul_rn(cs, w);
                    }
                    dst[a] = sn;
                    dst[din + a] = cs;
                    biggest = fmaxf(biggest, fmaxf(fabsf(sn), fabsf(cs)));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < ENC_ROWS; ++j) {
        const int s = (threadIdx.x >> 3) + j * (MLP_THREADS / 8);
        float* row = S.X + s * LDX;
        if (!replay) {
#pragma unroll
            for (int a = 0; a < ENC_DIMS; ++a) {
                const float x = S.pos[s * 8 + a];
                regs.raw[j][a] = normalise ? __fdiv_rn(x, p.size[a]) : x;
            }
        }
        if (part == 0) {
#pragma unroll
            for (int a = 0; a < ENC_DIMS; ++a) {
                row[a] = regs.raw[j][a];
                biggest = fmaxf(biggest, fabsf(regs.raw[j][a]));
            }
        }
        if (part == 1)
            for (int q = zero_from; q < pad; ++q) row[q] = 0.f;
#pragma unroll
        for (int slot = 0; slot < ENC_SLOTS; ++slot) {
            const int k = part + 8 * slot;
            if (k < octaves) {
                float* dst = row + ENC_DIMS + k * 2 * ENC_DIMS;
                if (!replay) {
                    const float f = ldexpf(1.0f, k);
                    const float w = octave_weights ? octave_weights[k] : 1.0f;
#pragma unroll
                    for (int a = 0; a < ENC_DIMS; ++a) {
                        const float arg = __fmul_rn(f, regs.raw[j][a]);
                        float sn = sinf(arg), cs = cosf(arg);
                        if (octave_weights) {
                            sn = __fmul_rn(sn, w);
                            cs = __fmul_rn(cs, w);
                        }
                        regs.sc[j][slot][a] = sn;
                        regs.sc[j][slot][ENC_DIMS + a] = cs;
                    }
                }
#pragma unroll
                for (int a = 0; a < 2 * ENC_DIMS; ++a) {
                    dst[a] = regs.sc[j][slot][a];
                    biggest = fmaxf(biggest, fabsf(regs.sc[j][slot][a]));
                }
            }
        }
    }
}

// input of the NeRF: PE of the (bent, normalised) position / of the skybox's [o / size, d / |d|]
// (max_slot: see commit_tile_max - the split-precision training forward only)
__device__ __forceinline__ void fill_nerf_input(Smem& S, const MlpParams& p, EncRegs& regs, bool replay, int* max_slot) {
    float biggest = 0.f;
    fill_encoding(S, p, p.din, p.octaves, p.enc, p.enc_pad, nullptr, p.kind == 0, regs, replay, biggest);
    commit_tile_max(biggest, max_slot);
}

// input of the ray bender: [annealed PE(x / size) | deformation code of the sample's frame], zero padded
__device__ __forceinline__ void fill_bender_input(Smem& S, const MlpParams& p, EncRegs& regs, bool replay, int* max_slot) {
    float biggest = 0.f;
    fill_encoding(S, p, /*din=*/3, p.b_octaves, p.benc + p.D, p.bin_pad, p.b_weights, /*normalise=*/true, regs, replay, biggest);
    for (int idx = threadIdx.x; idx < TILE_M * p.D; idx += MLP_THREADS) {
        const int s = idx / p.D, j = idx - s * p.D;
        const float v = p.deformation[(size_t)S.frame[s] * p.deformation_stride + j];
        S.X[s * LDX + p.benc + j] = v;
        biggest = fmaxf(biggest, fabsf(v));
    }
    commit_tile_max(biggest, max_slot);
}

// ``drain`` (backward chain): the operand tile in X is also WRITTEN OUT to global memory while a product runs - one 16-byte chunk
// per thread and loop iteration, so that the 64 KB of a tile reach the memory system spread over the K loop
struct Drain {
    float* dst;            // row `tile_base` of the (cap, ld) destination; NULL: nothing to write
    int ld, w4, rows_valid;
};
__device__ __forceinline__ void drain_chunk(const Drain& d, const float* X, int it) {
    const int idx = threadIdx.x + it * MLP_THREADS;
    int row = idx / d.w4;
    const int c = (idx - row * d.w4) * 4;
    // no branch in the K loop (a branch ends the basic block, and with it the scheduling groups that keep the operand requests of the
    // next step in front of this step's MFMAs): a thread whose row lies beyond the tile's real rows writes the LAST real row's chunk
    // of its columns again - the same bytes its owner writes (X is read-only while a product runs; a tile has >= 1 real row)
    row = row < d.rows_valid ? row : d.rows_valid - 1;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *reinterpret_cast<const v4f*>(X + row * LDX + c);
    // streamed once (the weight-gradient launch reads the stack from memory): non-temporal, so that the rows neither evict the weight
    // fragments from L2 nor wait for an L2 line - the loads of the following steps stand behind these stores in the wave's
    // request queue (one in-order counter on gfx9), and their waits end when the stores are acknowledged (measured: the NeRF chain
    // of the step 1.105 -> 1.051 ms; -DPR_DRAIN_PLAIN is the plain-store measurement build)
#ifdef PR_DRAIN_PLAIN
    *as_global(reinterpret_cast<v4f*>(d.dst + (size_t)row * d.ld + c)) = v;
#else
    __builtin_nontemporal_store(v, as_global(reinterpret_cast<v4f*>(d.dst + (size_t)row * d.ld + c)));
#endif
}

// The same write-out with the chunk index kept as a (row, column) cursor: drain_chunk's `idx / w4` is a division by a run-time value,
// ~20 VALU instructions per K step of a loop whose VALU work (the operand split) is what bounds it; the cursor advances by
// MLP_THREADS chunks per call with one carry (same chunks in the same order as drain_chunk(d, X, 0), (d, X, 1), ...)
#ifndef PR_DRAIN_CURSOR
#define PR_DRAIN_CURSOR 1   // 0: drain_chunk's division in every K step (A/B builds)
#endif
struct DrainCursor { int row, c4, drow, dc4; };
__device__ __forceinline__ DrainCursor drain_begin(const Drain& d) {
    DrainCursor k;
    k.row = (int)threadIdx.x / d.w4;
    k.c4 = (int)threadIdx.x - k.row * d.w4;
    k.drow = MLP_THREADS / d.w4;
    k.dc4 = MLP_THREADS - k.drow * d.w4;
    return k;
}
__device__ __forceinline__ void drain_next(const Drain& d, const float* X, DrainCursor& k) {
    const int row = k.row < d.rows_valid ? k.row : d.rows_valid - 1;      // (branch-free: see drain_chunk)
    const int c = 4 * k.c4;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *reinterpret_cast<const v4f*>(X + row * LDX + c);
#ifdef PR_DRAIN_PLAIN
    *as_global(reinterpret_cast<v4f*>(d.dst + (size_t)row * d.ld + c)) = v;
#else
    __builtin_nontemporal_store(v, as_global(reinterpret_cast<v4f*>(d.dst + (size_t)row * d.ld + c)));
#endif
    k.c4 += k.dc4;
    k.row += k.drow;
    const int wrap = k.c4 >= d.w4 ? 1 : 0;
    k.c4 -= wrap ? d.w4 : 0;
    k.row += wrap;
}

// The same product in SPLIT precision (PR_FLAG_SPLIT_BACKWARD; `sg.w` then points at the bf16-triple packing of the segment,
// k_pack kind 3): every fp32 operand as three bf16 terms, x = b1 + b2 + b3 exactly, a product as the six bf16 MFMAs whose terms
// are >= 2^-16 of it (see k_gemm_tn_all_bf16 in gemm.hip) - 16 K-values retire in 6 x 32 cycles where the fp32 pipe needs
// 8 x 64.  The operand tile stays fp32 in X (it is also the gradient that is written out): a lane reads its eight consecutive
// K-values of a step (two 16-byte LDS reads) and splits them in registers, behind the MFMAs of the previous step.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PR_MFMA_BF16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

struct Frag3 { bf16x8 p[3]; };
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// (plain v_sub_f32: hipcc packs adjacent subtractions into v_pk_add_f32, which is slow beside MFMAs)
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// two values -> their three bf16 terms (round to nearest even: v_cvt_pk_bf16_f32), x = b1 + b2 + b3 with |b2| <= 2^-9 |x|,
// |b3| <= 2^-18 |x| and residuals of either sign - the dropped product terms are below one fp32 rounding and unbiased (with
// truncated terms they were up to 2^-20 and all of one sign: a systematic error the float64 arbitration caught)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned int& p1, unsigned int& p2, unsigned int& p3) {
    const f32x2_t v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = {sub_f32(x0, __uint_as_float(p1 << 16)), sub_f32(x1, __uint_as_float(p1 & 0xffff0000u))};
    p2 = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t q = {sub_f32(r[0], __uint_as_float(p2 << 16)), sub_f32(r[1], __uint_as_float(p2 & 0xffff0000u))};
    p3 = __builtin_bit_cast(unsigned int, __builtin_convertvector(q, bf16x2_t));
}
__device__ __forceinline__ Frag3 split_fragment(const float4& lo, const float4& hi) {
    unsigned int a[4], b[4], c[4];
    split_pair(lo.x, lo.y, a[0], b[0], c[0]);
    split_pair(lo.z, lo.w, a[1], b[1], c[1]);
    split_pair(hi.x, hi.y, a[2], b[2], c[2]);
    split_pair(hi.z, hi.w, a[3], b[3], c[3]);
    Frag3 f;
    const u32x4 wa = {a[0], a[1], a[2], a[3]}, wb = {b[0], b[1], b[2], b[3]}, wc = {c[0], c[1], c[2], c[3]};
    f.p[0] = __builtin_bit_cast(bf16x8, wa);
    f.p[1] = __builtin_bit_cast(bf16x8, wb);
    f.p[2] = __builtin_bit_cast(bf16x8, wc);
    return f;
}
// six MFMAs of one 32 x 32 block, smallest terms first
__device__ __forceinline__ void mfma6(f32x16& acc, const Frag3& x, const bf16x8& w1, const bf16x8& w2, const bf16x8& w3) {
    PR_MFMA_BF16(acc, x.p[1], w2);
    PR_MFMA_BF16(acc, x.p[0], w3);
    PR_MFMA_BF16(acc, x.p[2], w1);
    PR_MFMA_BF16(acc, x.p[0], w2);
    PR_MFMA_BF16(acc, x.p[1], w1);
    PR_MFMA_BF16(acc, x.p[0], w1);
}

// the six terms of the blocks of one step, block by block inside a term (consecutive MFMAs write different accumulators)
#define PR_STEP_MFMAS(F0, F1, WA0, WA1, WA2, WB0, WB1, WB2)                                                                             \
    do {                                                                                                                             \
        if (two) {                                                                                                                   \
            PR_MFMA_BF16(a00, F0.p[1], WA1); PR_MFMA_BF16(a01, F1.p[1], WA1); PR_MFMA_BF16(a10, F0.p[1], WB1); PR_MFMA_BF16(a11, F1.p[1], WB1); \
            PR_MFMA_BF16(a00, F0.p[0], WA2); PR_MFMA_BF16(a01, F1.p[0], WA2); PR_MFMA_BF16(a10, F0.p[0], WB2); PR_MFMA_BF16(a11, F1.p[0], WB2); \
            PR_MFMA_BF16(a00, F0.p[2], WA0); PR_MFMA_BF16(a01, F1.p[2], WA0); PR_MFMA_BF16(a10, F0.p[2], WB0); PR_MFMA_BF16(a11, F1.p[2], WB0); \
            PR_MFMA_BF16(a00, F0.p[0], WA1); PR_MFMA_BF16(a01, F1.p[0], WA1); PR_MFMA_BF16(a10, F0.p[0], WB1); PR_MFMA_BF16(a11, F1.p[0], WB1); \
            PR_MFMA_BF16(a00, F0.p[1], WA0); PR_MFMA_BF16(a01, F1.p[1], WA0); PR_MFMA_BF16(a10, F0.p[1], WB0); PR_MFMA_BF16(a11, F1.p[1], WB0); \
            PR_MFMA_BF16(a00, F0.p[0], WA0); PR_MFMA_BF16(a01, F1.p[0], WA0); PR_MFMA_BF16(a10, F0.p[0], WB0); PR_MFMA_BF16(a11, F1.p[0], WB0); \
        } else {                                                                                                                     \
            PR_MFMA_BF16(a00, F0.p[1], WA1); PR_MFMA_BF16(a01, F1.p[1], WA1);                                                        \
            PR_MFMA_BF16(a00, F0.p[0], WA2); PR_MFMA_BF16(a01, F1.p[0], WA2);                                                        \
            PR_MFMA_BF16(a00, F0.p[2], WA0); PR_MFMA_BF16(a01, F1.p[2], WA0);                                                        \
            PR_MFMA_BF16(a00, F0.p[0], WA1); PR_MFMA_BF16(a01, F1.p[0], WA1);                                                        \
            PR_MFMA_BF16(a00, F0.p[1], WA0); PR_MFMA_BF16(a01, F1.p[1], WA0);                                                        \
            PR_MFMA_BF16(a00, F0.p[0], WA0); PR_MFMA_BF16(a01, F1.p[0], WA0);                                                        \
        }                                                                                                                            \
    } while (0)

__device__ __forceinline__ void tile_products_bf16(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                                   f32x16& a11, const Drain* drain = nullptr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int cbA = wave, cbB = wave + MLP_WAVES;
    if (cbA >= nblk) return;
    const bool two = cbB < nblk;
    __builtin_amdgcn_s_setprio(1);
    const int ks = sg.kq >> 1;                       // K steps of 16 (even: the padded widths are multiples of 32)
    const float* ap = X + r * LDX + 8 * half;
    // [column block][step][plane][lane] fragments of 16 bytes
    const auto* wpA = as_global(reinterpret_cast<const bf16x8*>(sg.w)) + (size_t)cbA * ks * 192 + lane;
    const auto* wpB = as_global(reinterpret_cast<const bf16x8*>(sg.w)) + (size_t)(two ? cbB : cbA) * ks * 192 + lane;
    // software pipeline with NAMED even / odd register sets (a rotating set costs a register copy per value and step: 6 moves
    // per MFMA, measured): the MFMAs of a step run on fragments that were split during the previous step; while they execute, the
    // raw operands of the next step (requested in front of them) are split - the conversions sit in the shadow of the MFMAs
    float4 xl, xh, yl, yh;
    xl = *reinterpret_cast<const float4*>(ap); xh = *reinterpret_cast<const float4*>(ap + 4);
    yl = *reinterpret_cast<const float4*>(ap + 32 * LDX); yh = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4);
    Frag3 e0 = split_fragment(xl, xh), e1 = split_fragment(yl, yh), o0, o1;
    bf16x8 ea0 = wpA[0], ea1 = wpA[64], ea2 = wpA[128], eb0 = wpB[0], eb1 = wpB[64], eb2 = wpB[128];
    bf16x8 oa0, oa1, oa2, ob0 = eb0, ob1 = eb1, ob2 = eb2;
    for (int s = 0; s < ks; s += 2) {
        // ---- even step: request the odd step's operands, multiply the even fragments, split the odd ones
        {
            const float* an = ap + 16 * (s + 1);
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)(s + 1) * 192;
            oa0 = wpA[at]; oa1 = wpA[at + 64]; oa2 = wpA[at + 128];
            ob0 = wpB[at]; ob1 = wpB[at + 64]; ob2 = wpB[at + 128];      // (unconditional: see wpB)
            __builtin_amdgcn_sched_barrier(0);      // the requests stay in FRONT of the step's MFMAs (hipcc sank them behind: L2 latency exposed every step)
            PR_STEP_MFMAS(e0, e1, ea0, ea1, ea2, eb0, eb1, eb2);
            o0 = split_fragment(xl, xh); o1 = split_fragment(yl, yh);
            if (drain) drain_chunk(*drain, X, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- odd step
        {
            const int sn = (s + 2 < ks) ? s + 2 : s;
            const float* an = ap + 16 * sn;
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)sn * 192;
            ea0 = wpA[at]; ea1 = wpA[at + 64]; ea2 = wpA[at + 128];
            eb0 = wpB[at]; eb1 = wpB[at + 64]; eb2 = wpB[at + 128];
            __builtin_amdgcn_sched_barrier(0);
            PR_STEP_MFMAS(o0, o1, oa0, oa1, oa2, ob0, ob1, ob2);
            e0 = split_fragment(xl, xh); e1 = split_fragment(yl, yh);
            if (drain) drain_chunk(*drain, X, s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// The FORWARD products of a split-precision training call (phase 1): operands as fp16 pairs, x = hi + lo (hi = fp16(x), lo =
// fp16(x - hi): ~22 significant bits; forward activations are O(1) - the range that rules fp16 out for gradients is not an issue
// here, and the split evaluation kernel passes the fp32 parity tolerance with this representation), a product as the THREE
// v_mfma_f32_32x32x16_f16 hi x hi + hi x lo + lo x hi: half the matrix time and two thirds of the weight bytes of the bf16-triple
// form, which the backward pass keeps (gradients need the fp32 exponent range).  `sg.w`: the segment as k_pack kind 2 fragments
// ([column block][K step][hi 64 lanes x 16 B | lo 64 lanes x 16 B]); the operand tile stays fp32 in X (it is what gets saved) and
// is split in registers at the top of every step from the raw operands requested during the previous one (one converted set: the
// forward carries its input encoding and its batch-statistics sums in registers across the layers).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#ifndef PR_TRAIN_SPLIT_SCALE
#define PR_TRAIN_SPLIT_SCALE 8
#endif
constexpr int TRAIN_SPLIT_WEIGHT_SCALE_LOG2 = PR_TRAIN_SPLIT_SCALE;    // the packed fp16 pairs hold w x 2^8 (add_seg3 in mlp.hip)
#define PR_MFMA_F16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
struct FragH { f16x8_t hi, lo; };
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned int& ph, unsigned int& pl) {
    // fp16 range guard (never reached by sane activations; the split evaluation kernel has the same one)
    const f32x2_t v = {__builtin_amdgcn_fmed3f(x0, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(x1, -65504.0f, 65504.0f)};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f32x2_t back = __builtin_convertvector(h, f32x2_t);
    const f32x2_t r = {sub_f32(v[0], back[0]), sub_f32(v[1], back[1])};
    ph = __builtin_bit_cast(unsigned int, h);
    pl = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ FragH split_fragment_h(const float4& lo, const float4& hi) {
    unsigned int a[4], b[4];
    split_pair_h(lo.x, lo.y, a[0], b[0]);
    split_pair_h(lo.z, lo.w, a[1], b[1]);
    split_pair_h(hi.x, hi.y, a[2], b[2]);
    split_pair_h(hi.z, hi.w, a[3], b[3]);
    FragH f;
    const u32x4 wa = {a[0], a[1], a[2], a[3]}, wb = {b[0], b[1], b[2], b[3]};
    f.hi = __builtin_bit_cast(f16x8_t, wa);
    f.lo = __builtin_bit_cast(f16x8_t, wb);
    return f;
}
// the three terms of the blocks of one step, smallest first, block by block inside a term
#define PR_STEP_MFMAS_H(F0, F1, WAH, WAL, WBH, WBL)                                                                          \
    do {                                                                                                                    \
        PR_MFMA_F16(a00, F0.lo, WAH); PR_MFMA_F16(a01, F1.lo, WAH);                                                         \
        if (two) { PR_MFMA_F16(a10, F0.lo, WBH); PR_MFMA_F16(a11, F1.lo, WBH); }                                            \
        PR_MFMA_F16(a00, F0.hi, WAL); PR_MFMA_F16(a01, F1.hi, WAL);                                                         \
        if (two) { PR_MFMA_F16(a10, F0.hi, WBL); PR_MFMA_F16(a11, F1.hi, WBL); }                                            \
        PR_MFMA_F16(a00, F0.hi, WAH); PR_MFMA_F16(a01, F1.hi, WAH);                                                         \
        if (two) { PR_MFMA_F16(a10, F0.hi, WBH); PR_MFMA_F16(a11, F1.hi, WBH); }                                            \
    } while (0)

__device__ __forceinline__ FragH split_fragment_scaled_h(const float4& lo, const float4& hi, float scale);
// timing builds only (results are wrong): PR_LEAN_ABLATE 1 = every step reads the first step's weight fragments, 2 = no operand
// conversions, 4 = a third of the MFMAs
#ifndef PR_LEAN_ABLATE
#define PR_LEAN_ABLATE 0
#endif
#if PR_LEAN_ABLATE & 1
#define PR_LEAN_WSTEP(s) ((s) & 1)
#else
#define PR_LEAN_WSTEP(s) (s)
#endif
#if PR_LEAN_ABLATE & 2
__device__ __forceinline__ FragH raw_fragment_h(const float4& lo, const float4& hi) {
    FragH f;
    f.hi = __builtin_bit_cast(f16x8_t, lo);
    f.lo = __builtin_bit_cast(f16x8_t, hi);
    return f;
}
#define PR_LEAN_SPLIT(l, h) raw_fragment_h(l, h)
#else
#define PR_LEAN_SPLIT(l, h) split_fragment_scaled_h(l, h, scale)
#endif
#if PR_LEAN_ABLATE & 4
#define PR_LEAN_MFMAS(F0, F1, WAH, WAL, WBH, WBL)                                  \
    do {                                                                          \
        PR_MFMA_F16(a00, F0.lo, WAH); PR_MFMA_F16(a01, F1.hi, WAL);               \
        if (two) { PR_MFMA_F16(a10, F0.hi, WBH); PR_MFMA_F16(a11, F1.lo, WBL); }  \
    } while (0)
#else
#define PR_LEAN_MFMAS(F0, F1, WAH, WAL, WBH, WBL) PR_STEP_MFMAS_H(F0, F1, WAH, WAL, WBH, WBL)
#endif
__device__ __forceinline__ void tile_products_f16x3_lean(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                                         f32x16& a11, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int cbA = wave, cbB = wave + MLP_WAVES;
    if (cbA >= nblk) return;
    const bool two = cbB < nblk;
    __builtin_amdgcn_s_setprio(1);
    const int ks = sg.kq >> 1;
    const float* ap = X + r * LDX + 8 * half;
    // per (column block, step): hi fragment (64 lanes x 16 B), then lo fragment
    const auto* wpA = as_global(reinterpret_cast<const f16x8_t*>(sg.w)) + (size_t)cbA * ks * 128 + lane;
    // (a wave without a second column block requests its first one again: requests under `if (two)` leave the number of requests in
    // flight unknown at the step's s_waitcnt, and hipcc then waits for one of the requests it has just issued)
    const auto* wpB = as_global(reinterpret_cast<const f16x8_t*>(sg.w)) + (size_t)(two ? cbB : cbA) * ks * 128 + lane;
    float4 xl = *reinterpret_cast<const float4*>(ap), xh = *reinterpret_cast<const float4*>(ap + 4);
    float4 yl = *reinterpret_cast<const float4*>(ap + 32 * LDX), yh = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4);
    f16x8_t eah = wpA[0], eal = wpA[64], ebh = wpB[0], ebl = wpB[64];
    f16x8_t oah, oal, obh = ebh, obl = ebl;
    for (int s = 0; s < ks; s += 2) {
        {
            const FragH f0 = PR_LEAN_SPLIT(xl, xh), f1 = PR_LEAN_SPLIT(yl, yh);
            const float* an = ap + 16 * (s + 1);
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)PR_LEAN_WSTEP(s + 1) * 128;
            oah = wpA[at]; oal = wpA[at + 64];
            obh = wpB[at]; obl = wpB[at + 64];      // (unconditional: see wpB)
            __builtin_amdgcn_sched_barrier(0);      // the requests stay in front of the step's MFMAs
            PR_LEAN_MFMAS(f0, f1, eah, eal, ebh, ebl);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const FragH f0 = PR_LEAN_SPLIT(xl, xh), f1 = PR_LEAN_SPLIT(yl, yh);
            const int sn = (s + 2 < ks) ? s + 2 : s;
            const float* an = ap + 16 * sn;
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)PR_LEAN_WSTEP(sn) * 128;
            eah = wpA[at]; eal = wpA[at + 64];
            ebh = wpB[at]; ebl = wpB[at + 64];
            __builtin_amdgcn_sched_barrier(0);
            PR_LEAN_MFMAS(f0, f1, oah, oal, obh, obl);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// The same products for the BACKWARD chain (gradient tiles): the operand is multiplied by `scale` (a power of two chosen per tile so
// that its largest entry sits just under 2^15: gradients of ~1e-7 are far below fp16's range) while it is split; the caller
// multiplies the accumulators by 1 / (scale x 2^TRAIN_SPLIT_WEIGHT_SCALE_LOG2) behind the loop.  Entries within 2^-16 of the tile's
// largest keep 22 significant bits, smaller ones an absolute error of 2^-39 of it.  With the gradient write-out of tile_products.
#ifndef PR_SPLIT_MIX
#define PR_SPLIT_MIX 1      // 0: the multiply / clamp / convert / convert back / subtract / convert sequence (A/B builds)
#endif
// Four operands x `scale` -> their packed fp16 hi halves (h01, h23) and lo halves (l01, l23) in EIGHT instructions: v_fma_mixlo/hi_f16
// evaluate x * scale + c in fp32 and round the result to fp16 into one half of the destination, and take c as either half of a packed
// fp16 register.  hi = fp16(x * scale) (the product by a power of two is exact), lo = fp16(x * scale - hi) (the difference has <= 13
// significant bits: exact in fp32) - bit for bit what multiply, v_cvt_pk_f16_f32, two v_cvt_f32_f16, two subtractions and a second
// v_cvt_pk_f16_f32 produce (10 instructions per pair with the range guard; the K loops of the split-precision training kernels issued
// 115 VALU instructions per twelve MFMAs and were bound by them).  No range guard: the scale puts the tile's largest entry under 2^15; a
// tile with a non-finite entry yields NaN products like the fp32 kernels.  A write to one half of a register must not be followed
// directly by a read of that register on gfx940+ (destination-select forwarding hazard; hipcc does not look inside the asm block): every
// v_fma_mixhi is one instruction away from the first reader of its destination, and the block ends with two wait states (what a VALU
// result needs in front of a matrix instruction that reads it, should hipcc ever schedule one right behind the block).
__device__ __forceinline__ void split_quad_scaled_h(float x0, float x1, float x2, float x3, float scale, unsigned int& h01,
                                                    unsigned int& h23, unsigned int& l01, unsigned int& l23) {
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
__device__ __forceinline__ FragH split_fragment_scaled_h(const float4& lo, const float4& hi, float scale) {
#if PR_SPLIT_MIX
    unsigned int a[4], b[4];
    split_quad_scaled_h(lo.x, lo.y, lo.z, lo.w, scale, a[0], a[1], b[0], b[1]);
    split_quad_scaled_h(hi.x, hi.y, hi.z, hi.w, scale, a[2], a[3], b[2], b[3]);
    FragH f;
    const u32x4 wa = {a[0], a[1], a[2], a[3]}, wb = {b[0], b[1], b[2], b[3]};
    f.hi = __builtin_bit_cast(f16x8_t, wa);
    f.lo = __builtin_bit_cast(f16x8_t, wb);
    return f;
#else
    const float4 l = make_float4(lo.x * scale, lo.y * scale, lo.z * scale, lo.w * scale);
    const float4 h = make_float4(hi.x * scale, hi.y * scale, hi.z * scale, hi.w * scale);
    return split_fragment_h(l, h);
#endif
}
__device__ __forceinline__ void tile_products_f16x3(const Seg& sg, int nblk, const float* X, f32x16& a00, f32x16& a01, f32x16& a10,
                                                    f32x16& a11, const Drain* drain, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int cbA = wave, cbB = wave + MLP_WAVES;
    if (cbA >= nblk) return;
    const bool two = cbB < nblk;
    __builtin_amdgcn_s_setprio(1);
    const int ks = sg.kq >> 1;
    const float* ap = X + r * LDX + 8 * half;
    const auto* wpA = as_global(reinterpret_cast<const f16x8_t*>(sg.w)) + (size_t)cbA * ks * 128 + lane;
    const auto* wpB = as_global(reinterpret_cast<const f16x8_t*>(sg.w)) + (size_t)(two ? cbB : cbA) * ks * 128 + lane;
    float4 xl = *reinterpret_cast<const float4*>(ap), xh = *reinterpret_cast<const float4*>(ap + 4);
    float4 yl = *reinterpret_cast<const float4*>(ap + 32 * LDX), yh = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4);
    FragH e0 = split_fragment_scaled_h(xl, xh, scale), e1 = split_fragment_scaled_h(yl, yh, scale), o0, o1;
    f16x8_t eah = wpA[0], eal = wpA[64], ebh = wpB[0], ebl = wpB[64];
    f16x8_t oah, oal, obh = ebh, obl = ebl;
    DrainCursor cursor = {0, 0, 0, 0};
    if (drain) cursor = drain_begin(*drain);
    for (int s = 0; s < ks; s += 2) {
        {   // even step: request the odd step's operands, multiply the even fragments, split the odd ones behind the MFMAs
            const float* an = ap + 16 * (s + 1);
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)(s + 1) * 128;
            oah = wpA[at]; oal = wpA[at + 64];
            obh = wpB[at]; obl = wpB[at + 64];      // (unconditional: see wpB)
            __builtin_amdgcn_sched_barrier(0);
            PR_STEP_MFMAS_H(e0, e1, eah, eal, ebh, ebl);
            o0 = split_fragment_scaled_h(xl, xh, scale); o1 = split_fragment_scaled_h(yl, yh, scale);
            if (drain) { if (PR_DRAIN_CURSOR) drain_next(*drain, X, cursor); else drain_chunk(*drain, X, s); }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const int sn = (s + 2 < ks) ? s + 2 : s;
            const float* an = ap + 16 * sn;
            xl = *reinterpret_cast<const float4*>(an); xh = *reinterpret_cast<const float4*>(an + 4);
            yl = *reinterpret_cast<const float4*>(an + 32 * LDX); yh = *reinterpret_cast<const float4*>(an + 32 * LDX + 4);
            const size_t at = (size_t)sn * 128;
            eah = wpA[at]; eal = wpA[at + 64];
            ebh = wpB[at]; ebl = wpB[at + 64];
            __builtin_amdgcn_sched_barrier(0);
            PR_STEP_MFMAS_H(o0, o1, oah, oal, obh, obl);
            e0 = split_fragment_scaled_h(xl, xh, scale); e1 = split_fragment_scaled_h(yl, yh, scale);
            if (drain) { if (PR_DRAIN_CURSOR) drain_next(*drain, X, cursor); else drain_chunk(*drain, X, s + 1); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <bool BWD, bool BITS, bool STATS, bool SPLIT>
__device__ __forceinline__ void run_layer(const Layer& L, Smem& S, const MlpParams& p, int tile_base, int input_kind, EncRegs& enc,
                                          const BwdEpilogue* bwd, unsigned long long* bits_out, ColumnStats* stats, int* cur_slot) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, half = lane >> 5;
    const int nblk = L.nblk;
    // SPLIT: `slot` names the tile_max word of the tile this layer reads, `k` its scale exponent (commit_tile_max)
    int slot = SPLIT ? *cur_slot : 0;
    int k = SPLIT ? tile_scale_log2(S.tile_max[slot]) : 0;
    const int cbA = wave, cbB = wave + MLP_WAVES;
    const bool active = cbA < nblk;      // this wave has a first column block
    const bool two = cbB < nblk;         // ... and a second one
    PR_PHASE_T0();
    f32x16 a00, a01, a10, a11;           // [column block A / B][row block 0 / 1]
    {
        float biasA = (L.bias != nullptr && active) ? as_global(L.bias)[cbA * 32 + r] : 0.f;
        float biasB = (L.bias != nullptr && two) ? as_global(L.bias)[cbB * 32 + r] : 0.f;
        if (SPLIT) {        // the split-precision segments hold w x 2^8, the operand is multiplied by 2^k: the accumulators run at that scale
            biasA = ldexpf(biasA, k + TRAIN_SPLIT_WEIGHT_SCALE_LOG2);
            biasB = ldexpf(biasB, k + TRAIN_SPLIT_WEIGHT_SCALE_LOG2);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a00[i] = biasA;
            a01[i] = biasA;
            a10[i] = biasB;
            a11[i] = biasB;
        }
    }
    for (int sidx = 0; sidx < L.nseg; ++sidx) {
        const Seg& sg = L.seg[sidx];
        if (!BWD && sg.src == 1 && sidx > 0) {
            // second K segment of a skip layer: its operand is the network input, re-encoded over the
            // (now dead) activations of the first segment
            __syncthreads();
            int* refill_slot = SPLIT ? &S.tile_max[slot ^ 1] : nullptr;      // (cleared by the first segment's product)
            if (input_kind == 1) fill_bender_input(S, p, enc, true, refill_slot); else fill_nerf_input(S, p, enc, true, refill_slot);
            __syncthreads();
            if (SPLIT) {         // the re-encoded input has its own scale: bring the accumulators over
                slot ^= 1;
                const int k1 = tile_scale_log2(S.tile_max[slot]);
                const float shift = ldexpf(1.0f, k1 - k);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    a00[i] *= shift;
                    a01[i] *= shift;
                    a10[i] *= shift;
                    a11[i] *= shift;
                }
                k = k1;
            }
        }
        if (SPLIT && threadIdx.x == 0) S.tile_max[slot ^ 1] = 0;       // the next producer's word (nobody reads it any more)
        if (!active) continue;
#if defined(PR_MLP_ABLATE) && (PR_MLP_ABLATE & 128)
        continue;   // measurement build: no matrix work (results are wrong)
#endif
        if (SPLIT) {             // phase 1 of a training call with PR_FLAG_SPLIT_BACKWARD: fp16 pairs
            PR_PHASE_COUNT(8, 1000000ull * (sg.kq >> 1));       // K steps of 16 (read as p8 x 1e6)
            PR_PHASE_COUNT(9, 1000000ull);                      // products
            tile_products_f16x3_lean(sg, nblk, S.X, a00, a01, a10, a11, ldexpf(1.0f, k));
            continue;
        }
        PR_PHASE_COUNT(8, 1000000ull * (sg.kq >> 1));
        PR_PHASE_COUNT(9, 1000000ull);
        // matrix work outranks the other resident tile's serial phases in the per-SIMD issue arbitration
        __builtin_amdgcn_s_setprio(1);
        const int kq = sg.kq;   // even (K is padded to a multiple of 16)
        const float* ap = S.X + r * LDX + half * 4 * kq;
        const auto* wpA = as_global(reinterpret_cast<const f32x4_t*>(sg.w)) + (size_t)cbA * kq * 64 + lane;
        // two steps in flight: even/odd fragments live in their own registers and are re-loaded
        // right after their last use, a full step before they are needed again
        float4 x0e = *reinterpret_cast<const float4*>(ap);
        float4 x1e = *reinterpret_cast<const float4*>(ap + 32 * LDX);
        float4 x0o = *reinterpret_cast<const float4*>(ap + 4);
        float4 x1o = *reinterpret_cast<const float4*>(ap + 32 * LDX + 4);
        if (two) {
            // (the first requests in the order the loop repeats them - even A, even B, odd A, odd B: hipcc merges the request queue of
            // the loop entry with that of the back edge, and with A, A, B, B in front of the loop the wait for the even B fragment
            // became vmcnt(0), i.e. every iteration waited for the odd fragments it had requested a moment before)
            const auto* wpB = as_global(reinterpret_cast<const f32x4_t*>(sg.w)) + (size_t)cbB * kq * 64 + lane;
            f32x4_t wAe = wpA[0], wBe = wpB[0];
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t wAo = wpA[64], wBo = wpB[64];
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
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else {
            f32x4_t wAe = wpA[0];
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t wAo = wpA[64];
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
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    if (SPLIT) {
        *cur_slot = slot ^ 1;        // (the word the epilogue below commits this layer's output tile to)
        const float back = ldexpf(1.0f, -k - TRAIN_SPLIT_WEIGHT_SCALE_LOG2);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a00[i] *= back;
            a01[i] *= back;
            a10[i] *= back;
            a11[i] *= back;
        }
    }
#if defined(PR_MLP_ABLATE) && (PR_MLP_ABLATE & 4)
    return;   // measurement build: no barriers, no epilogue (results are wrong)
#endif
    PR_PHASE(3);
    if (BWD) {
        // Backward chain (k_chain_bwd): the tile holds d loss / d pre-activation of a layer, the product is its input
        // gradient.  EPI_BWD_GLOBAL: the rows go straight to global memory (the gradient of the network input: X keeps the
        // operand, which the next product of the same layer still needs); EPI_BWD_MASK: ReLU backward with the saved
        // post-ReLU activation of the previous layer as the mask (a bit image of the tile in LDS), result back into X.
        const int rows_valid = bwd->rows_valid;
        if (L.epi == EPI_BWD_GLOBAL) {
            if (active) {
                for (int blk = 0; blk < (two ? 2 : 1); ++blk) {
                    const int col = (blk ? cbB : cbA) * 32 + r;
                    const f32x16& lo = blk ? a10 : a00;
                    const f32x16& hi = blk ? a11 : a01;
                    if (col < bwd->n_real) {
                        // one base pointer per lane, row offsets are wave-uniform multiples of the leading dimension
                        auto* base = as_global(bwd->gout) + (size_t)(tile_base + 4 * half) * bwd->ldg + col;
                        const int ldg = bwd->ldg;
                        const int limit = rows_valid - 4 * half;       // rows of this lane: PR_ACC_ROW(i) (+ 32) < limit
                        float old_lo[16], old_hi[16];
                        if (bwd->accumulate) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                old_lo[i] = PR_ACC_ROW(i) < limit ? base[PR_ACC_ROW(i) * ldg] : 0.f;
                                old_hi[i] = PR_ACC_ROW(i) + 32 < limit ? base[(PR_ACC_ROW(i) + 32) * ldg] : 0.f;
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; ++i) old_lo[i] = old_hi[i] = 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            if (PR_ACC_ROW(i) < limit) base[PR_ACC_ROW(i) * ldg] = lo[i] + old_lo[i];
                            if (PR_ACC_ROW(i) + 32 < limit) base[(PR_ACC_ROW(i) + 32) * ldg] = hi[i] + old_hi[i];
                        }
                    }
                }
            }
            return;    // X untouched: no barrier needed
        }
        __syncthreads();  // every wave has finished reading X (and the mask bits of this layer are complete)
        if (active) {
            // bit (row, col) of the ReLU mask: byte row * (width / 8) + col / 8 of the tile's bit image (built by
            // build_relu_mask_bits before the product), bit col % 8
            // (the bit image lives in Smem::pos - named here, not read through the pointer in *bwd, which would be a flat pointer)
            const unsigned char* bits = reinterpret_cast<const unsigned char*>(S.pos);
            const int bpr = bwd->mask_bytes_per_row;
            for (int blk = 0; blk < (two ? 2 : 1); ++blk) {
                const int col = (blk ? cbB : cbA) * 32 + r;
                const f32x16& lo = blk ? a10 : a00;
                const f32x16& hi = blk ? a11 : a01;
                float* x0 = S.X + (4 * half) * LDX + col;
                const unsigned char* b0 = bits + (4 * half) * bpr + (col >> 3);
                const int bit = col & 7;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = PR_ACC_ROW(i);
                    x0[ro * LDX] = ((b0[ro * bpr] >> bit) & 1) ? lo[i] : 0.f;
                    x0[(ro + 32) * LDX] = ((b0[(ro + 32) * bpr] >> bit) & 1) ? hi[i] : 0.f;
                }
            }
        }
        __syncthreads();
        return;
    }
    __syncthreads();  // every wave has finished reading X
    PR_PHASE(4);
#if defined(PR_MLP_ABLATE) && (PR_MLP_ABLATE & 8)
    if (false) {   // measurement build: no epilogue (results are wrong)
#else
    if (active) {
#endif
        for (int blk = 0; blk < (two ? 2 : 1); ++blk) {
            const int col = (blk ? cbB : cbA) * 32 + r;
            const f32x16& lo = blk ? a10 : a00;   // rows 0..31
            const f32x16& hi = blk ? a11 : a01;   // rows 32..63
            float* x0 = S.X + (4 * half) * LDX + col;
            if (L.epi == EPI_RELU) {
                store_relu(lo, x0);
                store_relu(hi, x0 + 32 * LDX);
                if (SPLIT) {        // the next layer's operand tile: its largest entry
                    float biggest = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) biggest = fmaxf(biggest, fmaxf(lo[i], hi[i]));
                    commit_tile_max(fmaxf(biggest, 0.f), &S.tile_max[slot ^ 1]);
                }
                if (BITS && bits_out) {
                    // this lane holds column `col` for the rows PR_ACC_ROW(i) + 4 half (+ 32): its half of the column's word, the
                    // other half sits in lane ^ 32
                    unsigned int mlo = 0u, mhi = 0u;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        mlo |= (lo[i] > 0.f ? 1u : 0u) << PR_ACC_ROW(i);
                        mhi |= (hi[i] > 0.f ? 1u : 0u) << PR_ACC_ROW(i);
                    }
                    mlo <<= 4 * half;
                    mhi <<= 4 * half;
                    mlo |= (unsigned int)__shfl_xor((int)mlo, 32, 64);
                    mhi |= (unsigned int)__shfl_xor((int)mhi, 32, 64);
                    if (half == 0) bits_out[col] = ((unsigned long long)mhi << 32) | mlo;
                }
            } else if (L.epi == EPI_ADAIN_RELU) {
                const int bofs = L.nblk * 32;
                if (S.uniform_frame) {
                    const float* tab = p.adain + (size_t)S.frame[0] * p.adain_stride + L.adain_off;
                    const float g = tab[col], b = tab[bofs + col];
                    store_adain_uniform(lo, x0, g, b);
                    store_adain_uniform(hi, x0 + 32 * LDX, g, b);
                } else {
                    store_adain_rows(lo, S, p, 4 * half, col, L.adain_off + col, L.adain_off + bofs + col);
                    store_adain_rows(hi, S, p, 4 * half + 32, col, L.adain_off + col, L.adain_off + bofs + col);
                }
            } else {
                // last layer: stage the tile in X, the caller writes it out with coalesced 16-byte stores
                store_plain(lo, x0);
                store_plain(hi, x0 + 32 * LDX);
                if (STATS && stats) {
                    // rows that enter the statistics: bit ro of `mine` <-> tile row ro + 4 half.  double: var = E[x^2] - mean^2
                    // cancels badly in fp32 when |mean| >> std
                    const unsigned long long mine = __ballot((S.flags[lane] & 3) == 3) >> (4 * half);
                    const unsigned int mlo = (unsigned int)mine, mhi = (unsigned int)(mine >> 32);
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        // (selects, not branches: the two halves of a wave hold different rows - 32 divergent branches per block;
                        // a row outside the statistics adds exact zeros)
                        const double x0 = ((mlo >> PR_ACC_ROW(i)) & 1u) ? (double)lo[i] : 0.0;
                        s1 += x0;
                        s2 = fma(x0, x0, s2);
                        const double x1 = ((mhi >> PR_ACC_ROW(i)) & 1u) ? (double)hi[i] : 0.0;
                        s1 += x1;
                        s2 = fma(x1, x1, s2);
                    }
                    s1 += __shfl_xor(s1, 32, 64);      // the other half of the column
                    s2 += __shfl_xor(s2, 32, 64);
                    if (blk == 0) {
                        stats->s1[0] += s1;
                        stats->s2[0] += s2;
                    } else {
                        stats->s1[1] += s1;
                        stats->s2[1] += s2;
                    }
                }
            }
        }
    }
    PR_PHASE(5);
    __syncthreads();
    PR_PHASE(6);
}

// dot products of every tile row with `nout` (<= 3) weight rows of length `width` (raw, padded),
// for tile row `s`: 8 threads per row (callers loop s = tid / 8, + MLP_THREADS / 8, ...); the result is valid in all 8.
// (W: an LDS array or an as_global() pointer - a plain `const float*` chosen between the two at run time would be a flat pointer)
template <class W>
__device__ __forceinline__ void row_dots(const Smem& S, int s, W w, int width, int wstride, int nout, float* out) {
    const int part = threadIdx.x & 7;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = part; k < width; k += 8) {
        const float x = S.X[s * LDX + k];
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

// Tile rows staged in X -> HBM with coalesced 16-byte stores (width % 4 == 0) or scalar stores.
// zero_dead: rows that failed the second AABB test are written as zeros (feature rows).
__device__ __forceinline__ void write_tile_rows(const Smem& S, float* dst, int width, int stride, int tile_base, bool zero_dead) {
    const int tid = threadIdx.x;
    const float* src = S.X;
    const int ld = LDX;
    if ((width & 3) == 0 && (stride & 3) == 0) {
        // Four chunks per thread at a time: row flags, then the four LDS reads, then the four stores.  One chunk per iteration was a
        // chain of an integer division, an LDS read (flags), an LDS read (the chunk) and the store - ~500 cycles per iteration, 16
        // iterations per 256-wide tile, 0.18 of a wave's time in the training forward (phase timers).
        const int w4 = width >> 2;
        const int count = TILE_M * w4;
        constexpr int BATCH = 4;
        const bool aligned = (MLP_THREADS % w4) == 0;        // a thread keeps its columns, its rows advance by MLP_THREADS / w4
        const int rstep = MLP_THREADS / w4;
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        for (int base = tid; base < count; base += MLP_THREADS * BATCH) {
            int row[BATCH], c[BATCH], fl[BATCH];
            const int row0 = base / w4, c0 = (base - row0 * w4) * 4;
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                const int idx = base + q * MLP_THREADS;
                if (aligned) {
                    row[q] = row0 + q * rstep;
                    c[q] = c0;
                } else {
                    row[q] = idx / w4;
                    c[q] = (idx - row[q] * w4) * 4;
                }
                if (idx >= count) row[q] = 0, c[q] = 0;
                fl[q] = idx < count ? S.flags[row[q]] : 0;
            }
            f32x4_nt v[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) v[q] = *reinterpret_cast<const f32x4_nt*>(src + row[q] * ld + c[q]);
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                if (fl[q] & 1) {
                    if (zero_dead && !(fl[q] & 2)) v[q] = f32x4_nt{0.f, 0.f, 0.f, 0.f};
                    // streamed once, read next by another kernel: keep the rows from evicting the weight fragments in L2
                    __builtin_nontemporal_store(v[q], reinterpret_cast<f32x4_nt*>(dst + (size_t)(tile_base + row[q]) * stride + c[q]));
                }
            }
        }
    } else {
        for (int idx = tid; idx < TILE_M * width; idx += MLP_THREADS) {
            const int row = idx / width, c = idx - row * width;
            const int fl = S.flags[row];
            if (fl & 1) dst[(size_t)(tile_base + row) * stride + c] = (zero_dead && !(fl & 2)) ? 0.f : src[row * ld + c];
        }
    }
}

// Feature rows staged in X -> HBM rows S.dest[row] (rows with dest < 0 are skipped); see write_tile_rows.
__device__ __forceinline__ void write_rows_indirect(const Smem& S, float* dst, int width, int stride) {
    const int tid = threadIdx.x;
    if ((width & 3) == 0 && (stride & 3) == 0) {
        const int w4 = width >> 2;
        for (int idx = tid; idx < TILE_M * w4; idx += MLP_THREADS) {
            const int row = idx / w4, c = (idx - row * w4) * 4;
            const int d = S.dest[row];
            if (d >= 0) {
                const float4 v = *reinterpret_cast<const float4*>(S.X + row * LDX + c);
                typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                f32x4_nt nt = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(nt, reinterpret_cast<f32x4_nt*>(dst + (size_t)d * stride + c));
            }
        }
    } else {
        for (int idx = tid; idx < TILE_M * width; idx += MLP_THREADS) {
            const int row = idx / width, c = idx - row * width;
            const int d = S.dest[row];
            if (d >= 0) dst[(size_t)d * stride + c] = S.X[row * LDX + c];
        }
    }
}

// ReLU mask of a tile as one bit per element: bit (row, col) = saved post-ReLU activation > 0.  Coalesced 16-byte loads of the
// 64 x width activation rows, eight columns -> one byte of the bit image (2 KB at width 256; it lives in Smem::pos, which
// the backward chain does not use otherwise).  Rows beyond the tile's real rows get zero bits.
__device__ __forceinline__ void build_relu_mask_bits(unsigned char* bits, const float* acts, int ld, int width_pad, int tile_base,
                                                     int rows_valid) {
    const int c8n = width_pad >> 3;
    for (int idx = threadIdx.x; idx < TILE_M * c8n; idx += MLP_THREADS) {
        const int row = idx / c8n, c8 = idx - row * c8n;
        unsigned int byte = 0;
        if (row < rows_valid) {
            const float* src = acts + (size_t)(tile_base + row) * ld + 8 * c8;
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            byte = (a.x > 0.f ? 1u : 0u) | (a.y > 0.f ? 2u : 0u) | (a.z > 0.f ? 4u : 0u) | (a.w > 0.f ? 8u : 0u) |
                   (b.x > 0.f ? 16u : 0u) | (b.y > 0.f ? 32u : 0u) | (b.z > 0.f ? 64u : 0u) | (b.w > 0.f ? 128u : 0u);
        }
        bits[idx] = (unsigned char)byte;
    }
}

}  // namespace pr
