// Internal declarations shared by the libplayrender translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "playrender.h"

namespace pr {

// ---------------------------------------------------------------------------------------------
// Errors
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define PR_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            pr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PR_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define PR_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            pr::set_error(__VA_ARGS__);                                                 \
            return PR_ERR_INVALID;                                                      \
        }                                                                               \
    } while (0)

#define PR_LAUNCH_CHECK() PR_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// Geometry of the MLP tiling
// ---------------------------------------------------------------------------------------------
constexpr int TILE_M = 64;        // samples per workgroup tile (two 32-row MFMA blocks)
constexpr int MLP_WAVES = 4;      // 256 threads; every wave owns two 32-column blocks x both row blocks
constexpr int MLP_BLOCKS_PER_CU = 2;   // two independent tiles per CU (LDS ~70 KB each): one computes while the other
                                       // is in a prologue / epilogue / barrier
constexpr int MLP_THREADS = MLP_WAVES * 64;
// measurement builds only: delay every second workgroup (rule 1: the second half of the grid; 2: bit 3 of the block index) at launch
#ifndef PR_STAGGER_RULE
#define PR_STAGGER_RULE 0
#endif
#ifndef PR_STAGGER_KERNELS
#define PR_STAGGER_KERNELS 0      // 1 head forward phases, 2 head backward phases, 4 phase 1 of the training forward
#endif
#ifndef PR_STAGGER_SLEEPS
#define PR_STAGGER_SLEEPS 2       // x 127 x 64 clocks (~3.4 us each)
#endif
__device__ __forceinline__ void pr_stagger(int kernel_bit) {
#if PR_STAGGER_RULE != 0
    if (!(PR_STAGGER_KERNELS & kernel_bit)) return;
    const unsigned b = blockIdx.x;
    const bool late = PR_STAGGER_RULE == 1 ? (b >= gridDim.x / 2) : (((b >> 3) & 1u) != 0u);
    if (late) {
#pragma unroll 1
        for (int i = 0; i < PR_STAGGER_SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);
    }
#else
    (void)kernel_bit;
#endif
}

// A device pointer that reaches a kernel through a table (a Layer / Seg copied out of the kernel arguments, a pointer chosen at run
// time) has lost its address space: hipcc then emits flat_load, and - because a flat access may also be an LDS access - waits for
// EVERY outstanding memory and LDS operation (s_waitcnt vmcnt(0) lgkmcnt(0)) in front of the first use of any loaded value.  In a
// software-pipelined K loop that wait includes the requests just issued for the NEXT step: the pipelining is gone and every step
// pays a full L2 round trip.  as_global() states what the host code guarantees - the pointer names global memory.
#define PR_GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const PR_GLOBAL_AS T* as_global(const T* p) {
    return (const PR_GLOBAL_AS T*)(p);
}
template <class T>
__device__ __forceinline__ PR_GLOBAL_AS T* as_global(T* p) {
    return (PR_GLOBAL_AS T*)(p);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));     // (a native vector: loadable through an address-space pointer, float4 is a class)

constexpr int MAX_WIDTH = 256;    // padded layer width limit (8 column blocks of 32)
constexpr int LDX = MAX_WIDTH + 4;  // activation row stride (floats): conflict-free ds_read_b128
constexpr int MAX_RESIDENT_TILES = 1024;   // upper bound of the persistent MLP grid (2 workgroups x CUs; 512 on MI355X)
constexpr int MAX_ENC = 128;      // padded encoding width limit
constexpr int LDE = MAX_ENC + 4;

__host__ __device__ static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// A K-segment of a layer: packed weights [nblk][kq][64 lanes][4] (see pack kernels).
struct Seg {
    const float* w;     // packed, device
    int kq;             // K_pad / 8
    int src;            // 0 = activation buffer X, 1 = encoding buffer E
};

enum Epilogue { EPI_RELU = 0, EPI_ADAIN_RELU = 1, EPI_FEATURES = 2, EPI_BWD_MASK = 3, EPI_BWD_GLOBAL = 4 };

struct Layer {
    Seg seg[2];
    int nseg;
    const float* bias;  // padded to nblk*32, device (NULL = no bias)
    int nblk;           // padded out features / 32
    int epi;
    int adain_off;      // offset (floats) of [g | b] of this layer inside one AdaIN table row
    int n_real;         // real out features
};

// Epilogue operands of the backward chain kernel (run_layer<true>, mlp.hip).
struct BwdEpilogue {
    const unsigned char* mask_bits; int mask_bytes_per_row;   // EPI_BWD_MASK: bit image (LDS) of "saved activation > 0" of the tile
    float* gout; int ldg;         // EPI_BWD_GLOBAL: destination rows in global memory
    int accumulate;               // EPI_BWD_GLOBAL: add to the destination
    int n_real;                   // real (unpadded) width of the product
    int rows_valid;               // real rows of the tile
};

// Backward chain of a ReLU MLP with one skip concatenation: all input-gradient products in one launch (k_chain_bwd).
struct ChainBwdParams {
    const int32_t* total;
    int count, skip, W, Wpad, in_pad, in_real;
    Layer act_layers[PR_MAX_LAYERS];   // l = 1 .. count - 1: W_l[:, :W]^T
    Layer in0_skip, in0_first;         // W_skip[:, W:]^T and W_0^T (gradient of the network input)
    const float* g_last;               // (cap, Wpad) d loss / d pre-activation of the last layer
    const float* acts; size_t act_stride;   // saved post-ReLU outputs of layers 0 .. count - 1, (cap, Wpad) each
    const unsigned char* bits; size_t bits_stride;   // their ReLU masks as bit images (cap, Wpad / 8 bytes) each, written by the forward pass
    float* gstack; size_t g_stride;    // out: pre-activation gradients of layers 0 .. count - 2, (cap, Wpad) each
    float* g_in; int ld_in;            // out: gradient of the network input (cap, ld_in)
};
size_t chain_bwd_packed_bytes(int count, int width, int in_features);
int prepare_chain_bwd(const pr_linear_t* layers, int count, int skip, int width, int in_features, float* packed,
                      ChainBwdParams* c, hipStream_t s);
int launch_chain_bwd(const ChainBwdParams& c, int max_rows, hipStream_t s);

// Offsets (in floats) inside the packed buffer of one model.
struct PackedLayout {
    // bender
    int b_seg_off[PR_MAX_LAYERS][2];
    int b_bias_off[PR_MAX_LAYERS];
    int b_out_off;                     // raw copy (3, BWpad)
    // nerf backbone
    int n_seg_off[PR_MAX_LAYERS][2];
    int n_bias_off[PR_MAX_LAYERS];
    int sigma_off;                     // raw copy (Wpad) + bias at [Wpad]
    int h0_off, h3_off, h6_off, h6_bias_off;
    // backward pass (fp32 packing only): W^T as fragment-ordered segments, out[m][n] = sum_k G[m][k] W[k][col_off + n]
    int t_b_act[PR_MAX_LAYERS], t_b_skip, t_b_first;   // bender chain: W_l[:, :BW]^T (l >= 1), W_skip[:, BW:]^T, W_0^T
    int t_n_act[PR_MAX_LAYERS], t_n_skip, t_n_first;   // NeRF backbone chain
    int t_h0, t_h3, t_h6;                              // head0^T (W -> W), head3^T (W/2 -> W), head6^T (F -> W/2)
    // the same transposed segments as bf16 TRIPLES (PR_FLAG_SPLIT_BACKWARD): w = b1 + b2 + b3, fragments of v_mfma_f32_32x32x16_bf16
    // - [column block][16-wide K step][plane][64 lanes][8 bf16] - 1.5 x the floats of the fp32 segment
    int t3_b_act[PR_MAX_LAYERS], t3_b_skip, t3_b_first;
    int t3_n_act[PR_MAX_LAYERS], t3_n_skip, t3_n_first;
    int t3_h0, t3_h3, t3_h6;
    // ... and the FORWARD segments of phase 1 of a training call (ray bender, backbone, head layer 0) as bf16 triples
    int b_seg3[PR_MAX_LAYERS][2], n_seg3[PR_MAX_LAYERS][2], h0_3;
    int total;
};

struct ModelDims {
    int enc;        // real NeRF encoding size  (din + 2*din*octaves)
    int enc_pad;    // multiple of 8
    int din;        // 3 (AdaIN) or 6 (skybox)
    int W, Wpad;    // backbone width
    int W2, W2pad;  // W / 2
    int F, Fpad;
    int benc;       // bender PE size (3 + 6*boct)
    int bin, bin_pad; // bender input = benc + D
    int BW, BWpad;
};

int compute_dims(const pr_object_model_t& m, ModelDims* d);
int compute_layout(const pr_object_model_t& m, const ModelDims& d, PackedLayout* l);

// Kernel parameters of the fused per-object MLP (passed by value).
struct MlpParams {
    // compact sample records
    const float* rec_pos;      // (cap, 3) object-frame positions ; unused for skybox
    const int32_t* rec_flat;   // (cap) flat sample index n*R*P + r*P + i
    const int32_t* total;      // device scalar: number of records
    int samples_per_frame;     // R * P
    int positions;             // P
    int rays;                  // R
    // object
    int kind;
    int has_bender;
    int canonical;
    float lo[3], hi[3], size[3];
    float empty_alpha;
    const uint8_t* in_scene;     // (N,K) base offset to this object: the density of an absent object's samples is empty_alpha
    int in_scene_stride;         // K                               (object_composer.py:546-547: overridden AFTER the network ran)
    // skybox inputs
    const float* ray_directions; // (N,R,3)
    const float* ray_origins;    // (N,3)
    const float* w2o;            // (N,K,3,4) base already offset to this object; stride below
    int w2o_stride;              // floats between frames (K*12)
    // bender
    int b_octaves, benc, bin_pad, D;
    float b_weights[PR_MAX_OCTAVES];
    const float* deformation;    // base offset to object k
    int deformation_stride;      // floats between frames (K*D)
    Layer b_layers[PR_MAX_LAYERS];
    int b_count;
    const float* b_out;          // raw (3, BWpad)
    int BW, BWpad;
    // nerf
    int octaves, din, enc, enc_pad;
    Layer layers[PR_MAX_LAYERS + 3];
    int n_layers;                // backbone + 3 head layers
    int n_backbone;
    const float* sigma_w;        // (Wpad) + bias
    int W, Wpad;
    const float* adain;          // table base for this object; row = frame
    int adain_stride;            // floats between frames
    int F;
    // train-mode BatchNorm (batch statistics sit between the head matmuls -> three phases, see mlp.hip)
    int phase;                   // 0 = eval (everything fused); 1 = ... -> raw h1; 2 = h1 -> raw h2; 3 = h2 -> features
    float* h_out;                // phase 1/2: raw (pre-BN) head activations, (cap, h_out_width)
    const float* h_in;           // phase 2/3: the previous phase's h_out
    int h_out_width, h_in_width; // padded widths
    int32_t* row_flags;          // (cap) bit 0 valid, bit 1 passed every AABB test; written in phase 1
    double* stats;               // phase 1/2: [sum(h_out_width) | sum of squares(h_out_width)] over the alive rows
    int32_t* stat_count;         // number of alive rows (written in phase 1)
    // saved for the backward pass (phase 1 only; NULL = not saved): compact rows like `feat`
    float* save_enc;             // (cap, enc_pad) NeRF input encoding
    float* save_act;             // n_backbone blocks of (cap, Wpad): post-ReLU output of every backbone layer
    size_t save_act_stride;      // floats between the blocks
    float* save_bin;             // (cap, bin_pad) bender input [annealed PE | deformation]
    float* save_bact;            // b_count blocks of (cap, BWpad)
    size_t save_bact_stride;
    unsigned char* save_bits;    // n_backbone blocks of relu_bits_bytes(cap, Wpad): per 64-row tile and column one 64-bit word of
    size_t save_bits_stride;     // (post-ReLU activation > 0) bits - the ReLU masks the backward chain reads (bytes between the blocks)
    unsigned char* save_bbits;   // the same for the bender layers, (cap, BWpad / 8)
    size_t save_bbits_stride;
    float* save_braw;            // (cap, 3) bender head output before * size and the clamp
    float* save_delta;           // (cap, 3) final displacement (after clamp / canonical_pose)
    float* delta_dense;          // (N,R,P,3) the same, scattered to the sample grid (optional export)
    // sigma-gated feature head (eval, no noise; see gated_head in mlp.hip)
    int gate;                    // 1 = run the feature head on the samples with density > 0 only
    float* pend_act;             // (MAX_RESIDENT_TILES, TILE_M, Wpad floats) per-workgroup stacks of pending live rows
    int32_t* pend_meta;          // (MAX_RESIDENT_TILES, TILE_M, 2) [compact feature row, frame] of the pending rows
    int32_t* head_count;         // device counter: rows sent through the feature head (NULL = not counted)
    int32_t* tile_counter;       // zeroed device counter: tiles beyond the first of a workgroup are claimed from it (NULL: strided)
    int split3;                  // 1 (phase 1 of a training call): the layer segments are bf16-triple packings (six bf16 MFMAs per product)
    // outputs
    float* sigma;                // dense (N,R,P)
    float* dispmag;              // dense (N,R,P) or NULL
    float* feat;                 // compact (cap, F)
};

// Grouped launch of the fused MLP: up to MLP_GROUP_MAX object models, one persistent launch (k_mlp_mfma_group in mlp.hip).
// The job descriptors travel BY VALUE in the kernel argument segment and are addressed with constant indices, like the
// single launch's parameters.
constexpr int MLP_GROUP_MAX = 4;
struct MlpGroupParams {
    MlpParams jobs[MLP_GROUP_MAX];
    int count;
};

// bytes of the ReLU bit images of one layer: a 64-bit word per column and 64-row tile (written by the training forward's ReLU epilogues, run_layer<false, true> in mlp_tile.h)
static inline size_t relu_bits_bytes(size_t rows, int width_pad) { return ((rows + TILE_M - 1) / TILE_M) * (size_t)width_pad * 8; }

// AdaIN table row layout for one (frame, object): g1[Wpad] b1[Wpad] g2[W2pad] b2[W2pad]
static inline int adain_row_floats(const ModelDims& d) { return 2 * d.Wpad + 2 * d.W2pad; }

// ---------------------------------------------------------------------------------------------
// Noise sources.  A noise tensor of the reference (torch.rand / torch.randn draws: stratified jitter, density noise, inverse-
// CDF positions, Hutchinson probes) is either an explicit device array (replayed draws: parity tests, the oracle's order) or
// - PR_FLAG_DEVICE_NOISE - a counter-based generator evaluated where the value is needed: Philox4x32-10 keyed by
// (call seed, stream id), counter = element index, so the backward pass regenerates exactly what the forward pass used
// and nothing of size (N, R, P) is materialised.  Stream ids: kind * 16 + model type * 8 + object.
// ---------------------------------------------------------------------------------------------
enum NoiseKind { NOISE_JITTER = 0, NOISE_ALPHA = 1, NOISE_PDF = 2, NOISE_INTEGRATE = 3, NOISE_INTEGRATE_GLOBAL = 4, NOISE_DIVERGENCE = 5 };

struct NoiseRef {
    const float* ptr;            // explicit values, or NULL
    unsigned int key0, key1;     // Philox key of the stream (generate != 0)
    int generate;                // 1: values come from the generator
    int rays, ray_offset, total_rays;   // ray r of frame n of this call is ray (ray_offset + r) of total_rays in the stream's index space
    const unsigned long long* seed_dev; // the call's seed as a device word (pr_call_t.noise_seed_device), or NULL: key0 / key1 are final
    unsigned long long salt;            // with seed_dev: key = mix(*seed_dev ^ salt), what the host computes from noise_seed otherwise
};

__host__ __device__ static inline unsigned long long noise_mix(unsigned long long x) {   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline NoiseRef make_noise(const float* ptr, const pr_call_t& c, int kind, int type, int object) {
    NoiseRef n;
    memset(&n, 0, sizeof(n));
    n.ptr = ptr;
    n.rays = c.rays;
    n.ray_offset = c.noise_ray_offset;
    n.total_rays = c.noise_total_rays > 0 ? c.noise_total_rays : c.rays;
    if (!ptr && (c.flags & PR_FLAG_DEVICE_NOISE)) {
        n.salt = noise_mix((unsigned long long)(kind * 16 + type * 8 + object) + 1);
        const unsigned long long k = noise_mix(c.noise_seed ^ n.salt);
        n.key0 = (unsigned int)k;
        n.key1 = (unsigned int)(k >> 32);
        n.seed_dev = reinterpret_cast<const unsigned long long*>(c.noise_seed_device);
        n.generate = 1;
    }
    return n;
}

// noise of the perturbation path (jitter, density noise, inverse-CDF positions): explicit tensors are used whenever given,
// generated values only with PR_FLAG_PERTURB
static inline NoiseRef perturb_noise(const float* ptr, const pr_call_t& c, int kind, int type, int object) {
    NoiseRef n = make_noise(ptr, c, kind, type, object);
    if (!(c.flags & PR_FLAG_PERTURB)) n.generate = 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// Stage launchers (each enqueues on `stream`, returns pr_status)
// ---------------------------------------------------------------------------------------------
struct PlaceParams {
    int frames, rays, positions, objects, object_index;
    const float* ray_origins;     // (N,3)
    const float* ray_directions;  // (N,R,3)
    const float* w2o;             // (N,K,3,4)
    const uint8_t* in_scene;      // (N,K)
    float lo[3], hi[3];
    float z_near_min, z_far_max, empty_alpha;
    const float* linspace;        // (P)
    NoiseRef jitter;              // U[0,1) per sample (N,R,P), or absent
    float* t;                     // (N,R,P) out
    float* sigma;                 // (N,R,P) out, filled with empty_alpha
    float* dispmag;               // (N,R,P) out zeros or NULL
    int32_t* block_sums;          // (ceil(N*R/256)) in-box counts per block of 256 rays
};
int launch_place_coarse(const PlaceParams& p, hipStream_t s);

struct FillParams {
    int frames, rays, positions, objects, object_index;
    const float* ray_origins;
    const float* ray_directions;
    const float* w2o;
    const uint8_t* in_scene;
    float lo[3], hi[3];
    const float* t;               // (N,R,P)
    const int32_t* block_offsets; // exclusive scan of block_sums
    float* rec_pos;               // (cap,3)
    int32_t* rec_flat;            // (cap)
    int32_t* slot;                // (N,R,P) compact row or -1
};
int launch_fill(const FillParams& p, hipStream_t s);
int launch_placement_group(const PlaceParams* pp, const FillParams* fp, int32_t* const* totals, int count, hipStream_t s);

// exclusive scan of n block sums (single workgroup), writes total to *total
int launch_scan(const int32_t* sums, int32_t* offsets, int32_t* total, int n, hipStream_t s);

struct ResampleParams {
    int frames, rays, objects, object_index;
    int pc, pf;
    const float* ray_origins;
    const float* ray_directions;
    const float* w2o;
    const uint8_t* in_scene;
    float lo[3], hi[3];
    float empty_alpha;
    const float* t_coarse;        // (N,R,Pc)
    const float* sigma_coarse;    // (N,R,Pc)
    NoiseRef alpha_noise;         // N(0,1) (N,R,Pc) or absent
    const float* u_fixed;         // linspace(0,1,Pf) (Pf)
    NoiseRef u_random;            // U[0,1) (N,R,Pf) or absent
    float* t_fine;                // (N,R,Pc+Pf) out
    float* sigma_fine;            // (N,R,Pc+Pf) out, filled with empty_alpha
    float* dispmag_fine;          // or NULL
    int32_t* block_sums;
};
int launch_resample(const ResampleParams& p, hipStream_t s);

struct FoldParams {
    int frames, objects, object_index;
    const float* style;           // (N,K,S)
    int S;
    pr_linear_t affine1; const float* bn1_mean; const float* bn1_var;
    pr_linear_t affine4; const float* bn4_mean; const float* bn4_var;
    float eps;
    int W, Wpad, W2, W2pad;
    float* table;                 // (N, row) rows of this object
    int row_floats;
};
int launch_adain_fold(const FoldParams& p, hipStream_t s);
int launch_adain_fold_group(const FoldParams* jobs, int count, hipStream_t s);     // the objects of an evaluation call: one launch

int launch_mlp(const MlpParams& p, int max_rows, bool naive, const pr_object_model_t* raw, hipStream_t s);
// evaluation launches of several objects as one
int launch_mlp_group(const MlpParams* host_jobs, const int* max_rows, int count, hipStream_t s);
// PR_PRECISION_F16X3 (terms = 3) / PR_PRECISION_F16 (terms = 1), eval only
int launch_mlp_split(const MlpParams& p, int max_rows, int terms, hipStream_t s);
int launch_mlp_split_group(const MlpParams* host_jobs, const int* max_rows, int count, int terms, hipStream_t s);

// BatchNorm1d(affine=False) in training mode: batch mean / biased variance from the accumulated sums,
// running statistics updated in place with momentum 0.1 and the unbiased variance, num_batches_tracked += 1
// (torch.nn.functional.batch_norm semantics, model/layers/adain.py:47,58).
struct BnFinalizeParams {
    const double* stats;          // [sum(width_pad) | sumsq(width_pad)]
    const int32_t* count;
    int width, width_pad;
    float momentum;
    float* running_mean;          // nn buffers, updated in place
    float* running_var;
    long long* num_batches_tracked;
    float* batch_mean;            // out (width)
    float* batch_var;             // out (width), biased
    int frozen;                   // eval mode: hand the running statistics through, update nothing
};
int launch_bn_finalize(const BnFinalizeParams& p, hipStream_t s);

// batch statistics of one AdaIN layer + their fold into the AdaIN table (k_bn_fold_group: every object of a training call)
struct BnFoldJob {
    const double* stats; const int32_t* count; int width, width_pad; float momentum;
    float* running_mean; float* running_var; long long* num_batches_tracked;
    float* batch_mean; float* batch_var; int frozen;
    const float* style; int style_stride, S, frames;      // style code of frame n: style + n * style_stride
    pr_linear_t affine; float eps;
    float* table; int row_floats, g_off, b_off;           // table row n: [.. g (g_off) .. b (b_off) ..]
    int32_t* normalised_out;                              // or NULL: receives the number of rows in the statistics
};
struct BnFoldJobs { BnFoldJob job[PR_MAX_OBJECTS]; };
int launch_bn_fold_group(const BnFoldJobs& jobs, int count, hipStream_t s);

// Hutchinson divergence estimate of the ray benders of a training call as one tile kernel (k_div_chain_group, train_bwd.hip):
// the tangent of the bender input along the probe goes through the bender's layers like the sample itself (the forward
// fragments, the saved ReLU bit images as masks), the 3-wide output head and the clamp cases follow per row.
struct DivChainJob {
    const int32_t* total; const int32_t* rec_flat; const int32_t* row_flags; const float* rec_pos;
    NoiseRef noise; int positions;
    const float* bin; int bin_pad, benc, b_octaves;
    const unsigned char* bbits; size_t bbits_stride;
    int BW, BWpad, b_count, b_skip;
    Seg seg0[PR_MAX_LAYERS]; Seg seg1;          // forward fragments: first K segment of every layer, second segment of the skip layer
    const float* w_out;                         // packed raw copy of the output head (3, BWpad)
    const float* braw;
    float lo[3], hi[3];
    int canonical;
    float* div;                                 // (N,R,P), zeroed by the caller
    int32_t* tile_counter;                      // zeroed
};
bool div_chain_supported(int BWpad, int bin_pad);
int launch_div_chain_group(const DivChainJob* jobs, const long* max_rows, int count, hipStream_t s);

struct CompositeObject {
    const float* t;
    const float* sigma;
    const int32_t* slot;
    const float* divergence;  // (N,R,P) Hutchinson divergence estimate, or NULL (zeros)
    const float* dispmag;   // or NULL
    const float* feat;      // compact rows
    NoiseRef noise;         // integrate noise (N,R,P) or absent
    int positions;
    pr_entry_t out;
};
struct CompositeParams {
    int frames, rays, objects, static_objects, F;
    int fix_overlaps;
    int any_divergence;          // some object carries a divergence estimate (differentiable training calls)
    int sigmoid;                 // PR_FLAG_SIGMOID_FEATURES
    int total_positions;         // sum P_k
    int sort_size;               // next pow2 >= total_positions
    const float* ray_directions; // (N,R,3) world
    NoiseRef noise_global;       // (N,R,sumP) or absent
    CompositeObject obj[PR_MAX_OBJECTS];
    pr_entry_t global;
    pr_decoder_layout_t decoder;   // global features additionally as channels-first maps per ray group (groups = 0: off)
};
int launch_composite(const CompositeParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Workspace plan of one call (render.hip); pr_render_backward recomputes it from the same call
// ---------------------------------------------------------------------------------------------
#define PR_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != PR_OK) return _r;  \
    } while (0)

struct SavedPlan {   // PR_FLAG_SAVE_FOR_BACKWARD: per object instance and model type
    size_t rec_pos, rec_flat, row_flags, enc, act, h1, h2, batch, stat_count, bin, bact, braw, delta, div, bits, bbits, stats;
};
struct TypePlan {
    size_t t[PR_MAX_OBJECTS], sigma[PR_MAX_OBJECTS], slot[PR_MAX_OBJECTS], dispmag[PR_MAX_OBJECTS];
    size_t adain[PR_MAX_OBJECTS];
    size_t feat[PR_MAX_OBJECTS];
    int positions[PR_MAX_OBJECTS];
    size_t totals;  // K ints
    size_t head_counts;  // PR_MAX_OBJECTS ints: rows sent through the feature head (sigma-gated head), then PR_MAX_OBJECTS tile
                         // counters of the evaluation launches, then PR_MAX_OBJECTS tile counters of the divergence launch
    size_t zero_begin, zero_bytes;   // ONE fill per model type: the counters above, the batch-statistics accumulators and the
                                     // divergence arrays of every object
    SavedPlan saved[PR_MAX_OBJECTS];
};
struct Plan {
    TypePlan type[2];
    size_t block_sums, block_offsets;
    size_t rec_pos, rec_flat;
    size_t pend_act, pend_meta;   // pending stacks of the sigma-gated head (gate_active calls only)
    // train-mode BatchNorm scratch (shared by all objects, they are processed one after the other)
    size_t h1, h2, row_flags, stats, stat_count, batch_stats;
    size_t div_t0, div_ta, div_tb;   // divergence tangent scratch
    size_t rec_pos_k[PR_MAX_OBJECTS], rec_flat_k[PR_MAX_OBJECTS];   // grouped evaluation launches (group_active)
    size_t bytes;
    int nblocks256;
};
// Once per (kernel, device): raises the kernel's dynamic LDS limit on the CURRENT device; *cu_count (optional) receives
// that device's number of compute units.  Function attributes are per device: a process may drive several.
int prepare_kernel(const void* kernel, int lds_bytes, int* cu_count);
// zero `bytes` bytes (a multiple of 4, 4-byte aligned) with a kernel on `s`
int launch_zero_fill(void* dst, size_t bytes, hipStream_t s);
int validate_call(const pr_call_t& c, const pr_object_t* objs);
bool group_active(const pr_call_t& c);
bool group_train_active(const pr_call_t& c);
bool gate_active(const pr_call_t& c);
bool group_active(const pr_call_t& c);
int make_plan(const pr_call_t& c, const pr_object_t* objs, Plan* plan);
void bbox_split(const pr_object_model_t& m, float* lo, float* hi, float* size);
int build_mlp_layers(const pr_object_model_t& m, const ModelDims& d, const PackedLayout& l, const float* base,
                     MlpParams* p, bool split3 = false);

// ---------------------------------------------------------------------------------------------
// Backward pass building blocks (gemm.hip, backward.hip)
// ---------------------------------------------------------------------------------------------
struct GemmNN {            // C[M x n] (+)= A[M x k] . B[k x n], then optionally zeroed where mask <= 0
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const int32_t* rows;   // device scalar M
    int n, k;
    int accumulate;
    const float* mask; int ldm;
    int b_transposed;      // B element (k, n) is read from B[n * ldb + k] (C = A . W^T for a row-major W)
    int k_valid;           // rows of B beyond k_valid are zero (0 = all k rows exist)
};
int launch_gemm_nn(const GemmNN& p, int max_rows, hipStream_t s);

struct GemmTN {            // C[ni x nj] += sum_m A[m][i] B[m][j] ; bias[i] += sum_m A[m][i]
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    float* bias;           // or NULL
    const int32_t* rows;
    int ni, nj;
    int splits;
    float* partial;        // gemm_tn_scratch_floats(splits) floats
    float* bias_partial;   // inside the same scratch, or NULL
};
int launch_gemm_tn(const GemmTN& p, hipStream_t s);
size_t gemm_tn_scratch_floats(int splits);
// Several weight-gradient products over the same sample rows in one launch (+ one reduction launch): the layers of a chain.
constexpr int MAX_TN_GROUP = 16;
struct GemmTNGroup {
    GemmTN job[MAX_TN_GROUP];      // .partial / .bias_partial: one scratch region of gemm_tn_scratch_floats(splits) per job
    int count;
};
int launch_gemm_tn_group(const GemmTNGroup& g, hipStream_t s);

// Fused tile kernels of the backward pass (train_bwd.hip).  One job = one object instance of one model type.
struct HeadBwdJob {                 // feature-head backward, phase 1 (head layer 6 -> AdaIN 4) or 2 (head layer 3 -> AdaIN 1)
    const int32_t* total;           // device scalar: evaluated rows
    const int32_t* rec_flat;        // (cap) flat sample index of a row
    const int32_t* row_flags;       // (cap) bit 0 real sample, bit 1 passed the second AABB test
    int samples_per_frame;
    int phase;
    int frozen;                     // eval-mode BatchNorm: the statistics are constants
    const int32_t* stat_count;      // rows that entered the batch statistics
    float eps;
    const float* table; int table_stride;   // AdaIN table rows (frames) of the object; [g | b] of this phase's layer at goff / boff
    int goff, boff;
    // the operand.  phase 1: feature-row gradients of the compositing backward (rows of dead samples are cleared in place)
    float* g_in; int ld_gin, k_real;
    // phase 2: d loss / d x_hat of head layer 4 (phase 1's d_out) -> BatchNorm backward with the batch terms -> back in place
    float* d_in; const float* h_in;
    const float* mean_in; const float* var_in; const double* sums_in; int width_in;
    int kpad;                       // padded width of the operand (multiple of 16; phase 2: the row stride of d_in / h_in)
    Seg wt; int nblk;               // W^T fragments (K = kpad, N = nblk * 32): head6^T / head3^T
    // the layer that is differentiated
    const float* h; int ld;         // its raw (pre-BatchNorm) activations (cap, ld), ld = nblk * 32
    const float* mean; const float* var;
    int width;                      // real channels
    float* a_out;                   // (cap, ld) relu(AdaIN(h)), recomputed: right factor of the product's weight gradient
    float* d_out;                   // (cap, ld) d loss / d x_hat
    double* sums;                   // [sum d x_hat (ld) | sum d x_hat x_hat (ld)], zeroed by the caller
    float* dscale; float* dbias;    // (frames, MAX_WIDTH) d loss / d AdaIN scale / bias, zeroed by the caller
    int32_t* tile_counter;          // zeroed
    int split;                      // 1: `wt` is the bf16-triple packing of the segment, the product runs on six bf16 MFMAs
};
int launch_head_bwd_group(const HeadBwdJob* jobs, const long* max_rows, int count, hipStream_t s);

struct ChainBwdJob {                // backward chain of a ReLU MLP with one skip concatenation, entry fused in
    const int32_t* total; const int32_t* rec_flat; const int32_t* row_flags; int samples_per_frame;
    int entry;                      // 1: NeRF backbone behind head layer 0 and the sigma head; 0: ray bender behind its output head
    // entry 1: d loss / d x_hat of head layer 1 (phase 2's d_out) -> BatchNorm backward (in place) -> . head0 -> + sigma path
    float* d1; const float* h1; const float* mean1; const float* var1; const double* sums1;
    const int32_t* stat_count; int frozen; float eps;
    Seg w0t;
    const float* g_sigma;           // (N,R,P) d loss / d sigma from the compositing backward
    const uint8_t* in_scene; int in_scene_stride;
    const float* w_sigma;           // raw alpha_head.weight (W), NULL: no sigma head (skybox)
    float* gsr4;                    // (cap, 4) out: [d loss / d sigma, 0, 0, 0] per row (left factor of the sigma head's gradient), or NULL
    // entry 0: d loss / d raw bender output (cap, 4) and the raw output head (3, w_out_ld)
    const float* g_braw4; const float* w_out; int w_out_ld;
    // the chain
    int count, skip, W, Wpad, in_pad, in_real;
    Seg act_t[PR_MAX_LAYERS];       // l = 1 .. count - 1: W_l[:, :W]^T
    Seg in0_skip, in0_first;        // W_skip[:, W:]^T and W_0^T
    const unsigned char* bits; size_t bits_stride;   // ReLU masks of layers 0 .. count - 1 (bit images written by the forward pass)
    float* gstack; size_t g_stride; // out: pre-activation gradients of layers 0 .. count - 1, (cap, Wpad) each
    float* g_in; int ld_in;         // out: gradient of the network input (cap, ld_in)
    int32_t* tile_counter;          // zeroed
    int split;                      // 1: the segments are bf16-triple packings, every product runs on six bf16 MFMAs
};
int launch_chain_bwd_group(const ChainBwdJob* jobs, const long* max_rows, int count, hipStream_t s);

// Every weight-gradient product of a backward pass in one launch (k_gemm_tn_all in gemm.hip): the products of all layers of
// all objects as (job, split of TN_ALL_CHUNK sample rows, 128 x 128 tile) work items of a persistent grid.
constexpr int TN_ALL_MAX = 96;         // jobs per launch
#ifndef PR_TNALL_CHUNK
#define PR_TNALL_CHUNK 2048
#endif
constexpr int TN_ALL_CHUNK = PR_TNALL_CHUNK;     // sample rows per split
constexpr int TN_ALL_TILES = 6;        // claim slots per (job, split): the tiles of the largest gradient (256 x 384)
struct TnJob {             // C[ni x nj] += sum_m A[m][i] B[m][j] ; bias[i] += sum_m A[m][i]
    const float* A; const float* B;
    float* C; float* bias;             // bias: or NULL
    const int32_t* rows;               // device scalar: number of sample rows
    float* partial;                    // tn_all_partial_floats(ni, nj, row capacity) floats: [split][tile rows][tile cols], then the bias partials
    float* bias_partial;               // (set by the launcher)
    // optional side product on the tiles of the first row block: wgrad[j] += sum_m w[m] B[m][j], wbias += sum_m w[m] - the
    // gradient of a ONE-output layer that reads the same B (the density head beside head layer 0): as a product of its own it
    // would cost two full 128 x 128 tiles per 32 rows for 256 useful multiply-adds
    const float* w; int ldw;           // w[m * ldw]
    float* wgrad; float* wbias;        // (nj) and (1); wbias: or NULL
    float* w_partial;                  // (set by the launcher)
    int lda, ldb, ldc, ni, nj;
    int chain_next, head;              // (set by the launcher) jobs that share a destination: reduced together, in job order
};
struct TnAll {
    TnJob job[TN_ALL_MAX];
    int count;
    int32_t* counters;                 // 8 zeroed ints: the per-XCD claim counters
    int split_precision;               // 1: operands as fp16 pairs of scaled half slabs (k_gemm_tn_all_f16; -DPR_TNALL_F16=0: bf16 triples)
};
size_t tn_all_partial_floats(int ni, int nj, long max_rows);
int launch_gemm_tn_all(TnAll& g, const long* max_rows, hipStream_t s);

// Hutchinson divergence estimate e^T (d delta / d x) e of the ray bender (object_composer.py:582-601) as a
// forward-mode derivative through the saved bender activations; writes the dense (N,R,P) array `div`.
struct DivergenceParams {
    const int32_t* total; int max_rows;
    const int32_t* rec_flat; const int32_t* row_flags; const float* rec_pos;
    NoiseRef noise;                // N(0,1) (N,R,P,3)
    int positions;                 // P (index space of the probes)
    const float* bin; int bin_pad, benc, b_octaves;
    const float* bacts; size_t bact_stride; int BW, BWpad, b_count, b_skip, bin_real;
    const pr_linear_t* layers; pr_linear_t out_head;
    const float* braw;
    float lo[3], hi[3];
    int canonical;
    float* t0; float* ta; float* tb;   // scratch: (cap, bin_pad), (cap, BWpad) x 2
    float* tstack; size_t tstride; // or: every layer's tangent kept, layer l at tstack + l * tstride (ta / tb unused)
    float* div;                    // (N,R,P), zero-initialised by the caller; NULL: tangents only
};
int launch_divergence(const DivergenceParams& p, hipStream_t s);

// Kernel timing (bench.py): records an event pair on `s` around a launch when profiling is on.
struct ProfileScope {
    ProfileScope(int category, hipStream_t s);
    ~ProfileScope();
    int category_;
    hipStream_t stream_;
    hipEvent_t start_;
    bool active_;
};

// ---------------------------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): 128-bit counter, 64-bit key.
__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0,
                                              unsigned int k1, unsigned int* out) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned int n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ bool noise_present(const NoiseRef& n) { return n.ptr != nullptr || n.generate != 0; }

// element index of (ray g of THIS call, element e of `per_ray`) in the noise tensor's own index space: calls that were split
// along the rays address the rows of the unsplit tensor
__device__ __forceinline__ unsigned long long noise_index(const NoiseRef& n, long g, int per_ray, int e) {
    if (n.total_rays == n.rays) return (unsigned long long)g * per_ray + e;
    const long frame = g / n.rays, r = g - frame * n.rays;
    return (unsigned long long)(frame * n.total_rays + n.ray_offset + r) * per_ray + e;
}

// Philox key of a stream: fixed by the host from pr_call_t.noise_seed, or derived here from the seed word on the device
__device__ __forceinline__ void noise_key(const NoiseRef& n, unsigned int* k0, unsigned int* k1) {
    *k0 = n.key0;
    *k1 = n.key1;
    if (n.seed_dev) {
        const unsigned long long k = noise_mix(*n.seed_dev ^ n.salt);
        *k0 = (unsigned int)k;
        *k1 = (unsigned int)(k >> 32);
    }
}

__device__ __forceinline__ float noise_uniform(const NoiseRef& n, long g, int per_ray, int e) {   // U[0, 1)
    if (n.ptr) return n.ptr[(size_t)g * per_ray + e];      // explicit tensors arrive already sliced to the call's rays
    const unsigned long long idx = noise_index(n, g, per_ray, e);
    unsigned int x[4], k0, k1;
    noise_key(n, &k0, &k1);
    philox4x32_10((unsigned int)idx, (unsigned int)(idx >> 32), 0u, 0u, k0, k1, x);
    return (float)(x[0] >> 8) * 5.9604644775390625e-8f;   // 24 random bits * 2^-24
}

__device__ __forceinline__ float noise_normal(const NoiseRef& n, long g, int per_ray, int e) {    // N(0, 1), Box-Muller
    if (n.ptr) return n.ptr[(size_t)g * per_ray + e];
    const unsigned long long idx = noise_index(n, g, per_ray, e);
    unsigned int x[4], k0, k1;
    noise_key(n, &k0, &k1);
    philox4x32_10((unsigned int)idx, (unsigned int)(idx >> 32), 0u, 0u, k0, k1, x);
    const float u1 = (float)((x[0] >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
    const float u2 = (float)(x[1] >> 8) * 5.9604644775390625e-8f;          // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864769f * u2);
}

// torch.min / torch.max / clamp propagate NaN; fminf/fmaxf do not.
// Two persistent workgroups share a CU (and each of its four matrix pipes).  With EQUAL wave priorities the SIMD arbitrates
// between their K loops fairly, the two tiles stay in lock step, and their serial phases (epilogues, barriers, encodings)
// coincide - the matrix pipes then idle for the whole of those phases.  Resident workgroups therefore take alternating
// matrix-phase priorities: the arrival order on the CU (a counter per physical CU, keyed by the hardware id registers; only
// the parity matters, so it is never reset) decides which one outranks the other.  One table per translation unit.
static __device__ unsigned int g_cu_arrivals[4096];
__device__ __forceinline__ int cu_arrival_parity() {   // call from ONE thread of the workgroup
    const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID [3:0]
    return (int)(atomicAdd(&g_cu_arrivals[((xcc & 15u) << 8) | ((hw >> 8) & 255u)], 1u) & 1u);
}

__device__ __forceinline__ float nan_min(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ float nan_max(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float nan_clamp(float v, float lo, float hi) {
    // torch.clamp(v, min=lo, max=hi) = min(max(v, lo), hi), NaN stays NaN
    float r = (v < lo) ? lo : v;
    r = (r > hi) ? hi : r;
    return r;
}

// Object-frame ray of (frame, ray) exactly as RayHelper.transform_rays computes it:
// sum_j p_j * M[i][j] evaluated as ((p0*m0 + p1*m1) + p2*m2) (+ m3)   (ray_helper.py:1195-1199)
struct ObjRay { float o[3]; float d[3]; };
__device__ __forceinline__ ObjRay object_ray(const float* __restrict__ m /*3x4*/, const float* __restrict__ o,
                                            const float* __restrict__ d) {
    ObjRay r;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float m0 = m[i * 4 + 0], m1 = m[i * 4 + 1], m2 = m[i * 4 + 2], m3 = m[i * 4 + 3];
        r.o[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(o[0], m0), __fmul_rn(o[1], m1)), __fmul_rn(o[2], m2)), m3);
        r.d[i] = __fadd_rn(__fadd_rn(__fmul_rn(d[0], m0), __fmul_rn(d[1], m1)), __fmul_rn(d[2], m2));
    }
    return r;
}

__device__ __forceinline__ bool in_box(const float x, const float y, const float z, const float* lo, const float* hi) {
    return x >= lo[0] && x <= hi[0] && y >= lo[1] && y <= hi[1] && z >= lo[2] && z <= hi[2];
}
#endif

}  // namespace pr
