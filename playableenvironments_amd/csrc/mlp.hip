// Fused per-object MLP of the renderer on the gfx950 matrix cores (exact-fp32 MFMA).
//
//   [PositionalRayBender]  x/size -> annealed PE (+ deformation) -> 6 x (Linear 128 + ReLU, skip) ->
//                          Linear 128->3 -> * size -> clamp into the box -> x' = x + delta
//   [AdaInStyleNerfModel]  x'/size -> PE -> 8 x (Linear 256 + ReLU, skip) -> sigma head ;
//                          Linear 256->256 -> AdaIN -> ReLU -> Linear 256->128 -> AdaIN -> ReLU -> Linear 128->F
//   [SkyboxAdaInStyleNerfModelV3] same trunk on [o/size, d/|d|], sigma == 10
// (model/nerf_models/{positional_ray_bender_model,adain_style_nerf_model,skybox_adain_style_nerf_model_v3}.py).
//
// Structure (DESIGN.md "K3"): a persistent workgroup of 8 waves owns a tile of 64 compacted samples.
// Activations live in LDS as X[64][260] fp32 (row stride 260 -> conflict-free ds_read_b128), the
// positional encoding in E[64][132].  For a layer out = X . W^T each wave owns one 32-column block
// of the output for all 64 rows: A fragments (activations) come from LDS, B fragments (weights) are
// read straight from L2 in a host-prepared fragment order (one dwordx4 per lane per 4 MFMAs), the
// accumulators start from the bias, and the epilogue (ReLU / folded AdaIN + ReLU / feature store)
// runs on the accumulator registers.  v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain.
#include "pr_common.h"
#include "mlp_tile.h"

#ifndef PR_TRAINFWD_ABLATE
#define PR_TRAINFWD_ABLATE 0    // timing builds only (training forward): 1 = no activation saves, 2 = no ReLU bit images, 4 = no batch statistics
#endif

#include <cstddef>

#include <stdarg.h>
#include <stdlib.h>

namespace pr {


// ---------------------------------------------------------------------------------------------
// Dimensions and packed layout (host)
// ---------------------------------------------------------------------------------------------
int compute_dims(const pr_object_model_t& m, ModelDims* d) {
    PR_REQUIRE(m.kind == 0 || m.kind == 1, "unknown nerf model kind %d", m.kind);
    PR_REQUIRE(m.octaves >= 0 && m.octaves <= PR_MAX_OCTAVES, "octaves %d out of range", m.octaves);
    d->din = m.kind == 0 ? 3 : 6;
    d->enc = d->din + 2 * d->din * m.octaves;
    d->enc_pad = round_up(d->enc, 32);   // K loops advance two steps per iteration (2 x 8 fp32, 2 x 16 split)
    d->W = m.layers_width;
    d->Wpad = round_up(d->W, 32);
    d->W2 = m.layers_width / 2;
    d->W2pad = round_up(d->W2, 32);
    d->F = m.output_features;
    d->Fpad = round_up(d->F, 32);
    PR_REQUIRE(d->enc_pad <= MAX_ENC, "encoding width %d exceeds %d", d->enc, MAX_ENC);
    PR_REQUIRE(d->W >= 2 && d->Wpad <= MAX_WIDTH, "layers_width %d unsupported (max %d)", d->W, MAX_WIDTH);
    PR_REQUIRE(d->F >= 1 && d->Fpad <= MAX_WIDTH, "output_features %d unsupported (max %d)", d->F, MAX_WIDTH);
    PR_REQUIRE(m.backbone_count >= 2 && m.backbone_count <= PR_MAX_LAYERS, "backbone_layers_count %d unsupported",
               m.backbone_count);
    PR_REQUIRE(m.skip_layer_idx >= 1 && m.skip_layer_idx < m.backbone_count, "skip_layer_idx %d unsupported",
               m.skip_layer_idx);
    d->benc = d->bin = d->bin_pad = d->BW = d->BWpad = 0;
    if (m.has_bender) {
        PR_REQUIRE(m.kind == 0, "a ray bender on the skybox model is not supported");
        PR_REQUIRE(m.bender_octaves >= 0 && m.bender_octaves <= PR_MAX_OCTAVES, "bender octaves out of range");
        d->benc = 3 + 6 * m.bender_octaves;
        d->bin = d->benc + m.deformation_features;
        d->bin_pad = round_up(d->bin, 32);
        d->BW = m.bender_width;
        d->BWpad = round_up(d->BW, 32);
        PR_REQUIRE(d->bin_pad <= MAX_ENC, "bender input width %d exceeds %d", d->bin, MAX_ENC);
        PR_REQUIRE(d->BW >= 1 && d->BWpad <= MAX_WIDTH, "bender width %d unsupported", d->BW);
        PR_REQUIRE(m.bender_count >= 2 && m.bender_count <= PR_MAX_LAYERS, "bender layers_count %d unsupported",
                   m.bender_count);
        PR_REQUIRE(m.bender_skip >= 1 && m.bender_skip < m.bender_count, "bender skip_layer_idx %d unsupported",
                   m.bender_skip);
    }
    return PR_OK;
}

static int seg_floats(int nblk, int kpad) { return nblk * (kpad / 8) * 256; }

int compute_layout(const pr_object_model_t& m, const ModelDims& d, PackedLayout* l) {
    int off = 0;
    memset(l, 0, sizeof(*l));
    if (m.has_bender) {
        const int nb = d.BWpad / 32;
        for (int j = 0; j < m.bender_count; ++j) {
            l->b_seg_off[j][0] = off;
            off += seg_floats(nb, j == 0 ? d.bin_pad : d.BWpad);
            l->b_seg_off[j][1] = off;
            if (j == m.bender_skip) off += seg_floats(nb, d.bin_pad);
            l->b_bias_off[j] = off;
            off += d.BWpad;
        }
        l->b_out_off = off;
        off += round_up(3 * d.BWpad, 32);
    }
    const int nb = d.Wpad / 32;
    for (int i = 0; i < m.backbone_count; ++i) {
        l->n_seg_off[i][0] = off;
        off += seg_floats(nb, i == 0 ? d.enc_pad : d.Wpad);
        l->n_seg_off[i][1] = off;
        if (i == m.skip_layer_idx) off += seg_floats(nb, d.enc_pad);
        l->n_bias_off[i] = off;
        off += d.Wpad;
    }
    l->sigma_off = off;
    off += d.Wpad + 32;
    l->h0_off = off;
    off += seg_floats(d.Wpad / 32, d.Wpad);
    l->h3_off = off;
    off += seg_floats(d.W2pad / 32, d.Wpad);
    l->h6_off = off;
    off += seg_floats(d.Fpad / 32, d.W2pad);
    l->h6_bias_off = off;
    off += d.Fpad;
    // transposed segments of the backward pass: K = the layer's outputs, N = its inputs
    if (m.has_bender) {
        for (int j = 1; j < m.bender_count; ++j) {
            l->t_b_act[j] = off;
            off += seg_floats(d.BWpad / 32, d.BWpad);
        }
        l->t_b_skip = off;
        off += seg_floats(d.bin_pad / 32, d.BWpad);
        l->t_b_first = off;
        off += seg_floats(d.bin_pad / 32, d.BWpad);
    }
    for (int i = 1; i < m.backbone_count; ++i) {
        l->t_n_act[i] = off;
        off += seg_floats(d.Wpad / 32, d.Wpad);
    }
    l->t_n_skip = off;
    off += seg_floats(d.enc_pad / 32, d.Wpad);
    l->t_n_first = off;
    off += seg_floats(d.enc_pad / 32, d.Wpad);
    l->t_h0 = off;
    off += seg_floats(d.Wpad / 32, d.Wpad);
    l->t_h3 = off;
    off += seg_floats(d.Wpad / 32, d.W2pad);
    l->t_h6 = off;
    off += seg_floats(d.W2pad / 32, d.Fpad);
    // ... and as bf16 triples (split-precision backward): 3 planes x 2 bytes per weight = 1.5 floats
    auto take3 = [&](int nblk, int kpad) { const int at = off; off += seg_floats(nblk, kpad) / 2 * 3; return at; };
    if (m.has_bender) {
        for (int j = 1; j < m.bender_count; ++j) l->t3_b_act[j] = take3(d.BWpad / 32, d.BWpad);
        l->t3_b_skip = take3(d.bin_pad / 32, d.BWpad);
        l->t3_b_first = take3(d.bin_pad / 32, d.BWpad);
    }
    for (int i = 1; i < m.backbone_count; ++i) l->t3_n_act[i] = take3(d.Wpad / 32, d.Wpad);
    l->t3_n_skip = take3(d.enc_pad / 32, d.Wpad);
    l->t3_n_first = take3(d.enc_pad / 32, d.Wpad);
    l->t3_h0 = take3(d.Wpad / 32, d.Wpad);
    l->t3_h3 = take3(d.Wpad / 32, d.W2pad);
    l->t3_h6 = take3(d.W2pad / 32, d.Fpad);
    // forward segments of a training call's phase 1 in split precision: fp16 (hi, lo) fragment pairs, the layout of the split
    // evaluation kernel (k_pack kind 2: the same number of bytes as the fp32 fragments)
    auto take2 = [&](int nblk, int kpad) { const int at = off; off += seg_floats(nblk, kpad); return at; };
    if (m.has_bender) {
        const int nbb = d.BWpad / 32;
        for (int j = 0; j < m.bender_count; ++j) {
            l->b_seg3[j][0] = take2(nbb, j == 0 ? d.bin_pad : d.BWpad);
            if (j == m.bender_skip) l->b_seg3[j][1] = take2(nbb, d.bin_pad);
        }
    }
    for (int i = 0; i < m.backbone_count; ++i) {
        l->n_seg3[i][0] = take2(d.Wpad / 32, i == 0 ? d.enc_pad : d.Wpad);
        if (i == m.skip_layer_idx) l->n_seg3[i][1] = take2(d.Wpad / 32, d.enc_pad);
    }
    l->h0_3 = take2(d.Wpad / 32, d.Wpad);
    l->total = off;
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight packing.  Fragment order of one K-segment of a Linear (out = n, in = k):
//   dst[((nb * kq + q) * 64 + lane) * 4 + e] = W[nb*32 + (lane & 31)][col_off + (lane >> 5) * 4*kq + 4*q + e]
// i.e. lanes 0-31 sweep the first half of the (padded) K range and lanes 32-63 the second half,
// matching how the kernel reads the A operand; rows/cols beyond the real shape are zero.
// ---------------------------------------------------------------------------------------------
struct PackJob {
    const float* src;
    float* dst;
    int kind;       // 0 = fp32 fragment-ordered matrix segment, 1 = padded vector / raw row copy,
                    // 2 = fp16 hi/lo split fragments (same byte size as kind 0), 3 = bf16 triples (1.5 x the size of kind 0; `count`
                    // counts 32-bit words)
    int in_total;   // row stride of src
    int col_off;
    int k_real, n_real, kq, nblk;
    int count;      // elements of dst
    int transposed; // kind 0: element (n, k) is read from src[k * in_total + col_off + n] (the backward chain's W^T)
    int scale_log2; // kind 2: the weights are multiplied by 2^scale_log2 before they are split (0 for the evaluation packing)
};
constexpr int MAX_PACK_JOBS = 256;
struct PackJobs {
    PackJob job[MAX_PACK_JOBS];
    int n;
    int seg_kind;   // kind of the weight-segment jobs: 0 = fp32 fragments, 2 = fp16 hi / lo fragment pairs
};

__global__ __launch_bounds__(256) void k_pack(PackJobs jobs) {
    const PackJob& j = jobs.job[blockIdx.y];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < j.count; idx += gridDim.x * 256) {
        float v = 0.f;
        if (j.kind == 0) {
            const int e = idx & 3;
            const int lane = (idx >> 2) & 63;
            const int rest = idx >> 8;
            const int q = rest % j.kq;
            const int nb = rest / j.kq;
            const int n = nb * 32 + (lane & 31);
            const int k = (lane >> 5) * 4 * j.kq + 4 * q + e;
            if (n < j.n_real && k < j.k_real)
                v = j.transposed ? j.src[(size_t)k * j.in_total + j.col_off + n] : j.src[(size_t)n * j.in_total + j.col_off + k];
        } else if (j.kind == 2) {
            // split fragments: per (column block, 16-wide K step): 64 lanes x 8 halves "hi", then the same for
            // "lo" = w - hi (mostly an fp16 subnormal: the matrix pipe takes those exactly).  Lane l holds W[nb*32 + (l & 31)][16 s + 8 (l >> 5) + e], e = 0..7.
            unsigned short out[2];
            for (int t = 0; t < 2; ++t) {
                const int h = idx * 2 + t;
                const int e = h & 7;
                const int lane = (h >> 3) & 63;
                const int part = (h >> 9) & 1;
                const int rest = h >> 10;
                const int ks = j.kq >> 1;
                const int sidx = rest % ks;
                const int nb = rest / ks;
                const int n = nb * 32 + (lane & 31);
                const int k = 16 * sidx + 8 * (lane >> 5) + e;
                float w = 0.f;
                if (n < j.n_real && k < j.k_real)
                    w = ldexpf(j.transposed ? j.src[(size_t)k * j.in_total + j.col_off + n] : j.src[(size_t)n * j.in_total + j.col_off + k],
                               j.scale_log2);
                // an UNSCALED weight beyond fp16's range (|w| > 65504: evaluation packing) saturates; a weight whose training-time
                // scaled form w x 2^8 leaves the range (|w| >= 255.9) is left to overflow: hi = inf, lo = NaN - the step's outputs and
                // gradients turn NaN, which a trainer notices, instead of silently training against a saturated copy of the weight
                if (j.scale_log2 == 0) w = fminf(fmaxf(w, -65504.0f), 65504.0f);
                const _Float16 hi = (_Float16)w;
                const _Float16 lo = (_Float16)(w - (float)hi);
                const _Float16 sel = part ? lo : hi;
                out[t] = *reinterpret_cast<const unsigned short*>(&sel);
            }
            reinterpret_cast<unsigned int*>(j.dst)[idx] = (unsigned int)out[0] | ((unsigned int)out[1] << 16);
            continue;
        } else if (j.kind == 3) {
            // bf16 triples of a segment, w = b1 + b2 + b3 (each term what is left, rounded to the nearest bf16):
            // word idx holds the elements e, e + 1 of [column block][K step of 16][plane][lane]; lane l carries
            // W(n = nb*32 + (l & 31), k = 16 s + 8 (l >> 5) + e), e = 0..7 - the B fragment of v_mfma_f32_32x32x16_bf16
            unsigned int out[2];
            for (int t = 0; t < 2; ++t) {
                const int h = idx * 2 + t;
                const int e = h & 7;
                const int lane = (h >> 3) & 63;
                const int rest = h >> 9;
                const int plane = rest % 3;
                const int step = (rest / 3) % (j.kq >> 1);
                const int nb = (rest / 3) / (j.kq >> 1);
                const int n = nb * 32 + (lane & 31);
                const int k = 16 * step + 8 * (lane >> 5) + e;
                float w = 0.f;
                if (n < j.n_real && k < j.k_real)
                    w = j.transposed ? j.src[(size_t)k * j.in_total + j.col_off + n] : j.src[(size_t)n * j.in_total + j.col_off + k];
                // (round to nearest, like the activations' split in the kernels)
                const __bf16 b1 = (__bf16)w;
                const float r1 = w - (float)b1;
                const __bf16 b2 = (__bf16)r1;
                const __bf16 b3 = (__bf16)(r1 - (float)b2);
                const __bf16 sel = plane == 0 ? b1 : (plane == 1 ? b2 : b3);
                out[t] = *reinterpret_cast<const unsigned short*>(&sel);
            }
            reinterpret_cast<unsigned int*>(j.dst)[idx] = out[0] | (out[1] << 16);
            continue;
        } else {
            // rows of length kq (padded) from rows of length k_real; n_real rows
            const int row = idx / j.kq, c = idx % j.kq;
            if (j.src && row < j.n_real && c < j.k_real) v = j.src[(size_t)row * j.in_total + j.col_off + c];
        }
        j.dst[idx] = v;
    }
}

static int add_seg(PackJobs* js, const pr_linear_t& lin, int col_off, int k_real, int kpad, int npad, float* dst) {
    PR_REQUIRE(js->n < MAX_PACK_JOBS, "too many pack jobs");
    PR_REQUIRE(lin.weight != nullptr, "missing weight pointer");
    PackJob& j = js->job[js->n++];
    j.src = lin.weight;
    j.dst = dst;
    j.kind = js->seg_kind;
    j.in_total = lin.in_features;
    j.col_off = col_off;
    j.k_real = k_real;
    j.n_real = lin.out_features;
    j.kq = kpad / 8;
    j.nblk = npad / 32;
    j.count = seg_floats(j.nblk, kpad);
    j.transposed = 0;
    j.scale_log2 = 0;
    return PR_OK;
}

// W^T of a Linear as a segment: out[m][n] = sum_k G[m][k] W[k][col_off + n] - K = the layer's outputs (k_real of kpad), N = n_real
// of its inputs starting at column col_off (npad)
static int add_seg_t(PackJobs* js, const pr_linear_t& lin, int col_off, int k_real, int kpad, int n_real, int npad, float* dst) {
    PR_TRY(add_seg(js, lin, col_off, k_real, kpad, npad, dst));
    PackJob& j = js->job[js->n - 1];
    j.kind = 0;            // fp32 fragments in every packing (differentiable calls run on the exact kernel)
    j.n_real = n_real;
    j.transposed = 1;
    return PR_OK;
}

// the same W^T segment for the split-precision backward chains: fp16 (hi, lo) pairs of w x 2^8 (kind 2; the chains scale their
// gradient tiles into fp16's range, train_bwd.hip) - or bf16 triples (kind 3, -DPR_CHAIN_BF16: 1.5 x the bytes, six MFMAs per product)
static int add_seg_t3(PackJobs* js, const pr_linear_t& lin, int col_off, int k_real, int kpad, int n_real, int npad, float* dst) {
    PR_TRY(add_seg_t(js, lin, col_off, k_real, kpad, n_real, npad, dst));
    PackJob& j = js->job[js->n - 1];
#ifdef PR_CHAIN_BF16
    j.kind = 3;
    j.count = j.count / 2 * 3;      // 32-bit words: two bf16 each, three planes
#else
    j.kind = 2;                     // (in the region sized for the triples)
    j.scale_log2 = TRAIN_SPLIT_WEIGHT_SCALE_LOG2;
#endif
    return PR_OK;
}

// a forward segment as fp16 (hi, lo) fragment pairs (kind 2) inside the fp32 packing: phase 1 of a split-precision training call
static int add_seg3(PackJobs* js, const pr_linear_t& lin, int col_off, int k_real, int kpad, int npad, float* dst) {
    PR_TRY(add_seg(js, lin, col_off, k_real, kpad, npad, dst));
    js->job[js->n - 1].kind = 2;
    // weights of 0.01 - 0.1 have their lo half in fp16's subnormal range (relative error 2^-25 / |w|: harmless for a rendered value,
    // 10 x fp32's on the density head's bias gradient, which sums over every sample - measured); scaled by 2^8 the lo halves of
    // |w| > 0.001 are normal numbers and |w| < 255 stays in range (beyond, k_pack lets the scaled weight overflow: NaN gradients, loud).  The kernel starts its accumulators
    // at bias x 2^8 (x the operand tile's scale) and divides both out behind the K loops - exact.
    js->job[js->n - 1].scale_log2 = TRAIN_SPLIT_WEIGHT_SCALE_LOG2;
    return PR_OK;
}

static int add_vec(PackJobs* js, const float* src, int rows, int row_real, int row_stride, int row_pad, float* dst) {
    PR_REQUIRE(js->n < MAX_PACK_JOBS, "too many pack jobs");
    PackJob& j = js->job[js->n++];
    j.src = src;
    j.dst = dst;
    j.kind = 1;
    j.in_total = row_stride;
    j.col_off = 0;
    j.k_real = row_real;
    j.n_real = rows;
    j.kq = row_pad;
    j.nblk = 0;
    j.count = rows * row_pad;
    j.transposed = 0;
    return PR_OK;
}

#define PR_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != PR_OK) return _r;  \
    } while (0)

static int check_linear(const pr_linear_t& l, int out, int in, const char* name) {
    PR_REQUIRE(l.weight != nullptr, "%s: weight pointer is NULL", name);
    PR_REQUIRE(l.out_features == out && l.in_features == in, "%s: shape (%d, %d), expected (%d, %d)", name,
               l.out_features, l.in_features, out, in);
    return PR_OK;
}

static int build_pack_jobs(const pr_object_model_t& m, const ModelDims& d, const PackedLayout& l, float* base,
                           PackJobs* js) {
    js->n = 0;
    if (m.has_bender) {
        for (int j = 0; j < m.bender_count; ++j) {
            const int in = (j == 0) ? d.bin : (j == m.bender_skip ? d.BW + d.bin : d.BW);
            PR_TRY(check_linear(m.bender[j], d.BW, in, "ray_bender.backbone_layers"));
            PR_REQUIRE(m.bender[j].bias != nullptr, "ray_bender.backbone_layers: bias is NULL");
            if (j == 0) {
                PR_TRY(add_seg(js, m.bender[j], 0, d.bin, d.bin_pad, d.BWpad, base + l.b_seg_off[j][0]));
            } else {
                PR_TRY(add_seg(js, m.bender[j], 0, d.BW, d.BWpad, d.BWpad, base + l.b_seg_off[j][0]));
                if (j == m.bender_skip)
                    PR_TRY(add_seg(js, m.bender[j], d.BW, d.bin, d.bin_pad, d.BWpad, base + l.b_seg_off[j][1]));
            }
            PR_TRY(add_vec(js, m.bender[j].bias, 1, d.BW, d.BW, d.BWpad, base + l.b_bias_off[j]));
        }
        PR_TRY(check_linear(m.bender_out, 3, d.BW, "ray_bender.output_head"));
        PR_TRY(add_vec(js, m.bender_out.weight, 3, d.BW, d.BW, d.BWpad, base + l.b_out_off));
    }
    for (int i = 0; i < m.backbone_count; ++i) {
        const int in = (i == 0) ? d.enc : (i == m.skip_layer_idx ? d.W + d.enc : d.W);
        PR_TRY(check_linear(m.backbone[i], d.W, in, "nerf_model.backbone_layers"));
        PR_REQUIRE(m.backbone[i].bias != nullptr, "nerf_model.backbone_layers: bias is NULL");
        if (i == 0) {
            PR_TRY(add_seg(js, m.backbone[i], 0, d.enc, d.enc_pad, d.Wpad, base + l.n_seg_off[i][0]));
        } else {
            PR_TRY(add_seg(js, m.backbone[i], 0, d.W, d.Wpad, d.Wpad, base + l.n_seg_off[i][0]));
            if (i == m.skip_layer_idx)
                PR_TRY(add_seg(js, m.backbone[i], d.W, d.enc, d.enc_pad, d.Wpad, base + l.n_seg_off[i][1]));
        }
        PR_TRY(add_vec(js, m.backbone[i].bias, 1, d.W, d.W, d.Wpad, base + l.n_bias_off[i]));
    }
    if (m.kind == 0) {
        PR_TRY(check_linear(m.alpha_head, 1, d.W, "nerf_model.alpha_head"));
        PR_REQUIRE(m.alpha_head.bias != nullptr, "nerf_model.alpha_head: bias is NULL");
        PR_TRY(add_vec(js, m.alpha_head.weight, 1, d.W, d.W, d.Wpad, base + l.sigma_off));
        PR_TRY(add_vec(js, m.alpha_head.bias, 1, 1, 1, 32, base + l.sigma_off + d.Wpad));
    } else {
        PR_TRY(add_vec(js, nullptr, 1, 0, 0, d.Wpad + 32, base + l.sigma_off));
    }
    PR_TRY(check_linear(m.head0, d.W, d.W, "nerf_model.features_head.0"));
    PR_TRY(check_linear(m.head3, d.W2, d.W, "nerf_model.features_head.3"));
    PR_TRY(check_linear(m.head6, d.F, d.W2, "nerf_model.features_head.6"));
    PR_REQUIRE(m.head6.bias != nullptr, "nerf_model.features_head.6: bias is NULL");
    PR_TRY(add_seg(js, m.head0, 0, d.W, d.Wpad, d.Wpad, base + l.h0_off));
    PR_TRY(add_seg(js, m.head3, 0, d.W, d.Wpad, d.W2pad, base + l.h3_off));
    PR_TRY(add_seg(js, m.head6, 0, d.W2, d.W2pad, d.Fpad, base + l.h6_off));
    PR_TRY(add_vec(js, m.head6.bias, 1, d.F, d.F, d.Fpad, base + l.h6_bias_off));
    if (js->seg_kind != 0) return PR_OK;      // the backward pass runs on the fp32 packing only
    if (m.has_bender) {
        for (int j = 1; j < m.bender_count; ++j)
            PR_TRY(add_seg_t(js, m.bender[j], 0, d.BW, d.BWpad, d.BW, d.BWpad, base + l.t_b_act[j]));
        PR_TRY(add_seg_t(js, m.bender[m.bender_skip], d.BW, d.BW, d.BWpad, d.bin, d.bin_pad, base + l.t_b_skip));
        PR_TRY(add_seg_t(js, m.bender[0], 0, d.BW, d.BWpad, d.bin, d.bin_pad, base + l.t_b_first));
    }
    for (int i = 1; i < m.backbone_count; ++i)
        PR_TRY(add_seg_t(js, m.backbone[i], 0, d.W, d.Wpad, d.W, d.Wpad, base + l.t_n_act[i]));
    PR_TRY(add_seg_t(js, m.backbone[m.skip_layer_idx], d.W, d.W, d.Wpad, d.enc, d.enc_pad, base + l.t_n_skip));
    PR_TRY(add_seg_t(js, m.backbone[0], 0, d.W, d.Wpad, d.enc, d.enc_pad, base + l.t_n_first));
    PR_TRY(add_seg_t(js, m.head0, 0, d.W, d.Wpad, d.W, d.Wpad, base + l.t_h0));
    PR_TRY(add_seg_t(js, m.head3, 0, d.W2, d.W2pad, d.W, d.Wpad, base + l.t_h3));
    PR_TRY(add_seg_t(js, m.head6, 0, d.F, d.Fpad, d.W2, d.W2pad, base + l.t_h6));
    if (m.has_bender) {
        for (int j = 1; j < m.bender_count; ++j)
            PR_TRY(add_seg_t3(js, m.bender[j], 0, d.BW, d.BWpad, d.BW, d.BWpad, base + l.t3_b_act[j]));
        PR_TRY(add_seg_t3(js, m.bender[m.bender_skip], d.BW, d.BW, d.BWpad, d.bin, d.bin_pad, base + l.t3_b_skip));
        PR_TRY(add_seg_t3(js, m.bender[0], 0, d.BW, d.BWpad, d.bin, d.bin_pad, base + l.t3_b_first));
    }
    for (int i = 1; i < m.backbone_count; ++i)
        PR_TRY(add_seg_t3(js, m.backbone[i], 0, d.W, d.Wpad, d.W, d.Wpad, base + l.t3_n_act[i]));
    PR_TRY(add_seg_t3(js, m.backbone[m.skip_layer_idx], d.W, d.W, d.Wpad, d.enc, d.enc_pad, base + l.t3_n_skip));
    PR_TRY(add_seg_t3(js, m.backbone[0], 0, d.W, d.Wpad, d.enc, d.enc_pad, base + l.t3_n_first));
    PR_TRY(add_seg_t3(js, m.head0, 0, d.W, d.Wpad, d.W, d.Wpad, base + l.t3_h0));
    PR_TRY(add_seg_t3(js, m.head3, 0, d.W2, d.W2pad, d.W, d.Wpad, base + l.t3_h3));
    PR_TRY(add_seg_t3(js, m.head6, 0, d.F, d.Fpad, d.W2, d.W2pad, base + l.t3_h6));
    if (m.has_bender) {
        for (int j = 0; j < m.bender_count; ++j) {
            if (j == 0) {
                PR_TRY(add_seg3(js, m.bender[j], 0, d.bin, d.bin_pad, d.BWpad, base + l.b_seg3[j][0]));
            } else {
                PR_TRY(add_seg3(js, m.bender[j], 0, d.BW, d.BWpad, d.BWpad, base + l.b_seg3[j][0]));
                if (j == m.bender_skip) PR_TRY(add_seg3(js, m.bender[j], d.BW, d.bin, d.bin_pad, d.BWpad, base + l.b_seg3[j][1]));
            }
        }
    }
    for (int i = 0; i < m.backbone_count; ++i) {
        if (i == 0) {
            PR_TRY(add_seg3(js, m.backbone[i], 0, d.enc, d.enc_pad, d.Wpad, base + l.n_seg3[i][0]));
        } else {
            PR_TRY(add_seg3(js, m.backbone[i], 0, d.W, d.Wpad, d.Wpad, base + l.n_seg3[i][0]));
            if (i == m.skip_layer_idx) PR_TRY(add_seg3(js, m.backbone[i], d.W, d.enc, d.enc_pad, d.Wpad, base + l.n_seg3[i][1]));
        }
    }
    PR_TRY(add_seg3(js, m.head0, 0, d.W, d.Wpad, d.Wpad, base + l.h0_3));
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// AdaIN fold: for every (frame, object)  [scale | bias] = Linear(style)   (model/layers/adain.py:30-32)
// and eval-mode BatchNorm1d(affine=False) folded in:  y = h * g + b  with
//   g = scale / sqrt(running_var + eps),  b = bias - running_mean * g       (adain.py:47,58-59)
// ---------------------------------------------------------------------------------------------
// (one thread per channel, the channels of a frame spread over gridDim.y workgroups; the style dot products keep their
// left-to-right fmaf order, the weight rows are read 16 bytes at a time with eight loads in flight - as a scalar loop the
// kernel was one L2 round trip per style feature: 30 us for 49 k multiply-adds)
__device__ __forceinline__ void adain_fold_body(const FoldParams& p) {
    const int n = blockIdx.x;
    const float* style = p.style + ((size_t)n * p.objects + p.object_index) * p.S;
    float* row = p.table + (size_t)n * p.row_floats;
    const int total = p.Wpad + p.W2pad;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < total; c += gridDim.y * 256) {
        const bool first = c < p.Wpad;
        const int ch = first ? c : c - p.Wpad;
        const int width = first ? p.W : p.W2;
        const pr_linear_t& a = first ? p.affine1 : p.affine4;
        const float* mean = first ? p.bn1_mean : p.bn4_mean;
        const float* var = first ? p.bn1_var : p.bn4_var;
        float g = 0.f, b = 0.f;
        if (ch < width) {
            float scale = a.bias[ch], bias = a.bias[width + ch];
            const float* ws = a.weight + (size_t)ch * p.S;
            const float* wb = a.weight + (size_t)(width + ch) * p.S;
            int s = 0;
            if ((p.S & 3) == 0 && ((reinterpret_cast<size_t>(a.weight) | reinterpret_cast<size_t>(style)) & 15) == 0) {
                for (; s + 16 <= p.S; s += 16) {
                    float4 w1[4], w2[4], v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w1[q] = *reinterpret_cast<const float4*>(ws + s + 4 * q);
                        w2[q] = *reinterpret_cast<const float4*>(wb + s + 4 * q);
                        v[q] = *reinterpret_cast<const float4*>(style + s + 4 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        scale = fmaf(w1[q].x, v[q].x, scale); bias = fmaf(w2[q].x, v[q].x, bias);
                        scale = fmaf(w1[q].y, v[q].y, scale); bias = fmaf(w2[q].y, v[q].y, bias);
                        scale = fmaf(w1[q].z, v[q].z, scale); bias = fmaf(w2[q].z, v[q].z, bias);
                        scale = fmaf(w1[q].w, v[q].w, scale); bias = fmaf(w2[q].w, v[q].w, bias);
                    }
                }
            }
            for (; s < p.S; ++s) {
                scale = fmaf(ws[s], style[s], scale);
                bias = fmaf(wb[s], style[s], bias);
            }
            const float inv = 1.0f / sqrtf(var[ch] + p.eps);
            g = scale * inv;
            b = bias - mean[ch] * g;
        }
        if (first) {
            row[ch] = g;
            row[p.Wpad + ch] = b;
        } else {
            row[2 * p.Wpad + ch] = g;
            row[2 * p.Wpad + p.W2pad + ch] = b;
        }
    }
}

__global__ __launch_bounds__(256) void k_adain_fold(FoldParams p) { adain_fold_body(p); }

// the objects of an evaluation call in one launch (blockIdx.z = object): a fold is a few microseconds of work, so a launch per
// object is mostly launch gaps (4 of them = 2 % of a native 11 520-ray evaluation frame)
struct FoldGroup { FoldParams job[PR_MAX_OBJECTS]; };
__global__ __launch_bounds__(256) void k_adain_fold_group(FoldGroup g) {
    const FoldParams& p = g.job[blockIdx.z];
    if ((int)blockIdx.x >= p.frames || (int)blockIdx.y * 256 >= p.Wpad + p.W2pad) return;
    adain_fold_body(p);
}

int launch_adain_fold(const FoldParams& p, hipStream_t s) {
    PR_REQUIRE(p.affine1.weight && p.affine1.bias && p.affine4.weight && p.affine4.bias && p.bn1_mean && p.bn1_var &&
                   p.bn4_mean && p.bn4_var,
               "AdaIN parameters missing");
    PR_REQUIRE(p.affine1.out_features == 2 * p.W && p.affine1.in_features == p.S, "features_head.1 affine shape");
    PR_REQUIRE(p.affine4.out_features == 2 * p.W2 && p.affine4.in_features == p.S, "features_head.4 affine shape");
    hipLaunchKernelGGL(k_adain_fold, dim3(p.frames, (p.Wpad + p.W2pad + 255) / 256), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

int launch_adain_fold_group(const FoldParams* jobs, int count, hipStream_t s) {
    PR_REQUIRE(count >= 1 && count <= PR_MAX_OBJECTS, "adain fold group: %d objects", count);
    static thread_local FoldGroup g;
    int frames = 0, blocks = 0;
    for (int k = 0; k < count; ++k) {
        const FoldParams& p = jobs[k];
        PR_REQUIRE(p.affine1.weight && p.affine1.bias && p.affine4.weight && p.affine4.bias && p.bn1_mean && p.bn1_var &&
                       p.bn4_mean && p.bn4_var,
                   "AdaIN parameters missing");
        PR_REQUIRE(p.affine1.out_features == 2 * p.W && p.affine1.in_features == p.S, "features_head.1 affine shape");
        PR_REQUIRE(p.affine4.out_features == 2 * p.W2 && p.affine4.in_features == p.S, "features_head.4 affine shape");
        g.job[k] = p;
        frames = std::max(frames, p.frames);
        blocks = std::max(blocks, (p.Wpad + p.W2pad + 255) / 256);
    }
    hipLaunchKernelGGL(k_adain_fold_group, dim3(frames, blocks, count), dim3(256), 0, s, g);
    PR_LAUNCH_CHECK();
    return PR_OK;
}


// The three feature-head layers on the 64 rows in X whose destinations are in S.dest, then the write-out.
__device__ __forceinline__ void head_on_tile(Smem& S, const MlpParams& p, EncRegs& enc, int valid_rows) {
    for (int l = p.n_backbone; l < p.n_layers; ++l) run_layer(p.layers[l], S, p, 0, 0, enc);
    write_rows_indirect(S, p.feat, p.F, p.F);
    if (threadIdx.x == 0 && p.head_count) atomicAdd(p.head_count, valid_rows);
    __syncthreads();   // the next prologue overwrites flags / X
}

// Sigma-gated feature head (eval, no noise).  A sample whose raw density is <= 0 has alpha = 1 - exp(-relu(sigma) dt) = 0
// exactly, in its object's list and in the merged list alike, so its feature row is never read by the compositing
// kernel (composite.hip: take = row >= 0 && (w1 != 0 || w2 != 0)); neither are the rows of samples that failed the
// second AABB test.  The head (3 of the 11 matrix products, 20 % of the FLOPs of a sample) therefore runs on the LIVE rows
// only.  Rows are independent, so live rows of different tiles can share a head tile: every workgroup keeps a stack of
// up to 63 pending live rows (their 256-wide backbone outputs) in global memory.  After the sigma head of a tile with L
// live rows and P pending ones:
//   P + L >= 64: the dead slots of X are refilled with 64 - L pending rows (popped from the stack), the head runs on a
//                full tile, in place;
//   otherwise:   the L live rows are pushed on the stack and the tile is done.
// A tile that is entirely live (or dead) never touches the stack; what is left on it when the workgroup runs out of
// tiles is flushed through one partial head tile.  MFMA rows do not interact, so the results are bit-identical to the
// ungated kernel's for every row that is read downstream.
__device__ __forceinline__ int gated_head(Smem& S, const MlpParams& p, int tile_base, int pending, EncRegs& enc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long live = __ballot((S.flags[lane] & 4) != 0);   // the same value in every wave
    const int L = __popcll(live);
    float* pact = p.pend_act + (size_t)blockIdx.x * TILE_M * p.Wpad;
    int* pmeta = p.pend_meta + (size_t)blockIdx.x * TILE_M * 2;
    const unsigned long long below = (1ull << lane) - 1ull;   // lanes below this one (tid < 64: rows below this row)
    const int w4 = p.Wpad >> 2;
    if (L == 0) {
        __syncthreads();   // every wave has read the flags; the next prologue may overwrite them
        return pending;
    }
    if (pending + L >= TILE_M) {
        const int need = TILE_M - L;
        if (tid == 0) S.uniform_frame = 1;
        if (tid < TILE_M) {
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
        if (need) {
            for (int idx = tid; idx < TILE_M * w4; idx += MLP_THREADS) {
                const int row = idx / w4, c = (idx - row * w4) * 4;
                const int slot = S.src[row];
                if (slot >= 0)
                    *reinterpret_cast<float4*>(S.X + row * LDX + c) = *reinterpret_cast<const float4*>(pact + (size_t)slot * p.Wpad + c);
            }
        }
        if (tid < TILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
        __syncthreads();
        head_on_tile(S, p, enc, TILE_M);
        return pending - need;
    }
    if (tid < TILE_M) {
        int slot = -1;
        if ((live >> tid) & 1ull) {
            slot = pending + __popcll(live & below);
            pmeta[2 * slot] = tile_base + tid;
            pmeta[2 * slot + 1] = S.frame[tid];
        }
        S.src[tid] = slot;
    }
    __syncthreads();
    for (int idx = tid; idx < TILE_M * w4; idx += MLP_THREADS) {
        const int row = idx / w4, c = (idx - row * w4) * 4;
        const int slot = S.src[row];
        if (slot >= 0)
            *reinterpret_cast<float4*>(pact + (size_t)slot * p.Wpad + c) = *reinterpret_cast<const float4*>(S.X + row * LDX + c);
    }
    __syncthreads();   // the pushed rows are visible to the whole workgroup; X / flags may be overwritten
    return pending + L;
}

// What is left on the pending stack after the last tile of a workgroup.
__device__ __forceinline__ void gated_head_flush(Smem& S, const MlpParams& p, int pending, EncRegs& enc) {
    if (pending <= 0) return;
    const int tid = threadIdx.x;
    const float* pact = p.pend_act + (size_t)blockIdx.x * TILE_M * p.Wpad;
    const int* pmeta = p.pend_meta + (size_t)blockIdx.x * TILE_M * 2;
    const int w4 = p.Wpad >> 2;
    if (tid == 0) S.uniform_frame = 1;
    if (tid < TILE_M) {
        const bool has = tid < pending;
        S.dest[tid] = has ? pmeta[2 * tid] : -1;
        S.frame[tid] = pmeta[2 * (has ? tid : 0) + 1];
    }
    __syncthreads();
    for (int idx = tid; idx < pending * w4; idx += MLP_THREADS) {
        const int row = idx / w4, c = (idx - row * w4) * 4;
        *reinterpret_cast<float4*>(S.X + row * LDX + c) = *reinterpret_cast<const float4*>(pact + (size_t)row * p.Wpad + c);
    }
    if (tid < TILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;
    __syncthreads();
    head_on_tile(S, p, enc, pending);
}


// The batch-statistics sums of a training launch live in the lanes' registers across the tiles of a workgroup (ColumnStats,
// filled by run_layer<.., STATS>'s epilogue) and reach the global double accumulators once per workgroup and object: one atomic
// per column instead of one per column and tile.  The number of rows in the statistics is counted per tile (phase 1 only).
__device__ __forceinline__ void count_stat_rows(const Smem& S, const MlpParams& p) {
    if (p.phase == 1 && threadIdx.x < 64) {
        const int alive = __popcll(__ballot((S.flags[threadIdx.x] & 3) == 3));
        if (threadIdx.x == 0 && alive) atomicAdd(p.stat_count, alive);
    }
}

__device__ __forceinline__ void flush_column_stats(const ColumnStats& cs, const MlpParams& p, int nblk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane >= 32) return;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int cb = wave + blk * MLP_WAVES;
        if (cb >= nblk) continue;
        const int col = cb * 32 + lane;
        const double s1 = blk ? cs.s1[1] : cs.s1[0], s2 = blk ? cs.s2[1] : cs.s2[0];
        if (s1 != 0.0 || s2 != 0.0) {
            atomicAdd(p.stats + col, s1);
            atomicAdd(p.stats + p.h_out_width + col, s2);
        }
    }
}

#ifdef PR_MLP_TRACE
// Measurement build (-DPR_MLP_TRACE): per workgroup of the LAST grouped evaluation launch [start, end of slot 0..3] in 100 MHz ticks
// and the tiles it took per slot; read back with pr_debug_mlp_trace.
__device__ unsigned long long g_mlp_trace[1024][12];
#define PR_TRACE_MARK(i) if (threadIdx.x == 0) g_mlp_trace[blockIdx.x][i] = wall_clock64()
#else
#define PR_TRACE_MARK(i)
#endif

// TRAIN = false: the evaluation kernel (everything fused, optional sigma gate) - what the benchmark runs; none of the
// training-only code (saved activations, ReLU bit images, phase 1 of the batch-statistics launches) is compiled into it.
// TRAIN = true: phase 1 of the phased launches (train-mode BatchNorm and / or PR_FLAG_SAVE_FOR_BACKWARD).
// GROUP = true: one of several objects evaluated by the same launch (k_mlp_mfma_group) - the workgroups walk through the
// objects in order, EVERY tile is claimed from the object's counter, and a workgroup that finds an object's tiles
// exhausted moves on to the next object at once: small objects do not pay a launch of their own (a launch lasts at
// least one tile time and ends with idle CUs) and the tail of one object overlaps the start of the next.
template <bool TRAIN, bool GROUP, bool SPLIT = false>
__device__ __forceinline__ void mlp_tile_loop(const MlpParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& S = *reinterpret_cast<Smem*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = *p.total;
    PR_PHASE_BEGIN();
    if (GROUP) {
        __syncthreads();   // every wave has left the previous object's last tile (head weights, flags, X are reused)
        if (tid == 0) S.next_tile = atomicAdd(p.tile_counter, 1);
    }
    // stage the small head weights once: [0, Wpad] sigma weights + bias (the 3 rows of the bender head are read from L2: HEAD_BENDER)
    for (int i = tid; i <= p.Wpad; i += MLP_THREADS) S.head_w[i] = p.sigma_w[i];
    __syncthreads();
    int pending = 0;   // rows on this workgroup's pending stack (sigma-gated head), uniform across the workgroup
    EncRegs enc;       // this thread's share of the current network input (see fill_encoding)
    ColumnStats cstats;   // (training) batch-statistics sums of this lane's columns over the workgroup's tiles
    cstats.s1[0] = cstats.s1[1] = cstats.s2[0] = cstats.s2[1] = 0.0;
    // Tile order: the first tile of a workgroup is its block index; evaluation launches claim every further tile from a
    // device counter, so that a workgroup that drew cheaper tiles (sigma-gated head) or a faster CU simply takes more of them.
    // The claim is issued LATE - behind the backbone, in front of the density head (the atomic's latency hides behind that head;
    // what is left of the tile is the feature head, <= 20 %) - and parked in S.next_tile in front of the barriers of the feature
    // head: a claim at the top of a tile hoards - on a launch of ~2 tiles per workgroup (the evaluators' 11 520-ray frame:
    // 933 tiles on 512 workgroups) the first workgroups to reach an object took two of its tiles each, one after the other, while
    // a third of the chip had nothing left to take and left after one tile (1.08 ms for 0.7 ms of balanced work).
#ifdef PR_MLP_STATIC_TILES
    const bool dynamic_tiles = GROUP;     // measurement build: strided tile order
#else
    const bool dynamic_tiles = GROUP || (!TRAIN && p.tile_counter != nullptr);
#endif
    // (GROUP: the first claim was parked in LDS before the barrier that follows the head-weight staging)
    for (int tile = GROUP ? S.next_tile : (int)blockIdx.x; tile * TILE_M < total; tile = S.next_tile) {
        const int tile_base = tile * TILE_M;
        PR_PHASE_T0();
#ifdef PR_MLP_TRACE
        if (GROUP && !TRAIN && tid == 0) g_mlp_trace[blockIdx.x][5 + (p.positions == 4 ? 0 : 1)] += 1;     // tiles of 4-position / other objects
#endif
        int claimed = 0;
#ifdef PR_MLP_EARLY_CLAIM
        if (dynamic_tiles && tid == 0) claimed = atomicAdd(p.tile_counter, 1);      // measurement build: round 3's claim at the top of the tile
#endif
        if (tid == 0) S.uniform_frame = 1;
        if (SPLIT && tid == 0) S.tile_max[0] = S.tile_max[1] = 0;
        // ---- load the sample records of the tile --------------------------------------------
        if (tid < TILE_M) {
            const int idx = tile_base + tid;
            const bool valid = idx < total;
            const int src = valid ? idx : tile_base;   // padding rows replicate the first sample
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
                // skybox input [o / size, d / |d|]   (skybox_adain_style_nerf_model_v3.py:88-96)
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
        if (tid < TILE_M && S.frame[tid] != S.frame[0]) S.uniform_frame = 0;   // visible after the next barrier
#ifdef PR_MLP_EARLY_CLAIM
        if (tid == 0) S.next_tile = GROUP ? claimed : (dynamic_tiles ? (int)gridDim.x + claimed : tile + (int)gridDim.x);   // read at the end of the tile
#endif
        PR_PHASE(0);

        // ---- ray bender -----------------------------------------------------------------------
        // (split-precision training forward: `cur` names the tile_max word of the tile in X, see commit_tile_max in mlp_tile.h; both
        // words were cleared in front of the tile's first barrier)
        int cur = p.has_bender ? 0 : 1;
        if (p.has_bender) {
            fill_bender_input(S, p, enc, false, SPLIT ? &S.tile_max[0] : nullptr);
            __syncthreads();
            if (TRAIN && p.save_bin) write_tile_rows(S, p.save_bin, p.bin_pad, p.bin_pad, tile_base, false);
            for (int l = 0; l < p.b_count; ++l) {
                if (TRAIN) {
                    run_layer<false, true, false, SPLIT>(p.b_layers[l], S, p, tile_base, /*input_kind=*/1, enc, nullptr,
                                           p.save_bbits ? reinterpret_cast<unsigned long long*>(p.save_bbits + (size_t)l * p.save_bbits_stride) +
                                                              (size_t)tile * p.BWpad : nullptr, nullptr, &cur);
                } else {
                    run_layer(p.b_layers[l], S, p, tile_base, /*input_kind=*/1, enc);
                }
                // saved for the backward pass: the post-ReLU activations of every layer (the next layer only READS X until the
                // barrier in front of its epilogue: no barrier needed behind the copy)
                if (TRAIN && p.save_bact) write_tile_rows(S, p.save_bact + (size_t)l * p.save_bact_stride, p.BWpad, p.BWpad, tile_base, false);
            }
            if (SPLIT && tid == 0) S.tile_max[cur ^ 1] = 0;     // the NeRF input's word (its last reader was the last layer's product)
            // output head (no bias), * size, clamp into the box  (positional_ray_bender_model.py:108-140)
            for (int s = tid >> 3; s < TILE_M; s += MLP_THREADS / 8) {
                float out[3];
                row_dots(S, s, as_global(p.b_out), p.BWpad, p.BWpad, 3, out);
                if ((tid & 7) != 0) continue;
                float d[3], bent[3];
                if (TRAIN && p.save_braw && (S.flags[s] & 1))
                    for (int a = 0; a < 3; ++a) p.save_braw[(size_t)(tile_base + s) * 3 + a] = out[a];
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
                    if (p.dispmag)
                        p.dispmag[S.flat[s]] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])),
                                                               __fmul_rn(d[2], d[2])));
                    if (TRAIN && p.save_delta)
                        for (int a = 0; a < 3; ++a) p.save_delta[(size_t)(tile_base + s) * 3 + a] = d[a];
                    if (p.delta_dense)
                        for (int a = 0; a < 3; ++a) p.delta_dense[(size_t)S.flat[s] * 3 + a] = d[a];
                    // second AABB test on the bent position (adain_style_nerf_model.py:173-184)
                    if (!in_box(bent[0], bent[1], bent[2], p.lo, p.hi)) S.flags[s] &= ~2;
                }
            }
            __syncthreads();   // the bent positions and the row flags are complete; X may be overwritten
        }

        PR_PHASE(1);
        // ---- positional encoding of the NeRF input --------------------------------------------
        fill_nerf_input(S, p, enc, false, SPLIT ? &S.tile_max[cur ^ 1] : nullptr);
        cur ^= 1;
        __syncthreads();
        PR_PHASE(2);
        if (TRAIN && p.save_enc) write_tile_rows(S, p.save_enc, p.enc_pad, p.enc_pad, tile_base, false);

        // ---- backbone ---------------------------------------------------------------------------
        for (int l = 0; l < p.n_backbone; ++l) {
            if (TRAIN) {
                run_layer<false, true, false, SPLIT>(p.layers[l], S, p, tile_base, /*input_kind=*/0, enc, nullptr,
                                       (p.save_bits && !(PR_TRAINFWD_ABLATE & 2))
                                           ? reinterpret_cast<unsigned long long*>(p.save_bits + (size_t)l * p.save_bits_stride) + (size_t)tile * p.Wpad
                                           : nullptr, nullptr, &cur);
            } else {
                run_layer(p.layers[l], S, p, tile_base, /*input_kind=*/0, enc);
            }
            PR_PHASE(10);
            if (TRAIN && p.save_act && !(PR_TRAINFWD_ABLATE & 1))
                write_tile_rows(S, p.save_act + (size_t)l * p.save_act_stride, p.Wpad, p.Wpad, tile_base, false);
            PR_PHASE(11);
        }

        PR_PHASE(15);
#ifndef PR_MLP_EARLY_CLAIM
        if (dynamic_tiles && tid == 0) claimed = atomicAdd(p.tile_counter, 1);      // the next tile of this workgroup (see "Tile order")
#endif
        // ---- sigma head -------------------------------------------------------------------------
        if (p.kind == 0) {
            for (int s = tid >> 3; s < TILE_M; s += MLP_THREADS / 8) {
                float sg;
                row_dots(S, s, S.head_w, p.Wpad, p.Wpad, 1, &sg);
                if ((tid & 7) == 0 && (S.flags[s] & 3) == 3) {
                    const float sv = p.in_scene[(size_t)S.frame[s] * p.in_scene_stride] ? sg + S.head_w[p.Wpad] : p.empty_alpha;
                    p.sigma[S.flat[s]] = sv;
                    if (!(sv <= 0.f)) S.flags[s] |= 4;   // a NaN density stays live (relu(NaN) = NaN in the reference)
                }
            }
        } else if (tid < TILE_M) {
            if (S.flags[tid] & 1) {
                const bool present = p.in_scene[(size_t)S.frame[tid] * p.in_scene_stride] != 0;
                p.sigma[S.flat[tid]] = present ? 10.0f : p.empty_alpha;
                if (present || !(p.empty_alpha <= 0.f)) S.flags[tid] |= 4;
            }
        }

#ifndef PR_MLP_EARLY_CLAIM
        if (tid == 0) S.next_tile = GROUP ? claimed : (dynamic_tiles ? (int)gridDim.x + claimed : tile + (int)gridDim.x);   // read at the end of the tile
#endif
        PR_PHASE(7);
        // ---- style-modulated feature head -------------------------------------------------------
        if (!TRAIN && p.gate) {
            __syncthreads();   // the liveness bits are complete
            pending = gated_head(S, p, tile_base, pending, enc);
            PR_PHASE(8);
        } else if (!TRAIN) {
            for (int l = p.n_backbone; l < p.n_layers; ++l) run_layer(p.layers[l], S, p, tile_base, 0, enc);
            PR_PHASE(15);
            write_tile_rows(S, p.feat, p.F, p.F, tile_base, /*zero_dead=*/true);
            __syncthreads();   // the next tile's prologue overwrites flags / X
            PR_PHASE(8);
        } else {
            // train mode, phase 1: stop after the first head matmul; the raw activations go to HBM and
            // their per-channel sums feed the batch statistics
            Layer raw = p.layers[p.n_backbone];
            raw.epi = EPI_FEATURES;   // plain store into X
            run_layer<false, false, true, SPLIT>(raw, S, p, tile_base, 0, enc, nullptr, nullptr, (PR_TRAINFWD_ABLATE & 4) ? nullptr : &cstats, &cur);
            PR_PHASE(12);
            if (tid < TILE_M && (S.flags[tid] & 1)) p.row_flags[tile_base + tid] = S.flags[tid];
            write_tile_rows(S, p.h_out, p.h_out_width, p.h_out_width, tile_base, /*zero_dead=*/false);
            count_stat_rows(S, p);
            PR_PHASE(13);
            __syncthreads();
            PR_PHASE(14);
        }
    }
    if (!TRAIN && p.gate) gated_head_flush(S, p, pending, enc);
    if (TRAIN) flush_column_stats(cstats, p, p.layers[p.n_backbone].nblk);
    PR_PHASE_FLUSH();
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_mfma(MlpParams p) { mlp_tile_loop<false, false>(p); }
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_mfma_train(MlpParams p) { mlp_tile_loop<true, false>(p); }
// Several objects, one launch.  One copy of the tile loop per job slot: the parameters of a slot are kernel arguments at
// constant offsets, exactly as in k_mlp_mfma (a table indexed at run time would turn every parameter into a memory load
// and every pointer into a flat pointer - measured: 3 % slower).
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_mfma_group(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                                   int count) {
    PR_TRACE_MARK(0);
    mlp_tile_loop<false, true>(j0);
    PR_TRACE_MARK(1);
    if (count > 1) mlp_tile_loop<false, true>(j1);
    PR_TRACE_MARK(2);
    if (count > 2) mlp_tile_loop<false, true>(j2);
    PR_TRACE_MARK(3);
    if (count > 3) mlp_tile_loop<false, true>(j3);
    PR_TRACE_MARK(4);
}

#ifdef PR_MLP_TRACE
extern "C" int pr_debug_mlp_trace(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_trace), sizeof(unsigned long long) * 1024 * 12) == hipSuccess ? 0 : 1;
}
#endif

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_mfma_train_group(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                                         int count) {
    pr_stagger(4);
    mlp_tile_loop<true, true>(j0);
    if (count > 1) mlp_tile_loop<true, true>(j1);
    if (count > 2) mlp_tile_loop<true, true>(j2);
    if (count > 3) mlp_tile_loop<true, true>(j3);
}

// phase 1 of a training call in split precision (PR_FLAG_SPLIT_BACKWARD): the same tile loop on fp16-pair segments (tile_products_f16x3_lean)
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_mfma_train_group_split(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                                              int count) {
    pr_stagger(4);
    mlp_tile_loop<true, true, true>(j0);
    if (count > 1) mlp_tile_loop<true, true, true>(j1);
    if (count > 2) mlp_tile_loop<true, true, true>(j2);
    if (count > 3) mlp_tile_loop<true, true, true>(j3);
}

// Train-mode phases 2 and 3: re-load the raw head activations of the previous phase, apply the AdaIN
// affine built from the BATCH statistics + ReLU, run the next head matmul.
// GROUP: `first` is this workgroup's first tile of the object and comes back advanced by the object's tile count (mod grid): the
// objects of a launch are dealt to the workgroups as ONE round-robin sequence - strided from the block index per object, the
// workgroups 0 .. (tiles mod grid) of EVERY object took an extra tile (a few objects of a few hundred tiles each on 512 workgroups).
#ifndef PR_HEADF_ABLATE
#define PR_HEADF_ABLATE 0     // timing builds only (k_mlp_head*): 1 = no activation reads, 2 = no row write-out, 4 = no statistics flush
#endif
template <bool GROUP>
__device__ __forceinline__ void mlp_head_loop(const MlpParams& p, int& first) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& S = *reinterpret_cast<Smem*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = *p.total;
    if (GROUP) __syncthreads();   // every wave has left the previous object's last tile
    const Layer& prev = p.layers[p.n_backbone + p.phase - 2];   // the layer whose output is h_in
    const Layer& cur = p.layers[p.n_backbone + p.phase - 1];
    ColumnStats cstats;        // phase 2: batch-statistics sums of this lane's columns over the workgroup's tiles
    cstats.s1[0] = cstats.s1[1] = cstats.s2[0] = cstats.s2[1] = 0.0;
    const int first_tile = first;
    if (GROUP) {
        const int tiles = (total + TILE_M - 1) / TILE_M, grid = (int)gridDim.x;
        first = (first + grid - tiles % grid) % grid;       // the next object continues where this one's round robin stops
    }
    for (int tile = first_tile; tile * TILE_M < total; tile += gridDim.x) {
        const int tile_base = tile * TILE_M;
        EncRegs enc;   // this thread's share of the current network input (see fill_encoding)
        if (tid < TILE_M) {
            const int idx = tile_base + tid;
            const bool valid = idx < total;
            const int src = valid ? idx : tile_base;
            S.flat[tid] = p.rec_flat[src];
            S.frame[tid] = S.flat[tid] / p.samples_per_frame;
            S.flags[tid] = valid ? p.row_flags[src] : 0;
        }
        __syncthreads();
        const int wq = p.h_in_width >> 2;
        {
            // batches of loads first (raw activations + the AdaIN table rows of their frames), then the arithmetic and the LDS
            // stores: as one loop every iteration waited for its own loads (8 - 16 round trips to memory per tile)
            constexpr int BATCH = 4;
            const int count = TILE_M * wq;
            for (int base = tid; base < count; base += MLP_THREADS * BATCH) {
                float4 h[BATCH], g[BATCH], b[BATCH];
#pragma unroll
                for (int q = 0; q < BATCH; ++q) {
                    const int idx = base + q * MLP_THREADS;
                    const int row = idx < count ? idx / wq : 0, c = idx < count ? (idx - row * wq) * 4 : 0;
                    const int src = (tile_base + row < total) ? tile_base + row : tile_base;
#if PR_HEADF_ABLATE & 1
                    h[q] = make_float4(0.5f, 0.25f, -0.5f, 1.f);    // measurement build: no activation reads (results are wrong)
#else
                    h[q] = *reinterpret_cast<const float4*>(p.h_in + (size_t)src * p.h_in_width + c);
#endif
                    const float* tab = p.adain + (size_t)S.frame[row] * p.adain_stride + prev.adain_off;
                    g[q] = *reinterpret_cast<const float4*>(tab + c);
                    b[q] = *reinterpret_cast<const float4*>(tab + prev.nblk * 32 + c);
                }
#pragma unroll
                for (int q = 0; q < BATCH; ++q) {
                    const int idx = base + q * MLP_THREADS;
                    if (idx >= count) continue;
                    const int row = idx / wq, c = (idx - row * wq) * 4;
                    float4 y;
                    y.x = fmaf(h[q].x, g[q].x, b[q].x); y.y = fmaf(h[q].y, g[q].y, b[q].y);
                    y.z = fmaf(h[q].z, g[q].z, b[q].z); y.w = fmaf(h[q].w, g[q].w, b[q].w);
                    y.x = y.x > 0.f ? y.x : 0.f; y.y = y.y > 0.f ? y.y : 0.f; y.z = y.z > 0.f ? y.z : 0.f; y.w = y.w > 0.f ? y.w : 0.f;
                    *reinterpret_cast<float4*>(S.X + row * LDX + c) = y;
                }
            }
        }
        __syncthreads();
        Layer raw = cur;
        raw.epi = EPI_FEATURES;   // plain store into X
        run_layer<false, false, true>(raw, S, p, tile_base, 0, enc, nullptr, nullptr, p.phase == 2 ? &cstats : nullptr);
#if PR_HEADF_ABLATE & 2
        __syncthreads();      // measurement build: no row write-out
#else
        if (p.phase == 2) {
            write_tile_rows(S, p.h_out, p.h_out_width, p.h_out_width, tile_base, /*zero_dead=*/false);
            __syncthreads();
        } else {
            write_tile_rows(S, p.feat, p.F, p.F, tile_base, /*zero_dead=*/true);
            __syncthreads();
        }
#endif
    }
#if !(PR_HEADF_ABLATE & 4)
    if (p.phase == 2) flush_column_stats(cstats, p, cur.nblk);
#endif
}

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_head(MlpParams p) {
    int first = (int)blockIdx.x;
    mlp_head_loop<false>(p, first);
}
// the same phase of several objects in one launch: a workgroup takes its strided share of every object's tiles in turn
__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_mlp_head_group(MlpParams j0, MlpParams j1, MlpParams j2, MlpParams j3,
                                                                                   int count) {
    pr_stagger(1);
    int first = (int)blockIdx.x;
    mlp_head_loop<true>(j0, first);
    if (count > 1) mlp_head_loop<true>(j1, first);
    if (count > 2) mlp_head_loop<true>(j2, first);
    if (count > 3) mlp_head_loop<true>(j3, first);
}

__global__ __launch_bounds__(256) void k_bn_finalize(BnFinalizeParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.width) return;
    if (p.frozen) {   // eval mode: normalise with the running statistics (torch batch_norm with training=False)
        p.batch_mean[c] = p.running_mean[c];
        p.batch_var[c] = p.running_var[c];
        return;
    }
    const double n = (double)*p.count;
    const double mean = p.stats[c] / n;
    double var = p.stats[p.width_pad + c] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unbiased = var * n / (n - 1.0);   // NaN / inf for n <= 1, where the reference raises
    p.batch_mean[c] = (float)mean;
    p.batch_var[c] = (float)var;
    // nn.BatchNorm1d counts every training batch; the running statistics move for batches of two or more rows (an empty batch
    // leaves them alone, a single row makes torch raise - the host raises after the call)
    if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
    if (n > 1.0) {
        p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
        p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
    }
}

// k_bn_finalize and the fold of THAT layer's statistics into the AdaIN table, for every object of a training call in one
// launch (blockIdx.y = object): a thread owns one channel - batch statistics, running-statistics update, then
// [scale | bias] = affine(style) of every frame folded with them (the arithmetic of k_bn_finalize / k_adain_fold).
__global__ __launch_bounds__(256) void k_bn_fold_group(BnFoldJobs jobs) {
    const BnFoldJob& p = jobs.job[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && p.normalised_out) *p.normalised_out = *p.count;
    if (c >= p.width_pad) return;
    float mean_f = 0.f, var_f = 0.f;
    if (c < p.width) {
        if (p.frozen) {
            mean_f = p.running_mean[c];
            var_f = p.running_var[c];
        } else {
            const double n = (double)*p.count;
            const double mean = p.stats[c] / n;
            double var = p.stats[p.width_pad + c] / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const double unbiased = var * n / (n - 1.0);
            mean_f = (float)mean;
            var_f = (float)var;
            // nn.BatchNorm1d counts every training batch; the running statistics move for batches of two or more rows (an empty
            // batch leaves them alone, a single row makes torch raise - the host raises after the call)
            if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
            if (n > 1.0) {
                p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
                p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
            }
        }
        p.batch_mean[c] = mean_f;
        p.batch_var[c] = var_f;
    }
    const float* ws = p.affine.weight + (size_t)c * p.S;
    const float* wb = p.affine.weight + (size_t)(p.width + c) * p.S;
    const float inv = 1.0f / sqrtf(var_f + p.eps);
    const bool live = c < p.width;
    const float bias_s = live ? p.affine.bias[c] : 0.f, bias_b = live ? p.affine.bias[p.width + c] : 0.f;
    for (int n = 0; n < p.frames; ++n) {
        float g = 0.f, b = 0.f;
        if (live) {
            const float* style = p.style + (size_t)n * p.style_stride;
            float scale = bias_s, bias = bias_b;
            // eight terms' loads at a time, then their multiply-adds in the same (ascending) order: as one loop of dependent
            // load - fma pairs this kernel - 4 workgroups, on the critical path between two phases - took 19 - 26 us
            int q = 0;
            for (; q + 8 <= p.S; q += 8) {
                float w8[8], b8[8], s8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w8[k] = ws[q + k];
                    b8[k] = wb[q + k];
                    s8[k] = style[q + k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    scale = fmaf(w8[k], s8[k], scale);
                    bias = fmaf(b8[k], s8[k], bias);
                }
            }
            for (; q < p.S; ++q) {
                scale = fmaf(ws[q], style[q], scale);
                bias = fmaf(wb[q], style[q], bias);
            }
            g = scale * inv;
            b = bias - mean_f * g;
        }
        float* row = p.table + (size_t)n * p.row_floats;
        row[p.g_off + c] = g;
        row[p.b_off + c] = b;
    }
}

int launch_bn_fold_group(const BnFoldJobs& jobs, int count, hipStream_t s) {
    if (count <= 0) return PR_OK;
    hipLaunchKernelGGL(k_bn_fold_group, dim3((MAX_WIDTH + 255) / 256, count), dim3(256), 0, s, jobs);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

int launch_bn_finalize(const BnFinalizeParams& p, hipStream_t s) {
    PR_REQUIRE(p.running_mean && p.running_var && p.batch_mean && p.batch_var && p.stats && p.count, "bn finalize: NULL pointer");
    hipLaunchKernelGGL(k_bn_finalize, dim3((p.width + 255) / 256), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Scalar debugging kernel: one thread per sample, raw (reference-layout) weights.  Used by tests
// to separate packing / MFMA-layout bugs from pipeline bugs.  Never used by the product path
// unless PR_FLAG_NAIVE_MLP is set explicitly.
// ---------------------------------------------------------------------------------------------
__device__ void naive_linear(const pr_linear_t& l, const float* in, float* out, bool relu) {
    for (int n = 0; n < l.out_features; ++n) {
        float acc = l.bias ? l.bias[n] : 0.f;
        const float* w = l.weight + (size_t)n * l.in_features;
        for (int k = 0; k < l.in_features; ++k) acc = fmaf(in[k], w[k], acc);
        out[n] = relu ? (acc > 0.f ? acc : 0.f) : acc;
    }
}

__global__ __launch_bounds__(64) void k_mlp_naive(MlpParams p, pr_object_model_t m) {
    const int total = *p.total;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= total) return;
    float in[MAX_WIDTH + MAX_ENC], h[MAX_WIDTH], enc[MAX_ENC];
    const int flat = p.rec_flat[idx];
    const int frame = flat / p.samples_per_frame;
    float v[6];
    bool alive = true;
    if (p.kind == 0) {
        float x[3] = {p.rec_pos[(size_t)idx * 3], p.rec_pos[(size_t)idx * 3 + 1], p.rec_pos[(size_t)idx * 3 + 2]};
        if (p.has_bender) {
            float xn[3];
            for (int a = 0; a < 3; ++a) xn[a] = __fdiv_rn(x[a], p.size[a]);
            const int bin = p.benc + p.D;
            for (int j = 0; j < p.benc; ++j) enc[j] = pe_element(xn, 3, p.benc, j, p.b_weights);
            for (int j = 0; j < p.D; ++j) enc[p.benc + j] = p.deformation[(size_t)frame * p.deformation_stride + j];
            for (int j = 0; j < bin; ++j) in[j] = enc[j];
            for (int l = 0; l < p.b_count; ++l) {
                if (l == m.bender_skip) {
                    for (int j = 0; j < p.BW; ++j) in[j] = h[j];
                    for (int j = 0; j < bin; ++j) in[p.BW + j] = enc[j];
                } else if (l > 0) {
                    for (int j = 0; j < p.BW; ++j) in[j] = h[j];
                }
                naive_linear(m.bender[l], in, h, true);
            }
            float out[3];
            naive_linear(m.bender_out, h, out, false);
            float d[3];
            for (int a = 0; a < 3; ++a) {
                float dl = __fmul_rn(out[a], p.size[a]);
                dl = nan_max(dl, __fsub_rn(p.lo[a], x[a]));
                dl = nan_min(dl, __fsub_rn(p.hi[a], x[a]));
                if (p.canonical) dl = __fmul_rn(dl, 0.0f);
                d[a] = dl;
                x[a] = __fadd_rn(x[a], dl);
            }
            if (p.dispmag)
                p.dispmag[flat] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])),
                                                  __fmul_rn(d[2], d[2])));
            alive = in_box(x[0], x[1], x[2], p.lo, p.hi);
        }
        for (int a = 0; a < 3; ++a) v[a] = __fdiv_rn(x[a], p.size[a]);
    } else {
        const int ray = (flat - frame * p.samples_per_frame) / p.positions;
        const ObjRay rr = object_ray(p.w2o + (size_t)frame * p.w2o_stride, p.ray_origins + (size_t)frame * 3,
                                     p.ray_directions + ((size_t)frame * p.rays + ray) * 3);
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rr.d[0], rr.d[0]), __fmul_rn(rr.d[1], rr.d[1])),
                                          __fmul_rn(rr.d[2], rr.d[2])));
        for (int a = 0; a < 3; ++a) {
            v[a] = __fdiv_rn(rr.o[a], p.size[a]);
            v[3 + a] = __fdiv_rn(rr.d[a], nrm);
        }
    }
    for (int j = 0; j < p.enc; ++j) enc[j] = pe_element(v, p.din, p.enc, j, nullptr);
    for (int j = 0; j < p.enc; ++j) in[j] = enc[j];
    for (int l = 0; l < p.n_backbone; ++l) {
        if (l == m.skip_layer_idx) {
            for (int j = 0; j < p.W; ++j) in[j] = h[j];
            for (int j = 0; j < p.enc; ++j) in[p.W + j] = enc[j];
        } else if (l > 0) {
            for (int j = 0; j < p.W; ++j) in[j] = h[j];
        }
        naive_linear(m.backbone[l], in, h, true);
    }
    if (p.kind == 0) {
        float sg;
        naive_linear(m.alpha_head, h, &sg, false);
        if (alive) p.sigma[flat] = p.in_scene[(size_t)frame * p.in_scene_stride] ? sg : p.empty_alpha;
    } else {
        p.sigma[flat] = p.in_scene[(size_t)frame * p.in_scene_stride] ? 10.0f : p.empty_alpha;
    }
    const float* tab = p.adain + (size_t)frame * p.adain_stride;
    const int W2 = m.layers_width / 2;
    const int W2pad = round_up(W2, 32);
    naive_linear(m.head0, h, in, false);
    for (int j = 0; j < p.W; ++j) {
        const float y = fmaf(in[j], tab[j], tab[p.Wpad + j]);
        h[j] = y > 0.f ? y : 0.f;
    }
    naive_linear(m.head3, h, in, false);
    for (int j = 0; j < W2; ++j) {
        const float y = fmaf(in[j], tab[2 * p.Wpad + j], tab[2 * p.Wpad + W2pad + j]);
        h[j] = y > 0.f ? y : 0.f;
    }
    naive_linear(m.head6, h, in, false);
    for (int j = 0; j < p.F; ++j) p.feat[(size_t)idx * p.F + j] = alive ? in[j] : 0.f;
}


int launch_mlp(const MlpParams& p, int max_rows, bool naive, const pr_object_model_t* raw, hipStream_t s) {
    if (max_rows <= 0) return PR_OK;
    const int max_tiles = naive ? (max_rows + 63) / 64 : (max_rows + TILE_M - 1) / TILE_M;
    if (naive) {
        hipLaunchKernelGGL(k_mlp_naive, dim3(max_tiles), dim3(64), 0, s, p, *raw);
        PR_LAUNCH_CHECK();
        return PR_OK;
    }
    const MlpParams& pd = p;
    int cu_count = 0;
    const void* kernel = pd.phase >= 2 ? reinterpret_cast<const void*>(k_mlp_head)
                         : (pd.phase == 1 ? reinterpret_cast<const void*>(k_mlp_mfma_train) : reinterpret_cast<const void*>(k_mlp_mfma));
    PR_TRY(prepare_kernel(kernel, (int)sizeof(Smem), &cu_count));
    int resident = cu_count * MLP_BLOCKS_PER_CU;
    if (resident > MAX_RESIDENT_TILES) resident = MAX_RESIDENT_TILES;   // the pending stacks of the gated head are sized for this
    const int grid = max_tiles < resident ? max_tiles : resident;
    PR_REQUIRE(!pd.gate || (pd.pend_act && pd.pend_meta && pd.phase == 0), "gated head: pending buffers missing");
    ProfileScope scope(0, s);
    if (pd.phase >= 2) {
        hipLaunchKernelGGL(k_mlp_head, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, pd);
    } else if (pd.phase == 1) {
        hipLaunchKernelGGL(k_mlp_mfma_train, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, pd);
    } else {
        hipLaunchKernelGGL(k_mlp_mfma, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, pd);
    }
    PR_LAUNCH_CHECK();
#ifdef PR_MLP_TIMING
    {
        unsigned long long now[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(now, HIP_SYMBOL(g_mlp_phase), sizeof(now));
        fprintf(stderr, "[mlp phases, cumulative Mticks of thread 0 summed over workgroups]");
        for (int i = 0; i < 9; ++i) fprintf(stderr, " p%d=%.1f", i, (double)now[i] * 1e-6);
        fprintf(stderr, "\n");
    }
#endif
    return PR_OK;
}

int launch_mlp_group(const MlpParams* host_jobs, const int* max_rows, int count, hipStream_t s) {
    PR_REQUIRE(count >= 1, "grouped MLP launch: no jobs");
    static thread_local MlpGroupParams g;     // 18 KB: not on the stack
    for (int begin = 0; begin < count; begin += MLP_GROUP_MAX) {
        const int n = count - begin < MLP_GROUP_MAX ? count - begin : MLP_GROUP_MAX;
        long max_tiles = 0;
        for (int j = 0; j < n; ++j) {
            const MlpParams& q = host_jobs[begin + j];
            PR_REQUIRE(q.phase >= 0 && q.phase <= 3 && q.phase == host_jobs[0].phase && (q.phase >= 2 || q.tile_counter),
                       "grouped MLP launch: one phase for all jobs, a tile counter for the fused / first phase");
            PR_REQUIRE(!q.gate || (q.pend_act && q.pend_meta), "gated head: pending buffers missing");
            max_tiles += ((long)max_rows[begin + j] + TILE_M - 1) / TILE_M;
            g.jobs[j] = q;
        }
        if (max_tiles <= 0) continue;
        int cu_count = 0;
        const int phase = host_jobs[0].phase;
        const bool split3 = phase == 1 && host_jobs[begin].split3 != 0;
        for (int j = 0; j < n; ++j)
            PR_REQUIRE((host_jobs[begin + j].split3 != 0) == (host_jobs[begin].split3 != 0) && (phase == 1 || !host_jobs[begin + j].split3),
                       "grouped MLP launch: bf16-triple segments belong to phase 1, for every job of the launch or none");
        const void* kernel = phase >= 2 ? reinterpret_cast<const void*>(k_mlp_head_group)
                                        : (phase == 1 ? (split3 ? reinterpret_cast<const void*>(k_mlp_mfma_train_group_split)
                                                                : reinterpret_cast<const void*>(k_mlp_mfma_train_group))
                                                      : reinterpret_cast<const void*>(k_mlp_mfma_group));
        PR_TRY(prepare_kernel(kernel, (int)sizeof(Smem), &cu_count));
        int resident = cu_count * MLP_BLOCKS_PER_CU;
        if (resident > MAX_RESIDENT_TILES) resident = MAX_RESIDENT_TILES;
        const int grid = max_tiles < resident ? (int)max_tiles : resident;
        g.count = n;
        ProfileScope scope(0, s);
        if (phase >= 2)
            hipLaunchKernelGGL(k_mlp_head_group, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, g.jobs[0], g.jobs[1], g.jobs[2], g.jobs[3], n);
        else if (phase == 1 && split3)
            hipLaunchKernelGGL(k_mlp_mfma_train_group_split, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, g.jobs[0], g.jobs[1], g.jobs[2], g.jobs[3], n);
        else if (phase == 1)
            hipLaunchKernelGGL(k_mlp_mfma_train_group, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, g.jobs[0], g.jobs[1], g.jobs[2], g.jobs[3], n);
        else
            hipLaunchKernelGGL(k_mlp_mfma_group, dim3(grid), dim3(MLP_THREADS), sizeof(Smem), s, g.jobs[0], g.jobs[1], g.jobs[2], g.jobs[3], n);
        PR_LAUNCH_CHECK();
#ifdef PR_MLP_TIMING
        {
            static unsigned long long before[16];
            unsigned long long now[16];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(now, HIP_SYMBOL(g_mlp_phase), sizeof(now));
            fprintf(stderr, "[mlp group phases, phase %d%s, grid %d, Mticks of thread 0 summed over workgroups]", phase, split3 ? " split" : "", grid);
            for (int i = 0; i < 16; ++i) {
                if (now[i] != before[i]) fprintf(stderr, " p%d=%.2f", i, (double)(now[i] - before[i]) * 1e-6);
                before[i] = now[i];
            }
            fprintf(stderr, "\n");
        }
#endif
    }
    return PR_OK;
}

// Fills the layer tables of MlpParams from the packed buffer.
int build_mlp_layers(const pr_object_model_t& m, const ModelDims& d, const PackedLayout& l, const float* base,
                     MlpParams* p, bool split3) {
    // split3 (phase 1 of a training call with PR_FLAG_SPLIT_BACKWARD): ray bender, backbone and head layer 0 read their
    // fp16-pair packings (add_seg3); the head layers of the later phases and every small vector stay fp32
    p->split3 = split3 ? 1 : 0;
    p->kind = m.kind;
    p->has_bender = m.has_bender;
    p->b_count = 0;
    if (m.has_bender) {
        p->b_octaves = m.bender_octaves;
        p->benc = d.benc;
        p->bin_pad = d.bin_pad;
        p->D = m.deformation_features;
        for (int k = 0; k < PR_MAX_OCTAVES; ++k) p->b_weights[k] = m.bender_octave_weights[k];
        p->b_count = m.bender_count;
        p->BW = d.BW;
        p->BWpad = d.BWpad;
        p->b_out = base + l.b_out_off;
        for (int j = 0; j < m.bender_count; ++j) {
            Layer& L = p->b_layers[j];
            memset(&L, 0, sizeof(L));
            L.nblk = d.BWpad / 32;
            L.n_real = d.BW;
            L.bias = base + l.b_bias_off[j];
            L.epi = EPI_RELU;
            L.nseg = 1;
            if (j == 0) {
                L.seg[0] = Seg{base + (split3 ? l.b_seg3[j][0] : l.b_seg_off[j][0]), d.bin_pad / 8, 1};
            } else {
                L.seg[0] = Seg{base + (split3 ? l.b_seg3[j][0] : l.b_seg_off[j][0]), d.BWpad / 8, 0};
                if (j == m.bender_skip) {
                    L.seg[1] = Seg{base + (split3 ? l.b_seg3[j][1] : l.b_seg_off[j][1]), d.bin_pad / 8, 1};
                    L.nseg = 2;
                }
            }
        }
    }
    p->octaves = m.octaves;
    p->din = d.din;
    p->enc = d.enc;
    p->enc_pad = d.enc_pad;
    p->W = d.W;
    p->Wpad = d.Wpad;
    p->F = d.F;
    p->n_backbone = m.backbone_count;
    p->n_layers = m.backbone_count + 3;
    for (int i = 0; i < m.backbone_count; ++i) {
        Layer& L = p->layers[i];
        memset(&L, 0, sizeof(L));
        L.nblk = d.Wpad / 32;
        L.n_real = d.W;
        L.bias = base + l.n_bias_off[i];
        L.epi = EPI_RELU;
        L.nseg = 1;
        if (i == 0) {
            L.seg[0] = Seg{base + (split3 ? l.n_seg3[i][0] : l.n_seg_off[i][0]), d.enc_pad / 8, 1};
        } else {
            L.seg[0] = Seg{base + (split3 ? l.n_seg3[i][0] : l.n_seg_off[i][0]), d.Wpad / 8, 0};
            if (i == m.skip_layer_idx) {
                L.seg[1] = Seg{base + (split3 ? l.n_seg3[i][1] : l.n_seg_off[i][1]), d.enc_pad / 8, 1};
                L.nseg = 2;
            }
        }
    }
    p->sigma_w = base + l.sigma_off;
    Layer& h0 = p->layers[m.backbone_count];
    memset(&h0, 0, sizeof(h0));
    h0.seg[0] = Seg{base + (split3 ? l.h0_3 : l.h0_off), d.Wpad / 8, 0};
    h0.nseg = 1;
    h0.nblk = d.Wpad / 32;
    h0.n_real = d.W;
    h0.epi = EPI_ADAIN_RELU;
    h0.adain_off = 0;
    Layer& h3 = p->layers[m.backbone_count + 1];
    memset(&h3, 0, sizeof(h3));
    h3.seg[0] = Seg{base + l.h3_off, d.Wpad / 8, 0};
    h3.nseg = 1;
    h3.nblk = d.W2pad / 32;
    h3.n_real = d.W2;
    h3.epi = EPI_ADAIN_RELU;
    h3.adain_off = 2 * d.Wpad;
    Layer& h6 = p->layers[m.backbone_count + 2];
    memset(&h6, 0, sizeof(h6));
    h6.seg[0] = Seg{base + l.h6_off, d.W2pad / 8, 0};
    h6.nseg = 1;
    h6.nblk = d.Fpad / 32;
    h6.n_real = d.F;
    h6.bias = base + l.h6_bias_off;
    h6.epi = EPI_FEATURES;
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward chain of a ReLU MLP with one skip concatenation (NeRF backbone, ray bender): the input gradients of ALL its
// layers in one launch.  A persistent workgroup owns a 64-sample tile: G = d loss / d pre-activation of the last layer is
// loaded into X; for l = count - 1 .. 1:  dX = G . W_l[:, :width]  on the matrix cores (W^T as fragment-ordered segments,
// packed per backward call), masked with the saved post-ReLU activation of layer l - 1 -> the new G, which stays in LDS
// for the next layer and is written once to `gstack[l - 1]` for the weight-gradient products (one grouped launch over
// all layers afterwards).  The input parts of the skip layer and of layer 0 go straight to `g_in` (store, then
// accumulate).  Compared with one dX GEMM per layer this removes a read of dY and a separate mask pass per layer, and
// - what matters for calls with a few thousand samples per object - 2 x count launches whose tiles fill a fraction of
// the chip for a fraction of a round each.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(MLP_THREADS, MLP_BLOCKS_PER_CU) void k_chain_bwd(ChainBwdParams c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem& S = *reinterpret_cast<Smem*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = *c.total;
    const MlpParams& unused = *reinterpret_cast<const MlpParams*>(smem_raw);   // never dereferenced by run_layer<true>
    EncRegs enc;
    const int w4 = c.Wpad >> 2;
    for (int tile = blockIdx.x; tile * TILE_M < total; tile += gridDim.x) {
        const int tile_base = tile * TILE_M;
        const int rows_valid = (total - tile_base < TILE_M) ? total - tile_base : TILE_M;
        if (tid < TILE_M) S.flags[tid] = tid < rows_valid ? 1 : 0;
        // G of the last layer
        for (int idx = tid; idx < TILE_M * w4; idx += MLP_THREADS) {
            const int row = idx / w4, c4 = (idx - row * w4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            // (columns beyond the real width are padding that the producers of g_last do not write)
            if (row < rows_valid && c4 < c.W) v = *reinterpret_cast<const float4*>(c.g_last + (size_t)(tile_base + row) * c.Wpad + c4);
            *reinterpret_cast<float4*>(S.X + row * LDX + c4) = v;
        }
        __syncthreads();
        BwdEpilogue e;
        e.rows_valid = rows_valid;
        e.mask_bits = reinterpret_cast<const unsigned char*>(S.pos);
        e.mask_bytes_per_row = c.Wpad >> 3;
        bool g_in_written = false;
        for (int l = c.count - 1; l >= 1; --l) {
            // the ReLU mask of this layer's input (layer l - 1's output) -> bits, visible after the barrier inside run_layer
#if !(defined(PR_CHAIN_ABLATE) && (PR_CHAIN_ABLATE & 2))     // measurement builds: 2 = no mask bits, 1 = no gradient write-out
            if (c.bits) {
                // the forward pass left the bit image of this tile and layer behind: 64 x (width / 8) contiguous bytes
                const int bytes = TILE_M * (c.Wpad >> 3);
                const unsigned char* src = c.bits + (size_t)(l - 1) * c.bits_stride + (size_t)tile_base * (c.Wpad >> 3);
                const int live_bytes = rows_valid * (c.Wpad >> 3);
                for (int i = tid * 8; i < bytes; i += MLP_THREADS * 8) {
                    unsigned long long v = 0ull;
                    if (i + 8 <= live_bytes) {
                        v = *reinterpret_cast<const unsigned long long*>(src + i);
                    } else {
                        // the word straddles the end of the tile's real rows (rows narrower than 8 bytes): byte by byte
                        for (int b = 0; b < 8 && i + b < live_bytes; ++b) v |= (unsigned long long)src[i + b] << (8 * b);
                    }
                    *reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(S.pos) + i) = v;
                }
            } else {
                build_relu_mask_bits(reinterpret_cast<unsigned char*>(S.pos), c.acts + (size_t)(l - 1) * c.act_stride, c.Wpad, c.Wpad,
                                     tile_base, rows_valid);
            }
#endif
            if (l == c.skip) {
                e.gout = c.g_in; e.ldg = c.ld_in; e.accumulate = 0; e.n_real = c.in_real;
                run_layer<true>(c.in0_skip, S, unused, tile_base, 0, enc, &e);
                g_in_written = true;
            }
            e.gout = nullptr; e.n_real = c.W;
            run_layer<true>(c.act_layers[l], S, unused, tile_base, 0, enc, &e);
            // the pre-activation gradient of layer l - 1, for its weight-gradient product
#if !(defined(PR_CHAIN_ABLATE) && (PR_CHAIN_ABLATE & 1))
            write_tile_rows(S, c.gstack + (size_t)(l - 1) * c.g_stride, c.Wpad, c.Wpad, tile_base, false);
#endif
        }
        e.gout = c.g_in; e.ldg = c.ld_in; e.accumulate = g_in_written ? 1 : 0; e.n_real = c.in_real;
        run_layer<true>(c.in0_first, S, unused, tile_base, 0, enc, &e);
        __syncthreads();   // the next tile's loads overwrite X and the flags
    }
}

static size_t chain_packed_floats(int count, int Wpad, int in_pad) {
    return (size_t)(count - 1) * seg_floats(Wpad / 32, Wpad) + 2 * (size_t)seg_floats(in_pad / 32, Wpad);
}

size_t chain_bwd_packed_bytes(int count, int width, int in_features) {
    return sizeof(float) * chain_packed_floats(count, round_up(width, 32), round_up(in_features, 32));
}

// Packs W_l^T of every layer of the chain (fragment order of run_layer) into `packed` and fills the layer tables of `c`.
int prepare_chain_bwd(const pr_linear_t* layers, int count, int skip, int width, int in_features, float* packed,
                      ChainBwdParams* c, hipStream_t s) {
    const int Wpad = round_up(width, 32), in_pad = round_up(in_features, 32);
    PR_REQUIRE(count >= 2 && count <= PR_MAX_LAYERS && Wpad <= MAX_WIDTH && in_pad <= MAX_ENC, "backward chain: unsupported shape");
    PackJobs jobs;
    jobs.n = 0;
    jobs.seg_kind = 0;
    float* at = packed;
    auto seg = [&](const pr_linear_t& lin, int col_off, int n_real, int npad, Layer* L) -> int {
        // out[m][n] = sum_k G[m][k] W[k][col_off + n]:  K = the layer's outputs (width), N = its inputs
        PR_TRY(add_seg(&jobs, lin, col_off, width, Wpad, npad, at));
        PackJob& j = jobs.job[jobs.n - 1];
        j.n_real = n_real;
        j.transposed = 1;
        memset(L, 0, sizeof(*L));
        L->seg[0] = Seg{at, Wpad / 8, 0};
        L->nseg = 1;
        L->nblk = npad / 32;
        L->n_real = n_real;
        at += seg_floats(npad / 32, Wpad);
        return PR_OK;
    };
    for (int l = 1; l < count; ++l) {
        PR_REQUIRE(layers[l].out_features == width && layers[l].in_features == (l == skip ? width + in_features : width),
                   "backward chain: layer %d has shape (%d, %d)", l, layers[l].out_features, layers[l].in_features);
        PR_TRY(seg(layers[l], 0, width, Wpad, &c->act_layers[l]));
        c->act_layers[l].epi = EPI_BWD_MASK;
    }
    PR_REQUIRE(skip >= 1 && skip < count, "backward chain: skip layer %d", skip);
    PR_TRY(seg(layers[skip], width, in_features, in_pad, &c->in0_skip));
    c->in0_skip.epi = EPI_BWD_GLOBAL;
    PR_REQUIRE(layers[0].out_features == width && layers[0].in_features == in_features, "backward chain: first layer shape");
    PR_TRY(seg(layers[0], 0, in_features, in_pad, &c->in0_first));
    c->in0_first.epi = EPI_BWD_GLOBAL;
    c->count = count; c->skip = skip; c->W = width; c->Wpad = Wpad; c->in_pad = in_pad; c->in_real = in_features;
    hipLaunchKernelGGL(k_pack, dim3(64, jobs.n), dim3(256), 0, s, jobs);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

int launch_chain_bwd(const ChainBwdParams& c, int max_rows, hipStream_t s) {
    if (max_rows <= 0) return PR_OK;
    int cu_count = 0;
    PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_chain_bwd), (int)sizeof(Smem), &cu_count));
    const int max_tiles = (max_rows + TILE_M - 1) / TILE_M;
    const int resident = cu_count * MLP_BLOCKS_PER_CU;
    ProfileScope scope(2, s);
    hipLaunchKernelGGL(k_chain_bwd, dim3(max_tiles < resident ? max_tiles : resident), dim3(MLP_THREADS), sizeof(Smem), s, c);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 MFMA peak probe ("confirm the peak on the box with a pure-MFMA microbenchmark")
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_probe_mfma(int iterations, float* sink, int random_operands) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc0[i] = 0.f;
        acc1[i] = 1.f;
    }
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    unsigned int x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int it = 0; it < iterations; ++it) {
        if (random_operands) {
            // full-range mantissas that change every iteration (xorshift -> floats in [-1, 1))
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            a = __uint_as_float((x & 0x807FFFFFu) | 0x3F000000u);
            b = __uint_as_float(((x * 2654435761u) & 0x807FFFFFu) | 0x3F000000u);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc1, 0, 0, 0);
        if (random_operands && (it & 63) == 63) {
            // keep the accumulators bounded so that the data stays "ordinary"
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0[i] *= 0.001f;
                acc1[i] *= 0.001f;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 123.456f) sink[0] = s;  // keep the chain alive
}

}  // namespace pr

extern "C" int pr_probe_mfma_f32(int32_t iterations, int32_t random_operands, double* tflops, double* milliseconds, void* stream) {
    PR_REQUIRE(iterations > 0 && tflops, "pr_probe_mfma_f32: bad argument");
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* sink = nullptr;
    PR_CHECK_HIP(hipMalloc(&sink, sizeof(float)));
    hipEvent_t e0, e1;
    PR_CHECK_HIP(hipEventCreate(&e0));
    PR_CHECK_HIP(hipEventCreate(&e1));
    hipStream_t s = (hipStream_t)stream;
#ifndef PR_PROBE_WAVES
#define PR_PROBE_WAVES 8   // the renderer's occupancy; 4 = one wave per SIMD (experiments)
#endif
    hipLaunchKernelGGL(pr::k_probe_mfma, dim3(cus), dim3(64 * PR_PROBE_WAVES), 0, s, 16, sink, random_operands);  // warm-up
    PR_CHECK_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(pr::k_probe_mfma, dim3(cus), dim3(64 * PR_PROBE_WAVES), 0, s, iterations, sink, random_operands);
    PR_CHECK_HIP(hipEventRecord(e1, s));
    PR_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)cus * PR_PROBE_WAVES * (double)iterations * 8.0 * (2.0 * 32 * 32 * 2);
    *tflops = flop / (ms * 1e-3) / 1e12;
    if (milliseconds) *milliseconds = ms;
    PR_CHECK_HIP(hipEventDestroy(e0));
    PR_CHECK_HIP(hipEventDestroy(e1));
    PR_CHECK_HIP(hipFree(sink));
    return PR_OK;
}

namespace pr {
}  // namespace pr

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int pr_packed_size(const pr_object_model_t* model, size_t* bytes) {
    PR_REQUIRE(model && bytes, "pr_packed_size: NULL argument");
    pr::ModelDims d;
    pr::PackedLayout l;
    PR_TRY(pr::compute_dims(*model, &d));
    PR_TRY(pr::compute_layout(*model, d, &l));
    *bytes = (size_t)l.total * sizeof(float);
    return PR_OK;
}

extern "C" int pr_pack_model(const pr_object_model_t* model, int32_t precision, void* packed, size_t packed_bytes, void* stream) {
    PR_REQUIRE(precision == PR_PRECISION_FP32 || precision == PR_PRECISION_F16X3 || precision == PR_PRECISION_F16,
               "pr_pack_model: precision must be one of PR_PRECISION_*");
    PR_REQUIRE(model && packed, "pr_pack_model: NULL argument");
    PR_REQUIRE(((uintptr_t)packed & 15) == 0, "pr_pack_model: packed buffer must be 16-byte aligned");
    pr::ModelDims d;
    pr::PackedLayout l;
    PR_TRY(pr::compute_dims(*model, &d));
    PR_TRY(pr::compute_layout(*model, d, &l));
    PR_REQUIRE(packed_bytes >= (size_t)l.total * sizeof(float), "pr_pack_model: buffer too small (%zu < %zu)",
               packed_bytes, (size_t)l.total * sizeof(float));
    pr::PackJobs jobs;
    jobs.seg_kind = precision ? 2 : 0;   // fragment layout of the weight segments: exact fp32 or fp16 hi / lo pairs
    PR_TRY(pr::build_pack_jobs(*model, d, l, static_cast<float*>(packed), &jobs));
    hipLaunchKernelGGL(pr::k_pack, dim3(64, jobs.n), dim3(256), 0, (hipStream_t)stream, jobs);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// Several models in as few launches as their jobs allow (a training step re-packs every model after every optimiser step:
// three launches of ~12 us for the minecraft renderers' three models become one).
extern "C" int pr_pack_models(int32_t count, const pr_object_model_t* const* models, const int32_t* precisions, void* const* packed,
                              const size_t* packed_bytes, void* stream) {
    PR_REQUIRE(count >= 0 && (count == 0 || (models && precisions && packed && packed_bytes)), "pr_pack_models: NULL argument");
    static thread_local pr::PackJobs all, one;
    all.n = 0;
    all.seg_kind = 0;
    auto flush = [&]() -> int {
        if (all.n == 0) return PR_OK;
        hipLaunchKernelGGL(pr::k_pack, dim3(64, all.n), dim3(256), 0, (hipStream_t)stream, all);
        PR_LAUNCH_CHECK();
        all.n = 0;
        return PR_OK;
    };
    for (int i = 0; i < count; ++i) {
        PR_REQUIRE(precisions[i] == PR_PRECISION_FP32 || precisions[i] == PR_PRECISION_F16X3 || precisions[i] == PR_PRECISION_F16,
                   "pr_pack_models: precision must be one of PR_PRECISION_*");
        PR_REQUIRE(models[i] && packed[i], "pr_pack_models: NULL model or buffer");
        PR_REQUIRE(((uintptr_t)packed[i] & 15) == 0, "pr_pack_models: packed buffer must be 16-byte aligned");
        pr::ModelDims d;
        pr::PackedLayout l;
        PR_TRY(pr::compute_dims(*models[i], &d));
        PR_TRY(pr::compute_layout(*models[i], d, &l));
        PR_REQUIRE(packed_bytes[i] >= (size_t)l.total * sizeof(float), "pr_pack_models: buffer %d too small (%zu < %zu)", i, packed_bytes[i],
                   (size_t)l.total * sizeof(float));
        one.seg_kind = precisions[i] ? 2 : 0;
        PR_TRY(pr::build_pack_jobs(*models[i], d, l, static_cast<float*>(packed[i]), &one));
        if (all.n + one.n > pr::MAX_PACK_JOBS) PR_TRY(flush());
        for (int j = 0; j < one.n; ++j) all.job[all.n++] = one.job[j];
    }
    return flush();
}
