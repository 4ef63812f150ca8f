// Backward pass of the renderer (what autograd does for the reference graph when
// training/trainer_backpropagated_autoencoder.py:349 calls total_loss.backward()):
//
//   compositing backward  (object_composer.py:153-214, :399-447, :724-784 differentiated)
//     -> per object: feature-head / BatchNorm(train) / AdaIN backward, backbone, positional encoding,
//        ray bender (adain_style_nerf_model.py:57-145, layers/adain.py:5-61, positional_ray_bender_model.py:81-163)
//     -> sample placement / slab test backward into the object poses (ray_helper.py:1180-1282,
//        object_composer.py:104-151).
//
// The forward pass ran with PR_FLAG_SAVE_FOR_BACKWARD: every layer input is in the forward workspace as
// compact fp32 rows, so each layer's backward is two dense products - dX = dY.W and dW = dY^T.X (gemm.hip,
// exact fp32 MFMA) - plus thin elementwise kernels.  Gradient buffers are accumulated into.
#include "pr_common.h"

#include <algorithm>
#include <map>
#include <utility>
#include <mutex>
#include "composite_dev.h"

namespace pr {

constexpr int BWD_SPLITS = 128;       // sample-dimension split of the weight-gradient products
constexpr int MAX_FCHUNK_B = 4;

// ---------------------------------------------------------------------------------------------
// Compositing backward
// ---------------------------------------------------------------------------------------------
struct CompositeBwdObject {
    const float* t;
    const float* sigma;
    const int32_t* slot;
    const float* dispmag;    // or NULL
    const float* feat;       // compact rows
    NoiseRef noise;          // integrate noise (N,R,P) or absent
    int positions;
    pr_entry_grads_t g;      // gradients of results["object_k"]
    float* g_feat;           // (cap, F) compact rows, every in-box row is written
    float* g_sigma;          // (N,R,P)
    float* g_t;              // (N,R,P)
    float* g_dm;             // (N,R,P) or NULL
    const float* g_sample_t; // (N,R,P) gradient of the exported sample depths, or NULL
    const float* divergence; // (N,R,P) the forward pass's Hutchinson estimates (PR_FLAG_DIVERGENCE_GRAD), or NULL
    float* g_div;            // (N,R,P) d loss / d divergence estimate, or NULL
};
struct CompositeBwdParams {
    int frames, rays, objects, static_objects, F;
    int fix_overlaps;
    int total_positions;
    int sort_size;
    int div_grad;            // gradients of integrated_divergence are given: one more per-entry LDS column
    int sigmoid;             // PR_FLAG_SIGMOID_FEATURES: the composited features are sigmoid(row); samples without a row carry 0.5
    int Fs;                  // floats between the rows of g_feat (F rounded up to 16: the head products walk K in 16-deep slabs)
    const float* ray_directions;
    NoiseRef noise_global;
    CompositeBwdObject obj[PR_MAX_OBJECTS];
    pr_entry_grads_t global;
    float* g_norm;           // (N,R) d loss / d |d| through the sample spacings dt |d| of every entry, or NULL
};

struct BwdSmem {
    unsigned int* key;
    float *tt, *sg, *dm, *wo, *wg, *al, *gs, *gt, *gd, *Tj, *wv, *dw, *dd, *gv;
    int *sl, *mk;
};

// T_j = prod_{i<j} (1 - alpha_i + 1e-10) and w_j = alpha_j T_j, same association as the forward kernel
__device__ __forceinline__ void transmittance_scan(const float* al, float* Tj, float* wv, int n, int lane) {
    float carry = 1.0f;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const float a = (j < n) ? al[j] : 0.f;
        float incl = (j < n) ? __fadd_rn(__fsub_rn(1.0f, a), 1e-10f) : 1.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(incl, d, 64);
            if (lane >= d) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        if (j < n) {
            Tj[j] = carry * excl;
            wv[j] = a * (carry * excl);
        }
        carry *= __shfl(incl, 63, 64);
    }
}

// Backward of integrate() over one sample list of the ray.  sorted = false: entries off .. off + n in
// order; sorted = true: the merged list in key order.  Adds into gs / gt / gd (per concatenation entry).
__device__ __forceinline__ void entry_backward(const CompositeBwdParams& p, BwdSmem& sm, bool sorted, int off, int n,
                                               const NoiseRef& noise, float norm, const pr_entry_grads_t& g, long ray,
                                               float* weights_out, int lane, float& g_norm) {
    const bool noisy = noise_present(noise);
    const int F = p.F;
    auto entry_of = [&](int j) -> int { return sorted ? (int)sm.key[j] : off + j; };
    for (int j = lane; j < n; j += 64) {
        const int e = entry_of(j);
        const float dt = (j < n - 1) ? __fsub_rn(sm.tt[entry_of(j + 1)], sm.tt[e]) : 1e10f;
        float raw = sm.sg[e];
        if (noisy) raw = __fadd_rn(raw, noise_normal(noise, ray, n, j));
        const float a = alpha_of(raw, __fmul_rn(dt, norm));
        sm.al[j] = a;
        // integrated_divergence = mean_j (alpha_j |div_j|), alphas detached (object_composer.py:768-769); carved entries
        // of the merged list carry div = 0
        if (p.div_grad && g.integrated_divergence && !sm.mk[e]) sm.gv[e] += g.integrated_divergence[ray] * a / (float)n;
    }
    __syncthreads();
    transmittance_scan(sm.al, sm.Tj, sm.wv, n, lane);
    __syncthreads();
    for (int j = lane; j < n; j += 64) weights_out[entry_of(j)] = sm.wv[j];

    const float gO = g.opacity ? g.opacity[ray] : 0.f;
    const float gD = g.depth ? g.depth[ray] : 0.f;
    const float gM = g.integrated_displacements_magnitude ? g.integrated_displacements_magnitude[ray] : 0.f;
    const bool has_gf = g.integrated_features != nullptr;
    const float* gW = g.weights ? g.weights + (size_t)ray * n : nullptr;   // in list order (merged order for the global entry)
    if (!has_gf && !gW && gO == 0.f && gD == 0.f && gM == 0.f) {   // uniform: nothing flows into this entry
        __syncthreads();
        return;
    }
    // d loss / d w_j = gF . f_j + gO + gD t_j + gW_j
    if (has_gf && !p.sigmoid && (F & 3) == 0 && F <= 256) {
        // four entries at a time: 16 lanes per feature row (16-byte loads, up to four per lane), one reduction over the 16-lane
        // groups for all four dot products - the rows of a ray are read with four requests in flight instead of one after the other
        const int grp = lane >> 4, sub = lane & 15;
        const int f4n = F >> 2;
        float4 gF4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int q = sub + 16 * c;
            gF4[c] = q < f4n ? *reinterpret_cast<const float4*>(g.integrated_features + (size_t)ray * F + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int j0 = 0; j0 < n; j0 += 4) {
            const int j = j0 + grp;
            const float* f = nullptr;
            int e = 0;
            if (j < n) {
                e = entry_of(j);
                const int row = sm.sl[e];
                if (row >= 0 && sm.Tj[j] != 0.f) {
                    int k = 0, o2 = 0;
                    while (k + 1 < p.objects && e >= o2 + p.obj[k].positions) {
                        o2 += p.obj[k].positions;
                        ++k;
                    }
                    f = p.obj[k].feat + (size_t)row * F;
                }
            }
            float4 fv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int q = sub + 16 * c;
                fv[c] = (f && q < f4n) ? *reinterpret_cast<const float4*>(f + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                part = fmaf(gF4[c].x, fv[c].x, part);
                part = fmaf(gF4[c].y, fv[c].y, part);
                part = fmaf(gF4[c].z, fv[c].z, part);
                part = fmaf(gF4[c].w, fv[c].w, part);
            }
            part += __shfl_xor(part, 8, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 1, 64);
            if (sub == 0 && j < n) sm.dw[j] = part + gO + gD * sm.tt[e] + (gW ? gW[j] : 0.f);
        }
    } else {
    float gF[MAX_FCHUNK_B];
#pragma unroll
    for (int c = 0; c < MAX_FCHUNK_B; ++c) {
        const int ch = lane + 64 * c;
        gF[c] = (has_gf && ch < F) ? g.integrated_features[(size_t)ray * F + ch] : 0.f;
    }
    for (int j = 0; j < n; ++j) {
        const int e = entry_of(j);
        const int row = sm.sl[e];
        float dot = 0.f;
        if (has_gf && (row >= 0 || p.sigmoid) && sm.Tj[j] != 0.f) {
            int k = 0, o2 = 0;
            while (k + 1 < p.objects && e >= o2 + p.obj[k].positions) {
                o2 += p.obj[k].positions;
                ++k;
            }
            const float* f = row >= 0 ? p.obj[k].feat + (size_t)row * F : nullptr;
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK_B; ++c) {
                const int ch = lane + 64 * c;
                if (ch < F) {
                    float v = f ? f[ch] : 0.f;                           // (a sample without a row: raw feature 0)
                    if (p.sigmoid) v = 1.0f / (1.0f + expf(-v));
                    part = fmaf(gF[c], v, part);
                }
            }
            dot = wave_sum(part);
        }
        if (lane == 0) sm.dw[j] = dot + gO + gD * sm.tt[e] + (gW ? gW[j] : 0.f);
    }
    }
    __syncthreads();
    // d loss / d alpha_j = dw_j T_j - (sum_{i>j} dw_i w_i) / (1 - alpha_j + 1e-10): suffix sums by 64-entry blocks from the end of the
    // list (a wave-level scan per block, the blocks chained through `carry`)
    {
        float carry = 0.f;
        for (int base = ((n - 1) / 64) * 64; base >= 0; base -= 64) {
            const int j = base + lane;
            const float dwj = j < n ? sm.dw[j] : 0.f;
            float incl = j < n ? dwj * sm.wv[j] : 0.f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_down(incl, d, 64);
                if (lane + d < 64) incl += o;
            }
            float excl = __shfl_down(incl, 1, 64);
            if (lane == 63) excl = 0.f;
            const float total = __shfl(incl, 0, 64);
            if (j < n) sm.dw[j] = dwj * sm.Tj[j] - (excl + carry) / __fadd_rn(__fsub_rn(1.0f, sm.al[j]), 1e-10f);
            carry += total;
        }
    }
    __syncthreads();
    for (int j = lane; j < n; j += 64) {
        const int e = entry_of(j);
        const float dt = (j < n - 1) ? __fsub_rn(sm.tt[entry_of(j + 1)], sm.tt[e]) : 1e10f;
        const float dist = __fmul_rn(dt, norm);
        float raw = sm.sg[e];
        if (noisy) raw = __fadd_rn(raw, noise_normal(noise, ray, n, j));
        const float s = raw > 0.f ? raw : 0.f;
        const float E = expf(__fmul_rn(-s, dist));
        const float da = sm.dw[j];
        sm.dd[j] = (j < n - 1) ? da * s * E * norm : 0.f;
        if (j < n - 1) g_norm = fmaf(da * s * E, dt, g_norm);     // alpha_j = 1 - exp(-s dt |d|)
        if (!sm.mk[e]) {
            if (raw > 0.f) sm.gs[e] += da * dist * E;
            sm.gt[e] += gD * sm.wv[j];
            sm.gd[e] += gM * sm.wv[j] / (float)n;
        }
    }
    __syncthreads();
    for (int j = lane; j < n; j += 64) {
        const int e = entry_of(j);
        if (!sm.mk[e]) sm.gt[e] += ((j > 0) ? sm.dd[j - 1] : 0.f) - sm.dd[j];
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_composite_bwd(CompositeBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char raw_smem[];
    const int S = p.sort_size;
    const int A = (p.total_positions + 63) & ~63;
    BwdSmem sm;
    sm.key = reinterpret_cast<unsigned int*>(raw_smem);
    float* fp = reinterpret_cast<float*>(sm.key + S);
    sm.tt = fp; fp += A;
    sm.sg = fp; fp += A;
    sm.dm = fp; fp += A;
    sm.wo = fp; fp += A;
    sm.wg = fp; fp += A;
    sm.al = fp; fp += A;
    sm.gs = fp; fp += A;
    sm.gt = fp; fp += A;
    sm.gd = fp; fp += A;
    sm.Tj = fp; fp += A;
    sm.wv = fp; fp += A;
    sm.dw = fp; fp += A;
    sm.dd = fp; fp += A;
    sm.sl = reinterpret_cast<int*>(fp); fp += A;
    sm.mk = reinterpret_cast<int*>(fp); fp += A;
    sm.gv = nullptr;
    if (p.div_grad) {
        sm.gv = fp;
        fp += A;
    }
    // 64-bit sort scratch of the calls that always take the bitonic network (overlap fix); 4 S + 60 A bytes precede it
    unsigned long long* wide = p.fix_overlaps ? reinterpret_cast<unsigned long long*>(fp) : nullptr;

    const int lane = threadIdx.x;
    const long g = blockIdx.x;
    const float* d = p.ray_directions + (size_t)g * 3;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const int PT = p.total_positions;
    float g_norm = 0.f;      // this lane's share of d loss / d |d|

    int off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeBwdObject& o = p.obj[k];
        const int P = o.positions;
        const size_t base = (size_t)g * P;
        for (int i = lane; i < P; i += 64) {
            sm.tt[off + i] = o.t[base + i];
            sm.sg[off + i] = o.sigma[base + i];
            sm.sl[off + i] = o.slot[base + i];
            sm.dm[off + i] = o.dispmag ? o.dispmag[base + i] : 0.f;
            sm.gs[off + i] = 0.f;
            sm.gt[off + i] = 0.f;
            sm.gd[off + i] = 0.f;
            sm.mk[off + i] = 0;
            sm.wg[off + i] = 0.f;
            if (p.div_grad) sm.gv[off + i] = 0.f;
        }
        off += P;
    }
    __syncthreads();
    // ---- per-object entries (integrated before the overlap fix, as the reference does) ------------
    off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeBwdObject& o = p.obj[k];
        const int P = o.positions;
        entry_backward(p, sm, false, off, P, o.noise, norm, o.g, g, sm.wo, lane, g_norm);
        off += P;
    }
    // ---- overlap fix: carved static samples are constants (sigma = -10, t = 0, |delta| = 0) -----------
    if (p.fix_overlaps) {
        int dyn_off0 = 0;
        for (int k = 0; k < p.static_objects; ++k) dyn_off0 += p.obj[k].positions;
        int soff = 0;
        for (int s = 0; s < p.static_objects; ++s) {
            const int Ps = p.obj[s].positions;
            unsigned int masked_bits = 0;
            int doff = dyn_off0;
            for (int dd = p.static_objects; dd < p.objects; ++dd) {
                const float b0 = sm.tt[doff + 0];
                const float b1 = sm.tt[doff + Ps - 1];
                int lo0 = 0, hi0 = Ps;
                while (lo0 < hi0) {
                    const int mid = (lo0 + hi0) >> 1;
                    if (sm.tt[soff + mid] < b0) lo0 = mid + 1; else hi0 = mid;
                }
                int lo1 = 0, hi1 = Ps;
                while (lo1 < hi1) {
                    const int mid = (lo1 + hi1) >> 1;
                    if (sm.tt[soff + mid] < b1) lo1 = mid + 1; else hi1 = mid;
                }
                int m = 0;
                for (int i = lane; i < Ps; i += 64, ++m)
                    if (i >= lo0 && i < lo1) masked_bits |= 1u << m;
                doff += p.obj[dd].positions;
            }
            __syncthreads();
            int m = 0;
            for (int i = lane; i < Ps; i += 64, ++m) {
                if ((masked_bits >> m) & 1u) {
                    sm.tt[soff + i] = 0.f;
                    sm.sg[soff + i] = -10.0f;
                    sm.dm[soff + i] = 0.f;
                    sm.mk[soff + i] = 1;
                }
            }
            __syncthreads();
            soff += Ps;
        }
    }
    // ---- merged list ---------------------------------------------------------------------------------
    {
        int counts[PR_MAX_OBJECTS];
        for (int k = 0; k < p.objects; ++k) counts[k] = p.obj[k].positions;
        order_entries(sm.key, sm.tt, counts, p.objects, PT, S, !p.fix_overlaps, lane, 64, wide);
    }
    entry_backward(p, sm, true, 0, PT, p.noise_global, norm, p.global, g, sm.wg, lane, g_norm);
    if (p.g_norm) {
        const float total = wave_sum(g_norm);
        if (lane == 0) p.g_norm[g] = total;
    }

    // ---- write the per-sample gradients ----------------------------------------------------------------
    const int F = p.F;
    float gFg[MAX_FCHUNK_B];
#pragma unroll
    for (int c = 0; c < MAX_FCHUNK_B; ++c) {
        const int ch = lane + 64 * c;
        gFg[c] = (p.global.integrated_features && ch < F) ? p.global.integrated_features[(size_t)g * F + ch] : 0.f;
    }
    off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeBwdObject& o = p.obj[k];
        const int P = o.positions;
        const size_t base = (size_t)g * P;
        for (int i = lane; i < P; i += 64) {
            o.g_sigma[base + i] = sm.gs[off + i];
            o.g_t[base + i] = sm.gt[off + i] + (o.g_sample_t ? o.g_sample_t[base + i] : 0.f);
            if (o.g_dm) o.g_dm[base + i] = sm.gd[off + i];
            if (o.g_div) {     // d |div| / d div (0 at 0, like torch.abs)
                const float dv = o.divergence[base + i];
                o.g_div[base + i] = dv > 0.f ? sm.gv[off + i] : (dv < 0.f ? -sm.gv[off + i] : 0.f);
            }
        }
        float gFo[MAX_FCHUNK_B];
#pragma unroll
        for (int c = 0; c < MAX_FCHUNK_B; ++c) {
            const int ch = lane + 64 * c;
            gFo[c] = (o.g.integrated_features && ch < F) ? o.g.integrated_features[(size_t)g * F + ch] : 0.f;
        }
        for (int i = 0; i < P; ++i) {
            const int row = sm.sl[off + i];
            if (row < 0) continue;
            const float w1 = sm.wo[off + i], w2 = sm.wg[off + i];
            float* dst = o.g_feat + (size_t)row * p.Fs;
            const float* f = o.feat + (size_t)row * F;
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK_B; ++c) {
                const int ch = lane + 64 * c;
                if (ch < F) {
                    float gv = fmaf(w1, gFo[c], w2 * gFg[c]);
                    if (p.sigmoid) {                                     // d sigmoid(x) / dx = s (1 - s)
                        const float sv = 1.0f / (1.0f + expf(-f[ch]));
                        gv *= sv * (1.0f - sv);
                    }
                    dst[ch] = gv;
                } else if (ch < p.Fs) {
                    dst[ch] = 0.f;                                       // padding columns of the head products
                }
            }
        }
        off += P;
    }
}

static int launch_composite_bwd(const CompositeBwdParams& p, hipStream_t s) {
    PR_REQUIRE(p.F <= 64 * MAX_FCHUNK_B, "output_features %d exceeds %d", p.F, 64 * MAX_FCHUNK_B);
    const size_t lds = (size_t)p.sort_size * 4 + (size_t)((p.total_positions + 63) & ~63) * (p.div_grad ? 16 : 15) * 4 +
                       (p.fix_overlaps ? (size_t)p.sort_size * 8 : 0);
    PR_REQUIRE(lds <= 156 * 1024, "too many samples per ray for the compositing backward kernel (%d)", p.total_positions);
    PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_composite_bwd), 156 * 1024, nullptr));   // the block-wide vote of order_entries owns a little static LDS
    const long total = (long)p.frames * p.rays;
    ProfileScope scope(4, s);
    hipLaunchKernelGGL(k_composite_bwd, dim3((unsigned)total), dim3(64), lds, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-wise kernels of the per-object network backward.  "rows" = compact evaluated samples, M = *total.
// flags: bit 0 = real sample, bit 1 = passed the second AABB test (its outputs are used).
// ---------------------------------------------------------------------------------------------
struct RowCtx {
    const int32_t* total;
    const int32_t* rec_flat;
    const int32_t* row_flags;
    int samples_per_frame;
    const uint8_t* in_scene;     // (N,K) base offset to this object; stride K: the densities of an absent object are constants
    int in_scene_stride;
};

// gathers the dense sigma / |delta| gradients to rows and clears the feature gradients of unused rows
__global__ __launch_bounds__(256) void k_gather_rows(RowCtx r, const float* g_sigma, const float* g_dm, float* gsr, float* gdr,
                                                     float* g_feat, int F) {   // F: floats per row of g_feat (padded)
    const int M = *r.total;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const int fl = r.row_flags[m];
        const int flat = r.rec_flat[m];
        if (threadIdx.x == 0) {
            const bool present = r.in_scene[(size_t)(flat / r.samples_per_frame) * r.in_scene_stride] != 0;
            gsr[m] = ((fl & 3) == 3 && present) ? g_sigma[flat] : 0.f;
            if (gdr) gdr[m] = (fl & 1) ? g_dm[flat] : 0.f;
        }
        if ((fl & 3) != 3)
            for (int c = threadIdx.x; c < F; c += 256) g_feat[(size_t)m * F + c] = 0.f;
    }
}

// a = relu(h * g[frame] + b[frame]) for the alive rows, 0 otherwise (train-mode AdaIN table: g = scale * rstd)
__global__ __launch_bounds__(256) void k_adain_recompute(RowCtx r, const float* h, int ld, int width, const float* table,
                                                         int table_stride, int goff, int boff, float* a) {
    const int M = *r.total;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const bool alive = (r.row_flags[m] & 3) == 3;
        const float* tab = table + (size_t)(r.rec_flat[m] / r.samples_per_frame) * table_stride;
        for (int c = threadIdx.x; c < width; c += 256) {
            float v = 0.f;
            if (alive) {
                v = fmaf(h[(size_t)m * ld + c], tab[goff + c], tab[boff + c]);
                v = v > 0.f ? v : 0.f;
            }
            a[(size_t)m * ld + c] = v;
        }
    }
}

struct AdainBwd {
    RowCtx r;
    const float* h; int ld; int width;
    const float* table; int table_stride, goff, boff;
    const float* mean; const float* var; float eps;
    float* g;               // in: d loss / d a (post-ReLU); out (reduce): d x_hat; out (apply): d h
    double* sums;           // [sum d x_hat (width) | sum d x_hat * x_hat (width)]
    float* dscale;          // (frames, width) accumulated
    float* dbias;           // (frames, width)
    const int32_t* count;   // rows that entered the batch statistics
    int frozen;             // eval mode: the statistics are constants (running buffers), no batch terms
};

// Block = 64 channels (blockIdx.y) x 4 row groups over 256 rows; thread (c, rg) walks rows rg, rg + 4, ...
// The four partial sums of a channel are combined through LDS before the atomics.
__global__ __launch_bounds__(256) void k_adain_bwd_reduce(AdainBwd p) {
    __shared__ float sh_ds[4][64], sh_db[4][64];
    __shared__ double sh_s1[4][64], sh_s2[4][64];
    __shared__ int sh_frame[4][64];
    const int M = *p.r.total;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool live = c < p.width;
    const float mu = live ? p.mean[c] : 0.f, rstd = live ? 1.0f / sqrtf(p.var[c] + p.eps) : 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int blk = blockIdx.x; blk * 256 < M; blk += gridDim.x) {
        const int m0 = blk * 256, m1 = (m0 + 256 < M) ? m0 + 256 : M;
        int cur = -1;
        float ds = 0.f, db = 0.f;
        if (live) {
            for (int m = m0 + rg; m < m1; m += 4) {
                const size_t at = (size_t)m * p.ld + c;
                float dxh = 0.f;
                if ((p.r.row_flags[m] & 3) == 3) {
                    const int frame = p.r.rec_flat[m] / p.r.samples_per_frame;
                    if (frame != cur) {
                        if (cur >= 0) {
                            atomicAdd(p.dscale + (size_t)cur * p.width + c, ds);
                            atomicAdd(p.dbias + (size_t)cur * p.width + c, db);
                        }
                        cur = frame;
                        ds = 0.f;
                        db = 0.f;
                    }
                    const float* tab = p.table + (size_t)frame * p.table_stride;
                    const float gg = tab[p.goff + c];
                    const float hv = p.h[at];
                    const float y = fmaf(hv, gg, tab[p.boff + c]);
                    const float dy = y > 0.f ? p.g[at] : 0.f;
                    const float xh = (hv - mu) * rstd;
                    ds = fmaf(dy, xh, ds);
                    db += dy;
                    dxh = dy * (gg / rstd);   // scale = g / rstd
                    s1 += (double)dxh;
                    s2 += (double)dxh * (double)xh;
                }
                p.g[at] = dxh;
            }
        }
        sh_frame[rg][cl] = cur;
        sh_ds[rg][cl] = ds;
        sh_db[rg][cl] = db;
        __syncthreads();
        if (rg == 0 && live) {
            // frames end where the last rows of the block are: groups usually agree on it
            for (int q = 0; q < 4; ++q) {
                const int f = sh_frame[q][cl];
                if (f < 0) continue;
                float a = sh_ds[q][cl], b = sh_db[q][cl];
                for (int q2 = q + 1; q2 < 4; ++q2)
                    if (sh_frame[q2][cl] == f) {
                        a += sh_ds[q2][cl];
                        b += sh_db[q2][cl];
                        sh_frame[q2][cl] = -1;
                    }
                atomicAdd(p.dscale + (size_t)f * p.width + c, a);
                atomicAdd(p.dbias + (size_t)f * p.width + c, b);
            }
        }
        __syncthreads();
    }
    sh_s1[rg][cl] = s1;
    sh_s2[rg][cl] = s2;
    __syncthreads();
    if (rg == 0 && live) {
        atomicAdd(p.sums + c, sh_s1[0][cl] + sh_s1[1][cl] + sh_s1[2][cl] + sh_s1[3][cl]);
        atomicAdd(p.sums + p.width + c, sh_s2[0][cl] + sh_s2[1][cl] + sh_s2[2][cl] + sh_s2[3][cl]);
    }
}

// BatchNorm (batch statistics) backward: dh = rstd (dxh - mean(dxh) - xh mean(dxh xh))
__global__ __launch_bounds__(256) void k_adain_bwd_apply(AdainBwd p) {
    const int M = *p.r.total;
    const int c = threadIdx.x;
    if (c >= p.width) return;
    const double n = (double)*p.count;
    const float mu = p.mean[c], rstd = 1.0f / sqrtf(p.var[c] + p.eps);
    const float m1 = p.frozen ? 0.f : (float)(p.sums[c] / n), m2 = p.frozen ? 0.f : (float)(p.sums[p.width + c] / n);
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const size_t at = (size_t)m * p.ld + c;
        float v = 0.f;
        if ((p.r.row_flags[m] & 3) == 3) {
            const float xh = (p.h[at] - mu) * rstd;
            v = rstd * (p.g[at] - m1 - xh * m2);
        }
        p.g[at] = v;
    }
}

// d act7 = (d act7 + gsr * w_sigma) masked by act7 > 0; d w_sigma += sum gsr * act7, d b_sigma += sum gsr
// (64 channels x 4 row groups per block, like k_adain_bwd_reduce)
__global__ __launch_bounds__(256) void k_sigma_bwd(RowCtx r, float* g, const float* act, int ld, int width, const float* gsr,
                                                   const float* w_sigma, float* dw, float* db) {
    __shared__ float sh[4][64], shb[4][64];
    const int M = *r.total;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool live = c < width;
    const float ws = (live && gsr && w_sigma) ? w_sigma[c] : 0.f;
    float acc = 0.f, accb = 0.f;
    if (live) {
        for (int blk = blockIdx.x; blk * 256 < M; blk += gridDim.x) {
            const int m0 = blk * 256, m1 = (m0 + 256 < M) ? m0 + 256 : M;
            for (int m = m0 + rg; m < m1; m += 4) {
                const size_t at = (size_t)m * ld + c;
                const float gs = gsr ? gsr[m] : 0.f;
                const float a = act[at];
                acc = fmaf(gs, a, acc);
                accb += gs;
                const float v = fmaf(gs, ws, g[at]);
                g[at] = a > 0.f ? v : 0.f;
            }
        }
    }
    sh[rg][cl] = acc;
    shb[rg][cl] = accb;
    __syncthreads();
    if (rg == 0 && live && gsr) {
        if (dw) atomicAdd(dw + c, sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]);
        if (db && c == 0) atomicAdd(db, shb[0][cl] + shb[1][cl] + shb[2][cl] + shb[3][cl]);
    }
}

// Positional-encoding backward from the saved encoding (sin / cos values are reused):
//   d v_a = g[a] + sum_k 2^k (cos_ka g[sin_ka] - sin_ka g[cos_ka])
// out[m][a] (ld_out floats per row) = d v_a / divide[a]; rows that are not `need` get zeros.
__global__ __launch_bounds__(256) void k_pe_bwd(RowCtx r, const float* enc, const float* g_enc, int ld, int din, int octaves,
                                                float d0, float d1, float d2, int need_mask, float* out, int ld_out, int accumulate) {
    const int M = *r.total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const bool need = (r.row_flags[m] & need_mask) == need_mask;
    const float* e = enc + (size_t)m * ld;
    const float* g = g_enc + (size_t)m * ld;
    const float div[3] = {d0, d1, d2};
    for (int a = 0; a < din; ++a) {
        float v = 0.f;
        if (need) {
            v = g[a];
            for (int k = 0; k < octaves; ++k) {
                const int sn = din + k * 2 * din + a, cs = sn + din;
                v += ldexpf(1.0f, k) * (e[cs] * g[sn] - e[sn] * g[cs]);
            }
            if (a < 3 && div[a] != 0.f) v /= div[a];
        }
        float* dst = out + (size_t)m * ld_out + a;
        *dst = accumulate ? *dst + v : v;
    }
}

// Bender output backward: bent = x + delta, delta = clamp(raw * size, lo - x, hi - x) (* 0 in canonical pose),
// |delta| feeds integrated_displacements_magnitude.  In: g_bent (M,3) = d loss / d bent.  Out: g_x (M,3)
// (the direct paths), g_braw (M,3).
__global__ __launch_bounds__(256) void k_bender_out_bwd(RowCtx r, const float* g_bent, const float* gdr, const float* g_delta_dense,
                                                        const float* delta, const float* braw, const float* pos, float lo0,
                                                        float lo1, float lo2, float hi0, float hi1, float hi2, int canonical,
                                                        float* g_x, float* g_braw) {
    const int M = *r.total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const bool real = (r.row_flags[m] & 1) != 0;
    float dl[3], nrm = 0.f;
    for (int a = 0; a < 3; ++a) {
        dl[a] = delta[(size_t)m * 3 + a];
        nrm = fmaf(dl[a], dl[a], nrm);
    }
    nrm = sqrtf(nrm);
    const float gd = (real && gdr) ? gdr[m] : 0.f;
    for (int a = 0; a < 3; ++a) {
        const float gb = real ? g_bent[(size_t)m * 3 + a] : 0.f;
        float g_delta = gb + (nrm > 0.f ? gd * dl[a] / nrm : 0.f);
        // gradient of the exported displacement vector itself (sample_delta, on the sample grid)
        if (real && g_delta_dense) g_delta += g_delta_dense[(size_t)r.rec_flat[m] * 3 + a];
        if (canonical) g_delta = 0.f;
        const float x = pos[(size_t)m * 3 + a];
        const float pre = braw[(size_t)m * 3 + a] * (hi[a] - lo[a]);
        const float lob = lo[a] - x, hib = hi[a] - x;
        const float m1 = pre > lob ? pre : lob;
        float g_pre = 0.f, gx = gb;
        if (m1 > hib) gx -= g_delta;            // delta = hi - x
        else if (pre >= lob) g_pre = g_delta;   // unclamped
        else gx -= g_delta;                     // delta = lo - x
        g_x[(size_t)m * 3 + a] = real ? gx : 0.f;
        g_braw[(size_t)m * 3 + a] = real ? g_pre * (hi[a] - lo[a]) : 0.f;
    }
}

// Bender head (3, BW), no bias: d act = (g_braw . W) masked by act > 0 ; dW[a][c] += sum_m g_braw[m][a] act[m][c]
// `x`: left factor of dW when it is not the activation itself (the probe tangents of the divergence estimate)
__global__ __launch_bounds__(256) void k_bender_head_bwd(RowCtx r, const float* g_braw, const float* act, const float* x, int ld, int width,
                                                         const float* w_out, int w_ld, float* g_act, float* dw) {
    __shared__ float sh[3][4][64];
    const int M = *r.total;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool live = c < width;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (live) {
        const float w0 = w_out[c], w1 = w_out[w_ld + c], w2 = w_out[2 * w_ld + c];
        for (int blk = blockIdx.x; blk * 256 < M; blk += gridDim.x) {
            const int m0 = blk * 256, m1 = (m0 + 256 < M) ? m0 + 256 : M;
            for (int m = m0 + rg; m < m1; m += 4) {
                const float g0 = g_braw[(size_t)m * 3], g1 = g_braw[(size_t)m * 3 + 1], g2 = g_braw[(size_t)m * 3 + 2];
                const float a = act[(size_t)m * ld + c];
                const float xv = x ? x[(size_t)m * ld + c] : a;
                a0 = fmaf(g0, xv, a0);
                a1 = fmaf(g1, xv, a1);
                a2 = fmaf(g2, xv, a2);
                g_act[(size_t)m * ld + c] = a > 0.f ? fmaf(g0, w0, fmaf(g1, w1, g2 * w2)) : 0.f;
            }
        }
    }
    sh[0][rg][cl] = a0;
    sh[1][rg][cl] = a1;
    sh[2][rg][cl] = a2;
    __syncthreads();
    if (rg == 0 && live && dw) {
        for (int a = 0; a < 3; ++a) atomicAdd(dw + a * w_ld + c, sh[a][0][cl] + sh[a][1][cl] + sh[a][2][cl] + sh[a][3][cl]);
    }
}

// d deformation[frame][j] += sum over the rows of the frame of g_bin[m][benc + j]
__global__ __launch_bounds__(64) void k_deformation_bwd(RowCtx r, const float* g_bin, int ld, int benc, int D, float* d_def,
                                                        int def_stride) {
    const int M = *r.total;
    const int j = threadIdx.x;
    if (j >= D) return;
    for (int blk = blockIdx.x; blk * 256 < M; blk += gridDim.x) {
        const int m0 = blk * 256, m1 = (m0 + 256 < M) ? m0 + 256 : M;
        int cur = -1;
        float acc = 0.f;
        for (int m = m0; m < m1; ++m) {
            if (!(r.row_flags[m] & 1)) continue;
            const int frame = r.rec_flat[m] / r.samples_per_frame;
            if (frame != cur) {
                if (cur >= 0) atomicAdd(d_def + (size_t)cur * def_stride + j, acc);
                cur = frame;
                acc = 0.f;
            }
            acc += g_bin[(size_t)m * ld + benc + j];
        }
        if (cur >= 0) atomicAdd(d_def + (size_t)cur * def_stride + j, acc);
    }
}

// Style affine backward: [scale | bias] = A style + b  (layers/adain.py:30-33).  d_out (frames, 2 width) is
// given as two tables dscale, dbias (frames, width).  Blocks [0, frames): d_style of one frame (the 2 width
// rows are spread over the threads); remaining blocks: dA / db elements.
__global__ __launch_bounds__(256) void k_style_bwd(int frames, int width, int S, const float* dscale, const float* dbias,
                                                   const float* style, int style_stride, const float* A, float* dA, float* db,
                                                   float* d_style) {
    __shared__ float sh[256];
    const int rows = 2 * width;
    if ((int)blockIdx.x < frames) {   // d_style[f][s] += sum_r A[r][s] d_out[f][r]
        const int f = blockIdx.x;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + (threadIdx.x & 63), part = threadIdx.x >> 6;
            float acc = 0.f;
            if (s < S)
                for (int rr = part; rr < rows; rr += 4) {
                    const float dv = rr < width ? dscale[(size_t)f * width + rr] : dbias[(size_t)f * width + rr - width];
                    acc = fmaf(A[(size_t)rr * S + s], dv, acc);
                }
            sh[threadIdx.x] = acc;
            __syncthreads();
            if (part == 0 && s < S && d_style)
                d_style[(size_t)f * style_stride + s] += sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
            __syncthreads();
        }
        return;
    }
    const long idx = (long)(blockIdx.x - frames) * 256 + threadIdx.x;
    if (idx < (long)rows * S) {   // dA[r][s] += sum_f d_out[f][r] style[f][s]
        const int rr = (int)(idx / S), s = (int)(idx - (long)rr * S);
        const float* src = rr < width ? dscale + rr : dbias + (rr - width);
        float acc = 0.f;
        for (int f = 0; f < frames; ++f) acc = fmaf(src[(size_t)f * width], style[(size_t)f * style_stride + s], acc);
        if (dA) dA[idx] += acc;
        if (s == 0 && db) {
            float b = 0.f;
            for (int f = 0; f < frames; ++f) b += src[(size_t)f * width];
            db[rr] += b;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sample placement backward: one thread per (frame, ray) of one object.
//   x_i = o + d t_i,  t_i = stratified(near, far),  near / far = clamped slab test of (o, d),
//   o = R o_w + T, d = R d_w   ->   d loss / d w2o (3 x 4)
// ---------------------------------------------------------------------------------------------
struct GeometryBwd {
    int frames, rays, positions, objects, object_index, kind;
    const float* ray_origins;
    const float* ray_directions;
    const float* w2o;
    const uint8_t* in_scene;
    float lo[3], hi[3];
    float z_near_min, z_far_max;
    const float* linspace;
    NoiseRef jitter;
    const float* t;          // (N,R,P) forward sample depths
    const int32_t* slot;     // (N,R,P)
    const float* g_t;        // (N,R,P) from the compositing backward
    const float* g_x;        // (cap,3) rows: d loss / d x (object frame), or NULL
    const float* g_in6;      // skybox: (cap,6) rows: d loss / d [o / size, d / |d|]
    float* d_w2o;            // (N,K,3,4), or NULL
    float* d_ray_origins;    // (N,3) accumulated, or NULL
    float* d_ray_directions; // (N,R,3) accumulated (launches of one stream follow each other: plain read-modify-write), or NULL
    const float* g_norm;     // (N,R) d loss / d |d_world| from the compositing backward: added by ONE launch per model type
    // hierarchical pass: `positions` = Pc + Pf merged depths; the entries that are coarse depths (matched by value
    // against t_coarse, both lists are sorted) carry the near / far dependence, the resampled ones are constants
    const float* t_coarse;   // (N,R,Pc) or NULL (coarse pass)
    int pc;
};

__device__ __forceinline__ void geometry_bwd_body(const GeometryBwd& p) {
    __shared__ float red[15 * 4];
    const int n = blockIdx.y;                              // frame
    const int ray = blockIdx.x * 256 + threadIdx.x;
    const long g = (long)n * p.rays + ray;
    float dM[15];            // 12 matrix entries, then the frame's ray origin
#pragma unroll
    for (int i = 0; i < 15; ++i) dM[i] = 0.f;
    if (ray < p.rays) {
        const float* M = p.w2o + ((size_t)n * p.objects + p.object_index) * 12;
        const float* ow = p.ray_origins + (size_t)n * 3;
        const float* dw = p.ray_directions + (size_t)g * 3;
        const ObjRay rr = object_ray(M, ow, dw);
        const bool present = p.in_scene[(size_t)n * p.objects + p.object_index] != 0;
        // forward slab test, remembering the selected candidates
        float zmin[3], zmax[3];
        int min_hi[3];   // 1: the high corner gave the minimum along this axis
        for (int a = 0; a < 3; ++a) {
            const float den = __fadd_rn(rr.d[a], 1e-6f);
            const float tl = __fdiv_rn(__fsub_rn(p.lo[a], rr.o[a]), den);
            const float th = __fdiv_rn(__fsub_rn(p.hi[a], rr.o[a]), den);
            min_hi[a] = th < tl;
            zmin[a] = min_hi[a] ? th : tl;
            zmax[a] = min_hi[a] ? tl : th;
        }
        int an = 0, af = 0;
        for (int a = 1; a < 3; ++a) {
            if (zmin[a] > zmin[an]) an = a;
            if (zmax[a] < zmax[af]) af = a;
        }
        float near = zmin[an], far = zmax[af];
        const bool hit = present && !(far <= near);
        bool near_free = hit && near >= p.z_near_min && near <= p.z_far_max;
        bool far_free = hit && far >= p.z_near_min && far <= p.z_far_max;
        // accumulate over the samples
        float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, g_near = 0.f, g_far = 0.f;
        const int P = p.positions;
        const size_t base = (size_t)g * P;
        const int Pc = p.t_coarse ? p.pc : P;
        const size_t cbase = (size_t)g * Pc;
        int jc = 0;   // next unmatched coarse depth (hierarchical pass)
        for (int i = 0; i < P; ++i) {
            float gt = p.g_t[base + i];
            const int row = p.slot[base + i];
            if (row >= 0 && p.g_x) {
                const float ti = p.t[base + i];
                for (int a = 0; a < 3; ++a) {
                    const float gx = p.g_x[(size_t)row * 3 + a];
                    go[a] += gx;
                    gd[a] = fmaf(gx, ti, gd[a]);
                    gt = fmaf(gx, rr.d[a], gt);
                }
            }
            if (row >= 0 && p.g_in6) {   // skybox input [o / size, d / |d|]
                const float* gi = p.g_in6 + (size_t)row * 6;
                const float nrm = sqrtf(rr.d[0] * rr.d[0] + rr.d[1] * rr.d[1] + rr.d[2] * rr.d[2]);
                float dotg = 0.f;
                for (int a = 0; a < 3; ++a) dotg = fmaf(gi[3 + a], rr.d[a] / nrm, dotg);
                for (int a = 0; a < 3; ++a) {
                    go[a] += gi[a] / (p.hi[a] - p.lo[a]);
                    gd[a] += (gi[3 + a] - dotg * rr.d[a] / nrm) / nrm;
                }
            }
            // t_i = near A_i + far B_i  (stratified_positions: linspace placement, optional jitter between midpoints)
            int ci = i;   // index of the coarse depth this sample is
            if (p.t_coarse) {
                const float ti = p.t[base + i];
                while (jc < Pc && p.t_coarse[cbase + jc] < ti) ++jc;
                if (jc < Pc && p.t_coarse[cbase + jc] == ti) ci = jc++;
                else continue;   // resampled depth: detached (ray_helper.py:1340)
            }
            const float s_i = p.linspace[ci];
            float An = 1.0f - s_i, Bf = s_i;
            if (noise_present(p.jitter)) {
                const float u = noise_uniform(p.jitter, (long)(cbase / Pc), Pc, ci);
                const float s_lo = ci > 0 ? 0.5f * (p.linspace[ci - 1] + s_i) : s_i;
                const float s_hi = ci < Pc - 1 ? 0.5f * (p.linspace[ci + 1] + s_i) : s_i;
                Bf = s_lo + (s_hi - s_lo) * u;
                An = 1.0f - Bf;
            }
            g_near = fmaf(gt, An, g_near);
            g_far = fmaf(gt, Bf, g_far);
        }
        // slab test backward: tb = (c - o_a) / (d_a + eps)
        if (near_free) {
            const int a = an;
            const float c = min_hi[a] ? p.hi[a] : p.lo[a];
            const float den = __fadd_rn(rr.d[a], 1e-6f);
            go[a] -= g_near / den;
            gd[a] -= g_near * (c - rr.o[a]) / (den * den);
        }
        if (far_free) {
            const int a = af;
            const float c = min_hi[a] ? p.lo[a] : p.hi[a];
            const float den = __fadd_rn(rr.d[a], 1e-6f);
            go[a] -= g_far / den;
            gd[a] -= g_far * (c - rr.o[a]) / (den * den);
        }
        // o_i = sum_j M[i][j] ow_j + M[i][3],  d_i = sum_j M[i][j] dw_j
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) dM[i * 4 + j] = go[i] * ow[j] + gd[i] * dw[j];
            dM[i * 4 + 3] = go[i];
        }
        // world-frame ray: o = M[:, :3] ow + M[:, 3], d = M[:, :3] dw  ->  d ow = M^T go, d dw = M^T gd (+ the |d| term)
        if (p.d_ray_origins)
            for (int j = 0; j < 3; ++j) dM[12 + j] = M[j] * go[0] + M[4 + j] * go[1] + M[8 + j] * go[2];
        if (p.d_ray_directions) {
            float gn = 0.f, nrm = 1.f;
            if (p.g_norm) {
                gn = p.g_norm[g];
                nrm = sqrtf(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
            }
            for (int j = 0; j < 3; ++j) {
                float v = M[j] * gd[0] + M[4 + j] * gd[1] + M[8 + j] * gd[2];
                if (p.g_norm) v = fmaf(gn, dw[j] / nrm, v);
                atomicAdd(p.d_ray_directions + (size_t)g * 3 + j, v);     // (objects on two lanes may add to the same ray)
            }
        }
    }
    // block reduction (blockIdx.y = frame), then one atomic per matrix entry
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const float v = wave_sum(dM[i]);
        if (lane == 0) red[i * 4 + wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        const float v = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
        if (threadIdx.x < 12) {
            if (p.d_w2o) atomicAdd(p.d_w2o + ((size_t)n * p.objects + p.object_index) * 12 + threadIdx.x, v);
        } else if (p.d_ray_origins) {
            atomicAdd(p.d_ray_origins + (size_t)n * 3 + (threadIdx.x - 12), v);
        }
    }
}

__global__ __launch_bounds__(256) void k_geometry_bwd(GeometryBwd p) { geometry_bwd_body(p); }

// ---- gradient of the Hutchinson divergence estimate (object_composer.py:582-601, create_graph=True) -----------------
// the three probe components of compact sample `flat` (flat = ray * P + sample): element (flat % P) * 3 + a of ray flat / P
__device__ __forceinline__ void probe_of(const NoiseRef& noise, int flat, int positions, float* e) {
    const long ray = flat / positions;
    const int sample = flat - (int)ray * positions;
    for (int a = 0; a < 3; ++a) e[a] = noise_normal(noise, ray, positions * 3, sample * 3 + a);
}

// div = sum_a e_a (J e)_a, (J e)_a = size_a (W_out t_last)_a where the clamp passes the network's output (a constant
// otherwise): d loss / d (W_out t_last)_a = g_div e_a size_a there, 0 elsewhere.  Written as the "raw output" gradient
// the bender head's backward takes.
__global__ __launch_bounds__(256) void k_div_out_bwd(RowCtx r, NoiseRef noise, int positions, const float* g_div, const float* braw,
                                                     const float* pos, float lo0, float lo1, float lo2, float hi0, float hi1,
                                                     float hi2, int canonical, float* g_out) {
    const int M = *r.total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const int flat = r.rec_flat[m];
    const bool real = (r.row_flags[m] & 1) != 0 && !canonical;
    const float gd = real ? g_div[flat] : 0.f;
    float e[3] = {0.f, 0.f, 0.f};
    if (gd != 0.f) probe_of(noise, flat, positions, e);
    for (int a = 0; a < 3; ++a) {
        const float x = pos[(size_t)m * 3 + a];
        const float size = hi[a] - lo[a];
        const float pre = braw[(size_t)m * 3 + a] * size;
        const float lob = lo[a] - x, hib = hi[a] - x;
        const float m1 = pre > lob ? pre : lob;
        const bool passes = !(m1 > hib) && pre >= lob;
        g_out[(size_t)m * 3 + a] = passes ? gd * e[a] * size : 0.f;
    }
}

// The tangent of the bender input along the probe: t[a] = dv_a, t[sin slot] = 2^k c dv_a, t[cos slot] = -2^k s dv_a with
// dv_a = e_a / size_a and (s, c) the saved (annealed) encoding values.  d s / d v = 2^k c, d c / d v = -2^k s, so
// d loss / d v_a = - sum_k 4^k dv_a (g[sin slot] s + g[cos slot] c); added to the position gradient (/ size_a).
__global__ __launch_bounds__(256) void k_div_tangent_in_bwd(RowCtx r, NoiseRef noise, int positions, const float* bin, const float* g_t0,
                                                            int ld, int octaves, float s0, float s1, float s2, float* g_x) {
    const int M = *r.total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M || !(r.row_flags[m] & 1)) return;
    const float size[3] = {s0, s1, s2};
    float e[3];
    probe_of(noise, r.rec_flat[m], positions, e);
    const float* b = bin + (size_t)m * ld;
    const float* g = g_t0 + (size_t)m * ld;
    for (int a = 0; a < 3; ++a) {
        const float dv = e[a] / size[a];
        float acc = 0.f;
        for (int k = 0; k < octaves; ++k) {
            const float f = ldexpf(1.0f, k);
            const int sn = 3 + k * 6 + a, cs = sn + 3;
            acc = fmaf(-f * f * dv, fmaf(g[sn], b[sn], g[cs] * b[cs]), acc);
        }
        g_x[(size_t)m * 3 + a] += acc / size[a];
    }
}


// ---------------------------------------------------------------------------------------------
// Row kernels of the grouped backward pass: the objects of a call share every launch (blockIdx.y = job)
// ---------------------------------------------------------------------------------------------
constexpr int MAX_ROW_JOBS = 2 * PR_MAX_OBJECTS;

// After the NeRF chain: positional-encoding backward of the network input, then - objects with a ray bender - the bender
// output's backward (k_pe_bwd + k_bender_out_bwd of the per-object path in one pass over the rows).
struct PostNerfJob {
    RowCtx r;
    int kind, has_bender, octaves, ld;          // ld: row stride of enc / g_enc
    const float* enc; const float* g_enc;
    float size[3], lo[3], hi[3];
    float* g_x;                                 // (cap,3) d loss / d object-frame position (kind 0)
    float* g_in6;                               // (cap,6) d loss / d [o / size, d / |d|] (kind 1)
    // ray bender
    const float* g_dm;                          // (N,R,P) d loss / d |delta| from the compositing backward
    const float* g_delta_dense;                 // (N,R,P,3) gradient of the exported displacement vectors, or NULL
    const float* delta; const float* braw; const float* pos;
    int canonical;
    float* g_braw4;                             // (cap,4) d loss / d raw bender output
};
struct PostNerfJobs { PostNerfJob job[MAX_ROW_JOBS]; };

// 64 rows per workgroup: the rows' encodings and encoding gradients go through LDS (whole rows with 16-byte loads - a thread
// that walks its own row in global memory touches a different cache line per load), then one thread per row.
constexpr int POST_ROWS = 64;
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld, int m0, int rows) {   // dst[row][ld + 1]
    const int l4 = ld >> 2;
    for (int idx = threadIdx.x; idx < POST_ROWS * l4; idx += 256) {
        const int row = idx / l4, c = (idx - row * l4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows) v = *reinterpret_cast<const float4*>(src + (size_t)(m0 + row) * ld + c);
        float* d = dst + row * (ld + 1) + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__global__ __launch_bounds__(256) void k_post_nerf_group(PostNerfJobs jobs) {
    extern __shared__ float post_smem[];
    const PostNerfJob& p = jobs.job[blockIdx.y];
    const int M = *p.r.total;
    const int m0 = blockIdx.x * POST_ROWS;
    if (m0 >= M) return;
    const int rows = M - m0 < POST_ROWS ? M - m0 : POST_ROWS;
    float* se = post_smem;
    float* sg = post_smem + POST_ROWS * (p.ld + 1);
    stage_rows(se, p.enc, p.ld, m0, rows);
    stage_rows(sg, p.g_enc, p.ld, m0, rows);
    __syncthreads();
    if ((int)threadIdx.x >= rows) return;
    const int m = m0 + threadIdx.x;
    const int fl = p.r.row_flags[m];
    const bool need = (fl & 3) == 3;
    const int din = p.kind == 0 ? 3 : 6;
    const float* e = se + threadIdx.x * (p.ld + 1);
    const float* g = sg + threadIdx.x * (p.ld + 1);
    float v[6];
    for (int a = 0; a < din; ++a) {
        float acc = 0.f;
        if (need) {
            acc = g[a];
            for (int k = 0; k < p.octaves; ++k) {
                const int sn = din + k * 2 * din + a, cs = sn + din;
                acc += ldexpf(1.0f, k) * (e[cs] * g[sn] - e[sn] * g[cs]);
            }
            if (p.kind == 0) acc /= p.size[a];
        }
        v[a] = acc;
    }
    if (p.kind == 1) {
        for (int a = 0; a < 6; ++a) p.g_in6[(size_t)m * 6 + a] = v[a];
        return;
    }
    if (!p.has_bender) {
        for (int a = 0; a < 3; ++a) p.g_x[(size_t)m * 3 + a] = v[a];
        return;
    }
    // bent = x + delta, delta = clamp(raw * size, lo - x, hi - x) (* 0 in canonical pose); |delta| feeds integrated_displacements_magnitude
    const bool real = (fl & 1) != 0;
    float dl[3], nrm = 0.f;
    for (int a = 0; a < 3; ++a) {
        dl[a] = p.delta[(size_t)m * 3 + a];
        nrm = fmaf(dl[a], dl[a], nrm);
    }
    nrm = sqrtf(nrm);
    const int flat = p.r.rec_flat[m];
    const float gd = real ? p.g_dm[flat] : 0.f;
    float gr[3];
    for (int a = 0; a < 3; ++a) {
        const float gb = real ? v[a] : 0.f;
        float g_delta = gb + (nrm > 0.f ? gd * dl[a] / nrm : 0.f);
        if (real && p.g_delta_dense) g_delta += p.g_delta_dense[(size_t)flat * 3 + a];
        if (p.canonical) g_delta = 0.f;
        const float x = p.pos[(size_t)m * 3 + a];
        const float pre = p.braw[(size_t)m * 3 + a] * (p.hi[a] - p.lo[a]);
        const float lob = p.lo[a] - x, hib = p.hi[a] - x;
        const float m1 = pre > lob ? pre : lob;
        float g_pre = 0.f, gx = gb;
        if (m1 > hib) gx -= g_delta;
        else if (pre >= lob) g_pre = g_delta;
        else gx -= g_delta;
        p.g_x[(size_t)m * 3 + a] = real ? gx : 0.f;
        gr[a] = real ? g_pre * (p.hi[a] - p.lo[a]) : 0.f;
    }
    *reinterpret_cast<float4*>(p.g_braw4 + (size_t)m * 4) = make_float4(gr[0], gr[1], gr[2], 0.f);
}

// After the bender chain: positional-encoding backward of the bender input (added to the position gradient) and the
// deformation code's gradient, summed per frame (k_pe_bwd + k_deformation_bwd of the per-object path).
struct PostBenderJob {
    RowCtx r;
    const float* bin; const float* g_bin; int ld, octaves, benc, D;
    float size[3];
    float* g_x;                                 // (cap,3) accumulated
    float* d_def; int def_stride;               // d deformation of this object, or NULL
};
struct PostBenderJobs { PostBenderJob job[MAX_ROW_JOBS]; };

__global__ __launch_bounds__(256) void k_post_bender_group(PostBenderJobs jobs) {
    extern __shared__ float post_smem[];
    __shared__ int sh_frame[2];
    const PostBenderJob& p = jobs.job[blockIdx.y];
    const int M = *p.r.total;
    const int m0 = blockIdx.x * POST_ROWS;
    if (m0 >= M) return;
    const int rows = M - m0 < POST_ROWS ? M - m0 : POST_ROWS;
    float* se = post_smem;
    float* sg = post_smem + POST_ROWS * (p.ld + 1);
    stage_rows(se, p.bin, p.ld, m0, rows);
    stage_rows(sg, p.g_bin, p.ld, m0, rows);
    if (threadIdx.x == 0) {
        sh_frame[0] = p.r.rec_flat[m0] / p.r.samples_per_frame;
        sh_frame[1] = p.r.rec_flat[m0 + rows - 1] / p.r.samples_per_frame;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < rows) {
        const int m = m0 + t;
        if (p.r.row_flags[m] & 1) {
            const float* e = se + t * (p.ld + 1);
            const float* g = sg + t * (p.ld + 1);
            for (int a = 0; a < 3; ++a) {
                float acc = g[a];
                for (int k = 0; k < p.octaves; ++k) {
                    const int sn = 3 + k * 6 + a, cs = sn + 3;
                    acc += ldexpf(1.0f, k) * (e[cs] * g[sn] - e[sn] * g[cs]);
                }
                p.g_x[(size_t)m * 3 + a] += acc / p.size[a];
            }
        }
    } else if (p.d_def && t >= 64 && t - 64 < p.D) {
        // d deformation[frame][j] += sum over the frame's rows of g_bin[m][benc + j]: wave 1 walks the tile's column j
        const int j = t - 64;
        const bool uniform = sh_frame[0] == sh_frame[1];
        float acc = 0.f;
        for (int row = 0; row < rows; ++row) {
            if (!(p.r.row_flags[m0 + row] & 1)) continue;
            const float gv = sg[row * (p.ld + 1) + p.benc + j];
            if (uniform) acc += gv;
            else atomicAdd(p.d_def + (size_t)(p.r.rec_flat[m0 + row] / p.r.samples_per_frame) * p.def_stride + j, gv);
        }
        if (uniform && acc != 0.f) atomicAdd(p.d_def + (size_t)sh_frame[0] * p.def_stride + j, acc);
    }
}

struct StyleBwdJob {
    int width, S;
    const float* dscale; const float* dbias;    // (frames, MAX_WIDTH)
    const float* style; int style_stride;
    const float* A; float* dA; float* db; float* d_style;
};
struct StyleBwdJobs { StyleBwdJob job[2 * PR_MAX_OBJECTS]; int frames; };

// k_style_bwd over (frames, MAX_WIDTH) tables for every AdaIN layer of every object (blockIdx.y = job)
__global__ __launch_bounds__(256) void k_style_bwd_group(StyleBwdJobs jobs) {
    __shared__ float sh[256];
    const StyleBwdJob& p = jobs.job[blockIdx.y];
    const int frames = jobs.frames, width = p.width, S = p.S;
    const int rows = 2 * width;
    if ((int)blockIdx.x < frames) {
        const int f = blockIdx.x;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + (threadIdx.x & 63), part = threadIdx.x >> 6;
            float acc = 0.f;
            if (s < S)
                for (int rr = part; rr < rows; rr += 4) {
                    const float dv = rr < width ? p.dscale[(size_t)f * MAX_WIDTH + rr] : p.dbias[(size_t)f * MAX_WIDTH + rr - width];
                    acc = fmaf(p.A[(size_t)rr * S + s], dv, acc);
                }
            sh[threadIdx.x] = acc;
            __syncthreads();
            // (instances that share a model write different d_style rows; the shared dA / db are accumulated with atomics below)
            if (part == 0 && s < S && p.d_style)      // (both AdaIN layers of an object add to its style row)
                atomicAdd(p.d_style + (size_t)f * p.style_stride + s, sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
            __syncthreads();
        }
        return;
    }
    const long idx = (long)(blockIdx.x - frames) * 256 + threadIdx.x;
    if (idx < (long)rows * S) {
        const int rr = (int)(idx / S), s = (int)(idx - (long)rr * S);
        const float* src = rr < width ? p.dscale + rr : p.dbias + (rr - width);
        float acc = 0.f;
        for (int f = 0; f < frames; ++f) acc = fmaf(src[(size_t)f * MAX_WIDTH], p.style[(size_t)f * p.style_stride + s], acc);
        if (p.dA) atomicAdd(p.dA + idx, acc);
        if (s == 0 && p.db) {
            float b = 0.f;
            for (int f = 0; f < frames; ++f) b += src[(size_t)f * MAX_WIDTH];
            atomicAdd(p.db + rr, b);
        }
    }
}

struct GeometryBwdJobs { GeometryBwd job[PR_MAX_OBJECTS]; };
__device__ __forceinline__ void geometry_bwd_body(const GeometryBwd& p);
__global__ __launch_bounds__(256) void k_geometry_bwd_group(GeometryBwdJobs jobs) { geometry_bwd_body(jobs.job[blockIdx.z]); }

}  // namespace pr

// ---------------------------------------------------------------------------------------------
// Orchestration
// ---------------------------------------------------------------------------------------------
namespace pr {

// Scratch of the grouped backward pass (backward_grouped): every object keeps its own buffers, the objects share the launches.
struct GroupPlan {
    size_t a2[PR_MAX_OBJECTS], d2[PR_MAX_OBJECTS], a1[PR_MAX_OBJECTS], d1[PR_MAX_OBJECTS];   // head layers: activations / gradients
    size_t gstack[PR_MAX_OBJECTS], g_enc[PR_MAX_OBJECTS], gsr4[PR_MAX_OBJECTS];
    size_t g_x[PR_MAX_OBJECTS], g_braw4[PR_MAX_OBJECTS], g_in6[PR_MAX_OBJECTS];
    size_t bgstack[PR_MAX_OBJECTS], g_benc[PR_MAX_OBJECTS];
    size_t zero_begin, zero_bytes;           // one fill: batch sums, AdaIN gradient tables, tile / claim counters
    size_t sums[PR_MAX_OBJECTS];             // doubles: [layer 4: sum | sum sq] [layer 1: sum | sum sq], 2 x MAX_WIDTH each
    size_t tables[PR_MAX_OBJECTS];           // frames x 4 x MAX_WIDTH floats: dscale1 | dbias1 | dscale2 | dbias2
    size_t counters;                         // ints: 4 per object, then the 8 claim counters of the weight-gradient launch
    size_t tn_partial, tn_partial_floats;    // partial tiles of the weight-gradient launch
    size_t end;                              // first byte behind the grouped scratch
    bool usable;
};
constexpr size_t GROUP_SCRATCH_LIMIT = (size_t)48 << 30;   // larger calls take the per-object path

struct BwdPlan {
    GroupPlan group;
    size_t front_bytes;               // per-sample gradients of the compositing backward: used by both paths
    size_t g_feat[PR_MAX_OBJECTS], g_sigma[PR_MAX_OBJECTS], g_t[PR_MAX_OBJECTS], g_dm[PR_MAX_OBJECTS];
    size_t bufA, bufB, act, g_enc, gsr, gdr, g_bent, g_x, g_braw, g_in6, partial, sums, tables;
    size_t gstack, chain_packed;      // layer-chained backward: per-layer pre-activation gradients, W^T fragments
    size_t gstack_bytes;              // 0: the chained path is off for this call (too large), layer-by-layer products instead
    size_t g_norm, g_norm_bytes;      // (N,R) d loss / d |d| (ray gradients)
    size_t g_div[PR_MAX_OBJECTS];     // PR_FLAG_DIVERGENCE_GRAD: d loss / d Hutchinson estimate (N,R,P) of the bender objects
    size_t div_t0, div_stack;         //   probe tangents of the bender input and of every bender layer's output
    size_t bytes;
    size_t max_cap;
    int lanes;                        // 2: the per-object scratch exists twice (lane_stride apart): two objects at a time
    size_t lane_stride;
};
constexpr size_t LANE_SCRATCH_LIMIT = (size_t)8 << 30;     // calls whose per-object scratch is larger keep one lane
constexpr size_t CHAIN_GSTACK_LIMIT = (size_t)16 << 30;   // calls whose chains would need more scratch go layer by layer

static size_t align_up_b(size_t v) { return (v + 255) & ~(size_t)255; }

static int make_bwd_plan(const pr_call_t& c, const pr_object_t* objs, BwdPlan* bp) {
    memset(bp, 0, sizeof(*bp));
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up_b(bytes);
        return at;
    };
    const size_t nr = (size_t)c.frames * c.rays;
    size_t max_cap = 0;
    for (int k = 0; k < c.objects; ++k) {
        const pr_object_model_t& m = c.use_fine ? objs[k].fine : objs[k].coarse;   // the fine pass has more positions
        const size_t cap = nr * m.positions;
        if (cap > max_cap) max_cap = cap;
        bp->g_feat[k] = take(sizeof(float) * cap * ((m.output_features + 15) & ~15));
        bp->g_sigma[k] = take(sizeof(float) * cap);
        bp->g_t[k] = take(sizeof(float) * cap);
        bp->g_dm[k] = take(sizeof(float) * cap);
    }
    bp->max_cap = max_cap;
    bp->g_norm_bytes = sizeof(float) * nr;
    bp->g_norm = take(bp->g_norm_bytes);
    if (c.flags & PR_FLAG_DIVERGENCE_GRAD)
        for (int k = 0; k < c.objects; ++k)
            if (objs[k].coarse.has_bender || (c.use_fine && objs[k].fine.has_bender))
                bp->g_div[k] = take(sizeof(float) * nr * std::max(objs[k].coarse.positions, c.use_fine ? objs[k].fine.positions : 0));
    // ---- everything below is scratch of ONE object's backward pass: a second copy lets two objects run side by side
    const size_t lane_begin = off;
    bp->front_bytes = off;
    bp->bufA = take(sizeof(float) * max_cap * MAX_WIDTH);
    bp->bufB = take(sizeof(float) * max_cap * MAX_WIDTH);
    bp->act = take(sizeof(float) * max_cap * MAX_WIDTH);
    bp->g_enc = take(sizeof(float) * max_cap * MAX_ENC);
    bp->gsr = take(sizeof(float) * max_cap);
    bp->gdr = take(sizeof(float) * max_cap);
    bp->g_bent = take(sizeof(float) * max_cap * 3);
    bp->g_x = take(sizeof(float) * max_cap * 3);
    bp->g_braw = take(sizeof(float) * max_cap * 3);
    bp->g_in6 = take(sizeof(float) * max_cap * 6);
    // layer-chained backward (k_chain_bwd): scratch for the largest chain of the call, one partial region per grouped product
    size_t gstack_need = 0, packed_need = 0;
    for (int k = 0; k < c.objects; ++k)
        for (int t = 0; t < (c.use_fine ? 2 : 1); ++t) {
            const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
            ModelDims d;
            PR_TRY(compute_dims(m, &d));
            const size_t cap = nr * m.positions;
            gstack_need = std::max(gstack_need, sizeof(float) * cap * d.Wpad * (size_t)(m.backbone_count - 1));
            packed_need = std::max(packed_need, chain_bwd_packed_bytes(m.backbone_count, d.W, d.enc));
            if (m.has_bender) {
                gstack_need = std::max(gstack_need, sizeof(float) * cap * d.BWpad * (size_t)(m.bender_count - 1));
                packed_need = std::max(packed_need, chain_bwd_packed_bytes(m.bender_count, d.BW, d.bin));
            }
        }
    if (c.flags & PR_FLAG_DIVERGENCE_GRAD) {
        size_t t0_need = 0, stack_need = 0;
        for (int k = 0; k < c.objects; ++k)
            for (int t = 0; t < (c.use_fine ? 2 : 1); ++t) {
                const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
                if (!m.has_bender) continue;
                ModelDims d;
                PR_TRY(compute_dims(m, &d));
                const size_t cap = nr * m.positions;
                t0_need = std::max(t0_need, sizeof(float) * cap * d.bin_pad);
                stack_need = std::max(stack_need, sizeof(float) * cap * d.BWpad * (size_t)m.bender_count);
            }
        bp->div_t0 = take(t0_need);
        bp->div_stack = take(stack_need);
    }
#ifdef PR_BWD_LAYERWISE
    gstack_need = CHAIN_GSTACK_LIMIT + 1;     // measurement build: one product per layer and launch
#endif
    const bool chained = gstack_need <= CHAIN_GSTACK_LIMIT;
    bp->gstack_bytes = chained ? gstack_need : 0;
    if (chained) {
        bp->gstack = take(gstack_need);
        bp->chain_packed = take(packed_need);
    }
    bp->partial = take(sizeof(float) * gemm_tn_scratch_floats(BWD_SPLITS) * (chained ? MAX_TN_GROUP : 1));
    bp->sums = take(sizeof(double) * 2 * MAX_WIDTH);
    bp->tables = take(sizeof(float) * (size_t)c.frames * 4 * MAX_WIDTH);
    const size_t lane_bytes = off - lane_begin;
    bp->lanes = 1;
#ifndef PR_BWD_ONE_LANE
    if (c.objects > 1 && lane_bytes <= LANE_SCRATCH_LIMIT) {
        bp->lanes = 2;
        bp->lane_stride = lane_bytes;
        off += lane_bytes;
    }
#endif
    bp->bytes = off;
    // ---- grouped path: the same front, then per-object buffers for every object at once
    {
        GroupPlan& gp = bp->group;
        size_t goff = lane_begin;
        auto gtake = [&](size_t bytes) {
            const size_t at = goff;
            goff += align_up_b(bytes);
            return at;
        };
        size_t tn_floats = 0;
        for (int k = 0; k < c.objects; ++k) {
            size_t cap = 0;
            int Wp = 0, W2p = 0, encp = 0, nb = 0, BWp = 0, binp = 0, bc = 0;
            size_t tn_k = 0;
            for (int t = 0; t < (c.use_fine ? 2 : 1); ++t) {      // the two model types run one after the other on the same scratch
                const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
                ModelDims d;
                PR_TRY(compute_dims(m, &d));
                const size_t cap_t = nr * m.positions;
                cap = std::max(cap, cap_t);
                Wp = std::max(Wp, d.Wpad); W2p = std::max(W2p, d.W2pad); encp = std::max(encp, d.enc_pad); nb = std::max(nb, m.backbone_count);
                if (m.has_bender) { BWp = std::max(BWp, d.BWpad); binp = std::max(binp, d.bin_pad); bc = std::max(bc, m.bender_count); }
                // partial tiles of every weight-gradient product of the model
                size_t f = 0;
                const long rows = (long)cap_t;
                f += tn_all_partial_floats(d.F, d.W2, rows) + tn_all_partial_floats(d.W2, d.W, rows) + tn_all_partial_floats(d.W, d.W, rows) +
                     tn_all_partial_floats(1, d.W, rows);
                f += tn_all_partial_floats(d.W, d.enc, rows) * 2 + tn_all_partial_floats(d.W, d.W, rows) * (size_t)(m.backbone_count - 1);
                if (m.has_bender)
                    f += tn_all_partial_floats(3, d.BW, rows) + tn_all_partial_floats(d.BW, d.bin, rows) * 2 +
                         tn_all_partial_floats(d.BW, d.BW, rows) * (size_t)(m.bender_count - 1);
                tn_k = std::max(tn_k, f);
            }
            tn_floats += tn_k;
            gp.a2[k] = gtake(sizeof(float) * cap * W2p);
            gp.d2[k] = gtake(sizeof(float) * cap * W2p);
            gp.a1[k] = gtake(sizeof(float) * cap * Wp);
            gp.d1[k] = gtake(sizeof(float) * cap * Wp);
            gp.gstack[k] = gtake(sizeof(float) * cap * Wp * (size_t)nb);
            gp.g_enc[k] = gtake(sizeof(float) * cap * encp);
            gp.gsr4[k] = gtake(sizeof(float) * cap * 4);
            gp.g_x[k] = gtake(sizeof(float) * cap * 3);
            gp.g_in6[k] = gtake(sizeof(float) * cap * 6);
            if (bc) {
                gp.g_braw4[k] = gtake(sizeof(float) * cap * 4);
                gp.bgstack[k] = gtake(sizeof(float) * cap * BWp * (size_t)bc);
                gp.g_benc[k] = gtake(sizeof(float) * cap * binp);
            }
        }
        gp.tn_partial_floats = tn_floats;
        gp.tn_partial = gtake(sizeof(float) * tn_floats);
        gp.zero_begin = goff;
        for (int k = 0; k < c.objects; ++k) {
            gp.sums[k] = gtake(sizeof(double) * 4 * MAX_WIDTH);
            gp.tables[k] = gtake(sizeof(float) * (size_t)c.frames * 4 * MAX_WIDTH);
        }
        gp.counters = gtake(sizeof(int32_t) * (4 * PR_MAX_OBJECTS + 8 * 4));
        gp.zero_bytes = goff - gp.zero_begin;
        gp.end = goff;
        gp.usable = !(c.flags & PR_FLAG_DIVERGENCE_GRAD) && (goff - lane_begin) <= GROUP_SCRATCH_LIMIT;
#ifdef PR_BWD_PER_OBJECT
        gp.usable = false;          // measurement build: the per-object path for every call
#endif
        if (gp.usable) bp->bytes = goff;     // (the per-object scratch is not needed)
    }
    return PR_OK;
}

// Second stream of the backward pass (one per device AND caller stream, created on first use, never destroyed): the per-object
// parts of different objects are independent between the compositing backward and the end of the call; most of their ~50 launches
// per object are small, so two objects side by side fill the GPU better than one after the other.  The caller's stream forks
// into it after the compositing backward and joins it before the call returns - for the caller everything is still
// enqueued on its own stream.  Keyed by the caller's stream: two host threads that run backward passes on one device do so on
// different streams (torch's autograd threads inherit the forward pass's stream), and one of them may be RECORDING its stream
// into a HIP graph (capture_error_mode = thread_local) - a second stream shared per device would be pulled into that capture by
// its fork and then receive the other thread's eager launches.
static int lane_stream(hipStream_t caller, hipStream_t* out) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, hipStream_t> streams;
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t& slot = streams[std::make_pair(dev, caller)];
    if (!slot) PR_CHECK_HIP(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
    *out = slot;
    return PR_OK;
}

// `waiter` continues after everything enqueued on `on` so far.  The events come from a small per-thread, per-device ring (a wait
// holds what the event had recorded when hipStreamWaitEvent was called; recording it again later does not disturb that wait), so a
// backward pass creates no events after its first calls.
static int stream_wait(hipStream_t waiter, hipStream_t on) {
    constexpr int RING = 8;
    struct Ring { hipEvent_t e[RING] = {}; int next = 0; };
    static thread_local std::map<int, Ring> rings;
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    Ring& ring = rings[dev];
    hipEvent_t& e = ring.e[ring.next];
    ring.next = (ring.next + 1) % RING;
    if (!e) PR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    PR_CHECK_HIP(hipEventRecord(e, on));
    PR_CHECK_HIP(hipStreamWaitEvent(waiter, e, 0));
    return PR_OK;
}

struct GemmCtx {
    const int32_t* rows;
    int max_rows;
    float* partial;
    hipStream_t s;
    float* gstack;          // layer-chained path (NULL: layer by layer)
    float* chain_packed;
    size_t cap;             // row capacity of the per-layer buffers
    // chain_backward with other left factors for the weight gradients than the saved activations (the probe tangents of
    // the divergence estimate): dW_l += dY_l^T X_l with X_0 = dw_in0, X_l = dw_acts + (l - 1) dw_stride; no bias gradients
    const float* dw_acts;
    size_t dw_stride;
    const float* dw_in0;
};

static int weight_grad(const GemmCtx& g, const float* dY, int ldy, int n_out, const float* X, int ldx, int n_in, float* dW,
                       int ldw, float* dbias) {
    if (!dW && !dbias) return PR_OK;
    GemmTN p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.lda = ldy; p.B = X; p.ldb = ldx;
    p.rows = g.rows; p.ni = n_out; p.nj = n_in; p.splits = BWD_SPLITS;
    p.partial = g.partial;
    p.bias_partial = dbias ? g.partial + (size_t)BWD_SPLITS * 256 * 384 : nullptr;
    p.bias = dbias;
    PR_REQUIRE(dW != nullptr, "a bias gradient buffer needs its weight gradient buffer");
    p.C = dW; p.ldc = ldw;
    return launch_gemm_tn(p, g.s);
}

static int input_grad(const GemmCtx& g, const float* dY, int ldy, int n_out, const float* W, int ldw, int n_in, float* dX,
                      int ldx, bool accumulate, const float* mask, int ldm) {
    GemmNN p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.lda = ldy; p.B = W; p.ldb = ldw; p.C = dX; p.ldc = ldx;
    p.rows = g.rows; p.n = n_in; p.k = n_out; p.accumulate = accumulate ? 1 : 0;
    if (n_out & 15) {      // the product walks K in 16-deep slabs: the tail reads padding columns of dY as zeros
        p.k = (n_out + 15) & ~15;
        p.k_valid = n_out;
        PR_REQUIRE(ldy >= p.k, "input gradient: %d output columns in rows of %d floats", n_out, ldy);
    }
    p.mask = mask; p.ldm = ldm;
    return launch_gemm_nn(p, g.max_rows, g.s);
}

// Layer-chained variant of chain_backward: ONE launch for every input-gradient product of the chain (k_chain_bwd keeps
// the running gradient in LDS), then ONE grouped launch (+ its reduction) for every weight / bias gradient.
static int chain_backward_fused(const GemmCtx& g, const pr_linear_t* layers, const pr_linear_grad_t* grads, int count, int skip,
                                int width, int width_pad, const float* acts, size_t act_stride, const float* in0, int ld_in0,
                                int n_in0, const float* cur, float* g_in, const unsigned char* bits) {
    ChainBwdParams cp;
    memset(&cp, 0, sizeof(cp));
    PR_TRY(prepare_chain_bwd(layers, count, skip, width, n_in0, g.chain_packed, &cp, g.s));
    PR_REQUIRE(cp.Wpad == width_pad && ld_in0 >= cp.in_pad, "backward chain: padded widths %d / %d do not match the saved rows", cp.Wpad, cp.in_pad);
    cp.total = g.rows;
    cp.g_last = cur;
    cp.acts = acts; cp.act_stride = act_stride;
    (void)bits;      // (the forward pass's bit images are column words for the grouped chain; this path builds its masks from the activations)
    cp.bits = nullptr; cp.bits_stride = 0;
    cp.gstack = g.gstack; cp.g_stride = g.cap * (size_t)width_pad;
    cp.g_in = g_in; cp.ld_in = ld_in0;
    PR_TRY(launch_chain_bwd(cp, g.max_rows, g.s));
    GemmTNGroup grp;
    memset(&grp, 0, sizeof(grp));
    const size_t region = gemm_tn_scratch_floats(BWD_SPLITS);
    auto add = [&](const float* dY, const float* X, int ldx, int n_in, float* dW, int ldw, float* dbias) -> int {
        if (!dW) return PR_OK;
        PR_REQUIRE(grp.count < MAX_TN_GROUP, "backward chain: too many weight-gradient products");
        GemmTN& p = grp.job[grp.count];
        p.A = dY; p.lda = width_pad; p.B = X; p.ldb = ldx;
        p.rows = g.rows; p.ni = width; p.nj = n_in; p.splits = BWD_SPLITS;
        p.partial = g.partial + (size_t)grp.count * region;
        p.bias_partial = dbias ? p.partial + (size_t)BWD_SPLITS * 256 * 384 : nullptr;
        p.bias = dbias;
        p.C = dW; p.ldc = ldw;
        ++grp.count;
        return PR_OK;
    };
    const bool sub = g.dw_acts != nullptr;
    const float* x_acts = sub ? g.dw_acts : acts;
    const size_t x_stride = sub ? g.dw_stride : act_stride;
    const float* x_in0 = sub ? g.dw_in0 : in0;
    for (int l = count - 1; l >= 0; --l) {
        const float* dY = (l == count - 1) ? cur : g.gstack + (size_t)l * cp.g_stride;
        PR_REQUIRE(!grads[l].bias || grads[l].weight, "a bias gradient buffer needs its weight gradient buffer");
        float* dbias = sub ? nullptr : grads[l].bias;
        if (l == 0) {
            PR_TRY(add(dY, x_in0, ld_in0, n_in0, grads[l].weight, layers[l].in_features, dbias));
        } else {
            PR_TRY(add(dY, x_acts + (size_t)(l - 1) * x_stride, width_pad, width, grads[l].weight, layers[l].in_features, dbias));
            if (l == skip)
                PR_TRY(add(dY, x_in0, ld_in0, n_in0, grads[l].weight ? grads[l].weight + width : nullptr, layers[l].in_features, nullptr));
        }
    }
    if (grp.count) PR_TRY(launch_gemm_tn_group(grp, g.s));
    return PR_OK;
}

// Backward through a ReLU MLP with one skip concatenation (backbone of the NeRF or of the ray bender).
// On entry `cur` holds d loss / d pre-activation of the last layer; `acts` are the saved post-ReLU outputs.
// Returns with the gradient of the network input [PE | extra] accumulated in g_in (n_in0 real columns).
static int chain_backward(const GemmCtx& g, const pr_linear_t* layers, const pr_linear_grad_t* grads, int count, int skip,
                          int width, int width_pad, const float* acts, size_t act_stride, const float* in0, int ld_in0,
                          int n_in0, float* cur, float* other, float* g_in, const unsigned char* bits) {
    if (g.gstack) return chain_backward_fused(g, layers, grads, count, skip, width, width_pad, acts, act_stride, in0, ld_in0, n_in0, cur, g_in, bits);
    bool g_in_written = false;
    const bool sub = g.dw_acts != nullptr;
    const float* x_in0 = sub ? g.dw_in0 : in0;
    for (int l = count - 1; l >= 0; --l) {
        const pr_linear_t& L = layers[l];
        const float* prev = l > 0 ? acts + (size_t)(l - 1) * act_stride : nullptr;
        const float* x_prev = (sub && l > 0) ? g.dw_acts + (size_t)(l - 1) * g.dw_stride : prev;
        float* dbias = sub ? nullptr : grads[l].bias;
        if (l == 0) {
            PR_TRY(weight_grad(g, cur, width_pad, width, x_in0, ld_in0, n_in0, grads[l].weight, L.in_features, dbias));
            PR_TRY(input_grad(g, cur, width_pad, width, L.weight, L.in_features, n_in0, g_in, ld_in0, g_in_written, nullptr, 0));
        } else {
            PR_TRY(weight_grad(g, cur, width_pad, width, x_prev, width_pad, width, grads[l].weight, L.in_features, dbias));
            if (l == skip) {
                PR_TRY(weight_grad(g, cur, width_pad, width, x_in0, ld_in0, n_in0,
                                   grads[l].weight ? grads[l].weight + width : nullptr, L.in_features, nullptr));
                PR_TRY(input_grad(g, cur, width_pad, width, L.weight + width, L.in_features, n_in0, g_in, ld_in0, false, nullptr, 0));
                g_in_written = true;
            }
            PR_TRY(input_grad(g, cur, width_pad, width, L.weight, L.in_features, width, other, width_pad, false, prev, width_pad));
            float* tmp = cur;
            cur = other;
            other = tmp;
        }
    }
    return PR_OK;
}

// compositing backward of one model type: per-sample gradients (feature rows, sigma, t, |delta|, divergence) of every object
struct CompositeBwdResult { bool div_grad, ray_grads; float* g_norm; };
static int composite_backward(const pr_call_t& c, const pr_object_t* objs, int t, const pr_output_grads_t& grads, const pr_input_grads_t& out,
                              char* fws, const Plan& plan, char* bws, const BwdPlan& bp, hipStream_t s, CompositeBwdResult* res) {
    const int K = c.objects;
    const TypePlan& tp = plan.type[t];
    const pr_noise_t& noise = t ? c.noise_fine : c.noise_coarse;
    const int F = objs[0].coarse.output_features;
    const int Fs = (F + 15) & ~15;     // row stride of the feature gradients

    // gradients of integrated_divergence flow only where the forward pass estimated a divergence: differentiable calls in
    // training mode (render.hip), objects with a ray bender, probes explicit or generated
    bool div_grad = (c.flags & PR_FLAG_DIVERGENCE_GRAD) && (c.flags & PR_FLAG_TRAIN_BN);
    if (div_grad) {
        bool any = grads.global.integrated_divergence != nullptr;
        for (int k = 0; k < K; ++k) any |= grads.object[k].integrated_divergence != nullptr;
        div_grad = any;
    }

    CompositeBwdParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.div_grad = div_grad ? 1 : 0;
    cp.sigmoid = (c.flags & PR_FLAG_SIGMOID_FEATURES) ? 1 : 0;
    cp.Fs = Fs;
    cp.frames = c.frames; cp.rays = c.rays; cp.objects = K; cp.static_objects = c.static_objects; cp.F = F;
    cp.fix_overlaps = (c.flags & PR_FLAG_FIX_OVERLAPS) ? 1 : 0;
    int total_positions = 0;
    for (int k = 0; k < K; ++k) {
        const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
        CompositeBwdObject& o = cp.obj[k];
        o.t = reinterpret_cast<const float*>(fws + tp.t[k]);
        o.sigma = reinterpret_cast<const float*>(fws + tp.sigma[k]);
        o.slot = reinterpret_cast<const int32_t*>(fws + tp.slot[k]);
        o.dispmag = m.has_bender ? reinterpret_cast<const float*>(fws + tp.dispmag[k]) : nullptr;
        o.feat = reinterpret_cast<const float*>(fws + tp.feat[k]);
        o.noise = perturb_noise(noise.integrate[k], c, NOISE_INTEGRATE, t, k);
        o.positions = m.positions;
        o.g = grads.object[k];
        o.g_feat = reinterpret_cast<float*>(bws + bp.g_feat[k]);
        o.g_sigma = reinterpret_cast<float*>(bws + bp.g_sigma[k]);
        o.g_t = reinterpret_cast<float*>(bws + bp.g_t[k]);
        o.g_dm = m.has_bender ? reinterpret_cast<float*>(bws + bp.g_dm[k]) : nullptr;
        o.g_sample_t = grads.sample_t[k];
        if (div_grad && m.has_bender) {
            o.divergence = reinterpret_cast<const float*>(fws + tp.saved[k].div);
            o.g_div = reinterpret_cast<float*>(bws + bp.g_div[k]);
        }
        total_positions += m.positions;
    }
    cp.total_positions = total_positions;
    int ss = 64;
    while (ss < total_positions) ss <<= 1;
    cp.sort_size = ss;
    cp.ray_directions = c.ray_directions;
    cp.noise_global = perturb_noise(noise.integrate_global, c, NOISE_INTEGRATE_GLOBAL, t, 0);
    cp.global = grads.global;
    res->div_grad = div_grad;
    res->ray_grads = out.ray_origins != nullptr || out.ray_directions != nullptr;
    res->g_norm = (out.ray_directions && bp.g_norm_bytes) ? reinterpret_cast<float*>(bws + bp.g_norm) : nullptr;
    cp.g_norm = res->g_norm;
    return launch_composite_bwd(cp, s);
}

// backward of one model type (t = 0: coarse models / results["coarse"], 1: fine), one object after the other on two lanes
static int backward(const pr_call_t& c, const pr_object_t* objs, int t, const pr_output_grads_t& grads, const pr_input_grads_t& out,
                    char* fws, const Plan& plan, char* bws, const BwdPlan& bp, hipStream_t s_caller) {
    const int K = c.objects;
    const TypePlan& tp = plan.type[t];
    const pr_noise_t& noise = t ? c.noise_fine : c.noise_coarse;
    int32_t* totals = reinterpret_cast<int32_t*>(fws + tp.totals);
    const int F = objs[0].coarse.output_features;
    const int Fs = (F + 15) & ~15;     // row stride of the feature gradients
    CompositeBwdResult cr;
    PR_TRY(composite_backward(c, objs, t, grads, out, fws, plan, bws, bp, s_caller, &cr));
    const bool div_grad = cr.div_grad, ray_grads = cr.ray_grads;
    float* g_norm = cr.g_norm;

    // ---- per object -------------------------------------------------------------------------------
    // Lanes: objects that share a model (the same gradient buffers are accumulated into) stay on one lane, in order; the
    // groups go to the lane with the smaller load so far (load = sample capacity, largest group first).
    int lane_of[PR_MAX_OBJECTS] = {};
    hipStream_t aux = nullptr;
    if (bp.lanes == 2) {
        int group_of[PR_MAX_OBJECTS];
        size_t group_cost[PR_MAX_OBJECTS] = {};
        int groups = 0;
        for (int k = 0; k < K; ++k) {
            // instances of one model (the same parameter storages) accumulate into the same gradient buffers - whichever of
            // them are trainable - with plain read-modify-writes: they stay on one lane
            const pr_object_model_t& Mk = t ? objs[k].fine : objs[k].coarse;
            group_of[k] = -1;
            for (int q = 0; q < k && group_of[k] < 0; ++q) {
                const pr_object_model_t& Mq = t ? objs[q].fine : objs[q].coarse;
                if (Mk.backbone[0].weight == Mq.backbone[0].weight) group_of[k] = group_of[q];
            }
            if (group_of[k] < 0) group_of[k] = groups++;
            group_cost[group_of[k]] += (size_t)c.frames * c.rays * (t ? objs[k].fine : objs[k].coarse).positions;
        }
        size_t load[2] = {0, 0};
        int group_lane[PR_MAX_OBJECTS];
        bool done[PR_MAX_OBJECTS] = {};
        for (int n = 0; n < groups; ++n) {
            int best = -1;
            for (int g = 0; g < groups; ++g)
                if (!done[g] && (best < 0 || group_cost[g] > group_cost[best])) best = g;
            done[best] = true;
            group_lane[best] = load[1] < load[0] ? 1 : 0;
            load[group_lane[best]] += group_cost[best];
        }
        bool any = false;
        for (int k = 0; k < K; ++k) {
            lane_of[k] = group_lane[group_of[k]];
            any |= lane_of[k] == 1;
        }
        if (any) {
            PR_TRY(lane_stream(s_caller, &aux));
            PR_TRY(stream_wait(aux, s_caller));          // fork: the second lane starts after the compositing backward
        }
    }

    // (a lambda: whatever happens inside - a failed launch, an unsupported configuration - the second lane is joined back
    // into the caller's stream before the call returns)
    auto run_objects = [&]() -> int {
    for (int k = 0; k < K; ++k) {
        const int lane = aux ? lane_of[k] : 0;
        hipStream_t s = lane ? aux : s_caller;           // (shadows the caller's stream for this object's launches)
        char* lws = bws + (size_t)lane * bp.lane_stride;
        float* bufA = reinterpret_cast<float*>(lws + bp.bufA);
        float* bufB = reinterpret_cast<float*>(lws + bp.bufB);
        float* actb = reinterpret_cast<float*>(lws + bp.act);
        float* g_enc = reinterpret_cast<float*>(lws + bp.g_enc);
        float* gsr = reinterpret_cast<float*>(lws + bp.gsr);
        float* gdr = reinterpret_cast<float*>(lws + bp.gdr);
        float* g_bent = reinterpret_cast<float*>(lws + bp.g_bent);
        float* g_x = reinterpret_cast<float*>(lws + bp.g_x);
        float* g_braw = reinterpret_cast<float*>(lws + bp.g_braw);
        float* g_in6 = reinterpret_cast<float*>(lws + bp.g_in6);
        double* sums = reinterpret_cast<double*>(lws + bp.sums);
        float* tables = reinterpret_cast<float*>(lws + bp.tables);
        const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
        const pr_model_grads_t& G = t ? out.model_fine[k] : out.model[k];
        const SavedPlan& sv = tp.saved[k];
        ModelDims d;
        PR_TRY(compute_dims(m, &d));
        const int P = m.positions;
        const size_t cap = (size_t)c.frames * c.rays * P;
        const int row_blocks = (int)((cap + 255) / 256);
        const int grid_rows = cap < 4096 ? (int)cap : 4096;
        const int grid_blk = row_blocks < 1024 ? row_blocks : 1024;
        RowCtx rc;
        rc.total = totals + k;
        rc.rec_flat = reinterpret_cast<const int32_t*>(fws + sv.rec_flat);
        rc.row_flags = reinterpret_cast<const int32_t*>(fws + sv.row_flags);
        rc.samples_per_frame = c.rays * P;
        rc.in_scene = c.object_in_scene + k; rc.in_scene_stride = K;
        GemmCtx gc;
        gc.rows = totals + k; gc.max_rows = (int)cap; gc.partial = reinterpret_cast<float*>(lws + bp.partial); gc.s = s;
        gc.gstack = bp.gstack_bytes ? reinterpret_cast<float*>(lws + bp.gstack) : nullptr;
        gc.chain_packed = bp.gstack_bytes ? reinterpret_cast<float*>(lws + bp.chain_packed) : nullptr;
        gc.cap = cap;
        gc.dw_acts = nullptr; gc.dw_stride = 0; gc.dw_in0 = nullptr;

        const float* rec_pos = reinterpret_cast<const float*>(fws + sv.rec_pos);
        const float* enc = reinterpret_cast<const float*>(fws + sv.enc);
        const float* acts = reinterpret_cast<const float*>(fws + sv.act);
        const size_t act_stride = cap * d.Wpad;
        const float* h1 = reinterpret_cast<const float*>(fws + sv.h1);
        const float* h2 = reinterpret_cast<const float*>(fws + sv.h2);
        const float* batch = reinterpret_cast<const float*>(fws + sv.batch);
        const int32_t* stat_count = reinterpret_cast<const int32_t*>(fws + sv.stat_count);
        const float* table = reinterpret_cast<const float*>(fws + tp.adain[k]);
        const int table_stride = adain_row_floats(d);
        float* g_feat = reinterpret_cast<float*>(bws + bp.g_feat[k]);
        const float* g_sigma = reinterpret_cast<const float*>(bws + bp.g_sigma[k]);
        const float* g_t = reinterpret_cast<const float*>(bws + bp.g_t[k]);
        const float* g_dm = m.has_bender ? reinterpret_cast<const float*>(bws + bp.g_dm[k]) : nullptr;
        float lo[3], hi[3], size[3];
        bbox_split(m, lo, hi, size);

        hipLaunchKernelGGL(k_gather_rows, dim3(grid_rows), dim3(256), 0, s, rc, g_sigma, g_dm, gsr, m.has_bender ? gdr : nullptr,
                           g_feat, Fs);
        PR_LAUNCH_CHECK();

        // ---- feature head, layer 6 -------------------------------------------------------------
        hipLaunchKernelGGL(k_adain_recompute, dim3(grid_rows), dim3(256), 0, s, rc, h2, d.W2pad, d.W2pad, table, table_stride,
                           2 * d.Wpad, 2 * d.Wpad + d.W2pad, actb);
        PR_LAUNCH_CHECK();
        PR_TRY(weight_grad(gc, g_feat, Fs, F, actb, d.W2pad, d.W2, G.head6.weight, d.W2, G.head6.bias));
        PR_TRY(input_grad(gc, g_feat, Fs, F, m.head6.weight, d.W2, d.W2, bufB, d.W2pad, false, nullptr, 0));
        // ---- AdaIN + BatchNorm (batch statistics) 4 -----------------------------------------------
        float* dscale1 = tables;
        float* dbias1 = tables + (size_t)c.frames * MAX_WIDTH;
        float* dscale2 = tables + (size_t)c.frames * 2 * MAX_WIDTH;
        float* dbias2 = tables + (size_t)c.frames * 3 * MAX_WIDTH;
        PR_TRY(launch_zero_fill(tables, sizeof(float) * (size_t)c.frames * 4 * MAX_WIDTH, s));
        AdainBwd ab;
        memset(&ab, 0, sizeof(ab));
        ab.r = rc; ab.h = h2; ab.ld = d.W2pad; ab.width = d.W2; ab.table = table; ab.table_stride = table_stride;
        ab.goff = 2 * d.Wpad; ab.boff = 2 * d.Wpad + d.W2pad;
        ab.mean = batch + 2 * MAX_WIDTH; ab.var = batch + 3 * MAX_WIDTH; ab.eps = m.bn_eps;
        ab.g = bufB; ab.sums = sums; ab.dscale = dscale2; ab.dbias = dbias2; ab.count = stat_count;
        ab.frozen = (c.flags & PR_FLAG_TRAIN_BN) ? 0 : 1;
        PR_TRY(launch_zero_fill(sums, sizeof(double) * 2 * MAX_WIDTH, s));
        hipLaunchKernelGGL(k_adain_bwd_reduce, dim3(grid_blk, (d.W2 + 63) / 64), dim3(256), 0, s, ab);
        PR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_adain_bwd_apply, dim3(grid_rows), dim3(256), 0, s, ab);
        PR_LAUNCH_CHECK();
        // ---- layer 3 ----------------------------------------------------------------------------
        hipLaunchKernelGGL(k_adain_recompute, dim3(grid_rows), dim3(256), 0, s, rc, h1, d.Wpad, d.Wpad, table, table_stride, 0,
                           d.Wpad, actb);
        PR_LAUNCH_CHECK();
        PR_TRY(weight_grad(gc, bufB, d.W2pad, d.W2, actb, d.Wpad, d.W, G.head3.weight, d.W, nullptr));
        PR_TRY(input_grad(gc, bufB, d.W2pad, d.W2, m.head3.weight, d.W, d.W, bufA, d.Wpad, false, nullptr, 0));
        // ---- AdaIN + BatchNorm 1 ------------------------------------------------------------------
        ab.h = h1; ab.ld = d.Wpad; ab.width = d.W; ab.goff = 0; ab.boff = d.Wpad;
        ab.mean = batch; ab.var = batch + MAX_WIDTH; ab.g = bufA; ab.dscale = dscale1; ab.dbias = dbias1;
        PR_TRY(launch_zero_fill(sums, sizeof(double) * 2 * MAX_WIDTH, s));
        hipLaunchKernelGGL(k_adain_bwd_reduce, dim3(grid_blk, (d.W + 63) / 64), dim3(256), 0, s, ab);
        PR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_adain_bwd_apply, dim3(grid_rows), dim3(256), 0, s, ab);
        PR_LAUNCH_CHECK();
        // ---- layer 0 of the head + sigma head ------------------------------------------------------
        const int nb = m.backbone_count;
        const float* act_last = acts + (size_t)(nb - 1) * act_stride;
        PR_TRY(weight_grad(gc, bufA, d.Wpad, d.W, act_last, d.Wpad, d.W, G.head0.weight, d.W, nullptr));
        PR_TRY(input_grad(gc, bufA, d.Wpad, d.W, m.head0.weight, d.W, d.W, bufB, d.Wpad, false, nullptr, 0));
        hipLaunchKernelGGL(k_sigma_bwd, dim3(grid_blk, (d.W + 63) / 64), dim3(256), 0, s, rc, bufB, act_last, d.Wpad, d.W,
                           m.kind == 0 ? gsr : nullptr, m.alpha_head.weight, G.alpha_head.weight, G.alpha_head.bias);
        PR_LAUNCH_CHECK();
        // ---- backbone -------------------------------------------------------------------------------
        PR_TRY(chain_backward(gc, m.backbone, G.backbone, nb, m.skip_layer_idx, d.W, d.Wpad, acts, act_stride, enc, d.enc_pad,
                              d.enc, bufB, bufA, g_enc, reinterpret_cast<const unsigned char*>(fws + sv.bits)));
        // ---- positional encoding ----------------------------------------------------------------------
        const float* gx_final = nullptr;
        if (m.kind == 0) {
            hipLaunchKernelGGL(k_pe_bwd, dim3(row_blocks), dim3(256), 0, s, rc, enc, g_enc, d.enc_pad, 3, m.octaves, size[0],
                               size[1], size[2], 3, g_bent, 3, 0);
            PR_LAUNCH_CHECK();
            gx_final = g_bent;
        } else {
            hipLaunchKernelGGL(k_pe_bwd, dim3(row_blocks), dim3(256), 0, s, rc, enc, g_enc, d.enc_pad, 6, m.octaves, 0.f, 0.f, 0.f,
                               3, g_in6, 6, 0);
            PR_LAUNCH_CHECK();
        }
        // ---- ray bender -----------------------------------------------------------------------------------
        if (m.has_bender) {
            PR_REQUIRE(m.kind == 0, "backward: a skybox model with a ray bender is not supported");
            const float* bin = reinterpret_cast<const float*>(fws + sv.bin);
            const float* bacts = reinterpret_cast<const float*>(fws + sv.bact);
            const size_t bact_stride = cap * d.BWpad;
            const float* braw = reinterpret_cast<const float*>(fws + sv.braw);
            const float* delta = reinterpret_cast<const float*>(fws + sv.delta);
            hipLaunchKernelGGL(k_bender_out_bwd, dim3(row_blocks), dim3(256), 0, s, rc, g_bent, gdr, grads.sample_delta[k], delta, braw, rec_pos, lo[0],
                               lo[1], lo[2], hi[0], hi[1], hi[2], (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0, g_x, g_braw);
            PR_LAUNCH_CHECK();
            const int bc = m.bender_count;
            hipLaunchKernelGGL(k_bender_head_bwd, dim3(grid_blk, (d.BW + 63) / 64), dim3(256), 0, s, rc, g_braw, bacts + (size_t)(bc - 1) * bact_stride,
                               nullptr, d.BWpad, d.BW, m.bender_out.weight, d.BW, bufA, G.bender_out.weight);
            PR_LAUNCH_CHECK();
            PR_TRY(chain_backward(gc, m.bender, G.bender, bc, m.bender_skip, d.BW, d.BWpad, bacts, bact_stride, bin, d.bin_pad,
                                  d.bin, bufA, bufB, g_enc, reinterpret_cast<const unsigned char*>(fws + sv.bbits)));
            hipLaunchKernelGGL(k_pe_bwd, dim3(row_blocks), dim3(256), 0, s, rc, bin, g_enc, d.bin_pad, 3, m.bender_octaves, size[0],
                               size[1], size[2], 1, g_x, 3, 1);
            PR_LAUNCH_CHECK();
            if (out.deformation) {
                hipLaunchKernelGGL(k_deformation_bwd, dim3(grid_blk), dim3(64), 0, s, rc, g_enc, d.bin_pad, d.benc,
                                   m.deformation_features, out.deformation + (size_t)k * m.deformation_features,
                                   K * m.deformation_features);
                PR_LAUNCH_CHECK();
            }
            // ---- Hutchinson divergence: div = e^T J e is linear in the bender's weights along the probe tangents
            // t_0 = d input / dx . e, t_{l+1} = relu'(z_l) (W_l t_l), div = sum_a e_a size_a (W_out t_last)_a  (where the
            // clamp passes the network's output).  Its backward is the SAME chain with the tangents in place of the
            // activations as left factors of the weight gradients (the ReLU masks are piecewise constant), and the
            // input's tangent depends on the position through the encoding's derivative only.
            NoiseRef probes = make_noise(noise.divergence[k], c, NOISE_DIVERGENCE, t, k);
            if (div_grad && (probes.ptr || probes.generate)) {
                float* t0 = reinterpret_cast<float*>(lws + bp.div_t0);
                float* tstack = reinterpret_cast<float*>(lws + bp.div_stack);
                const size_t tstride = cap * d.BWpad;
                DivergenceParams dp;
                memset(&dp, 0, sizeof(dp));
                dp.total = totals + k; dp.max_rows = (int)cap;
                dp.rec_flat = rc.rec_flat; dp.row_flags = rc.row_flags; dp.rec_pos = rec_pos;
                dp.noise = probes; dp.positions = P;
                dp.bin = bin; dp.bin_pad = d.bin_pad; dp.benc = d.benc; dp.b_octaves = m.bender_octaves;
                dp.bacts = bacts; dp.bact_stride = bact_stride; dp.BW = d.BW; dp.BWpad = d.BWpad;
                dp.b_count = bc; dp.b_skip = m.bender_skip; dp.bin_real = d.bin;
                dp.layers = m.bender; dp.out_head = m.bender_out; dp.braw = braw;
                for (int a = 0; a < 3; ++a) {
                    dp.lo[a] = lo[a];
                    dp.hi[a] = hi[a];
                }
                dp.canonical = (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0;
                dp.t0 = t0; dp.tstack = tstack; dp.tstride = tstride;
                dp.div = nullptr;                                   // tangents only
                PR_TRY(launch_divergence(dp, s));
                const float* g_div = reinterpret_cast<const float*>(bws + bp.g_div[k]);
                hipLaunchKernelGGL(k_div_out_bwd, dim3(row_blocks), dim3(256), 0, s, rc, probes, P, g_div, braw, rec_pos, lo[0], lo[1],
                                   lo[2], hi[0], hi[1], hi[2], dp.canonical, g_braw);
                PR_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_bender_head_bwd, dim3(grid_blk, (d.BW + 63) / 64), dim3(256), 0, s, rc, g_braw,
                                   bacts + (size_t)(bc - 1) * bact_stride, tstack + (size_t)(bc - 1) * tstride, d.BWpad, d.BW,
                                   m.bender_out.weight, d.BW, bufA, G.bender_out.weight);
                PR_LAUNCH_CHECK();
                GemmCtx tc = gc;
                tc.dw_acts = tstack; tc.dw_stride = tstride; tc.dw_in0 = t0;
                PR_TRY(chain_backward(tc, m.bender, G.bender, bc, m.bender_skip, d.BW, d.BWpad, bacts, bact_stride, bin, d.bin_pad,
                                      d.bin, bufA, bufB, g_enc, reinterpret_cast<const unsigned char*>(fws + sv.bbits)));
                hipLaunchKernelGGL(k_div_tangent_in_bwd, dim3(row_blocks), dim3(256), 0, s, rc, probes, P, bin, g_enc, d.bin_pad,
                                   m.bender_octaves, size[0], size[1], size[2], g_x);
                PR_LAUNCH_CHECK();
            }
            gx_final = g_x;
        }
        // ---- style affine -----------------------------------------------------------------------------------
        {
            const int S = m.style_features;
            const float* style_k = c.style + (size_t)k * S;
            float* d_style_k = out.style ? out.style + (size_t)k * S : nullptr;
            long n = (long)2 * d.W * S;
            hipLaunchKernelGGL(k_style_bwd, dim3((unsigned)(c.frames + (n + 255) / 256)), dim3(256), 0, s, c.frames, d.W, S, dscale1,
                               dbias1, style_k, K * S, m.affine1.weight, G.affine1.weight, G.affine1.bias, d_style_k);
            PR_LAUNCH_CHECK();
            n = (long)2 * d.W2 * S;
            hipLaunchKernelGGL(k_style_bwd, dim3((unsigned)(c.frames + (n + 255) / 256)), dim3(256), 0, s, c.frames, d.W2, S, dscale2,
                               dbias2, style_k, K * S, m.affine4.weight, G.affine4.weight, G.affine4.bias, d_style_k);
            PR_LAUNCH_CHECK();
        }
        // ---- sample placement -> object pose -----------------------------------------------------------------
        if (out.w2o || ray_grads) {
            GeometryBwd gb;
            memset(&gb, 0, sizeof(gb));
            gb.frames = c.frames; gb.rays = c.rays; gb.positions = P; gb.objects = K; gb.object_index = k; gb.kind = m.kind;
            gb.ray_origins = c.ray_origins; gb.ray_directions = c.ray_directions; gb.w2o = c.w2o; gb.in_scene = c.object_in_scene;
            for (int a = 0; a < 3; ++a) {
                gb.lo[a] = lo[a];
                gb.hi[a] = hi[a];
            }
            gb.z_near_min = m.z_near_min; gb.z_far_max = m.z_far_max;
            gb.linspace = c.linspace_coarse[k];
            gb.jitter = perturb_noise(c.noise_coarse.jitter[k], c, NOISE_JITTER, 0, k);
            if (t) {
                gb.t_coarse = reinterpret_cast<const float*>(fws + plan.type[0].t[k]);
                gb.pc = objs[k].coarse.positions;
            }
            gb.t = reinterpret_cast<const float*>(fws + tp.t[k]);
            gb.slot = reinterpret_cast<const int32_t*>(fws + tp.slot[k]);
            gb.g_t = g_t;
            gb.g_x = gx_final;
            gb.g_in6 = m.kind == 1 ? g_in6 : nullptr;
            gb.d_w2o = out.w2o;
            gb.d_ray_origins = out.ray_origins;
            gb.d_ray_directions = out.ray_directions;
            gb.g_norm = (k == 0) ? g_norm : nullptr;
            hipLaunchKernelGGL(k_geometry_bwd, dim3((c.rays + 255) / 256, c.frames), dim3(256), 0, s, gb);
            PR_LAUNCH_CHECK();
        }
    }
    return PR_OK;
    };
    int status = run_objects();
    if (aux) {                                           // join: the caller's stream continues after both lanes
        const int joined = stream_wait(s_caller, aux);
        if (status == PR_OK) status = joined;
    }
    return status;
}


// ---------------------------------------------------------------------------------------------
// Grouped backward pass of one model type: the objects of the call share every launch.
//   compositing backward -> head phase 1 -> head phase 2 -> NeRF chains (head layer 0 and the sigma head fused in) -> row
//   kernel (positional encoding, bender output) -> bender chains -> row kernel (bender encoding, deformation) -> EVERY weight
//   gradient of every object (one launch + its reduction) -> style affines -> sample placement
// ---------------------------------------------------------------------------------------------
static int backward_grouped(const pr_call_t& c, const pr_object_t* objs, int t, const pr_output_grads_t& grads, const pr_input_grads_t& out,
                            char* fws, const Plan& plan, char* bws, const BwdPlan& bp, hipStream_t s) {
    const int K = c.objects;
    const TypePlan& tp = plan.type[t];
    const GroupPlan& gp = bp.group;
    int32_t* totals = reinterpret_cast<int32_t*>(fws + tp.totals);
    const int F = objs[0].coarse.output_features;
    const int Fs = (F + 15) & ~15;
    PR_TRY(launch_zero_fill(bws + gp.zero_begin, gp.zero_bytes, s));
    CompositeBwdResult cr;
    PR_TRY(composite_backward(c, objs, t, grads, out, fws, plan, bws, bp, s, &cr));

    static thread_local HeadBwdJob h1[PR_MAX_OBJECTS], h2[PR_MAX_OBJECTS];
    static thread_local ChainBwdJob cn[PR_MAX_OBJECTS], cb[PR_MAX_OBJECTS];
    static thread_local PostNerfJobs pn;
    static thread_local PostBenderJobs pb;
    static thread_local StyleBwdJobs sj;
    static thread_local GeometryBwdJobs gj;
    // products per object: backbone layers + the skip layer's second product + density head + three feature-head layers, and the
    // same for the ray bender (layers + skip + output head); four launches of TN_ALL_MAX products have claim counters (make_bwd_plan)
    constexpr int TN_LAUNCHES_MAX = 4;
    constexpr int MAX_TN_JOBS_CALL = TN_LAUNCHES_MAX * TN_ALL_MAX;
    static_assert(PR_MAX_OBJECTS * (2 * PR_MAX_LAYERS + 8) <= MAX_TN_JOBS_CALL, "weight-gradient job table smaller than the ABI's largest call");
    static thread_local TnJob tn_jobs[MAX_TN_JOBS_CALL];
    static thread_local TnAll tn;
    long rows[PR_MAX_OBJECTS], rows_b[PR_MAX_OBJECTS], tn_rows[MAX_TN_JOBS_CALL];
    int tn_count = 0;
    int benders = 0, style_jobs = 0;
    int style_blocks = 0;
    int32_t* counters = reinterpret_cast<int32_t*>(bws + gp.counters);
    float* tn_at = reinterpret_cast<float*>(bws + gp.tn_partial);
    size_t tn_used = 0;
    memset(&sj, 0, sizeof(sj));
    sj.frames = c.frames;
    long max_cap = 0;
    int max_ld_n = 0, max_ld_b = 0;      // widest encoding rows (LDS of the row kernels)

    for (int k = 0; k < K; ++k) {
        const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
        const pr_model_grads_t& G = t ? out.model_fine[k] : out.model[k];
        const float* packed = static_cast<const float*>(t ? objs[k].packed_fine : objs[k].packed_coarse);
        const SavedPlan& sv = tp.saved[k];
        ModelDims d;
        PackedLayout l;
        PR_TRY(compute_dims(m, &d));
        PR_TRY(compute_layout(m, d, &l));
        PR_REQUIRE(d.Fpad % 16 == 0 && Fs <= d.Fpad, "backward: feature rows of %d floats", Fs);
        const int P = m.positions;
        const size_t cap = (size_t)c.frames * c.rays * P;
        rows[k] = (long)cap;
        if ((long)cap > max_cap) max_cap = (long)cap;
        RowCtx rc;
        rc.total = totals + k;
        rc.rec_flat = reinterpret_cast<const int32_t*>(fws + sv.rec_flat);
        rc.row_flags = reinterpret_cast<const int32_t*>(fws + sv.row_flags);
        rc.samples_per_frame = c.rays * P;
        rc.in_scene = c.object_in_scene + k; rc.in_scene_stride = K;
        const float* rec_pos = reinterpret_cast<const float*>(fws + sv.rec_pos);
        const float* enc = reinterpret_cast<const float*>(fws + sv.enc);
        const float* acts = reinterpret_cast<const float*>(fws + sv.act);
        const size_t act_stride = cap * d.Wpad;
        const float* h1v = reinterpret_cast<const float*>(fws + sv.h1);
        const float* h2v = reinterpret_cast<const float*>(fws + sv.h2);
        const float* batch = reinterpret_cast<const float*>(fws + sv.batch);
        const int32_t* stat_count = reinterpret_cast<const int32_t*>(fws + sv.stat_count);
        const float* table = reinterpret_cast<const float*>(fws + tp.adain[k]);
        const int table_stride = adain_row_floats(d);
        float* g_feat = reinterpret_cast<float*>(bws + bp.g_feat[k]);
        float* a2 = reinterpret_cast<float*>(bws + gp.a2[k]);
        float* d2 = reinterpret_cast<float*>(bws + gp.d2[k]);
        float* a1 = reinterpret_cast<float*>(bws + gp.a1[k]);
        float* d1 = reinterpret_cast<float*>(bws + gp.d1[k]);
        float* gstack = reinterpret_cast<float*>(bws + gp.gstack[k]);
        float* g_enc = reinterpret_cast<float*>(bws + gp.g_enc[k]);
        float* gsr4 = reinterpret_cast<float*>(bws + gp.gsr4[k]);
        float* g_x = reinterpret_cast<float*>(bws + gp.g_x[k]);
        float* g_in6 = reinterpret_cast<float*>(bws + gp.g_in6[k]);
        double* sums = reinterpret_cast<double*>(bws + gp.sums[k]);
        float* tables = reinterpret_cast<float*>(bws + gp.tables[k]);
        float* dscale1 = tables;
        float* dbias1 = tables + (size_t)c.frames * MAX_WIDTH;
        float* dscale2 = tables + (size_t)c.frames * 2 * MAX_WIDTH;
        float* dbias2 = tables + (size_t)c.frames * 3 * MAX_WIDTH;
        const int frozen = (c.flags & PR_FLAG_TRAIN_BN) ? 0 : 1;
#ifdef PR_CHAIN_BF16
        const int split_bwd = (c.flags & PR_FLAG_SPLIT_BACKWARD) ? 1 : 0;      // products on bf16 triples (t3_* segments)
#else
        const int split_bwd = (c.flags & PR_FLAG_SPLIT_BACKWARD) ? 2 : 0;      // products on fp16 pairs of scaled tiles (t3_* segments)
#endif
        // (the two head phases keep the fp32 product: with the phase's raw-activation prefetch registers the bf16 variant of
        // k_head_bwd_group spills 125 VGPRs; they are 0.4 ms of the step)
        const int split_head = 0;
        float lo[3], hi[3], size[3];
        bbox_split(m, lo, hi, size);
        const int nb = m.backbone_count;

        // ---- feature head, phases 1 and 2 ------------------------------------------------------------
        HeadBwdJob& a = h1[k];
        memset(&a, 0, sizeof(a));
        a.total = totals + k; a.rec_flat = rc.rec_flat; a.row_flags = rc.row_flags; a.samples_per_frame = rc.samples_per_frame;
        a.phase = 1; a.frozen = frozen; a.stat_count = stat_count; a.eps = m.bn_eps;
        a.table = table; a.table_stride = table_stride; a.goff = 2 * d.Wpad; a.boff = 2 * d.Wpad + d.W2pad;
        a.g_in = g_feat; a.ld_gin = Fs; a.k_real = F; a.kpad = d.Fpad;
        a.wt = Seg{packed + (split_head ? l.t3_h6 : l.t_h6), d.Fpad / 8, 0}; a.nblk = d.W2pad / 32; a.split = split_head;
        a.h = h2v; a.ld = d.W2pad; a.mean = batch + 2 * MAX_WIDTH; a.var = batch + 3 * MAX_WIDTH; a.width = d.W2;
        a.a_out = a2; a.d_out = d2; a.sums = sums; a.dscale = dscale2; a.dbias = dbias2;
        a.tile_counter = counters + 4 * k;
        HeadBwdJob& b = h2[k];
        b = a;
        b.phase = 2;
        b.goff = 0; b.boff = d.Wpad;
        b.g_in = nullptr;
        b.d_in = d2; b.h_in = h2v; b.mean_in = batch + 2 * MAX_WIDTH; b.var_in = batch + 3 * MAX_WIDTH; b.sums_in = sums; b.width_in = d.W2;
        b.kpad = d.W2pad;
        b.wt = Seg{packed + (split_head ? l.t3_h3 : l.t_h3), d.W2pad / 8, 0}; b.nblk = d.Wpad / 32; b.split = split_head;
        b.h = h1v; b.ld = d.Wpad; b.mean = batch; b.var = batch + MAX_WIDTH; b.width = d.W;
        b.a_out = a1; b.d_out = d1; b.sums = sums + 2 * MAX_WIDTH; b.dscale = dscale1; b.dbias = dbias1;
        b.tile_counter = counters + 4 * k + 1;

        // ---- NeRF chain ---------------------------------------------------------------------------------
        ChainBwdJob& n = cn[k];
        memset(&n, 0, sizeof(n));
        n.total = totals + k; n.rec_flat = rc.rec_flat; n.row_flags = rc.row_flags; n.samples_per_frame = rc.samples_per_frame;
        n.entry = 1;
        n.d1 = d1; n.h1 = h1v; n.mean1 = batch; n.var1 = batch + MAX_WIDTH; n.sums1 = sums + 2 * MAX_WIDTH;
        n.stat_count = stat_count; n.frozen = frozen; n.eps = m.bn_eps;
        n.w0t = Seg{packed + (split_bwd ? l.t3_h0 : l.t_h0), d.Wpad / 8, 0}; n.split = split_bwd;
        n.g_sigma = reinterpret_cast<const float*>(bws + bp.g_sigma[k]);
        n.in_scene = rc.in_scene; n.in_scene_stride = K;
        n.w_sigma = m.kind == 0 ? m.alpha_head.weight : nullptr;
        n.gsr4 = gsr4;
        n.count = nb; n.skip = m.skip_layer_idx; n.W = d.W; n.Wpad = d.Wpad; n.in_pad = d.enc_pad; n.in_real = d.enc;
        for (int i = 1; i < nb; ++i) n.act_t[i] = Seg{packed + (split_bwd ? l.t3_n_act[i] : l.t_n_act[i]), d.Wpad / 8, 0};
        n.in0_skip = Seg{packed + (split_bwd ? l.t3_n_skip : l.t_n_skip), d.Wpad / 8, 0};
        n.in0_first = Seg{packed + (split_bwd ? l.t3_n_first : l.t_n_first), d.Wpad / 8, 0};
        n.bits = reinterpret_cast<const unsigned char*>(fws + sv.bits); n.bits_stride = relu_bits_bytes(cap, d.Wpad);
        n.gstack = gstack; n.g_stride = cap * (size_t)d.Wpad;
        n.g_in = g_enc; n.ld_in = d.enc_pad;
        n.tile_counter = counters + 4 * k + 2;

        // ---- rows after the NeRF chain ---------------------------------------------------------------------
        PostNerfJob& q = pn.job[k];
        memset(&q, 0, sizeof(q));
        q.r = rc; q.kind = m.kind; q.has_bender = m.has_bender; q.octaves = m.octaves; q.ld = d.enc_pad;
        q.enc = enc; q.g_enc = g_enc;
        if (d.enc_pad > max_ld_n) max_ld_n = d.enc_pad;
        for (int ax = 0; ax < 3; ++ax) { q.size[ax] = size[ax]; q.lo[ax] = lo[ax]; q.hi[ax] = hi[ax]; }
        q.g_x = g_x; q.g_in6 = g_in6;

        // ---- weight gradients --------------------------------------------------------------------------------
        auto add = [&](const float* dY, int ldy, int ni, const float* X, int ldx, int nj, float* dW, int ldw, float* dbias,
                       const float* side_w = nullptr, int side_ld = 0, float* side_grad = nullptr, float* side_bias = nullptr) -> int {
            if (!dW) {
                PR_REQUIRE(!dbias, "a bias gradient buffer needs its weight gradient buffer");
                return PR_OK;
            }
            PR_REQUIRE(tn_count < MAX_TN_JOBS_CALL, "backward: too many weight-gradient products");
            TnJob& j = tn_jobs[tn_count];
            memset(&j, 0, sizeof(j));
            j.A = dY; j.lda = ldy; j.B = X; j.ldb = ldx; j.C = dW; j.ldc = ldw; j.bias = dbias;
            j.rows = totals + k; j.ni = ni; j.nj = nj;
            if (side_grad) { j.w = side_w; j.ldw = side_ld; j.wgrad = side_grad; j.wbias = side_bias; }
            const size_t need = tn_all_partial_floats(ni, nj, (long)cap);
            PR_REQUIRE(tn_used + need <= gp.tn_partial_floats, "backward: weight-gradient scratch exhausted");
            j.partial = tn_at + tn_used;
            tn_used += need;
            tn_rows[tn_count] = (long)cap;
            ++tn_count;
            return PR_OK;
        };
        PR_TRY(add(g_feat, Fs, F, a2, d.W2pad, d.W2, G.head6.weight, d.W2, G.head6.bias));
        PR_TRY(add(d2, d.W2pad, d.W2, a1, d.Wpad, d.W, G.head3.weight, d.W, nullptr));
        const float* act_last = acts + (size_t)(nb - 1) * act_stride;
        // the density head reads the same activations as head layer 0: its (1 x W) gradient rides on that product's tiles as a
        // weighted column sum (as a product of its own it costs two full tiles per 32 rows)
        if (m.kind == 0 && G.head0.weight && G.alpha_head.weight) {
            PR_TRY(add(d1, d.Wpad, d.W, act_last, d.Wpad, d.W, G.head0.weight, d.W, nullptr, gsr4, 4, G.alpha_head.weight, G.alpha_head.bias));
        } else {
            PR_TRY(add(d1, d.Wpad, d.W, act_last, d.Wpad, d.W, G.head0.weight, d.W, nullptr));
            if (m.kind == 0) PR_TRY(add(gsr4, 4, 1, act_last, d.Wpad, d.W, G.alpha_head.weight, d.W, G.alpha_head.bias));
        }
        for (int i = nb - 1; i >= 0; --i) {
            const float* dY = gstack + (size_t)i * n.g_stride;
            const int inf = m.backbone[i].in_features;
            if (i == 0) {
                PR_TRY(add(dY, d.Wpad, d.W, enc, d.enc_pad, d.enc, G.backbone[i].weight, inf, G.backbone[i].bias));
            } else {
                PR_TRY(add(dY, d.Wpad, d.W, acts + (size_t)(i - 1) * act_stride, d.Wpad, d.W, G.backbone[i].weight, inf, G.backbone[i].bias));
                if (i == m.skip_layer_idx)
                    PR_TRY(add(dY, d.Wpad, d.W, enc, d.enc_pad, d.enc, G.backbone[i].weight ? G.backbone[i].weight + d.W : nullptr, inf, nullptr));
            }
        }

        // ---- ray bender ----------------------------------------------------------------------------------------
        if (m.has_bender) {
            PR_REQUIRE(m.kind == 0, "backward: a skybox model with a ray bender is not supported");
            const float* bin = reinterpret_cast<const float*>(fws + sv.bin);
            const float* bacts = reinterpret_cast<const float*>(fws + sv.bact);
            const size_t bact_stride = cap * d.BWpad;
            float* g_braw4 = reinterpret_cast<float*>(bws + gp.g_braw4[k]);
            float* bgstack = reinterpret_cast<float*>(bws + gp.bgstack[k]);
            float* g_benc = reinterpret_cast<float*>(bws + gp.g_benc[k]);
            const int bc = m.bender_count;
            q.g_dm = reinterpret_cast<const float*>(bws + bp.g_dm[k]);
            q.g_delta_dense = grads.sample_delta[k];
            q.delta = reinterpret_cast<const float*>(fws + sv.delta);
            q.braw = reinterpret_cast<const float*>(fws + sv.braw);
            q.pos = rec_pos;
            q.canonical = (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0;
            q.g_braw4 = g_braw4;
            ChainBwdJob& e = cb[benders];
            memset(&e, 0, sizeof(e));
            e.total = totals + k; e.rec_flat = rc.rec_flat; e.row_flags = rc.row_flags; e.samples_per_frame = rc.samples_per_frame;
            e.entry = 0;
            e.g_braw4 = g_braw4; e.w_out = m.bender_out.weight; e.w_out_ld = d.BW;
            e.count = bc; e.skip = m.bender_skip; e.W = d.BW; e.Wpad = d.BWpad; e.in_pad = d.bin_pad; e.in_real = d.bin;
            for (int i = 1; i < bc; ++i) e.act_t[i] = Seg{packed + (split_bwd ? l.t3_b_act[i] : l.t_b_act[i]), d.BWpad / 8, 0};
            e.split = split_bwd;
            e.in0_skip = Seg{packed + (split_bwd ? l.t3_b_skip : l.t_b_skip), d.BWpad / 8, 0};
            e.in0_first = Seg{packed + (split_bwd ? l.t3_b_first : l.t_b_first), d.BWpad / 8, 0};
            e.bits = reinterpret_cast<const unsigned char*>(fws + sv.bbits); e.bits_stride = relu_bits_bytes(cap, d.BWpad);
            e.gstack = bgstack; e.g_stride = cap * (size_t)d.BWpad;
            e.g_in = g_benc; e.ld_in = d.bin_pad;
            e.tile_counter = counters + 4 * k + 3;
            rows_b[benders] = (long)cap;
            PostBenderJob& w = pb.job[benders];
            memset(&w, 0, sizeof(w));
            w.r = rc; w.bin = bin; w.g_bin = g_benc; w.ld = d.bin_pad; w.octaves = m.bender_octaves; w.benc = d.benc;
            w.D = m.deformation_features;
            if (d.bin_pad > max_ld_b) max_ld_b = d.bin_pad;
            PR_REQUIRE(w.D <= 192, "backward: %d deformation features", w.D);
            for (int ax = 0; ax < 3; ++ax) w.size[ax] = size[ax];
            w.g_x = g_x;
            w.d_def = out.deformation ? out.deformation + (size_t)k * m.deformation_features : nullptr;
            w.def_stride = K * m.deformation_features;
            ++benders;
            PR_TRY(add(g_braw4, 4, 3, bacts + (size_t)(bc - 1) * bact_stride, d.BWpad, d.BW, G.bender_out.weight, d.BW, nullptr));
            for (int i = bc - 1; i >= 0; --i) {
                const float* dY = bgstack + (size_t)i * e.g_stride;
                const int inf = m.bender[i].in_features;
                if (i == 0) {
                    PR_TRY(add(dY, d.BWpad, d.BW, bin, d.bin_pad, d.bin, G.bender[i].weight, inf, G.bender[i].bias));
                } else {
                    PR_TRY(add(dY, d.BWpad, d.BW, bacts + (size_t)(i - 1) * bact_stride, d.BWpad, d.BW, G.bender[i].weight, inf, G.bender[i].bias));
                    if (i == m.bender_skip)
                        PR_TRY(add(dY, d.BWpad, d.BW, bin, d.bin_pad, d.bin, G.bender[i].weight ? G.bender[i].weight + d.BW : nullptr, inf, nullptr));
                }
            }
        }

        // ---- style affines -------------------------------------------------------------------------------------
        {
            const int S = m.style_features;
            StyleBwdJob& s1 = sj.job[style_jobs++];
            s1.width = d.W; s1.S = S; s1.dscale = dscale1; s1.dbias = dbias1;
            s1.style = c.style + (size_t)k * S; s1.style_stride = K * S;
            s1.A = m.affine1.weight; s1.dA = G.affine1.weight; s1.db = G.affine1.bias;
            s1.d_style = out.style ? out.style + (size_t)k * S : nullptr;
            StyleBwdJob& s2 = sj.job[style_jobs++];
            s2 = s1;
            s2.width = d.W2; s2.dscale = dscale2; s2.dbias = dbias2;
            s2.A = m.affine4.weight; s2.dA = G.affine4.weight; s2.db = G.affine4.bias;
            const int blocks = c.frames + (int)(((long)2 * d.W * S + 255) / 256);
            if (blocks > style_blocks) style_blocks = blocks;
        }

        // ---- sample placement -> object pose, camera rays ----------------------------------------------------------
        {
            GeometryBwd& gb = gj.job[k];
            memset(&gb, 0, sizeof(gb));
            gb.frames = c.frames; gb.rays = c.rays; gb.positions = P; gb.objects = K; gb.object_index = k; gb.kind = m.kind;
            gb.ray_origins = c.ray_origins; gb.ray_directions = c.ray_directions; gb.w2o = c.w2o; gb.in_scene = c.object_in_scene;
            for (int ax = 0; ax < 3; ++ax) { gb.lo[ax] = lo[ax]; gb.hi[ax] = hi[ax]; }
            gb.z_near_min = m.z_near_min; gb.z_far_max = m.z_far_max;
            gb.linspace = c.linspace_coarse[k];
            gb.jitter = perturb_noise(c.noise_coarse.jitter[k], c, NOISE_JITTER, 0, k);
            if (t) {
                gb.t_coarse = reinterpret_cast<const float*>(fws + plan.type[0].t[k]);
                gb.pc = objs[k].coarse.positions;
            }
            gb.t = reinterpret_cast<const float*>(fws + tp.t[k]);
            gb.slot = reinterpret_cast<const int32_t*>(fws + tp.slot[k]);
            gb.g_t = reinterpret_cast<const float*>(bws + bp.g_t[k]);
            gb.g_x = m.kind == 0 ? g_x : nullptr;
            gb.g_in6 = m.kind == 1 ? g_in6 : nullptr;
            gb.d_w2o = out.w2o;
            gb.d_ray_origins = out.ray_origins;
            gb.d_ray_directions = out.ray_directions;
            gb.g_norm = (k == 0) ? cr.g_norm : nullptr;
        }
    }

    // Two small-grid kernels leave the caller's stream: the style affines' backward (536 workgroups, ~58 us) needs the head phases' d scale /
    // d bias only and runs beside the NeRF chains; the sample-placement backward (144 workgroups, ~47 us) needs the position gradients and
    // runs beside the weight-gradient launch.  Both fit next to the persistent tile kernels (a few waves, ~1 KB of LDS per workgroup); on
    // the caller's stream each was ~50 us of a mostly idle chip plus a launch gap.  The second stream (one per device and caller stream, lane_stream)
    // forks from and joins the caller's stream through events: for the caller everything is still ordered on its own stream.
    hipStream_t aux = nullptr;
    auto run = [&]() -> int {
        PR_TRY(launch_head_bwd_group(h1, rows, K, s));
        PR_TRY(launch_head_bwd_group(h2, rows, K, s));
#ifndef PR_BWD_ONE_STREAM
        PR_TRY(lane_stream(s, &aux));
        PR_TRY(stream_wait(aux, s));                 // fork: behind the head phases
#endif
        hipLaunchKernelGGL(k_style_bwd_group, dim3(style_blocks, style_jobs), dim3(256), 0, aux ? aux : s, sj);
        PR_LAUNCH_CHECK();
        PR_TRY(launch_chain_bwd_group(cn, rows, K, s));
        const int row_blocks = (int)((max_cap + POST_ROWS - 1) / POST_ROWS);
        if (row_blocks > 0) {
            PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_post_nerf_group), (int)(sizeof(float) * 2 * POST_ROWS * (MAX_ENC + 1)), nullptr));
            hipLaunchKernelGGL(k_post_nerf_group, dim3(row_blocks, K), dim3(256), sizeof(float) * 2 * POST_ROWS * (size_t)(max_ld_n + 1), s, pn);
            PR_LAUNCH_CHECK();
        }
        if (benders) {
            PR_TRY(launch_chain_bwd_group(cb, rows_b, benders, s));
            long cap_b = 0;
            for (int i = 0; i < benders; ++i) cap_b = std::max(cap_b, rows_b[i]);
            PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_post_bender_group), (int)(sizeof(float) * 2 * POST_ROWS * (MAX_ENC + 1)), nullptr));
            hipLaunchKernelGGL(k_post_bender_group, dim3((unsigned)((cap_b + POST_ROWS - 1) / POST_ROWS), benders), dim3(256),
                               sizeof(float) * 2 * POST_ROWS * (size_t)(max_ld_b + 1), s, pb);
            PR_LAUNCH_CHECK();
        }
        if (out.w2o || cr.ray_grads) {
            if (aux) PR_TRY(stream_wait(aux, s));    // the position gradients are complete
            hipLaunchKernelGGL(k_geometry_bwd_group, dim3((c.rays + 255) / 256, c.frames, K), dim3(256), 0, aux ? aux : s, gj);
            PR_LAUNCH_CHECK();
        }
        // every weight gradient of the call: one launch (+ its reduction) per TN_ALL_MAX products, each with its own claim counters;
        // instances of one model that land in different launches accumulate one after the other (same stream)
        for (int begin = 0, chunk = 0; begin < tn_count; begin += TN_ALL_MAX, ++chunk) {
            tn.count = std::min(TN_ALL_MAX, tn_count - begin);
            memcpy(tn.job, tn_jobs + begin, sizeof(TnJob) * tn.count);
            tn.counters = counters + 4 * PR_MAX_OBJECTS + 8 * chunk;
            tn.split_precision = (c.flags & PR_FLAG_SPLIT_BACKWARD) ? 1 : 0;
            PR_TRY(launch_gemm_tn_all(tn, tn_rows + begin, s));
        }
        return PR_OK;
    };
    int status = run();
    if (aux) {                                       // join, whatever happened above: the caller's stream continues behind both
        const int joined = stream_wait(s, aux);
        if (status == PR_OK) status = joined;
    }
    return status;
}

}  // namespace pr

static int check_backward_call(const pr_call_t* call, const pr_object_t* objects) {
    PR_REQUIRE(call && objects, "NULL argument");
    PR_TRY(pr::validate_call(*call, objects));
    PR_REQUIRE(call->flags & PR_FLAG_SAVE_FOR_BACKWARD, "pr_render_backward: the forward call must set PR_FLAG_SAVE_FOR_BACKWARD");
    for (int k = 0; k < call->objects; ++k)
        for (int t = 0; t < (call->use_fine ? 2 : 1); ++t) {
            const pr_object_model_t& m = t ? objects[k].fine : objects[k].coarse;
            PR_REQUIRE(m.skip_layer_idx > 0 && (!m.has_bender || m.bender_skip > 0), "pr_render_backward: skip_layer_idx 0 is not supported");
        }
    return PR_OK;
}

extern "C" int pr_backward_workspace_size(const pr_call_t* call, const pr_object_t* objects, size_t* bytes) {
    PR_REQUIRE(bytes, "pr_backward_workspace_size: NULL argument");
    PR_TRY(check_backward_call(call, objects));
    pr::BwdPlan bp;
    PR_TRY(pr::make_bwd_plan(*call, objects, &bp));
    *bytes = bp.bytes;
    return PR_OK;
}

extern "C" int pr_render_backward(const pr_call_t* call, const pr_object_t* objects, const pr_output_grads_t* grads,
                                  const pr_output_grads_t* grads_fine, const pr_input_grads_t* out, void* forward_workspace,
                                  size_t forward_workspace_bytes, void* backward_workspace, size_t backward_workspace_bytes,
                                  void* stream) {
    PR_REQUIRE(grads && out && forward_workspace && backward_workspace, "pr_render_backward: NULL argument");
    PR_TRY(check_backward_call(call, objects));
    PR_REQUIRE(!call->use_fine || grads_fine, "pr_render_backward: use_fine calls need grads_fine");
    pr::Plan plan;
    PR_TRY(pr::make_plan(*call, objects, &plan));
    pr::BwdPlan bp;
    PR_TRY(pr::make_bwd_plan(*call, objects, &bp));
    if (forward_workspace_bytes < plan.bytes || backward_workspace_bytes < bp.bytes) {
        pr::set_error("workspace too small: forward %zu of %zu bytes, backward %zu of %zu bytes", forward_workspace_bytes, plan.bytes,
                      backward_workspace_bytes, bp.bytes);
        return PR_ERR_WORKSPACE;
    }
    PR_REQUIRE((((uintptr_t)forward_workspace | (uintptr_t)backward_workspace) & 255) == 0, "workspaces must be 256-byte aligned");
    // calls with gradients of the divergence estimate, or whose scratch would not fit, take the per-object path
    auto pass = bp.group.usable ? pr::backward_grouped : pr::backward;
    PR_TRY(pass(*call, objects, 0, *grads, *out, static_cast<char*>(forward_workspace), plan,
                static_cast<char*>(backward_workspace), bp, (hipStream_t)stream));
    if (call->use_fine)
        PR_TRY(pass(*call, objects, 1, *grads_fine, *out, static_cast<char*>(forward_workspace), plan,
                    static_cast<char*>(backward_workspace), bp, (hipStream_t)stream));
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Hutchinson divergence estimate (train mode with a graph only; object_composer.py:582-601)
// ---------------------------------------------------------------------------------------------
namespace pr {

// tangent of the bender input [annealed PE(x / size) | deformation] along e: d v_a = e_a / size_a;
// sin slot: 2^k cos_saved d v ; cos slot: -2^k sin_saved d v (the saved values carry the annealing weight)
__global__ __launch_bounds__(256) void k_div_tangent_in(const int32_t* total, const int32_t* rec_flat, NoiseRef noise, int positions,
                                                        const float* bin, int ld, int benc, int octaves, float s0, float s1,
                                                        float s2, float* t0) {
    const int M = *total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float size[3] = {s0, s1, s2};
    float e[3];
    probe_of(noise, rec_flat[m], positions, e);
    const float* b = bin + (size_t)m * ld;
    float* t = t0 + (size_t)m * ld;
    float dv[3];
    for (int a = 0; a < 3; ++a) {
        dv[a] = e[a] / size[a];
        t[a] = dv[a];
    }
    for (int k = 0; k < octaves; ++k) {
        const float f = ldexpf(1.0f, k);
        for (int a = 0; a < 3; ++a) {
            const int sn = 3 + k * 6 + a, cs = sn + 3;
            t[sn] = f * b[cs] * dv[a];
            t[cs] = -f * b[sn] * dv[a];
        }
    }
    for (int j = benc; j < ld; ++j) t[j] = 0.f;
}

// div = sum_a e_a (J e)_a with (J e)_a = size_a * (W_out t)_a where the clamp passes the network output,
// -e_a where delta = lo - x or hi - x, 0 in canonical pose
__global__ __launch_bounds__(256) void k_div_out(const int32_t* total, const int32_t* rec_flat, const int32_t* row_flags,
                                                 NoiseRef noise, int positions, const float* tlast, int ld, int width, const float* w_out,
                                                 const float* braw, const float* pos, float lo0, float lo1, float lo2, float hi0,
                                                 float hi1, float hi2, int canonical, float* div) {
    const int M = *total;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M || !(row_flags[m] & 1)) return;
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const int flat = rec_flat[m];
    float e[3];
    probe_of(noise, flat, positions, e);
    const float* t = tlast + (size_t)m * ld;
    float tan[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < width; ++c) {
        const float tv = t[c];
        tan[0] = fmaf(tv, w_out[c], tan[0]);
        tan[1] = fmaf(tv, w_out[width + c], tan[1]);
        tan[2] = fmaf(tv, w_out[2 * width + c], tan[2]);
    }
    float acc = 0.f;
    for (int a = 0; a < 3; ++a) {
        const float x = pos[(size_t)m * 3 + a];
        const float size = hi[a] - lo[a];
        const float pre = braw[(size_t)m * 3 + a] * size;
        const float lob = lo[a] - x, hib = hi[a] - x;
        const float m1 = pre > lob ? pre : lob;
        float je;
        if (m1 > hib) je = -e[a];
        else if (pre >= lob) je = size * tan[a];
        else je = -e[a];
        if (canonical) je = 0.f;
        acc = fmaf(e[a], je, acc);
    }
    div[flat] = acc;
}

int launch_divergence(const DivergenceParams& p, hipStream_t s) {
    if (p.max_rows <= 0) return PR_OK;
    const int blocks = (p.max_rows + 255) / 256;
    hipLaunchKernelGGL(k_div_tangent_in, dim3(blocks), dim3(256), 0, s, p.total, p.rec_flat, p.noise, p.positions, p.bin, p.bin_pad, p.benc,
                       p.b_octaves, p.hi[0] - p.lo[0], p.hi[1] - p.lo[1], p.hi[2] - p.lo[2], p.t0);
    PR_LAUNCH_CHECK();
    float* cur = p.tstack ? p.tstack : p.ta;
    float* other = p.tb;
    const float* prev = p.t0;
    for (int l = 0; l < p.b_count; ++l) {
        const pr_linear_t& L = p.layers[l];
        if (p.tstack) cur = p.tstack + (size_t)l * p.tstride;     // every layer's tangent is kept (backward of the estimate)
        GemmNN g;
        memset(&g, 0, sizeof(g));
        g.rows = p.total; g.C = cur; g.ldc = p.BWpad; g.n = p.BW;
        g.mask = p.bacts + (size_t)l * p.bact_stride; g.ldm = p.BWpad;
        g.b_transposed = 1; g.B = L.weight; g.ldb = L.in_features;
        if (l == 0) {
            g.A = p.t0; g.lda = p.bin_pad; g.k = p.bin_pad; g.k_valid = p.bin_real;
            PR_TRY(launch_gemm_nn(g, p.max_rows, s));
        } else if (l == p.b_skip) {
            // input = [h | bender input]: two products, the ReLU mask goes with the second
            GemmNN g1 = g;
            g1.A = prev; g1.lda = p.BWpad; g1.k = p.BWpad; g1.k_valid = p.BW; g1.mask = nullptr;
            PR_TRY(launch_gemm_nn(g1, p.max_rows, s));
            g.A = p.t0; g.lda = p.bin_pad; g.k = p.bin_pad; g.k_valid = p.bin_real; g.B = L.weight + p.BW; g.accumulate = 1;
            PR_TRY(launch_gemm_nn(g, p.max_rows, s));
        } else {
            g.A = prev; g.lda = p.BWpad; g.k = p.BWpad; g.k_valid = p.BW;
            PR_TRY(launch_gemm_nn(g, p.max_rows, s));
        }
        prev = cur;
        float* tmp = cur;
        cur = other;
        other = tmp;
    }
    if (!p.div) return PR_OK;
    hipLaunchKernelGGL(k_div_out, dim3(blocks), dim3(256), 0, s, p.total, p.rec_flat, p.row_flags, p.noise, p.positions, prev, p.BWpad, p.BW,
                       p.out_head.weight, p.braw, p.rec_pos, p.lo[0], p.lo[1], p.lo[2], p.hi[0], p.hi[1], p.hi[2], p.canonical, p.div);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
