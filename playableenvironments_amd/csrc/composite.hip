// Alpha compositing: per-object integration and the cross-object merge (ObjectComposer.integrate,
// compose, fix_object_overlap; model/object_composer.py:153-214, :295-447, :724-784).
//
// One workgroup per ray (one wave for short lists, four for long ones).  All K per-object sample lists of the ray are
// staged in LDS; the merged list is ordered by (t, concatenation index) - a rank merge of the already sorted lists, a
// bitonic network as the fallback - i.e. the merge is STABLE in object order, which is how ties are defined for this
// renderer (the reference calls torch.sort without stable=True; ties only occur between samples that carry zero alpha,
// see DESIGN.md).  The per-sample 192-channel features are never materialised per ray: they are read exactly once
// from the compact MLP output rows of the in-box samples and accumulated against both the per-object and the global
// weights.  The exclusive cumulative product of the transmittances is a wave-level multiplicative scan (the
// reference's cumprod is sequential; the association differs at the ulp level only).
#include "pr_common.h"
#include "composite_dev.h"

namespace pr {

struct CompositeSmem {
    float* tt;      // t per concatenated entry (after the overlap fix)
    float* sg;      // raw sigma per entry (after the overlap fix)
    float* dm;      // |displacement| per entry (after the overlap fix)
    float* wo;      // per-object weight per entry
    float* wg;      // global weight per entry (indexed by concatenation index)
    float* al;      // alpha scratch
    float* dv;      // |divergence estimate| per entry (after the overlap fix)
    int* sl;        // compact feature row per entry (-1: outside the box)
    unsigned int* key;   // entry index per merged rank
};

constexpr int MAX_FCHUNK = 4;  // F <= 256 channels, 64 lanes

// weights[j] = alpha[j] * prod_{i<j} (1 - alpha[i] + 1e-10)  over al[0..n), in place.
// Blocks of 64 consecutive entries are scanned across the wave (log steps), the running product is
// carried from block to block.  (The reference's cumprod is sequential; the association differs at
// the ulp level only.)
__device__ __forceinline__ void transmittance_weights(float* al, int n, int lane) {
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
        if (j < n) al[j] = a * (carry * excl);
        carry *= __shfl(incl, 63, 64);
    }
}

#ifndef ROWS_IN_FLIGHT
#define ROWS_IN_FLIGHT 8
#endif

// weights[j] = alpha[j] * prod_{i<j} (...) over al[0..n) by CW waves: wave w scans the contiguous segment
// [w * seg, (w + 1) * seg) (seg a multiple of 64) from a unit carry, the segment products are exchanged through
// `totals` and folded in afterwards.
__device__ __forceinline__ float sigmoid_of(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int CW>
__device__ __forceinline__ void transmittance_weights_block(float* al, int n, float* totals, int wave, int lane) {
    if (CW == 1) {
        transmittance_weights(al, n, lane);
        return;
    }
    const int seg = (((n + CW - 1) / CW) + 63) & ~63;
    const int begin = wave * seg;
    const int end = (begin + seg < n) ? begin + seg : n;
    float carry = 1.0f;
    for (int base = begin; base < end; base += 64) {
        const int j = base + lane;
        const float a = (j < end) ? al[j] : 0.f;
        float incl = (j < end) ? __fadd_rn(__fsub_rn(1.0f, a), 1e-10f) : 1.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(incl, d, 64);
            if (lane >= d) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        if (j < end) al[j] = a * (carry * excl);
        carry *= __shfl(incl, 63, 64);
    }
    if (lane == 0) totals[wave] = carry;
    __syncthreads();
    float before = 1.0f;
    for (int w = 0; w < wave; ++w) before *= totals[w];
    if (wave > 0)
        for (int j = begin + lane; j < end; j += 64) al[j] *= before;
    __syncthreads();
}

// sum over the workgroup of a per-thread value, in a fixed order (waves 0 .. CW - 1); every thread receives it
template <int CW>
__device__ __forceinline__ float block_sum(float v, float* scratch, int wave, int lane) {
    v = wave_sum(v);
    if (CW == 1) return v;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float total = 0.f;
    for (int w = 0; w < CW; ++w) total += scratch[w];
    return total;
}

// CW waves per ray.  Lists of a few hundred entries per ray keep the LDS footprint of a ray at tens of KB, i.e. a
// handful of rays per CU: with one wave per ray the latency-bound phases (staging, scans, merge) and the number of
// feature rows in flight were what set the kernel's time.  Wave w integrates the objects w, w + CW, ...; staging, the
// overlap fix, the merge, the global list and the feature rows are shared by all waves.
template <int CW>
__global__ __launch_bounds__(64 * CW) void k_composite(CompositeParams p) {
    extern __shared__ __attribute__((aligned(16))) char raw_smem[];
    constexpr int CT = 64 * CW;
    const int S = p.sort_size;
    CompositeSmem sm;
    const int A = (p.total_positions + 63) & ~63;   // per-entry arrays (the sort keys need the power of two S)
    sm.key = reinterpret_cast<unsigned int*>(raw_smem);
    sm.tt = reinterpret_cast<float*>(sm.key + S);
    sm.sg = sm.tt + A;
    sm.dm = sm.sg + A;
    sm.wo = sm.dm + A;
    sm.wg = sm.sg;       // the raw sigmas are dead once the merged list's alphas exist, which is before its weights do
    sm.al = sm.wo + A;
    // the divergence column exists only in differentiable training calls (keeps the eval footprint at 6 arrays)
    const bool use_div = p.any_divergence != 0;
    sm.dv = use_div ? sm.al + A : nullptr;
    sm.sl = reinterpret_cast<int*>(sm.al + (use_div ? 2 * A : A));
    float* scratch = reinterpret_cast<float*>(sm.sl + A);   // [CW * 64] cross-wave reductions
    // 64-bit sort scratch of the calls that always take the bitonic network (overlap fix), 8-byte aligned
    unsigned long long* wide = p.fix_overlaps ? reinterpret_cast<unsigned long long*>(scratch + 4 * 64) : nullptr;

    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction: as a scalar it keeps the per-object parameter reads (p.obj[k]) on the scalar path
    const int wave = (CW == 1) ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const long g = blockIdx.x;
    const float* d = p.ray_directions + (size_t)g * 3;
    // |d| of the world-frame direction (integrate receives the untransformed directions, :880/:886)
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const int PT = p.total_positions;

    // ---- stage every object's list ---------------------------------------------------------------
    {
        int off = 0;
        for (int k = 0; k < p.objects; ++k) {
            const CompositeObject& o = p.obj[k];
            const int P = o.positions;
            const size_t base = (size_t)g * P;
            for (int i = tid; i < P; i += CT) {
                sm.tt[off + i] = o.t[base + i];
                sm.sg[off + i] = o.sigma[base + i];
                sm.sl[off + i] = o.slot[base + i];
                sm.dm[off + i] = o.dispmag ? o.dispmag[base + i] : 0.f;
                if (use_div) sm.dv[off + i] = o.divergence ? fabsf(o.divergence[base + i]) : 0.f;
            }
            off += P;
        }
    }
    __syncthreads();

    // ---- integrate every object on its own: wave w takes the objects w, w + CW, ... ------------------
    // (wave-private data between the barriers: the alphas live in al[off ..), the object's own slice)
    for (int round = 0; round * CW < p.objects; ++round) {
        const int k = round * CW + wave;
        const bool mine = k < p.objects;
        int off = 0;
        for (int q = 0; q < k && q < p.objects; ++q) off += p.obj[q].positions;
        const CompositeObject& o = p.obj[mine ? k : 0];
        const int P = mine ? o.positions : 0;
        const size_t base = (size_t)g * P;
        float adiv = 0.f;   // sum alpha |div|  (object_composer.py:768-769, alphas detached)
        for (int i = lane; i < P; i += 64) {
            const float dt = (i < P - 1) ? __fsub_rn(sm.tt[off + i + 1], sm.tt[off + i]) : 1e10f;
            float raw = sm.sg[off + i];
            if (noise_present(o.noise)) raw = __fadd_rn(raw, noise_normal(o.noise, g, P, i));
            const float a = alpha_of(raw, __fmul_rn(dt, norm));
            sm.al[off + i] = a;
            if (use_div) adiv += a * sm.dv[off + i];
        }
        adiv = wave_sum(adiv);
        __syncthreads();
        if (mine) transmittance_weights(sm.al + off, P, lane);
        __syncthreads();
        float depth = 0.f, opacity = 0.f, dmag = 0.f;
        for (int i = lane; i < P; i += 64) {
            const float w = sm.al[off + i];
            sm.wo[off + i] = w;
            if (o.out.weights) o.out.weights[base + i] = w;
            depth += w * sm.tt[off + i];
            opacity += w;
            dmag += w * sm.dm[off + i];
        }
        depth = wave_sum(depth);
        opacity = wave_sum(opacity);
        dmag = wave_sum(dmag);
        if (mine && lane == 0) {
            if (o.out.depth) o.out.depth[g] = depth;
            if (o.out.opacity) o.out.opacity[g] = opacity;
            if (o.out.disparity) o.out.disparity[g] = disparity_of(depth, opacity);
            if (o.out.integrated_displacements_magnitude) o.out.integrated_displacements_magnitude[g] = dmag / (float)P;
            if (o.out.integrated_divergence) o.out.integrated_divergence[g] = adiv / (float)P;
        }
    }
    __syncthreads();

    // ---- static/dynamic overlap fix (fix_object_overlap, :295-397) ------------------------------
    // Static samples whose ORIGINAL t lies in [t_dyn[0], t_dyn[P_static - 1]) (searchsorted, left)
    // get sigma = -10, t = 0, displacement = 0.  The upper bound indexes the dynamic list with the
    // STATIC object's P - 1: reference behaviour, reproduced on purpose.
    if (p.fix_overlaps) {
        int dyn_off0 = 0;
        for (int k = 0; k < p.static_objects; ++k) dyn_off0 += p.obj[k].positions;
        int soff = 0;
        for (int s = 0; s < p.static_objects; ++s) {
            const int Ps = p.obj[s].positions;
            unsigned int masked_bits = 0;  // bit m <-> entry tid + CT m of this static list (Ps <= 2048)
            int doff = dyn_off0;
            for (int dd = p.static_objects; dd < p.objects; ++dd) {
                const float b0 = sm.tt[doff + 0];
                const float b1 = sm.tt[doff + Ps - 1];
                int lo0 = 0, hi0 = Ps;   // lower_bound(t_static, b0)
                while (lo0 < hi0) {
                    const int mid = (lo0 + hi0) >> 1;
                    if (sm.tt[soff + mid] < b0) lo0 = mid + 1; else hi0 = mid;
                }
                int lo1 = 0, hi1 = Ps;   // lower_bound(t_static, b1)
                while (lo1 < hi1) {
                    const int mid = (lo1 + hi1) >> 1;
                    if (sm.tt[soff + mid] < b1) lo1 = mid + 1; else hi1 = mid;
                }
                int m = 0;
                for (int i = tid; i < Ps; i += CT, ++m)
                    if (i >= lo0 && i < lo1) masked_bits |= 1u << m;
                doff += p.obj[dd].positions;
            }
            __syncthreads();  // every thread has finished searching the ORIGINAL t of this list
            int m = 0;
            for (int i = tid; i < Ps; i += CT, ++m) {
                if ((masked_bits >> m) & 1u) {
                    sm.tt[soff + i] = 0.f;
                    sm.sg[soff + i] = -10.0f;
                    sm.dm[soff + i] = 0.f;
                    if (use_div) sm.dv[soff + i] = 0.f;
                }
            }
            __syncthreads();
            soff += Ps;
        }
    }

    // ---- merge: order by (t, concatenation index) ------------------------------------------------
    {
        int counts[PR_MAX_OBJECTS];
        for (int k = 0; k < p.objects; ++k) counts[k] = p.obj[k].positions;
#if defined(PR_COMPOSITE_ABLATE) && PR_COMPOSITE_ABLATE >= 2
        for (int e = tid; e < PT; e += CT) sm.key[e] = (unsigned int)e;   // measurement build: no merge (wrong order)
        __syncthreads();
#else
        order_entries(sm.key, sm.tt, counts, p.objects, PT, S, !p.fix_overlaps, tid, CT, wide);
#endif
    }

    // ---- global alphas / weights in sorted order -------------------------------------------------
    const size_t gbase = (size_t)g * PT;
    float gdiv = 0.f;
    for (int j = tid; j < PT; j += CT) {
        const int e = (int)sm.key[j];
        float dt = 1e10f;
        if (j < PT - 1) {
            const int en = (int)sm.key[j + 1];
            dt = __fsub_rn(sm.tt[en], sm.tt[e]);
        }
        float raw = sm.sg[e];
        if (noise_present(p.noise_global)) raw = __fadd_rn(raw, noise_normal(p.noise_global, g, PT, j));
        sm.al[j] = alpha_of(raw, __fmul_rn(dt, norm));
        if (use_div) gdiv += sm.al[j] * sm.dv[e];
    }
    __syncthreads();
    transmittance_weights_block<CW>(sm.al, PT, scratch, wave, lane);   // al now holds the sorted-order weights
    if (CW == 1) __syncthreads();
    {
        float depth = 0.f, opacity = 0.f, dmag = 0.f;
        for (int j = tid; j < PT; j += CT) {
            const int e = (int)sm.key[j];
            const float w = sm.al[j];
            sm.wg[e] = w;
            if (p.global.weights) p.global.weights[gbase + j] = w;
            depth += w * sm.tt[e];
            opacity += w;
            dmag += w * sm.dm[e];
        }
        depth = block_sum<CW>(depth, scratch, wave, lane);
        opacity = block_sum<CW>(opacity, scratch, wave, lane);
        dmag = block_sum<CW>(dmag, scratch, wave, lane);
        if (use_div) gdiv = block_sum<CW>(gdiv, scratch, wave, lane);
        if (tid == 0) {
            if (p.global.depth) p.global.depth[g] = depth;
            if (p.global.opacity) p.global.opacity[g] = opacity;
            if (p.global.disparity) p.global.disparity[g] = disparity_of(depth, opacity);
            if (p.global.integrated_displacements_magnitude)
                p.global.integrated_displacements_magnitude[g] = dmag / (float)PT;
            if (p.global.integrated_divergence) p.global.integrated_divergence[g] = gdiv / (float)PT;
        }
    }
    __syncthreads();

    // ---- features: one pass over the compact MLP rows --------------------------------------------
    // Per object the contributing samples (inside the box, non-zero weight) are compacted into per-wave lists (wave w
    // takes the w-th contiguous part of the object's samples; the sort keys are dead by now, their storage is reused),
    // then consumed ROWS_IN_FLIGHT rows at a time.  Every wave accumulates all F channels of its part; the parts are added
    // in wave order.
    const int F = p.F;
    int* lrow = reinterpret_cast<int*>(sm.key);   // the ranks and the depths are dead by now
    float* lw1 = sm.tt;
    float accg[MAX_FCHUNK];
#pragma unroll
    for (int c = 0; c < MAX_FCHUNK; ++c) accg[c] = 0.f;
    int off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeObject& o = p.obj[k];
        const int P = o.positions;
        const int part = (((P + CW - 1) / CW) + 63) & ~63;
        const int begin = wave * part;
        const int end = (begin + part < P) ? begin + part : P;
        __syncthreads();
        int count = 0;
        float bare1 = 0.f, bare2 = 0.f;   // PR_FLAG_SIGMOID_FEATURES: weights of the samples without a feature row (raw feature 0)
        for (int base = begin; base < end; base += 64) {
            const int i = base + lane;
            bool take = false;
            int row = -1;
            float w1 = 0.f, w2 = 0.f;
            if (i < end) {
                row = sm.sl[off + i];
                w1 = sm.wo[off + i];
                w2 = sm.wg[off + i];
                take = row >= 0 && (w1 != 0.f || w2 != 0.f);
                if (p.sigmoid && row < 0) {
                    bare1 += w1;
                    bare2 += w2;
                }
            }
            const unsigned long long m = __ballot(take);
            if (take) {
                const int pos = off + begin + count + __popcll(m & ((1ull << lane) - 1ull));
                lrow[pos] = row;
                lw1[pos] = w1;
                sm.al[pos] = w2;
            }
            count += __popcll(m);
        }
#if defined(PR_COMPOSITE_ABLATE) && PR_COMPOSITE_ABLATE >= 1
        count = 0;   // measurement build: no feature rows are read
#endif
        // (the list of a wave is written and read by that wave only: no barrier needed in between)
        const int* rows_w = lrow + off + begin;
        const float* w1_w = lw1 + off + begin;
        const float* w2_w = sm.al + off + begin;
        float acco[MAX_FCHUNK];
#pragma unroll
        for (int c = 0; c < MAX_FCHUNK; ++c) acco[c] = 0.f;
        int i = 0;
        for (; i + ROWS_IN_FLIGHT <= count; i += ROWS_IN_FLIGHT) {
            float v[ROWS_IN_FLIGHT][MAX_FCHUNK];
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                const float* f = o.feat + (size_t)rows_w[i + u] * F;
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {
                    const int ch = lane + 64 * c;
                    v[u][c] = (ch < F) ? f[ch] : 0.f;
                    if (p.sigmoid) v[u][c] = sigmoid_of(v[u][c]);
                }
            }
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                const float w1 = w1_w[i + u], w2 = w2_w[i + u];
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {
                    acco[c] = __fadd_rn(acco[c], __fmul_rn(w1, v[u][c]));
                    accg[c] = __fadd_rn(accg[c], __fmul_rn(w2, v[u][c]));
                }
            }
        }
        for (; i < count; ++i) {
            const float* f = o.feat + (size_t)rows_w[i] * F;
            const float w1 = w1_w[i], w2 = w2_w[i];
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                float v = (ch < F) ? f[ch] : 0.f;
                if (p.sigmoid) v = sigmoid_of(v);
                acco[c] = __fadd_rn(acco[c], __fmul_rn(w1, v));
                accg[c] = __fadd_rn(accg[c], __fmul_rn(w2, v));
            }
        }
        if (p.sigmoid) {      // sigmoid(0) = 0.5 for the samples that have no row; every lane owns channels, so all of them add
            bare1 = wave_sum(bare1);
            bare2 = wave_sum(bare2);
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                acco[c] = __fadd_rn(acco[c], 0.5f * bare1);
                accg[c] = __fadd_rn(accg[c], 0.5f * bare2);
            }
        }
        if (o.out.integrated_features) {
            if (CW > 1) {
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {      // one 64-channel chunk at a time: 1 KB of scratch
                    if (64 * c >= F) break;
                    __syncthreads();
                    scratch[wave * 64 + lane] = acco[c];
                    __syncthreads();
                    if (wave == 0)
                        for (int w = 1; w < CW; ++w) acco[c] = __fadd_rn(acco[c], scratch[w * 64 + lane]);
                }
            }
            if (wave == 0) {
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < F) o.out.integrated_features[(size_t)g * F + ch] = acco[c];
                }
            }
        }
        off += P;
    }
    if (p.global.integrated_features) {
        if (CW > 1) {
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                if (64 * c >= F) break;
                __syncthreads();
                scratch[wave * 64 + lane] = accg[c];
                __syncthreads();
                if (wave == 0)
                    for (int w = 1; w < CW; ++w) accg[c] = __fadd_rn(accg[c], scratch[w * 64 + lane]);
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                if (ch < F) p.global.integrated_features[(size_t)g * F + ch] = accg[c];
            }
        }
    }
    if (p.decoder.groups > 0) {
        // decoder layout: this ray is cell (r / width, r % width) of its group's grid; its channels go to the channels-first
        // map of the group (neighbouring rays = neighbouring workgroups fill neighbouring columns of every plane)
        if (CW > 1 && !p.global.integrated_features) {     // the cross-wave sums have not been formed yet
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                if (64 * c >= F) break;
                __syncthreads();
                scratch[wave * 64 + lane] = accg[c];
                __syncthreads();
                if (wave == 0)
                    for (int w = 1; w < CW; ++w) accg[c] = __fadd_rn(accg[c], scratch[w * 64 + lane]);
            }
        }
        if (wave == 0) {
            const int frame = (int)(g / p.rays);
            int r = (int)(g - (long)frame * p.rays);
            int grp = 0;
            while (grp < p.decoder.groups - 1 && r >= p.decoder.rays[grp]) r -= p.decoder.rays[grp++];
            const int c0 = p.decoder.channel_begin[grp], c1 = p.decoder.channel_end[grp];
            const size_t cells = (size_t)p.decoder.rays[grp];
            float* map = p.decoder.map[grp] + (size_t)frame * (c1 - c0) * cells + r;
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                if (ch >= c0 && ch < c1 && ch < F) map[(size_t)(ch - c0) * cells] = accg[c];
            }
        }
    }
}

int launch_composite(const CompositeParams& p, hipStream_t s) {
    PR_REQUIRE(p.F <= 64 * MAX_FCHUNK, "output_features %d exceeds %d", p.F, 64 * MAX_FCHUNK);
    PR_REQUIRE(p.sort_size >= p.total_positions && (p.sort_size & (p.sort_size - 1)) == 0, "bad sort size");
    for (int k = 0; k < p.objects; ++k)
        PR_REQUIRE(p.obj[k].positions <= 64 * 32, "positions per ray %d too large for the overlap mask", p.obj[k].positions);
    const size_t lds = (size_t)p.sort_size * 4 + (size_t)((p.total_positions + 63) & ~63) * (p.any_divergence ? 7 : 6) * 4 +
                       sizeof(float) * 4 * 64 + (p.fix_overlaps ? (size_t)p.sort_size * 8 : 0);
    PR_REQUIRE(lds <= 156 * 1024, "too many samples per ray for the compositing kernel (%d)", p.total_positions);
    if (p.decoder.groups > 0) {
        PR_REQUIRE(p.decoder.groups <= PR_MAX_DECODER_GROUPS, "decoder layout: %d groups (max %d)", p.decoder.groups, PR_MAX_DECODER_GROUPS);
        long sum = 0;
        for (int i = 0; i < p.decoder.groups; ++i) {
            PR_REQUIRE(p.decoder.rays[i] > 0 && p.decoder.width[i] > 0 && p.decoder.rays[i] % p.decoder.width[i] == 0,
                       "decoder layout: group %d has %d rays in rows of %d", i, p.decoder.rays[i], p.decoder.width[i]);
            PR_REQUIRE(p.decoder.channel_begin[i] >= 0 && p.decoder.channel_begin[i] < p.decoder.channel_end[i] && p.decoder.channel_end[i] <= p.F,
                       "decoder layout: group %d channel range [%d, %d) outside 0..%d", i, p.decoder.channel_begin[i], p.decoder.channel_end[i], p.F);
            PR_REQUIRE(p.decoder.map[i] != nullptr, "decoder layout: group %d has no map", i);
            sum += p.decoder.rays[i];
        }
        PR_REQUIRE(sum == p.rays, "decoder layout: the groups hold %ld rays, the call has %d", sum, p.rays);
    }
    const long total = (long)p.frames * p.rays;
    ProfileScope scope(1, s);
    // four waves per ray once the lists are long enough to keep them busy; short lists (a few dozen entries) stay on one
    if (p.total_positions >= 256) {
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(&k_composite<4>), 156 * 1024, nullptr));   // the block-wide vote of order_entries owns a little static LDS
        hipLaunchKernelGGL(k_composite<4>, dim3((unsigned)total), dim3(256), lds, s, p);
    } else {
        PR_TRY(prepare_kernel(reinterpret_cast<const void*>(&k_composite<1>), 156 * 1024, nullptr));
        hipLaunchKernelGGL(k_composite<1>, dim3((unsigned)total), dim3(64), lds, s, p);
    }
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
