// Alpha compositing: per-object integration and the cross-object merge (ObjectComposer.integrate,
// compose, fix_object_overlap; model/object_composer.py:153-214, :295-447, :724-784).
//
// One 64-lane workgroup per ray.  All K per-object sample lists of the ray are staged in LDS; the
// merged list is ordered by a bitonic sort on (t, concatenation index) - i.e. the merge is STABLE
// in object order, which is how ties are defined for this renderer (the reference calls torch.sort
// without stable=True; ties only occur between samples that carry zero alpha).
// The per-sample 192-channel features are never materialised per ray: they are read exactly once
// from the compact MLP output rows of the in-box samples and accumulated against both the
// per-object and the global weights.  Sequential quantities (the exclusive cumulative product of
// transmittances) are evaluated by one lane in the reference's left-to-right order.
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
    unsigned long long* key;
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

__global__ __launch_bounds__(64) void k_composite(CompositeParams p) {
    extern __shared__ __attribute__((aligned(16))) char raw_smem[];
    const int S = p.sort_size;
    CompositeSmem sm;
    const int A = (p.total_positions + 63) & ~63;   // per-entry arrays (the sort keys need the power of two S)
    sm.key = reinterpret_cast<unsigned long long*>(raw_smem);
    sm.tt = reinterpret_cast<float*>(sm.key + S);
    sm.sg = sm.tt + A;
    sm.dm = sm.sg + A;
    sm.wo = sm.dm + A;
    sm.wg = sm.wo + A;
    sm.al = sm.wg + A;
    // the divergence column exists only in differentiable training calls (keeps the eval footprint at 7 arrays)
    const bool use_div = p.any_divergence != 0;
    sm.dv = use_div ? sm.al + A : nullptr;
    sm.sl = reinterpret_cast<int*>(sm.al + (use_div ? 2 * A : A));

    const int lane = threadIdx.x;
    const long g = blockIdx.x;
    const float* d = p.ray_directions + (size_t)g * 3;
    // |d| of the world-frame direction (integrate receives the untransformed directions, :880/:886)
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const int PT = p.total_positions;

    // ---- stage every object's list, integrate it on its own -----------------------------------
    int off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeObject& o = p.obj[k];
        const int P = o.positions;
        const size_t base = (size_t)g * P;
        for (int i = lane; i < P; i += 64) {
            sm.tt[off + i] = o.t[base + i];
            sm.sg[off + i] = o.sigma[base + i];
            sm.sl[off + i] = o.slot[base + i];
            sm.dm[off + i] = o.dispmag ? o.dispmag[base + i] : 0.f;
            if (use_div) sm.dv[off + i] = o.divergence ? fabsf(o.divergence[base + i]) : 0.f;
        }
        __syncthreads();
        float adiv = 0.f;   // sum alpha |div|  (object_composer.py:768-769, alphas detached)
        for (int i = lane; i < P; i += 64) {
            const float dt = (i < P - 1) ? __fsub_rn(sm.tt[off + i + 1], sm.tt[off + i]) : 1e10f;
            float raw = sm.sg[off + i];
            if (o.noise) raw = __fadd_rn(raw, o.noise[base + i]);
            sm.al[i] = alpha_of(raw, __fmul_rn(dt, norm));
            if (use_div) adiv += sm.al[i] * sm.dv[off + i];
        }
        adiv = wave_sum(adiv);
        __syncthreads();
        transmittance_weights(sm.al, P, lane);
        __syncthreads();
        for (int i = lane; i < P; i += 64) sm.wo[off + i] = sm.al[i];
        __syncthreads();
        float depth = 0.f, opacity = 0.f, dmag = 0.f;
        for (int i = lane; i < P; i += 64) {
            const float w = sm.wo[off + i];
            if (o.out.weights) o.out.weights[base + i] = w;
            depth += w * sm.tt[off + i];
            opacity += w;
            dmag += w * sm.dm[off + i];
        }
        depth = wave_sum(depth);
        opacity = wave_sum(opacity);
        dmag = wave_sum(dmag);
        if (lane == 0) {
            if (o.out.depth) o.out.depth[g] = depth;
            if (o.out.opacity) o.out.opacity[g] = opacity;
            if (o.out.disparity) o.out.disparity[g] = disparity_of(depth, opacity);
            if (o.out.integrated_displacements_magnitude) o.out.integrated_displacements_magnitude[g] = dmag / (float)P;
            if (o.out.integrated_divergence) o.out.integrated_divergence[g] = adiv / (float)P;
        }
        off += P;
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
            unsigned int masked_bits = 0;  // bit m <-> entry lane + 64 m of this static list (Ps <= 2048)
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
                for (int i = lane; i < Ps; i += 64, ++m)
                    if (i >= lo0 && i < lo1) masked_bits |= 1u << m;
                doff += p.obj[dd].positions;
            }
            __syncthreads();  // every lane has finished searching the ORIGINAL t of this list
            int m = 0;
            for (int i = lane; i < Ps; i += 64, ++m) {
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
        for (int e = lane; e < PT; e += 64) sm.key[e] = (unsigned long long)e;   // measurement build: no merge (wrong order)
        __syncthreads();
#else
        order_entries(sm.key, sm.tt, counts, p.objects, PT, S, !p.fix_overlaps, lane);
#endif
    }

    // ---- global alphas / weights in sorted order -------------------------------------------------
    const size_t gbase = (size_t)g * PT;
    float gdiv = 0.f;
    for (int j = lane; j < PT; j += 64) {
        const int e = (int)(sm.key[j] & 0xFFFFFFFFu);
        float dt = 1e10f;
        if (j < PT - 1) {
            const int en = (int)(sm.key[j + 1] & 0xFFFFFFFFu);
            dt = __fsub_rn(sm.tt[en], sm.tt[e]);
        }
        float raw = sm.sg[e];
        if (p.noise_global) raw = __fadd_rn(raw, p.noise_global[gbase + j]);
        sm.al[j] = alpha_of(raw, __fmul_rn(dt, norm));
        if (use_div) gdiv += sm.al[j] * sm.dv[e];
    }
    gdiv = wave_sum(gdiv);
    __syncthreads();
    transmittance_weights(sm.al, PT, lane);   // al now holds the sorted-order weights
    __syncthreads();
    {
        float depth = 0.f, opacity = 0.f, dmag = 0.f;
        for (int j = lane; j < PT; j += 64) {
            const int e = (int)(sm.key[j] & 0xFFFFFFFFu);
            const float w = sm.al[j];
            sm.wg[e] = w;
            if (p.global.weights) p.global.weights[gbase + j] = w;
            depth += w * sm.tt[e];
            opacity += w;
            dmag += w * sm.dm[e];
        }
        depth = wave_sum(depth);
        opacity = wave_sum(opacity);
        dmag = wave_sum(dmag);
        if (lane == 0) {
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
    // Per object the contributing samples (inside the box, non-zero weight) are first compacted into a
    // list (the sort keys are dead by now, their storage is reused), then consumed four rows at a time
    // so that several 768-byte row reads are in flight per wave.  Sums run in sample order.
    const int F = p.F;
    int* lrow = reinterpret_cast<int*>(sm.key);
    float* lw1 = reinterpret_cast<float*>(sm.key) + S;
    float accg[MAX_FCHUNK];
#pragma unroll
    for (int c = 0; c < MAX_FCHUNK; ++c) accg[c] = 0.f;
    off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeObject& o = p.obj[k];
        const int P = o.positions;
        __syncthreads();
        int count = 0;
        for (int base = 0; base < P; base += 64) {
            const int i = base + lane;
            bool take = false;
            int row = -1;
            float w1 = 0.f, w2 = 0.f;
            if (i < P) {
                row = sm.sl[off + i];
                w1 = sm.wo[off + i];
                w2 = sm.wg[off + i];
                take = row >= 0 && (w1 != 0.f || w2 != 0.f);
            }
            const unsigned long long m = __ballot(take);
            if (take) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                lrow[pos] = row;
                lw1[pos] = w1;
                sm.al[pos] = w2;
            }
            count += __popcll(m);
        }
        __syncthreads();
#if defined(PR_COMPOSITE_ABLATE) && PR_COMPOSITE_ABLATE >= 1
        count = 0;   // measurement build: no feature rows are read
#endif
        float acco[MAX_FCHUNK];
#pragma unroll
        for (int c = 0; c < MAX_FCHUNK; ++c) acco[c] = 0.f;
        int i = 0;
        // ROWS_IN_FLIGHT compact rows (768 B each) are requested before the first is consumed: a ray's LDS footprint
        // leaves room for ~5 waves per CU only, so the bytes in flight have to come from the depth of each wave's queue
        for (; i + ROWS_IN_FLIGHT <= count; i += ROWS_IN_FLIGHT) {
            float v[ROWS_IN_FLIGHT][MAX_FCHUNK];
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                const float* f = o.feat + (size_t)lrow[i + u] * F;
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {
                    const int ch = lane + 64 * c;
                    v[u][c] = (ch < F) ? f[ch] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
                const float w1 = lw1[i + u], w2 = sm.al[i + u];
#pragma unroll
                for (int c = 0; c < MAX_FCHUNK; ++c) {
                    acco[c] = __fadd_rn(acco[c], __fmul_rn(w1, v[u][c]));
                    accg[c] = __fadd_rn(accg[c], __fmul_rn(w2, v[u][c]));
                }
            }
        }
        for (; i < count; ++i) {
            const float* f = o.feat + (size_t)lrow[i] * F;
            const float w1 = lw1[i], w2 = sm.al[i];
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                const float v = (ch < F) ? f[ch] : 0.f;
                acco[c] = __fadd_rn(acco[c], __fmul_rn(w1, v));
                accg[c] = __fadd_rn(accg[c], __fmul_rn(w2, v));
            }
        }
        if (o.out.integrated_features) {
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                if (ch < F) o.out.integrated_features[(size_t)g * F + ch] = acco[c];
            }
        }
        off += P;
    }
    if (p.global.integrated_features) {
#pragma unroll
        for (int c = 0; c < MAX_FCHUNK; ++c) {
            const int ch = lane + 64 * c;
            if (ch < F) p.global.integrated_features[(size_t)g * F + ch] = accg[c];
        }
    }
}

int launch_composite(const CompositeParams& p, hipStream_t s) {
    PR_REQUIRE(p.F <= 64 * MAX_FCHUNK, "output_features %d exceeds %d", p.F, 64 * MAX_FCHUNK);
    PR_REQUIRE(p.sort_size >= p.total_positions && (p.sort_size & (p.sort_size - 1)) == 0, "bad sort size");
    for (int k = 0; k < p.objects; ++k)
        PR_REQUIRE(p.obj[k].positions <= 64 * 32, "positions per ray %d too large for the overlap mask", p.obj[k].positions);
    const size_t lds = (size_t)p.sort_size * 8 + (size_t)((p.total_positions + 63) & ~63) * (p.any_divergence ? 8 : 7) * 4;
    PR_REQUIRE(lds <= 160 * 1024, "too many samples per ray for the compositing kernel (%d)", p.total_positions);
    PR_TRY(prepare_kernel(reinterpret_cast<const void*>(k_composite), 160 * 1024, nullptr));
    const long total = (long)p.frames * p.rays;
    ProfileScope scope(1, s);
    hipLaunchKernelGGL(k_composite, dim3((unsigned)total), dim3(64), lds, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
