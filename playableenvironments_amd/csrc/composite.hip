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

namespace pr {

struct CompositeSmem {
    float* tt;      // t per concatenated entry (after the overlap fix)
    float* sg;      // raw sigma per entry (after the overlap fix)
    float* dm;      // |displacement| per entry (after the overlap fix)
    float* wo;      // per-object weight per entry
    float* wg;      // global weight per entry (indexed by concatenation index)
    float* al;      // alpha scratch
    int* sl;        // compact feature row per entry (-1: outside the box)
    unsigned long long* key;
};

__device__ __forceinline__ unsigned int float_order_bits(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// alpha = 1 - exp(-relu(raw) * dist)   (object_composer.py:197)
__device__ __forceinline__ float alpha_of(float raw, float dist) {
    const float relu = raw > 0.f ? raw : 0.f;
    return __fsub_rn(1.0f, expf(__fmul_rn(-relu, dist)));
}

// disparity = 1 / clamp(depth / opacity, min=1e-10), NaN-propagating  (object_composer.py:765)
__device__ __forceinline__ float disparity_of(float depth, float opacity) {
    float q = __fdiv_rn(depth, opacity);
    if (q < 1e-10f) q = 1e-10f;
    return __fdiv_rn(1.0f, q);
}

constexpr int MAX_FCHUNK = 4;  // F <= 256 channels, 64 lanes

__global__ __launch_bounds__(64) void k_composite(CompositeParams p) {
    extern __shared__ __attribute__((aligned(16))) char raw_smem[];
    const int S = p.sort_size;
    CompositeSmem sm;
    sm.key = reinterpret_cast<unsigned long long*>(raw_smem);
    sm.tt = reinterpret_cast<float*>(sm.key + S);
    sm.sg = sm.tt + S;
    sm.dm = sm.sg + S;
    sm.wo = sm.dm + S;
    sm.wg = sm.wo + S;
    sm.al = sm.wg + S;
    sm.sl = reinterpret_cast<int*>(sm.al + S);

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
        }
        __syncthreads();
        for (int i = lane; i < P; i += 64) {
            const float dt = (i < P - 1) ? __fsub_rn(sm.tt[off + i + 1], sm.tt[off + i]) : 1e10f;
            float raw = sm.sg[off + i];
            if (o.noise) raw = __fadd_rn(raw, o.noise[base + i]);
            sm.al[i] = alpha_of(raw, __fmul_rn(dt, norm));
        }
        __syncthreads();
        if (lane == 0) {
            float trans = 1.0f;
            for (int i = 0; i < P; ++i) {
                const float a = sm.al[i];
                sm.wo[off + i] = __fmul_rn(a, trans);
                trans = __fmul_rn(trans, __fadd_rn(__fsub_rn(1.0f, a), 1e-10f));
            }
        }
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
            if (o.out.integrated_divergence) o.out.integrated_divergence[g] = 0.f;
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
                }
            }
            __syncthreads();
            soff += Ps;
        }
    }

    // ---- merge: sort (t, concatenation index) ----------------------------------------------------
    for (int e = lane; e < S; e += 64) {
        sm.key[e] = (e < PT) ? (((unsigned long long)float_order_bits(sm.tt[e]) << 32) | (unsigned int)e)
                             : 0xFFFFFFFFFFFFFFFFull;
    }
    __syncthreads();
    for (int kk = 2; kk <= S; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < S; i += 64) {
                const int x = i ^ j;
                if (x > i) {
                    const unsigned long long a = sm.key[i], b = sm.key[x];
                    const bool up = ((i & kk) == 0);
                    if ((a > b) == up) {
                        sm.key[i] = b;
                        sm.key[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- global alphas / weights in sorted order -------------------------------------------------
    const size_t gbase = (size_t)g * PT;
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
    }
    __syncthreads();
    if (lane == 0) {
        float trans = 1.0f;
        for (int j = 0; j < PT; ++j) {
            const float a = sm.al[j];
            sm.al[j] = __fmul_rn(a, trans);   // al now holds the sorted-order weights
            trans = __fmul_rn(trans, __fadd_rn(__fsub_rn(1.0f, a), 1e-10f));
        }
    }
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
            if (p.global.integrated_divergence) p.global.integrated_divergence[g] = 0.f;
        }
    }
    __syncthreads();

    // ---- features: one pass over the compact MLP rows --------------------------------------------
    const int F = p.F;
    float accg[MAX_FCHUNK];
#pragma unroll
    for (int c = 0; c < MAX_FCHUNK; ++c) accg[c] = 0.f;
    off = 0;
    for (int k = 0; k < p.objects; ++k) {
        const CompositeObject& o = p.obj[k];
        const int P = o.positions;
        float acco[MAX_FCHUNK];
#pragma unroll
        for (int c = 0; c < MAX_FCHUNK; ++c) acco[c] = 0.f;
        for (int i = 0; i < P; ++i) {
            const int row = sm.sl[off + i];
            if (row < 0) continue;
            const float w1 = sm.wo[off + i], w2 = sm.wg[off + i];
            if (w1 == 0.f && w2 == 0.f) continue;
            const float* f = o.feat + (size_t)row * F;
#pragma unroll
            for (int c = 0; c < MAX_FCHUNK; ++c) {
                const int ch = lane + 64 * c;
                if (ch < F) {
                    const float v = f[ch];
                    acco[c] = __fadd_rn(acco[c], __fmul_rn(w1, v));
                    accg[c] = __fadd_rn(accg[c], __fmul_rn(w2, v));
                }
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
    const size_t lds = (size_t)p.sort_size * (8 + 7 * 4);
    PR_REQUIRE(lds <= 160 * 1024, "too many samples per ray for the compositing kernel (%d)", p.total_positions);
    static bool attr_set = false;
    if (!attr_set) {
        PR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_composite),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const long total = (long)p.frames * p.rays;
    ProfileScope scope(1, s);
    hipLaunchKernelGGL(k_composite, dim3((unsigned)total), dim3(64), lds, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr
