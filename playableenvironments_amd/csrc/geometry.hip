// Geometry stages of the renderer: camera rays, per-object sample placement (slab test +
// stratified t), AABB cull + deterministic compaction, hierarchical (inverse-CDF) resampling.
//
// All arithmetic that feeds a discontinuous decision (slab hit/miss, in-box tests) is written with
// explicit round-to-nearest mul/add/div in exactly the operation order of the reference's tensor
// expressions, so the decisions are bit-identical to the fp32 PyTorch path (this file is built with
// -ffp-contract=off).  Reference lines are cited at each step (paths relative to the reference).
#include "pr_common.h"

namespace pr {

// ---------------------------------------------------------------------------------------------
// Camera rays: RayHelper.create_camera_rays (utils/lib_3d/ray_helper.py:15-52), pixel selection
// (:433-482 / all pixels) and transform_rays with the camera-to-world matrix (:1203-1227).
// ---------------------------------------------------------------------------------------------
__global__ void k_camera_rays(int frames, int rays, int per_frame, float half_h, float half_w, const float* __restrict__ c2w,
                              const float* __restrict__ focals, const int32_t* __restrict__ rows,
                              const int32_t* __restrict__ cols, float* __restrict__ origins,
                              float* __restrict__ dirs, float* __restrict__ normals) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)frames * rays;
    if (g >= total) return;
    const int n = (int)(g / rays);
    const int r = (int)(g - (long)n * rays);
    const float* m = c2w + (size_t)n * 12;
    const float f = focals[n];
    const long pix = per_frame ? g : (long)r;
    const float dx = __fdiv_rn(__fsub_rn((float)cols[pix], half_w), f);
    const float dy = -__fdiv_rn(__fsub_rn((float)rows[pix], half_h), f);
    const float dz = -1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v = __fadd_rn(__fadd_rn(__fmul_rn(dx, m[i * 4 + 0]), __fmul_rn(dy, m[i * 4 + 1])),
                                  __fmul_rn(dz, m[i * 4 + 2]));
        dirs[g * 3 + i] = v;
    }
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            origins[n * 3 + i] = m[i * 4 + 3];
            normals[n * 3 + i] = -m[i * 4 + 2];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Block-level helpers (256 threads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan over the 256 threads of a block; `total` = block sum.  lds: >= 4 ints.
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// Per-ray slab test, model/object_composer.py:104-151 + clamp :522-523
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ray_bounds(const ObjRay& ray, const float* lo, const float* hi, bool valid,
                                           float zmin, float zmax, float* near_out, float* far_out) {
    float z_near = 0.f, z_far = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float den = __fadd_rn(ray.d[a], 1e-6f);
        const float z0 = __fdiv_rn(__fsub_rn(lo[a], ray.o[a]), den);
        const float z1 = __fdiv_rn(__fsub_rn(hi[a], ray.o[a]), den);
        const float mn = nan_min(z0, z1);
        const float mx = nan_max(z0, z1);
        if (a == 0) {
            z_near = mn;
            z_far = mx;
        } else {
            z_near = nan_max(z_near, mn);
            z_far = nan_min(z_far, mx);
        }
    }
    if ((z_far <= z_near) || !valid) {
        z_far = 0.f;
        z_near = 0.f;
    }
    *near_out = nan_clamp(z_near, zmin, zmax);
    *far_out = nan_clamp(z_far, zmin, zmax);
}

// ---------------------------------------------------------------------------------------------
// Coarse placement: RayHelper.create_ray_positions (utils/lib_3d/ray_helper.py:1229-1282).
// Phase 1: one lane per ray (slab test).  Phase 2: the wave walks its 64 rays together, lanes across the samples of
// one ray, so that t / sigma / displacement rows are written as whole cache lines (a lane-per-ray walk writes one
// float per line and lane: ~20x write amplification measured with WRITE_SIZE).  Also counts the in-box samples per
// 256-ray block for the compaction.
// ---------------------------------------------------------------------------------------------
struct WaveRay {
    float o[3], d[3];
    int valid;
};

__device__ __forceinline__ WaveRay broadcast_ray(const ObjRay& ray, bool valid, int src) {
    WaveRay w;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        w.o[a] = __shfl(ray.o[a], src, 64);
        w.d[a] = __shfl(ray.d[a], src, 64);
    }
    w.valid = __shfl((int)valid, src, 64);
    return w;
}

// Lane layout of the cooperative phase: P2 = min(64, next power of two >= P) lanes per ray, 64 / P2 rays per pass, so
// that short sample lists (16, 32 positions per ray) still use the whole wave.
struct RayLanes {
    int P2, per_pass, sub, idx;
    unsigned long long sub_mask;
    __device__ __forceinline__ RayLanes(int P, int lane) {
        P2 = 1;
        while (P2 < P && P2 < 64) P2 <<= 1;
        per_pass = 64 / P2;
        sub = lane / P2;
        idx = lane - sub * P2;
        sub_mask = (P2 == 64) ? ~0ull : ((1ull << P2) - 1ull);
    }
    // the bits of a wave ballot that belong to this lane's ray
    __device__ __forceinline__ unsigned long long mine(unsigned long long ballot) const { return (ballot >> (sub * P2)) & sub_mask; }
    // hands the per-ray value held by the lanes of ray (r0 + s) to lane r0 + s (the lane that owns that ray in phase 1)
    __device__ __forceinline__ void collect(int value, int r0, int lane, int* owner_value) const {
        const int v = __shfl(value, ((lane - r0) & (per_pass - 1)) * P2, 64);
        if (lane >= r0 && lane < r0 + per_pass) *owner_value = v;
    }
};

__device__ __forceinline__ void place_coarse_body(const PlaceParams& p) {
    __shared__ int lds[4];
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.frames * p.rays;
    ObjRay ray = {};
    bool valid = false;
    float z_near = 0.f, z_far = 0.f;
    if (g < total) {
        const int n = (int)(g / p.rays);
        const float* m = p.w2o + ((size_t)n * p.objects + p.object_index) * 12;
        ray = object_ray(m, p.ray_origins + (size_t)n * 3, p.ray_directions + (size_t)g * 3);
        valid = p.in_scene[(size_t)n * p.objects + p.object_index] != 0;
        ray_bounds(ray, p.lo, p.hi, valid, p.z_near_min, p.z_far_max, &z_near, &z_far);
    }
    const int P = p.positions;
    const long wave_first = g - lane;
    const RayLanes L(P, lane);
    int count = 0;
    for (int r0 = 0; r0 < 64 && wave_first + r0 < total; r0 += L.per_pass) {
        const int r = r0 + L.sub;
        const bool live = wave_first + r < total;
        const WaveRay w = broadcast_ray(ray, valid, r);
        const float zn = __shfl(z_near, r, 64), zf = __shfl(z_far, r, 64);
        const size_t base = (size_t)(wave_first + r) * P;
        // t_i = near * (1 - s_i) + far * s_i
        auto t_at = [&](int i) {
            const float s = p.linspace[i];
            return __fadd_rn(__fmul_rn(zn, __fsub_rn(1.0f, s)), __fmul_rn(zf, s));
        };
        int ray_count = 0;
        for (int i0 = 0; i0 < P; i0 += L.P2) {
            const int i = i0 + L.idx;
            bool inside = false;
            if (live && i < P) {
                const float t_cur = t_at(i);
                float t = t_cur;
                if (noise_present(p.jitter)) {
                    // mid points, upper = [mids, t_last], lower = [t_0, mids]   (:1267-1275)
                    const float upper = (i < P - 1) ? __fdiv_rn(__fadd_rn(t_at(i + 1), t_cur), 2.0f) : t_cur;
                    const float lower = (i > 0) ? __fdiv_rn(__fadd_rn(t_cur, t_at(i - 1)), 2.0f) : t_cur;
                    t = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), noise_uniform(p.jitter, wave_first + r, P, i)));
                }
                p.t[base + i] = t;
                p.sigma[base + i] = p.empty_alpha;
                if (p.dispmag) p.dispmag[base + i] = 0.f;
                const float x = __fadd_rn(w.o[0], __fmul_rn(w.d[0], t));
                const float y = __fadd_rn(w.o[1], __fmul_rn(w.d[1], t));
                const float z = __fadd_rn(w.o[2], __fmul_rn(w.d[2], t));
                inside = in_box(x, y, z, p.lo, p.hi);   // absent objects too: the reference evaluates their samples (near = far = 0, clamped) and
                                                        // only overrides their densities afterwards (object_composer.py:546-547)
            }
            ray_count += __popcll(L.mine(__ballot(inside)));
        }
        L.collect(ray_count, r0, lane, &count);
    }
    int block_total;
    block_exclusive_scan_256(count, lds, &block_total);
    if (threadIdx.x == 0) p.block_sums[blockIdx.x] = block_total;
}

__global__ __launch_bounds__(256) void k_place_coarse(PlaceParams p) { place_coarse_body(p); }
// the objects of a call in one launch: blockIdx.y = object
struct PlaceGroup { PlaceParams p[PR_MAX_OBJECTS]; };
__global__ __launch_bounds__(256) void k_place_coarse_group(PlaceGroup g) { place_coarse_body(g.p[blockIdx.y]); }

int launch_place_coarse(const PlaceParams& p, hipStream_t s) {
    const long total = (long)p.frames * p.rays;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_place_coarse, dim3(blocks), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of the per-block in-box counts (single workgroup; n is a few thousand at most).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_blocks_body(const int32_t* __restrict__ sums, int32_t* __restrict__ offsets,
                                                 int32_t* __restrict__ total, int n) {
    __shared__ int lds[4];
    int carry = 0;
    for (int start = 0; start < n; start += 256) {
        const int i = start + threadIdx.x;
        const int v = (i < n) ? sums[i] : 0;
        int chunk_total;
        const int ex = block_exclusive_scan_256(v, lds, &chunk_total);
        if (i < n) offsets[i] = carry + ex;
        carry += chunk_total;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_blocks(const int32_t* __restrict__ sums, int32_t* __restrict__ offsets,
                                                    int32_t* __restrict__ total, int n) {
    scan_blocks_body(sums, offsets, total, n);
}
struct ScanGroup { const int32_t* sums[PR_MAX_OBJECTS]; int32_t* offsets[PR_MAX_OBJECTS]; int32_t* total[PR_MAX_OBJECTS]; int n; };
__global__ __launch_bounds__(256) void k_scan_blocks_group(ScanGroup g) {
    scan_blocks_body(g.sums[blockIdx.x], g.offsets[blockIdx.x], g.total[blockIdx.x], g.n);
}

int launch_scan(const int32_t* sums, int32_t* offsets, int32_t* total, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, sums, offsets, total, n);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Compaction: the in-box samples, in flat (frame, ray, sample) order - the same order in which the
// reference's boolean-mask indexing compacts them (ray_bending_style_nerf_model.py:180-186).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fill_body(const FillParams& p) {
    __shared__ int lds[4];
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.frames * p.rays;
    const int P = p.positions;
    ObjRay ray = {};
    bool valid = false;
    if (g < total) {
        const int n = (int)(g / p.rays);
        const float* m = p.w2o + ((size_t)n * p.objects + p.object_index) * 12;
        ray = object_ray(m, p.ray_origins + (size_t)n * 3, p.ray_directions + (size_t)g * 3);
        valid = p.in_scene[(size_t)n * p.objects + p.object_index] != 0;
    }
    // lanes across the samples of a ray (coalesced reads of t, whole-line writes of slot / records)
    const long wave_first = g - lane;
    const RayLanes L(P, lane);
    auto walk = [&](auto&& per_ray) {
        for (int r0 = 0; r0 < 64 && wave_first + r0 < total; r0 += L.per_pass) {
            const int r = r0 + L.sub;
            per_ray(r0, r, wave_first + r < total, broadcast_ray(ray, valid, r), (size_t)(wave_first + r) * P);
        }
    };
    auto inside_at = [&](bool live, const WaveRay& w, size_t base, int i, float* x, float* y, float* z) {
        if (!live || i >= P) return false;
        const float t = p.t[base + i];
        *x = __fadd_rn(w.o[0], __fmul_rn(w.d[0], t));
        *y = __fadd_rn(w.o[1], __fmul_rn(w.d[1], t));
        *z = __fadd_rn(w.o[2], __fmul_rn(w.d[2], t));
        return in_box(*x, *y, *z, p.lo, p.hi);    // (absent objects included, see k_place_coarse)
    };
    int count = 0;
    walk([&](int r0, int r, bool live, const WaveRay& w, size_t base) {
        int ray_count = 0;
        for (int i0 = 0; i0 < P; i0 += L.P2) {
            float x, y, z;
            ray_count += __popcll(L.mine(__ballot(inside_at(live, w, base, i0 + L.idx, &x, &y, &z))));
        }
        L.collect(ray_count, r0, lane, &count);
    });
    int block_total;
    const int first_slot = p.block_offsets[blockIdx.x] + block_exclusive_scan_256(count, lds, &block_total);
    walk([&](int r0, int r, bool live, const WaveRay& w, size_t base) {
        int slot = __shfl(first_slot, r, 64);
        for (int i0 = 0; i0 < P; i0 += L.P2) {
            const int i = i0 + L.idx;
            float x = 0.f, y = 0.f, z = 0.f;
            const bool inside = inside_at(live, w, base, i, &x, &y, &z);
            const unsigned long long mask = L.mine(__ballot(inside));
            if (inside) {
                const int mine = slot + __popcll(mask & ((1ull << L.idx) - 1ull));
                p.rec_pos[(size_t)mine * 3 + 0] = x;
                p.rec_pos[(size_t)mine * 3 + 1] = y;
                p.rec_pos[(size_t)mine * 3 + 2] = z;
                p.rec_flat[mine] = (int32_t)(base + i);
                p.slot[base + i] = mine;
            } else if (live && i < P) {
                p.slot[base + i] = -1;
            }
            slot += __popcll(mask);
        }
    });
}

__global__ __launch_bounds__(256) void k_fill(FillParams p) { fill_body(p); }
struct FillGroup { FillParams p[PR_MAX_OBJECTS]; };
__global__ __launch_bounds__(256) void k_fill_group(FillGroup g) { fill_body(g.p[blockIdx.y]); }

// Coarse placement, block scan and compaction of `count` objects as three launches (each object with its own block sums /
// offsets and total): PlaceParams::block_sums, FillParams::block_offsets and totals[k] say where.
int launch_placement_group(const PlaceParams* pp, const FillParams* fp, int32_t* const* totals, int count, hipStream_t s) {
    if (count <= 0) return PR_OK;
    PR_REQUIRE(count <= PR_MAX_OBJECTS, "placement group: %d objects", count);
    static thread_local PlaceGroup pg;
    static thread_local FillGroup fg;
    ScanGroup sg;
    const long total = (long)pp[0].frames * pp[0].rays;
    const int blocks = (int)((total + 255) / 256);
    for (int k = 0; k < count; ++k) {
        pg.p[k] = pp[k];
        fg.p[k] = fp[k];
        sg.sums[k] = pp[k].block_sums;
        sg.offsets[k] = const_cast<int32_t*>(fp[k].block_offsets);
        sg.total[k] = totals[k];
    }
    sg.n = blocks;
    hipLaunchKernelGGL(k_place_coarse_group, dim3(blocks, count), dim3(256), 0, s, pg);
    PR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocks_group, dim3(count), dim3(256), 0, s, sg);
    PR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fill_group, dim3(blocks, count), dim3(256), 0, s, fg);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

int launch_fill(const FillParams& p, hipStream_t s) {
    const long total = (long)p.frames * p.rays;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

// ---------------------------------------------------------------------------------------------
// Hierarchical resampling: coarse alphas/weights (model/object_composer.py:552-554), inverse-CDF
// sampling (utils/lib_3d/ray_helper.py:1348-1403), merge + sort with the coarse t (:1337-1344).
// One 64-lane workgroup per ray.  Sequential parts (cumprod, cumsum) are done by one lane in the
// reference's left-to-right order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_f32(float* key, int n /*pow2*/, int lane) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n; i += 64) {
                const int x = i ^ j;
                if (x > i) {
                    const float a = key[i], b = key[x];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) {
                        key[i] = b;
                        key[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(64) void k_resample(ResampleParams p, int sort_size) {
    extern __shared__ float sm[];
    const int Pc = p.pc, Pf = p.pf;
    float* tc = sm;             // [Pc]
    float* al = tc + Pc;        // [Pc] alpha, then weights
    float* mids = al + Pc;      // [Pc]
    float* cdf = mids + Pc;     // [Pc]
    float* key = cdf + Pc;      // [sort_size]
    const int lane = threadIdx.x;
    const long g = blockIdx.x;
    const int n = (int)(g / p.rays);
    const float* m = p.w2o + ((size_t)n * p.objects + p.object_index) * 12;
    const ObjRay ray = object_ray(m, p.ray_origins + (size_t)n * 3, p.ray_directions + (size_t)g * 3);
    const bool valid = p.in_scene[(size_t)n * p.objects + p.object_index] != 0;
    // |d| in the object frame (forward_object passes the transformed directions, :552)
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ray.d[0], ray.d[0]), __fmul_rn(ray.d[1], ray.d[1])),
                                       __fmul_rn(ray.d[2], ray.d[2])));
    const size_t cbase = (size_t)g * Pc;
    for (int i = lane; i < Pc; i += 64) tc[i] = p.t_coarse[cbase + i];
    __syncthreads();
    for (int i = lane; i < Pc; i += 64) {
        float raw = valid ? p.sigma_coarse[cbase + i] : p.empty_alpha;
        if (noise_present(p.alpha_noise)) raw = __fadd_rn(raw, noise_normal(p.alpha_noise, g, Pc, i));
        const float dt = (i < Pc - 1) ? __fsub_rn(tc[i + 1], tc[i]) : 1e10f;
        const float dist = __fmul_rn(dt, norm);
        const float relu = raw > 0.f ? raw : 0.f;
        al[i] = __fsub_rn(1.0f, expf(__fmul_rn(-relu, dist)));
        if (i < Pc - 1) mids[i] = __fdiv_rn(__fadd_rn(tc[i + 1], tc[i]), 2.0f);
    }
    __syncthreads();
    if (lane == 0) {
        // weights = alpha * cumprod([1, 1 - alpha + 1e-10][:-1])
        float trans = 1.0f;
        for (int i = 0; i < Pc; ++i) {
            const float a = al[i];
            al[i] = __fmul_rn(a, trans);
            trans = __fmul_rn(trans, __fadd_rn(__fsub_rn(1.0f, a), 1e-10f));
        }
        // pdf over weights[1:-1] + 1e-5 ; cdf = [0, cumsum(pdf)]
        const int nb = Pc - 2;
        float sum = 0.f;
        for (int j = 0; j < nb; ++j) sum = __fadd_rn(sum, __fadd_rn(al[j + 1], 1e-5f));
        float c = 0.f;
        cdf[0] = 0.f;
        for (int j = 0; j < nb; ++j) {
            c = __fadd_rn(c, __fdiv_rn(__fadd_rn(al[j + 1], 1e-5f), sum));
            cdf[j + 1] = c;
        }
    }
    __syncthreads();
    const int ncdf = Pc - 1;
    for (int i = lane; i < sort_size; i += 64) key[i] = (i < Pc) ? tc[i] : __builtin_inff();
    for (int f = lane; f < Pf; f += 64) {
        const float u = noise_present(p.u_random) ? noise_uniform(p.u_random, g, Pf, f) : p.u_fixed[f];
        // searchsorted(cdf, u, right=True): first index with cdf[idx] > u
        int lo_i = 0, hi_i = ncdf;
        while (lo_i < hi_i) {
            const int mid = (lo_i + hi_i) >> 1;
            if (cdf[mid] > u) hi_i = mid; else lo_i = mid + 1;
        }
        const int below = lo_i - 1 < 0 ? 0 : lo_i - 1;
        const int above = lo_i > ncdf - 1 ? ncdf - 1 : lo_i;
        float den = __fsub_rn(cdf[above], cdf[below]);
        if (den < 1e-5f) den = 1.0f;
        const float frac = __fdiv_rn(__fsub_rn(u, cdf[below]), den);
        key[Pc + f] = __fadd_rn(mids[below], __fmul_rn(frac, __fsub_rn(mids[above], mids[below])));
    }
    __syncthreads();
    const int Pm = Pc + Pf;
    // Both lists are normally sorted already - the coarse depths by construction, the resampled ones whenever the
    // inverse CDF is evaluated at the fixed, increasing u (no perturbation) - and a merge by ranks costs a fraction of the
    // instructions of the bitonic network (this kernel is bound by VALU issue, most of it the network's).  The merged
    // VALUES are what the reference's sort returns, whatever it does with ties.
    float* merged = key + sort_size;
    bool sorted = true;
    for (int i = lane; i + 1 < Pc; i += 64) sorted = sorted && (key[i] <= key[i + 1]);
    for (int f = lane; f + 1 < Pf; f += 64) sorted = sorted && (key[Pc + f] <= key[Pc + f + 1]);
    if (__ballot(sorted) == ~0ull) {
        for (int i = lane; i < Pc; i += 64) {          // coarse entry i: ahead of it are the fine samples strictly below
            const float t = key[i];
            int lo = 0, hi = Pf;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (key[Pc + mid] < t) lo = mid + 1; else hi = mid;
            }
            merged[i + lo] = t;
        }
        for (int f = lane; f < Pf; f += 64) {          // fine sample f: ahead of it are the coarse depths below or equal
            const float t = key[Pc + f];
            int lo = 0, hi = Pc;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (key[mid] <= t) lo = mid + 1; else hi = mid;
            }
            merged[f + lo] = t;
        }
        __syncthreads();
    } else {
        bitonic_sort_f32(key, sort_size, lane);
        merged = key;
    }
    const size_t fbase = (size_t)g * Pm;
    int count = 0;
    for (int i = lane; i < Pm; i += 64) {
        const float t = merged[i];
        p.t_fine[fbase + i] = t;
        p.sigma_fine[fbase + i] = p.empty_alpha;
        if (p.dispmag_fine) p.dispmag_fine[fbase + i] = 0.f;
        const float x = __fadd_rn(ray.o[0], __fmul_rn(ray.d[0], t));
        const float y = __fadd_rn(ray.o[1], __fmul_rn(ray.d[1], t));
        const float z = __fadd_rn(ray.o[2], __fmul_rn(ray.d[2], t));
        if (in_box(x, y, z, p.lo, p.hi)) ++count;      // (absent objects included, see k_place_coarse)
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) count += __shfl_down(count, d, 64);
    if (lane == 0 && count) atomicAdd(&p.block_sums[g >> 8], count);
}

// expected[n][r] = sum_i w_i (o + d t_i + delta_i) / (sum_i w_i + 1e-8), torch op order of
// compute_expected_positions (object_composer.py:603-622): products first, then the sums over the samples
__global__ __launch_bounds__(256) void k_expected_positions(int frames, int rays, int objects, int object_index, int P,
                                                            const float* ray_origins, const float* ray_directions,
                                                            const float* w2o, const float* t, const float* weights,
                                                            const float* delta, float* expected) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long)frames * rays) return;
    const int n = (int)(g / rays);
    const ObjRay ray = object_ray(w2o + ((size_t)n * objects + object_index) * 12, ray_origins + (size_t)n * 3,
                                  ray_directions + (size_t)g * 3);
    const size_t base = (size_t)g * P;
    float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f;
    for (int i = 0; i < P; ++i) {
        const float ti = t[base + i], w = weights[base + i];
        for (int a = 0; a < 3; ++a) {
            float x = __fadd_rn(ray.o[a], __fmul_rn(ray.d[a], ti));
            if (delta) x = __fadd_rn(x, delta[(base + i) * 3 + a]);
            acc[a] = __fadd_rn(acc[a], __fmul_rn(x, w));
        }
        wsum = __fadd_rn(wsum, w);
    }
    for (int a = 0; a < 3; ++a) expected[(size_t)g * 3 + a] = __fdiv_rn(acc[a], __fadd_rn(wsum, 1e-8f));
}

static int next_pow2(int v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}

int launch_resample(const ResampleParams& p, hipStream_t s) {
    PR_REQUIRE(p.pc >= 3, "hierarchical sampling needs at least 3 coarse positions (got %d)", p.pc);
    const long total = (long)p.frames * p.rays;
    const int nblocks256 = (int)((total + 255) / 256);
    PR_TRY(launch_zero_fill(p.block_sums, sizeof(int32_t) * nblocks256, s));
    int sort_size = next_pow2(p.pc + p.pf);
    if (sort_size < 64) sort_size = 64;
    const size_t lds = sizeof(float) * (4 * (size_t)p.pc + 2 * (size_t)sort_size);   // inputs + sort keys + merged list
    PR_REQUIRE(lds <= 64 * 1024, "resample: too many positions per ray (%d + %d)", p.pc, p.pf);
    hipLaunchKernelGGL(k_resample, dim3((unsigned)total), dim3(64), lds, s, p, sort_size);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

}  // namespace pr

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int pr_expected_positions(int32_t frames, int32_t rays, int32_t objects, int32_t object_index, int32_t positions,
                                     const float* ray_origins, const float* ray_directions, const float* w2o, const float* t,
                                     const float* weights, const float* delta, float* expected, void* stream) {
    PR_REQUIRE(frames > 0 && rays > 0 && positions > 0 && objects > 0 && object_index >= 0 && object_index < objects,
               "pr_expected_positions: bad sizes");
    PR_REQUIRE(ray_origins && ray_directions && w2o && t && weights && expected, "pr_expected_positions: NULL pointer");
    const long total = (long)frames * rays;
    hipLaunchKernelGGL(pr::k_expected_positions, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames,
                       rays, objects, object_index, positions, ray_origins, ray_directions, w2o, t, weights, delta, expected);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

namespace pr {
// ---------------------------------------------------------------------------------------------
// RayHelper.sample_rays_strided_patch (utils/lib_3d/ray_helper.py:236-431) on the device: per frame one box-weighted random
// centre, clamped so that the patch stays inside the image and aligned to the grid of the largest stride, then a p_i x p_i pixel
// grid per stride (p_i = patch * s_0 / s_i), strides concatenated smallest first, row-major.  The reference builds the weight
// image (every object adds w_k / area_k over its pixel-aligned box), normalises, takes its cumulative sum and looks the draw up;
// the weight image is a sum of K box indicators, so its row-major prefix sum has a closed form and the lookup is a binary search
// with O(K) per probe - nothing of size H x W is touched, ~80 small tensor ops become one launch.
// ---------------------------------------------------------------------------------------------
struct PatchPixelsParams {
    int frames, objects, height, width, patch, nstrides;
    int strides[4];
    const float* boxes;      // (N, 4, K) normalised [left, top, right, bottom]
    const float* weights;    // (K)
    const float* u;          // (N) uniform draws
    int32_t* rows; int32_t* cols;   // (N, R)
    int rays;
};

__global__ __launch_bounds__(256) void k_patch_pixels(PatchPixelsParams p) {
    __shared__ int start[2];
    const int n = blockIdx.x;
    const int H = p.height, W = p.width, K = p.objects;
    const int s0 = p.strides[0], sm = p.strides[p.nstrides - 1];
    if (threadIdx.x < 64) {          // the first wave: every lane holds the boxes, the lookup probes 64 pixels at a time
        const int lane = threadIdx.x;
        float left[PR_MAX_OBJECTS], right[PR_MAX_OBJECTS], top[PR_MAX_OBJECTS], bottom[PR_MAX_OBJECTS], per[PR_MAX_OBJECTS];
        for (int k = 0; k < K; ++k) {
            const float* b = p.boxes + (size_t)n * 4 * K + k;
            left[k] = fminf(fmaxf(floorf(b[0] * W), 0.f), (float)W);
            top[k] = fminf(fmaxf(floorf(b[K] * H), 0.f), (float)H);
            right[k] = fminf(fmaxf(ceilf(b[2 * K] * W), 0.f), (float)W);
            bottom[k] = fminf(fmaxf(ceilf(b[3 * K] * H), 0.f), (float)H);
            per[k] = p.weights[k] / ((right[k] - left[k]) * (bottom[k] - top[k]));       // inf / nan for an empty box, as the reference
        }
        // inclusive prefix sum of the weight image up to flat pixel i (row-major)
        auto prefix = [&](long i) -> double {
            const int r = (int)(i / W), c = (int)(i - (long)r * W);
            double acc = 0.0;
            for (int k = 0; k < K; ++k) {
                const double w = right[k] - left[k];
                const double full = fmin((double)r, (double)bottom[k]) - top[k];          // complete rows of the box above row r
                double count = full > 0.0 ? full * w : 0.0;
                if (r >= top[k] && r < bottom[k]) {
                    const double part = fmin((double)(c + 1), (double)right[k]) - left[k];
                    if (part > 0.0) count += part;
                }
                if (count > 0.0) acc += (double)per[k] * count;
            }
            return acc;
        };
        const long total_px = (long)H * W;
        const double total = prefix(total_px - 1);
        long idx = total_px - 1;     // a weight image that does not normalise (empty boxes): the lookup runs off the end and is clamped
        if (total > 0.0 && total < 1e300) {
            // first pixel whose cumulative weight reaches the draw.  prefix() is monotone (a sum of non-negative terms that each
            // grow with the pixel index, rounded to nearest), so a 64-way search - lane l probes the end of the l-th of 64 equal
            // parts of [lo, hi], the first lane whose prefix reaches the target names the part - finds the pixel a binary search
            // finds, in 3 rounds of one probe per lane instead of 18 dependent probes of one thread
            const double target = (double)p.u[n] * total;
            long lo = 0, hi = total_px - 1;          // the answer lies in [lo, hi]; prefix(hi) >= target
            while (lo < hi) {
                const long step = (hi - lo + 64) / 64;                    // ceil((hi - lo + 1) / 64): 64 parts cover [lo, hi]
                long cand = lo + (long)(lane + 1) * step - 1;
                if (cand > hi) cand = hi;
                const unsigned long long reached = __ballot(prefix(cand) >= target);
                const int first = reached ? __ffsll((long long)reached) - 1 : 63;   // (the last lane probes hi: always set)
                long part_hi = lo + (long)(first + 1) * step - 1;
                if (part_hi > hi) part_hi = hi;
                lo = lo + (long)first * step;
                hi = part_hi;
            }
            idx = lo;
        }
        const int half = ((p.patch * s0) / sm) / 2;
        int row = (int)(idx / W), col = (int)(idx - (long)row * W);
        row = min(max(row, half * sm), H - sm * (half - 1) - 1);
        col = min(max(col, half * sm), W - sm * (half - 1) - 1);
        // snap the patch start to the grid of the largest stride (offset sm / 2), towards the reference's side (:372-396)
        auto align = [&](int st) {
            const int h = sm / 2, d = st % sm;
            if (d == h) return st;
            return st >= h ? st - (d + h) % sm : st + (sm + h - d);
        };
        if (lane == 0) {
            start[0] = align(row - half * sm);
            start[1] = align(col - half * sm);
        }
    }
    __syncthreads();
    int base = 0;
    for (int q = 0; q < p.nstrides; ++q) {
        const int s = p.strides[q], size = (p.patch * s0) / s, off = sm / 2 - s / 2;
        for (int i = threadIdx.x; i < size * size; i += 256) {
            const int a = i / size, b = i - a * size;
            p.rows[(size_t)n * p.rays + base + i] = start[0] - off + a * s;
            p.cols[(size_t)n * p.rays + base + i] = start[1] - off + b * s;
        }
        base += size * size;
    }
}
}  // namespace pr

extern "C" int pr_patch_pixels(int32_t frames, int32_t objects, int32_t height, int32_t width, int32_t patch_size, int32_t stride_count,
                               const int32_t* strides, const float* boxes, const float* weights, const float* u, int32_t* rows,
                               int32_t* cols, void* stream) {
    PR_REQUIRE(frames > 0 && objects >= 1 && objects <= PR_MAX_OBJECTS && height > 0 && width > 0, "pr_patch_pixels: bad sizes");
    PR_REQUIRE(stride_count >= 1 && stride_count <= 4 && strides, "pr_patch_pixels: 1..4 strides");
    PR_REQUIRE(boxes && weights && u && rows && cols, "pr_patch_pixels: NULL pointer");
    PR_REQUIRE(patch_size > 0 && patch_size % 2 == 0, "Patch size must be a multiple of 2");
    pr::PatchPixelsParams p;
    memset(&p, 0, sizeof(p));
    p.frames = frames; p.objects = objects; p.height = height; p.width = width; p.patch = patch_size; p.nstrides = stride_count;
    int rays = 0;
    for (int q = 0; q < stride_count; ++q) {
        PR_REQUIRE(strides[q] > 0 && (q == 0 || strides[q] >= strides[q - 1]), "pr_patch_pixels: strides must ascend");
        p.strides[q] = strides[q];
    }
    PR_REQUIRE((patch_size * strides[0]) % (2 * strides[stride_count - 1]) == 0,
               "Patch size is not compatible with the chosen strides. Make patch size divisible by a higher power of 2");
    for (int q = 0; q < stride_count; ++q) {
        const int size = (patch_size * strides[0]) / strides[q];
        rays += size * size;
    }
    p.boxes = boxes; p.weights = weights; p.u = u; p.rows = rows; p.cols = cols; p.rays = rays;
    hipLaunchKernelGGL(pr::k_patch_pixels, dim3(frames), dim3(256), 0, (hipStream_t)stream, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

namespace pr {
// (rotation, translation) -> [R t; 0 1] with R = Ry (Rx Rz) and its rigid inverse [R^T  -R^T t; 0 1], one thread per matrix:
// Transformations3D.homogeneous_rotation_translation (utils/lib_3d/transformations_3d.py:69-96) and the torch.inverse the
// reference applies to it (environment_model.py:221, :1078).  As torch ops this is ~30 launches per call (six sin / cos,
// stacks, two 3 x 3 products, slice assignments, the inverse's product and concatenations) for a few hundred FLOPs.
// m / v: 16 floats each (row-major 4 x 4)
__device__ __forceinline__ void pose_pair(float ax, float ay, float az, const float* t, float* m, float* v) {
    const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    const float rx[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
    const float ry[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
    const float rz[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
    float xz[9], r[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(rx[a * 3 + k], rz[k * 3 + b]));
            xz[a * 3 + b] = acc;
        }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(ry[a * 3 + k], xz[k * 3 + b]));
            r[a * 3 + b] = acc;
        }
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
            m[a * 4 + b] = r[a * 3 + b];
            v[a * 4 + b] = r[b * 3 + a];
        }
        m[a * 4 + 3] = t[a];
        float acc = 0.f;
        for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(r[k * 3 + a], t[k]));
        v[a * 4 + 3] = -acc;
    }
    for (int b = 0; b < 3; ++b) m[12 + b] = v[12 + b] = 0.f;
    m[15] = v[15] = 1.f;
}

__global__ void k_pose_matrices(int count, const float* __restrict__ rotations, const float* __restrict__ translations,
                                float* __restrict__ matrices, float* __restrict__ inverses) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float t[3] = {translations[i * 3 + 0], translations[i * 3 + 1], translations[i * 3 + 2]};
    float m[16], v[16];
    pose_pair(rotations[i * 3 + 0], rotations[i * 3 + 1], rotations[i * 3 + 2], t, m, v);
    for (int e = 0; e < 16; ++e) {
        matrices[(size_t)i * 16 + e] = m[e];
        inverses[(size_t)i * 16 + e] = v[e];
    }
}
}  // namespace pr

namespace pr {
// Projection of object-frame points into the cameras of their frame (EnvironmentModel.compute_object_bounding_boxes /
// compute_object_axes_projection, model/environment_model.py:234-404): world = R_o2w p + t, cam = R_w2c world + t,
// image-plane (x right, y down, relative to the image centre) = (-cam.x / cam.z f, cam.y / cam.z f), normalised to
// (v + size / 2) / size.  One 64-lane workgroup per (frame, camera, object); with `boxes` the lanes also reduce the
// points to [left, top, right, bottom], points behind the camera (cam.z > 0) counting as +-1e20, and both outputs are
// clamped to [0, 1] (the bounding-box variant); without, the points are left unclamped (the axes variant).
// one object's points through one camera, by the 64 lanes of a wave: mo / mc = 16-float matrices (o2w of the object, w2c of the
// camera); out_points / boxes already point at element [f][c] of their arrays
__device__ __forceinline__ void project_object(int lane, int k, int objects, int npoints, const float* __restrict__ pts, const float* mo,
                                               const float* mc, float focal, float width, float height, float* __restrict__ out_points,
                                               float* __restrict__ boxes) {
    float lo_x = 1e20f, lo_y = 1e20f, hi_x = -1e20f, hi_y = -1e20f;
    for (int i = lane; i < npoints; i += 64) {
        const float* pt = pts + (size_t)i * 3;
        float w[3], cam[3];
        for (int a = 0; a < 3; ++a)
            w[a] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pt[0], mo[a * 4 + 0]), __fmul_rn(pt[1], mo[a * 4 + 1])),
                                       __fmul_rn(pt[2], mo[a * 4 + 2])), mo[a * 4 + 3]);
        for (int a = 0; a < 3; ++a)
            cam[a] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w[0], mc[a * 4 + 0]), __fmul_rn(w[1], mc[a * 4 + 1])),
                                         __fmul_rn(w[2], mc[a * 4 + 2])), mc[a * 4 + 3]);
        const float px = __fmul_rn(__fdiv_rn(-cam[0], cam[2]), focal);
        const float py = -__fmul_rn(__fdiv_rn(-cam[1], cam[2]), focal);
        if (boxes) {
            const bool behind = cam[2] > 0.f;
            lo_x = fminf(lo_x, behind ? 1e20f : px);
            lo_y = fminf(lo_y, behind ? 1e20f : py);
            hi_x = fmaxf(hi_x, behind ? -1e20f : px);
            hi_y = fmaxf(hi_y, behind ? -1e20f : py);
        }
        float nx = __fdiv_rn(__fadd_rn(px, width / 2.0f), width), ny = __fdiv_rn(__fadd_rn(py, height / 2.0f), height);
        if (boxes) {
            nx = fminf(fmaxf(nx, 0.f), 1.f);
            ny = fminf(fmaxf(ny, 0.f), 1.f);
        }
        // (points, 2, objects)
        float* dst = out_points + ((size_t)i * 2) * objects + k;
        dst[0] = nx;
        dst[objects] = ny;
    }
    if (boxes) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo_x = fminf(lo_x, __shfl_down(lo_x, d, 64));
            lo_y = fminf(lo_y, __shfl_down(lo_y, d, 64));
            hi_x = fmaxf(hi_x, __shfl_down(hi_x, d, 64));
            hi_y = fmaxf(hi_y, __shfl_down(hi_y, d, 64));
        }
        if (lane == 0) {
            const float v[4] = {lo_x, lo_y, hi_x, hi_y};
            const float size[4] = {width, height, width, height};
            float* dst = boxes + k;      // (4, objects)
            for (int a = 0; a < 4; ++a) {
                const float n = __fdiv_rn(__fadd_rn(v[a], size[a] / 2.0f), size[a]);
                dst[(size_t)a * objects] = fminf(fmaxf(n, 0.f), 1.f);
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_project_points(int frames, int cameras, int objects, int npoints, const float* __restrict__ points,
                                                      const float* __restrict__ o2w, const float* __restrict__ w2c,
                                                      const float* __restrict__ focals, float width, float height,
                                                      float* __restrict__ out_points, float* __restrict__ boxes) {
    const int k = blockIdx.x % objects;
    const int c = (blockIdx.x / objects) % cameras;
    const int f = blockIdx.x / (objects * cameras);
    const size_t fc = (size_t)f * cameras + c;
    project_object(threadIdx.x, k, objects, npoints, points + (size_t)k * npoints * 3, o2w + ((size_t)f * objects + k) * 16, w2c + fc * 16,
                   focals[fc], width, height, out_points + fc * npoints * 2 * objects, boxes ? boxes + fc * 4 * objects : nullptr);
}

// Scene set-up of an evaluation call in ONE launch (EnvironmentModel.forward_from_scene_encoding without a graph): camera and
// object pose matrices with their rigid inverses, the projected boxes / box points / axes, and the renderer's inputs in the
// renderer's layouts - what used to be two pose launches, two projection launches and ~10 small copy kernels (permutes of the
// (..., 3 | S | D, K) scene tensors).  One workgroup of 256 threads per (frame, camera); the arithmetic is pose_pair /
// project_object, i.e. bit for bit what the separate launches compute.
struct SceneSetup {
    int frames, cameras, objects, points, S, D;
    float width, height, focal_multiplier, upsample;
    int axes_with_upsampled_focals;
    const float* cam_rot; const float* cam_tr; const float* focals;          // (frames, cameras, 3), (frames, cameras)
    const float* obj_rot; const float* obj_tr;                               // (frames, 3, objects)
    const float* style; const float* deformation; const uint8_t* in_scene;   // (frames, S | D, objects), (frames, objects)
    const float* box_points; const float* axes_points;                       // (objects, points, 3), (objects, 4, 3)
    float* boxes; float* projected; float* axes;                             // (frames, cameras, 4 | points x 2 | 4 x 2, objects)
    float* camera34; float* render_focals;                                   // (frames x cameras, 3, 4), (frames x cameras)
    float* w2o34; float* style_nks; float* deformation_nkd; uint8_t* present; // (frames x cameras, objects, 3 x 4 | S | D | 1)
};
__global__ __launch_bounds__(256) void k_scene_setup(SceneSetup p) {
    __shared__ float cam[32];                       // c2w, w2c
    __shared__ float obj[PR_MAX_OBJECTS][32];       // o2w, w2o per object
    __shared__ float foc[2];                        // rescaled focal, render focal
    const int f = blockIdx.x / p.cameras;
    const int tid = threadIdx.x, K = p.objects;
    const size_t fc = blockIdx.x;
    if (tid == 0) {
        const float* r = p.cam_rot + fc * 3;
        const float t[3] = {p.cam_tr[fc * 3 + 0], p.cam_tr[fc * 3 + 1], p.cam_tr[fc * 3 + 2]};
        pose_pair(r[0], r[1], r[2], t, cam, cam + 16);
        const float f1 = __fmul_rn(p.focals[fc], p.focal_multiplier);
        foc[0] = f1;
        foc[1] = p.upsample == 1.0f ? f1 : __fmul_rn(f1, p.upsample);
    } else if (tid >= 64 && tid < 64 + K) {
        const int k = tid - 64;
        const float* r = p.obj_rot + (size_t)f * 3 * K + k;
        const float* tr = p.obj_tr + (size_t)f * 3 * K + k;
        const float t[3] = {tr[0], tr[K], tr[2 * K]};
        pose_pair(r[0], r[K], r[2 * K], t, obj[k], obj[k] + 16);
    }
    __syncthreads();
    if (tid < 12) p.camera34[fc * 12 + tid] = cam[tid];
    if (tid == 12) p.render_focals[fc] = foc[1];
    for (int e = tid; e < K * 12; e += 256) p.w2o34[fc * K * 12 + e] = obj[e / 12][16 + e % 12];
    for (int e = tid; e < K * p.S; e += 256) p.style_nks[fc * K * p.S + e] = p.style[((size_t)f * p.S + e % p.S) * K + e / p.S];
    for (int e = tid; e < K * p.D; e += 256) p.deformation_nkd[fc * K * p.D + e] = p.deformation[((size_t)f * p.D + e % p.D) * K + e / p.D];
    if (tid < K) p.present[fc * K + tid] = p.in_scene[(size_t)f * K + tid] ? 1 : 0;
    // projections: wave w takes the objects w, w + 4, ...
    const int wave = tid >> 6, lane = tid & 63;
    for (int k = wave; k < K; k += 4) {
        project_object(lane, k, K, p.points, p.box_points + (size_t)k * p.points * 3, obj[k], cam + 16, foc[1], p.width, p.height,
                       p.projected + fc * p.points * 2 * K, p.boxes + fc * 4 * K);
        project_object(lane, k, K, 4, p.axes_points + (size_t)k * 12, obj[k], cam + 16, p.axes_with_upsampled_focals ? foc[1] : foc[0],
                       p.width, p.height, p.axes + fc * 8 * K, nullptr);
    }
}
}  // namespace pr

extern "C" int pr_project_points(int32_t frames, int32_t cameras, int32_t objects, int32_t points_per_object, const float* points,
                                 const float* o2w, const float* w2c, const float* focals, int32_t height, int32_t width,
                                 float* projected, float* boxes, void* stream) {
    PR_REQUIRE(frames >= 0 && cameras > 0 && objects > 0 && points_per_object > 0 && height > 0 && width > 0,
               "pr_project_points: bad sizes");
    if (frames == 0) return PR_OK;
    PR_REQUIRE(points && o2w && w2c && focals && projected, "pr_project_points: NULL pointer");
    hipLaunchKernelGGL(pr::k_project_points, dim3((unsigned)(frames * cameras * objects)), dim3(64), 0, (hipStream_t)stream, frames,
                       cameras, objects, points_per_object, points, o2w, w2c, focals, (float)width, (float)height, projected, boxes);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_scene_setup(const pr_scene_setup_t* q, void* stream) {
    PR_REQUIRE(q != nullptr, "pr_scene_setup: NULL argument");
    PR_REQUIRE(q->frames >= 0 && q->cameras > 0 && q->objects > 0 && q->objects <= PR_MAX_OBJECTS && q->box_points_per_object > 0 &&
                   q->height > 0 && q->width > 0 && q->style_features >= 0 && q->deformation_features >= 0,
               "pr_scene_setup: bad sizes");
    if (q->frames == 0) return PR_OK;
    PR_REQUIRE(q->camera_rotations && q->camera_translations && q->focals && q->object_rotations && q->object_translations && q->style &&
                   q->deformation && q->object_in_scene && q->box_points && q->axes_points && q->boxes && q->projected_points && q->axes &&
                   q->camera34 && q->render_focals && q->w2o34 && q->style_nks && q->deformation_nkd && q->present,
               "pr_scene_setup: NULL pointer");
    pr::SceneSetup p;
    p.frames = q->frames; p.cameras = q->cameras; p.objects = q->objects; p.points = q->box_points_per_object;
    p.S = q->style_features; p.D = q->deformation_features;
    p.width = (float)q->width; p.height = (float)q->height; p.focal_multiplier = q->focal_multiplier; p.upsample = q->upsample_factor;
    p.axes_with_upsampled_focals = q->axes_with_upsampled_focals;
    p.cam_rot = q->camera_rotations; p.cam_tr = q->camera_translations; p.focals = q->focals;
    p.obj_rot = q->object_rotations; p.obj_tr = q->object_translations;
    p.style = q->style; p.deformation = q->deformation; p.in_scene = q->object_in_scene;
    p.box_points = q->box_points; p.axes_points = q->axes_points;
    p.boxes = q->boxes; p.projected = q->projected_points; p.axes = q->axes;
    p.camera34 = q->camera34; p.render_focals = q->render_focals;
    p.w2o34 = q->w2o34; p.style_nks = q->style_nks; p.deformation_nkd = q->deformation_nkd; p.present = q->present;
    hipLaunchKernelGGL(pr::k_scene_setup, dim3((unsigned)(q->frames * q->cameras)), dim3(256), 0, (hipStream_t)stream, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

namespace pr {
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ float mat3_dot(const float* a, const float* b) {
    float acc = 0.f;
    for (int i = 0; i < 9; ++i) acc = fmaf(a[i], b[i], acc);
    return acc;
}

// Backward of k_pose_matrices: M = [R t; 0 1], M^-1 = [R^T u; 0 1] with u = -R^T t, R = Ry (Rx Rz).
//   d loss / d R[k][a] = gM[k][a] + gInv[a][k] - t[k] gu[a],  d loss / d t = gM[:, 3] - R gu,
//   d loss / d angle = <d loss / d R, d R / d angle>  with d R / d x = Ry (Rx' Rz), d R / d y = Ry' (Rx Rz), d R / d z = Ry (Rx Rz').
// gm / gv: 16-float gradients of the matrix / of its inverse (row-major 4 x 4; only the first three rows are read), or NULL
__device__ __forceinline__ void pose_backward(float ax, float ay, float az, const float* t, const float* gm, const float* gv,
                                              float* g_rot, float* g_tr) {
    const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    const float rx[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
    const float ry[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
    const float rz[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
    const float dx[9] = {0.f, 0.f, 0.f, 0.f, -sx, -cx, 0.f, cx, -sx};
    const float dy[9] = {-sy, 0.f, cy, 0.f, 0.f, 0.f, -cy, 0.f, -sy};
    const float dz[9] = {-sz, -cz, 0.f, cz, -sz, 0.f, 0.f, 0.f, 0.f};
    float xz[9], r[9], tmp[9], d[9];
    mat3_mul(rx, rz, xz);
    mat3_mul(ry, xz, r);
    float gu[3] = {0.f, 0.f, 0.f}, gr[9];
    if (gv)
        for (int a = 0; a < 3; ++a) gu[a] = gv[a * 4 + 3];
    for (int k = 0; k < 3; ++k) {
        for (int a = 0; a < 3; ++a) gr[k * 3 + a] = (gm ? gm[k * 4 + a] : 0.f) + (gv ? gv[a * 4 + k] : 0.f) - t[k] * gu[a];
        g_tr[k] = (gm ? gm[k * 4 + 3] : 0.f) - (r[k * 3] * gu[0] + r[k * 3 + 1] * gu[1] + r[k * 3 + 2] * gu[2]);
    }
    mat3_mul(dx, rz, tmp);
    mat3_mul(ry, tmp, d);
    g_rot[0] = mat3_dot(gr, d);
    mat3_mul(dy, xz, d);
    g_rot[1] = mat3_dot(gr, d);
    mat3_mul(rx, dz, tmp);
    mat3_mul(ry, tmp, d);
    g_rot[2] = mat3_dot(gr, d);
}

__global__ void k_pose_matrices_bwd(int count, const float* __restrict__ rotations, const float* __restrict__ translations,
                                    const float* __restrict__ g_matrices, const float* __restrict__ g_inverses,
                                    float* __restrict__ g_rotations, float* __restrict__ g_translations) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float t[3] = {translations[i * 3 + 0], translations[i * 3 + 1], translations[i * 3 + 2]};
    float g_rot[3], g_tr[3];
    pose_backward(rotations[i * 3 + 0], rotations[i * 3 + 1], rotations[i * 3 + 2], t, g_matrices ? g_matrices + (size_t)i * 16 : nullptr,
                  g_inverses ? g_inverses + (size_t)i * 16 : nullptr, g_rot, g_tr);
    for (int k = 0; k < 3; ++k) {
        g_rotations[i * 3 + k] = g_rot[k];
        g_translations[i * 3 + k] = g_tr[k];
    }
}

// Backward of k_scene_setup's renderer inputs in ONE launch (a training call through the fused scene set-up): the gradients the
// renderer's backward pass leaves in ITS layouts - d w2o (frames x cameras, objects, 3, 4), d style (.., objects, S), d deformation
// (.., objects, D) - summed over the cameras of a frame and taken back to the scene tensors' layouts: d object rotations /
// translations (frames, 3, objects) through pose_backward (w2o is the INVERSE of the pose matrix: gv), d style (frames, S, objects),
// d deformation (frames, D, objects).  One workgroup per frame.  (As tensor ops: two zero fills, four permuting copies, the
// pose-backward launch and its reshapes - ~10 launches of ~5 us at the end of every training step's backward pass.)
struct SceneSetupBwd {
    int frames, cameras, objects, S, D;
    const float* obj_rot; const float* obj_tr;                                 // (frames, 3, objects)
    const float* g_w2o34; const float* g_style_nks; const float* g_deformation_nkd;   // or NULL
    float* g_rot; float* g_tr; float* g_style; float* g_deformation;          // (frames, 3 | 3 | S | D, objects); NULL: not wanted
};
__global__ __launch_bounds__(256) void k_scene_setup_bwd(SceneSetupBwd p) {
    const int f = blockIdx.x, tid = threadIdx.x, K = p.objects, C = p.cameras;
    if (tid < K && p.g_rot && p.g_tr) {
        const int k = tid;
        float gv[16];
        for (int e = 0; e < 16; ++e) gv[e] = 0.f;
        if (p.g_w2o34)
            for (int c = 0; c < C; ++c) {
                const float* src = p.g_w2o34 + (((size_t)f * C + c) * K + k) * 12;
                for (int e = 0; e < 12; ++e) gv[e] += src[e];
            }
        const float* r = p.obj_rot + (size_t)f * 3 * K + k;
        const float* tr = p.obj_tr + (size_t)f * 3 * K + k;
        const float t[3] = {tr[0], tr[K], tr[2 * K]};
        float g_rot[3], g_tr[3];
        pose_backward(r[0], r[K], r[2 * K], t, nullptr, gv, g_rot, g_tr);
        for (int a = 0; a < 3; ++a) {
            p.g_rot[((size_t)f * 3 + a) * K + k] = g_rot[a];
            p.g_tr[((size_t)f * 3 + a) * K + k] = g_tr[a];
        }
    }
    if (p.g_style)
        for (int e = tid; e < K * p.S; e += 256) {
            const int k = e / p.S, s = e - k * p.S;
            float acc = 0.f;
            if (p.g_style_nks)
                for (int c = 0; c < C; ++c) acc += p.g_style_nks[(((size_t)f * C + c) * K + k) * p.S + s];
            p.g_style[((size_t)f * p.S + s) * K + k] = acc;
        }
    if (p.g_deformation)
        for (int e = tid; e < K * p.D; e += 256) {
            const int k = e / p.D, d = e - k * p.D;
            float acc = 0.f;
            if (p.g_deformation_nkd)
                for (int c = 0; c < C; ++c) acc += p.g_deformation_nkd[(((size_t)f * C + c) * K + k) * p.D + d];
            p.g_deformation[((size_t)f * p.D + d) * K + k] = acc;
        }
}
}  // namespace pr

extern "C" int pr_scene_setup_backward(int32_t frames, int32_t cameras, int32_t objects, int32_t style_features, int32_t deformation_features,
                                       const float* object_rotations, const float* object_translations, const float* g_w2o34,
                                       const float* g_style_nks, const float* g_deformation_nkd, float* g_rotations, float* g_translations,
                                       float* g_style, float* g_deformation, void* stream) {
    PR_REQUIRE(frames >= 0 && cameras > 0 && objects > 0 && objects <= PR_MAX_OBJECTS && style_features >= 0 && deformation_features >= 0,
               "pr_scene_setup_backward: bad sizes");
    if (frames == 0) return PR_OK;
    PR_REQUIRE((g_rotations == nullptr) == (g_translations == nullptr), "pr_scene_setup_backward: rotation and translation gradients come together");
    PR_REQUIRE(!g_rotations || (object_rotations && object_translations), "pr_scene_setup_backward: NULL pose pointer");
    pr::SceneSetupBwd p{frames, cameras, objects, style_features, deformation_features, object_rotations, object_translations, g_w2o34,
                        g_style_nks, g_deformation_nkd, g_rotations, g_translations, g_style, g_deformation};
    hipLaunchKernelGGL(pr::k_scene_setup_bwd, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream, p);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_pose_matrices_backward(int32_t count, const float* rotations, const float* translations, const float* g_matrices,
                                         const float* g_inverses, float* g_rotations, float* g_translations, void* stream) {
    PR_REQUIRE(count >= 0, "pr_pose_matrices_backward: bad count %d", count);
    if (count == 0) return PR_OK;
    PR_REQUIRE(rotations && translations && g_rotations && g_translations, "pr_pose_matrices_backward: NULL pointer");
    hipLaunchKernelGGL(pr::k_pose_matrices_bwd, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, count, rotations,
                       translations, g_matrices, g_inverses, g_rotations, g_translations);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_pose_matrices(int32_t count, const float* rotations, const float* translations, float* matrices,
                                float* inverses, void* stream) {
    PR_REQUIRE(count >= 0, "pr_pose_matrices: bad count %d", count);
    if (count == 0) return PR_OK;
    PR_REQUIRE(rotations && translations && matrices && inverses, "pr_pose_matrices: NULL pointer");
    hipLaunchKernelGGL(pr::k_pose_matrices, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, count, rotations, translations,
                       matrices, inverses);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_camera_rays(int32_t frames, int32_t rays, int32_t height, int32_t width, int32_t per_frame_pixels,
                              const float* c2w, const float* focals, const int32_t* rows, const int32_t* cols, float* ray_origins,
                              float* ray_directions, float* focal_normals, void* stream) {
    PR_REQUIRE(frames > 0 && rays > 0, "pr_camera_rays: empty call (%d frames, %d rays)", frames, rays);
    PR_REQUIRE(c2w && focals && rows && cols && ray_origins && ray_directions && focal_normals,
               "pr_camera_rays: NULL pointer");
    const long total = (long)frames * rays;
    const int blocks = (int)((total + 255) / 256);
    // width / 2 and height / 2 are Python floats in the reference (true division)
    hipLaunchKernelGGL(pr::k_camera_rays, dim3(blocks), dim3(256), 0, (hipStream_t)stream, frames, rays, per_frame_pixels,
                       (float)height / 2.0f, (float)width / 2.0f, c2w, focals, rows, cols, ray_origins,
                       ray_directions, focal_normals);
    PR_LAUNCH_CHECK();
    return PR_OK;
}
