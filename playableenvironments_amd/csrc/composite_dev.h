// Device helpers shared by the compositing kernel and its backward pass.
#pragma once
#include "pr_common.h"

namespace pr {

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

// Orders the concatenated per-object lists by (t, concatenation index) into key[0 .. total): key[rank] = entry.  Each object's list is normally already sorted (linspace placement, or the
// output of the resampler's sort), so the rank of an entry is its own index plus, for every other object, the
// number of entries that precede it - binary searches instead of a bitonic network over all entries.  Falls back
// to the bitonic sort when a list is not non-decreasing (overlap-fixed lists, or depths whose spacing is below one
// ulp).  Workgroup of `threads` threads (a multiple of 64, every thread calls); `sort_size` = power of two >= total.
// `wide` (sort_size 64-bit words, or NULL): scratch for the bitonic network - calls that always take it (overlap fix)
// provide it; without it the network compares through the depth array (slower, rare).
__device__ __forceinline__ void order_entries(unsigned int* key, const float* tt, const int* positions, int objects,
                                              int total, int sort_size, bool lists_may_be_sorted, int tid, int threads,
                                              unsigned long long* wide) {
    int merge = lists_may_be_sorted ? 1 : 0;
    if (merge) {
        int ok = 1;
        int off = 0;
        for (int k = 0; k < objects; ++k) {
            const int P = positions[k];
            for (int i = tid; i + 1 < P; i += threads) ok = ok && (tt[off + i] <= tt[off + i + 1]);
            off += P;
        }
        merge = __syncthreads_and(ok);
    }
    if (merge) {
        // Each thread owns a run of consecutive entries of a list: their ranks in another (sorted) list are non-decreasing,
        // so every search after the first starts where the previous one ended and first probes a window of 8 entries.
        int off = 0;
        for (int k = 0; k < objects; ++k) {
            const int P = positions[k];
            const int run = (P + threads - 1) / threads;
            const int first = tid * run;
            int resume[PR_MAX_OBJECTS];
#pragma unroll
            for (int q = 0; q < PR_MAX_OBJECTS; ++q) resume[q] = 0;
            for (int i = first; i < first + run && i < P; ++i) {
                const float t = tt[off + i];
                int rank = i;
                int o2 = 0;
#pragma unroll
                for (int k2 = 0; k2 < PR_MAX_OBJECTS; ++k2) {
                    if (k2 >= objects) break;
                    const int P2 = positions[k2];
                    if (k2 != k) {
                        // entries of an earlier object win ties (stable in object order)
                        auto before = [&](int idx) {
                            const float v = tt[o2 + idx];
                            return (k2 < k) ? (v <= t) : (v < t);
                        };
                        int lo = resume[k2], hi = P2;
                        if (lo + 8 < hi) {
                            if (before(lo + 7)) lo += 8; else hi = lo + 7;
                        }
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (before(mid)) lo = mid + 1; else hi = mid;
                        }
                        resume[k2] = lo;
                        rank += lo;
                    }
                    o2 += P2;
                }
                key[rank] = (unsigned int)(off + i);
            }
            off += P;
        }
        __syncthreads();
        return;
    }
    if (wide != nullptr) {
        for (int e = tid; e < sort_size; e += threads)
            wide[e] = (e < total) ? (((unsigned long long)float_order_bits(tt[e]) << 32) | (unsigned int)e) : 0xFFFFFFFFFFFFFFFFull;
        __syncthreads();
        for (int kk = 2; kk <= sort_size; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < sort_size; i += threads) {
                    const int x = i ^ j;
                    if (x > i) {
                        const unsigned long long a = wide[i], b = wide[x];
                        const bool up = ((i & kk) == 0);
                        if ((a > b) == up) {
                            wide[i] = b;
                            wide[x] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int e = tid; e < total; e += threads) key[e] = (unsigned int)(wide[e] & 0xFFFFFFFFu);
        __syncthreads();
        return;
    }
    // bitonic network on entry indices; the sort key of an entry is (order bits of its t, index), padding sorts last
    for (int e = tid; e < sort_size; e += threads) key[e] = (e < total) ? (unsigned int)e : 0xFFFFFFFFu;
    __syncthreads();
    auto sort_key = [&](unsigned int e) -> unsigned long long {
        return e == 0xFFFFFFFFu ? 0xFFFFFFFFFFFFFFFFull : (((unsigned long long)float_order_bits(tt[e]) << 32) | e);
    };
    for (int kk = 2; kk <= sort_size; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < sort_size; i += threads) {
                const int x = i ^ j;
                if (x > i) {
                    const unsigned int a = key[i], b = key[x];
                    const bool up = ((i & kk) == 0);
                    if ((sort_key(a) > sort_key(b)) == up) {
                        key[i] = b;
                        key[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace pr
