// Host orchestration of one renderer call (the body of ObjectComposer.forward,
// model/object_composer.py:786-892) and the small introspection entry points of the C ABI.
#include "pr_common.h"

#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace pr {

// ---------------------------------------------------------------------------------------------
// Kernel timing
// ---------------------------------------------------------------------------------------------
struct ProfileRecord { int category; hipEvent_t start, stop; };
static std::atomic<bool> g_profile_on{false};   // bench-only switch; launches read it relaxed, no lock on the product path
static std::vector<ProfileRecord> g_profile;
static std::mutex g_profile_mutex;

int prepare_kernel(const void* kernel, int lds_bytes, int* cu_count) {
    struct Entry { const void* kernel; int device; int cus; };
    static std::vector<Entry> done;
    static std::mutex guard;
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(guard);
    for (const Entry& e : done)
        if (e.kernel == kernel && e.device == dev) {
            if (cu_count) *cu_count = e.cus;
            return PR_OK;
        }
    PR_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipDeviceProp_t prop;
    PR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    done.push_back(Entry{kernel, dev, prop.multiProcessorCount});
    if (cu_count) *cu_count = prop.multiProcessorCount;
    return PR_OK;
}

// Every zero fill of the library is this kernel, never hipMemsetAsync: recorded into a HIP graph a memset becomes a memset NODE,
// and on ROCm 7.0.2 memset nodes stop executing after a few back-to-back replays followed by a host synchronisation (with the
// runtime's AQL packet capture on; DESIGN.md "Recorded training step") - a kernel node does not.
__global__ __launch_bounds__(256) void k_zero_fill(uint4* dst16, size_t n16, uint32_t* tail, size_t ntail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst16[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ntail; i += (size_t)gridDim.x * 256) tail[i] = 0u;
}

int launch_zero_fill(void* dst, size_t bytes, hipStream_t s) {
    if (bytes == 0) return PR_OK;
    PR_REQUIRE((bytes & 3) == 0 && ((uintptr_t)dst & 3) == 0, "zero fill: %zu bytes at a misaligned address", bytes);
    // 16-byte stores where the address allows them, 4-byte stores for the rest (and for small misaligned regions as a whole)
    const bool aligned = ((uintptr_t)dst & 15) == 0;
    const size_t n16 = aligned ? bytes / 16 : 0;
    const size_t ntail = (bytes - n16 * 16) / 4;
    size_t blocks = ((n16 > ntail ? n16 : ntail) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint4*>(dst), n16,
                       reinterpret_cast<uint32_t*>(static_cast<char*>(dst) + n16 * 16), ntail);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

ProfileScope::ProfileScope(int category, hipStream_t s) : category_(category), stream_(s), start_(nullptr), active_(false) {
    if (!g_profile_on.load(std::memory_order_relaxed)) return;
    if (hipEventCreate(&start_) != hipSuccess) return;
    if (hipEventRecord(start_, stream_) != hipSuccess) {
        (void)hipEventDestroy(start_);
        return;
    }
    active_ = true;
}

ProfileScope::~ProfileScope() {
    if (!active_) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) return;
    if (hipEventRecord(stop, stream_) != hipSuccess) {
        (void)hipEventDestroy(stop);
        (void)hipEventDestroy(start_);
        return;
    }
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    g_profile.push_back(ProfileRecord{category_, start_, stop});
}

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int build_mlp_layers(const pr_object_model_t& m, const ModelDims& d, const PackedLayout& l, const float* base,
                     MlpParams* p, bool split3);

// ---------------------------------------------------------------------------------------------
// Workspace plan (structs in pr_common.h)
// ---------------------------------------------------------------------------------------------
static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int validate_call(const pr_call_t& c, const pr_object_t* objs) {
    PR_REQUIRE(c.frames > 0 && c.rays > 0, "empty call: %d frames x %d rays", c.frames, c.rays);
    PR_REQUIRE(c.objects >= 1 && c.objects <= PR_MAX_OBJECTS, "objects %d out of range 1..%d", c.objects, PR_MAX_OBJECTS);
    PR_REQUIRE(c.static_objects >= 0 && c.static_objects <= c.objects, "static_objects %d out of range", c.static_objects);
    PR_REQUIRE((long)c.frames * c.rays < (1L << 31), "too many rays in one call");
    PR_REQUIRE(c.ray_origins && c.ray_directions && c.w2o && c.style && c.deformation && c.object_in_scene,
               "NULL input pointer");
    PR_REQUIRE(c.precision == PR_PRECISION_FP32 || c.precision == PR_PRECISION_F16X3 || c.precision == PR_PRECISION_F16,
               "unknown precision %d", c.precision);
    PR_REQUIRE(!(c.precision != PR_PRECISION_FP32 && (c.flags & PR_FLAG_TRAIN_BN)),
               "the fp16 kernels have no train-mode BatchNorm phases: use PR_PRECISION_FP32 for training");
    PR_REQUIRE(!(c.precision != PR_PRECISION_FP32 && (c.flags & PR_FLAG_SAVE_FOR_BACKWARD)),
               "differentiable calls run on the exact kernel: use PR_PRECISION_FP32 with PR_FLAG_SAVE_FOR_BACKWARD");
    PR_REQUIRE(!(c.flags & PR_FLAG_SAVE_FOR_BACKWARD) || !(c.flags & PR_FLAG_NAIVE_MLP), "the scalar debugging kernel saves nothing");
    for (int k = 0; k < c.objects; ++k) {
        const pr_object_model_t& m = objs[k].coarse;
        PR_REQUIRE(m.positions >= 1, "object %d: positions_count_coarse %d", k, m.positions);
        PR_REQUIRE(c.linspace_coarse[k] != nullptr, "object %d: linspace_coarse missing", k);
        PR_REQUIRE(objs[k].packed_coarse != nullptr, "object %d: packed coarse weights missing", k);
        PR_REQUIRE((long)c.frames * c.rays * (long)m.positions < (1L << 31), "too many samples in one call");
        if (c.use_fine) {
            PR_REQUIRE(c.positions_fine[k] >= 1, "object %d: positions_count_fine %d", k, c.positions_fine[k]);
            PR_REQUIRE(objs[k].fine.positions == m.positions + c.positions_fine[k],
                       "object %d: fine model positions %d != %d + %d", k, objs[k].fine.positions, m.positions,
                       c.positions_fine[k]);
            PR_REQUIRE(objs[k].packed_fine != nullptr, "object %d: packed fine weights missing", k);
            PR_REQUIRE(c.linspace_fine[k] != nullptr || c.noise_coarse.pdf[k] != nullptr ||
                           ((c.flags & PR_FLAG_DEVICE_NOISE) && (c.flags & PR_FLAG_PERTURB)),
                       "object %d: linspace_fine missing", k);
            PR_REQUIRE((long)c.frames * c.rays * (long)objs[k].fine.positions < (1L << 31), "too many samples in one call");
        }
    }
    if (c.flags & PR_FLAG_FIX_OVERLAPS) {
        for (int t = 0; t < (c.use_fine ? 2 : 1); ++t)
            for (int s = 0; s < c.static_objects; ++s)
                for (int d = c.static_objects; d < c.objects; ++d) {
                    const int ps = t ? objs[s].fine.positions : objs[s].coarse.positions;
                    const int pd = t ? objs[d].fine.positions : objs[d].coarse.positions;
                    // the reference indexes the dynamic list with the static P - 1 (object_composer.py:322)
                    PR_REQUIRE(ps <= pd, "overlap fix: static object %d has more positions (%d) than dynamic object %d (%d); "
                               "the reference raises IndexError here", s, ps, d, pd);
                }
    }
    return PR_OK;
}

// The sigma-gated feature head applies to calls whose compositing weights are a function of the raw densities alone:
// evaluation without perturbation noise, nothing saved for a backward pass, fused (unphased) MLP launches.
bool gate_active(const pr_call_t& c) {
    if (!(c.flags & PR_FLAG_GATE_HEAD)) return false;
    if (c.flags & (PR_FLAG_PERTURB | PR_FLAG_TRAIN_BN | PR_FLAG_SAVE_FOR_BACKWARD | PR_FLAG_NAIVE_MLP)) return false;
    for (int k = 0; k < c.objects; ++k)
        if (c.noise_coarse.integrate[k] || c.noise_fine.integrate[k]) return false;
    return !c.noise_coarse.integrate_global && !c.noise_fine.integrate_global;
}

int make_plan(const pr_call_t& c, const pr_object_t* objs, Plan* plan) {
    memset(plan, 0, sizeof(*plan));
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes);
        return at;
    };
    const size_t nr = (size_t)c.frames * c.rays;
    plan->nblocks256 = (int)((nr + 255) / 256);
    plan->block_sums = take(sizeof(int32_t) * plan->nblocks256 * c.objects);       // one region per object: the objects of a
    plan->block_offsets = take(sizeof(int32_t) * plan->nblocks256 * c.objects);    // call are placed / compacted by shared launches
    size_t max_cap = 0;
    size_t feat_bytes[2] = {0, 0};
    size_t div_t0 = 0, div_t = 0;   // divergence tangent scratch (shared by the bender objects)
    const int ntypes = c.use_fine ? 2 : 1;
    for (int t = 0; t < ntypes; ++t) {
        TypePlan& tp = plan->type[t];
        tp.totals = take(sizeof(int32_t) * PR_MAX_OBJECTS);
        tp.zero_begin = off;
        tp.head_counts = take(sizeof(int32_t) * 3 * PR_MAX_OBJECTS);
        if (c.flags & (PR_FLAG_TRAIN_BN | PR_FLAG_SAVE_FOR_BACKWARD))
            for (int k = 0; k < c.objects; ++k) {
                const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
                SavedPlan& sv = tp.saved[k];
                sv.stats = take(sizeof(double) * 4 * MAX_WIDTH);
                sv.stat_count = take(sizeof(int32_t) * 4);
                if ((c.flags & PR_FLAG_SAVE_FOR_BACKWARD) && m.has_bender) sv.div = take(sizeof(float) * nr * m.positions);
            }
        tp.zero_bytes = off - tp.zero_begin;
        for (int k = 0; k < c.objects; ++k) {
            const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
            ModelDims d;
            PR_TRY(compute_dims(m, &d));
            tp.positions[k] = m.positions;
            const size_t cap = nr * m.positions;
            if (cap > max_cap) max_cap = cap;
            tp.t[k] = take(sizeof(float) * cap);
            tp.sigma[k] = take(sizeof(float) * cap);
            tp.slot[k] = take(sizeof(int32_t) * cap);
            tp.dispmag[k] = m.has_bender ? take(sizeof(float) * cap) : (size_t)-1;
            tp.adain[k] = take(sizeof(float) * (size_t)c.frames * adain_row_floats(d));
            tp.feat[k] = feat_bytes[t];  // relative to the feature arena
            feat_bytes[t] += align_up(sizeof(float) * cap * m.output_features);
            if (c.flags & PR_FLAG_SAVE_FOR_BACKWARD) {
                // everything the backward pass re-reads, per object instance (compact rows, worst-case capacity)
                SavedPlan& sv = tp.saved[k];
                sv.rec_pos = take(sizeof(float) * 3 * cap);
                sv.rec_flat = take(sizeof(int32_t) * cap);
                sv.row_flags = take(sizeof(int32_t) * cap);
                sv.enc = take(sizeof(float) * cap * d.enc_pad);
                sv.act = take(sizeof(float) * cap * d.Wpad * m.backbone_count);
                sv.bits = take(relu_bits_bytes(cap, d.Wpad) * (size_t)m.backbone_count);
                sv.h1 = take(sizeof(float) * cap * d.Wpad);
                sv.h2 = take(sizeof(float) * cap * d.W2pad);
                sv.batch = take(sizeof(float) * 4 * MAX_WIDTH);
                if (m.has_bender) {
                    if (cap * d.bin_pad > div_t0) div_t0 = cap * d.bin_pad;
                    if (cap * d.BWpad > div_t) div_t = cap * d.BWpad;
                    sv.bin = take(sizeof(float) * cap * d.bin_pad);
                    sv.bact = take(sizeof(float) * cap * d.BWpad * m.bender_count);
                    sv.bbits = take(relu_bits_bytes(cap, d.BWpad) * (size_t)m.bender_count);
                    sv.braw = take(sizeof(float) * 3 * cap);
                    sv.delta = take(sizeof(float) * 3 * cap);
                }
            }
        }
    }
    plan->rec_pos = take(sizeof(float) * 3 * max_cap);
    plan->rec_flat = take(sizeof(int32_t) * max_cap);
    if (group_active(c)) {
        // one launch evaluates every object of a model type: each object keeps its own compact sample records
        // (the coarse and the fine pass of an object share them: the coarse launch has finished before the fine fill)
        for (int k = 0; k < c.objects; ++k) {
            size_t cap = nr * objs[k].coarse.positions;
            if (c.use_fine && nr * objs[k].fine.positions > cap) cap = nr * objs[k].fine.positions;
            plan->rec_pos_k[k] = take(sizeof(float) * 3 * cap);
            plan->rec_flat_k[k] = take(sizeof(int32_t) * cap);
        }
    }
    if (gate_active(c)) {   // per-workgroup stacks of pending live rows (sigma-gated head)
        plan->pend_act = take(sizeof(float) * (size_t)MAX_RESIDENT_TILES * TILE_M * MAX_WIDTH);
        plan->pend_meta = take(sizeof(int32_t) * (size_t)MAX_RESIDENT_TILES * TILE_M * 2);
    }
    if (c.flags & (PR_FLAG_TRAIN_BN | PR_FLAG_SAVE_FOR_BACKWARD)) {   // the phased launch structure (see render())
        plan->h1 = take(sizeof(float) * max_cap * MAX_WIDTH);
        plan->h2 = take(sizeof(float) * max_cap * (MAX_WIDTH / 2 + 32));
        plan->row_flags = take(sizeof(int32_t) * max_cap);
        plan->batch_stats = take(sizeof(float) * 4 * MAX_WIDTH);
    }
    // coarse and fine feature rows share one arena: the coarse rows are dead once the coarse
    // compositing pass has run, before the first fine MLP is launched.
    if (div_t) {
        plan->div_t0 = take(sizeof(float) * div_t0);
        plan->div_ta = take(sizeof(float) * div_t);
        plan->div_tb = take(sizeof(float) * div_t);
    }
    if (c.flags & PR_FLAG_SAVE_FOR_BACKWARD) {
        // the backward pass needs the feature rows of both model types
        for (int t = 0; t < ntypes; ++t) {
            const size_t arena = take(feat_bytes[t]);
            for (int k = 0; k < c.objects; ++k) plan->type[t].feat[k] += arena;
        }
    } else {
        const size_t arena = take(feat_bytes[0] > feat_bytes[1] ? feat_bytes[0] : feat_bytes[1]);
        for (int t = 0; t < ntypes; ++t)
            for (int k = 0; k < c.objects; ++k) plan->type[t].feat[k] += arena;
    }
    plan->bytes = off;
    return PR_OK;
}

// Evaluation calls run the objects of a model type as ONE grouped launch (k_mlp_mfma_group / k_mlp_split_group).
bool group_active(const pr_call_t& c) {
#ifdef PR_MLP_UNGROUPED
    return false;      // measurement build: one launch per object
#else
    return !(c.flags & (PR_FLAG_TRAIN_BN | PR_FLAG_SAVE_FOR_BACKWARD | PR_FLAG_NAIVE_MLP)) && c.objects > 1;
#endif
}

// Differentiable calls (everything saved per object) with several objects: phase 1 of the phased launches is grouped as well.
bool group_train_active(const pr_call_t& c) {
#ifdef PR_MLP_UNGROUPED
    return false;
#else
    return (c.flags & PR_FLAG_SAVE_FOR_BACKWARD) && !(c.flags & PR_FLAG_NAIVE_MLP) && c.objects > 1;
#endif
}

void bbox_split(const pr_object_model_t& m, float* lo, float* hi, float* size) {
    for (int a = 0; a < 3; ++a) {
        lo[a] = m.bbox[2 * a];
        hi[a] = m.bbox[2 * a + 1];
        if (size) size[a] = m.bbox[2 * a + 1] - m.bbox[2 * a];   // BoundingBox.get_size, fp32 subtraction
    }
}

static int render(const pr_call_t& c, const pr_object_t* objs, const pr_outputs_t* outs[2], char* ws, const Plan& plan,
                  hipStream_t s) {
    const int ntypes = c.use_fine ? 2 : 1;
    const int K = c.objects;
    int32_t* block_sums_all = reinterpret_cast<int32_t*>(ws + plan.block_sums);
    int32_t* block_offsets_all = reinterpret_cast<int32_t*>(ws + plan.block_offsets);
    // coarse placement / block scan / compaction of several objects: three launches for all of them
    const bool place_grouped = c.objects > 1 && ((c.flags & PR_FLAG_SAVE_FOR_BACKWARD) || group_active(c));   // (every object has its own sample records)
    static thread_local PlaceParams place_jobs[PR_MAX_OBJECTS];
    static thread_local FillParams fill_jobs[PR_MAX_OBJECTS];
    int32_t* total_ptrs[PR_MAX_OBJECTS];
    float* rec_pos = reinterpret_cast<float*>(ws + plan.rec_pos);
    int32_t* rec_flat = reinterpret_cast<int32_t*>(ws + plan.rec_flat);
    const bool naive = (c.flags & PR_FLAG_NAIVE_MLP) != 0;
    const bool gate = gate_active(c);
    const bool grouped = group_active(c);
    MlpParams jobs[PR_MAX_OBJECTS];
    int job_rows[PR_MAX_OBJECTS];
    FoldParams fold_jobs[PR_MAX_OBJECTS];
    // differentiable calls with several objects: phase 1 of every object (the whole network up to the first BatchNorm) runs as
    // one grouped launch too; what follows it per object (statistics, head phases, divergence) is deferred until after it
    const bool train_grouped = group_train_active(c);
    struct TrainJob {
        MlpParams mp; FoldParams fo; ModelDims d; const pr_object_model_t* m; const SavedPlan* sv;
        double* stats; int32_t* stat_count; float* batch; float* h1; float* h2; float* rec_pos; int32_t* rec_flat;
        int k, P, max_tiles; bool frozen, save;
    };
    static thread_local TrainJob train_jobs[PR_MAX_OBJECTS];

    for (int t = 0; t < ntypes; ++t) {
        const TypePlan& tp = plan.type[t];
        int32_t* totals = reinterpret_cast<int32_t*>(ws + tp.totals);
        int32_t* head_counts = reinterpret_cast<int32_t*>(ws + tp.head_counts);
        // one fill: feature-head / tile counters, and - training / differentiable calls - every object's batch-statistics
        // accumulators and divergence array
        PR_TRY(launch_zero_fill(ws + tp.zero_begin, tp.zero_bytes, s));
        const pr_noise_t& noise = t ? c.noise_fine : c.noise_coarse;
        int total_positions = 0;
        // what follows phase 1 of an object's phased launches (training / differentiable calls): batch statistics, the two head
        // phases, the divergence estimate
        // stage 1: statistics of phase 1 -> phase 2; stage 2: statistics of phase 2 -> phase 3; stage 3: counts, divergence.
        // `launch`: enqueue the stage's MLP phase here (one object at a time) - false when the caller groups the objects' launches
        auto finish_object = [&](TrainJob& J, int stage, bool launch) -> int {
            MlpParams& mp = J.mp; FoldParams& fo = J.fo; const ModelDims& d = J.d; const pr_object_model_t& m = *J.m;
            const SavedPlan& sv = *J.sv;
            double* stats = J.stats; int32_t* stat_count = J.stat_count; float* batch = J.batch; float* h1 = J.h1; float* h2 = J.h2;
            float* rec_pos = J.rec_pos; int32_t* rec_flat = J.rec_flat;
            const int k = J.k, P = J.P, max_tiles = J.max_tiles;
            const bool frozen = J.frozen, save = J.save;
            BnFinalizeParams bf;
            memset(&bf, 0, sizeof(bf));
            bf.count = stat_count; bf.momentum = 0.1f;
            bf.frozen = frozen ? 1 : 0;
            if (stage == 1) {
                bf.stats = stats; bf.width = d.W; bf.width_pad = d.Wpad;
                bf.running_mean = m.bn1_mean; bf.running_var = m.bn1_var; bf.num_batches_tracked = (long long*)m.bn1_batches;
                bf.batch_mean = batch; bf.batch_var = batch + MAX_WIDTH;
                PR_TRY(launch_bn_finalize(bf, s));
                fo.bn1_mean = batch; fo.bn1_var = batch + MAX_WIDTH;
                PR_TRY(launch_adain_fold(fo, s));          // second layer still folded with placeholders
                mp.phase = 2; mp.h_in = h1; mp.h_in_width = d.Wpad; mp.h_out = h2; mp.h_out_width = d.W2pad;
                mp.stats = stats + 2 * MAX_WIDTH;
                if (launch) PR_TRY(launch_mlp(mp, max_tiles, false, &m, s));
                return PR_OK;
            }
            if (stage == 2) {
                bf.stats = stats + 2 * MAX_WIDTH; bf.width = d.W2; bf.width_pad = d.W2pad;
                bf.running_mean = m.bn4_mean; bf.running_var = m.bn4_var; bf.num_batches_tracked = (long long*)m.bn4_batches;
                bf.batch_mean = batch + 2 * MAX_WIDTH; bf.batch_var = batch + 3 * MAX_WIDTH;
                PR_TRY(launch_bn_finalize(bf, s));
                fo.bn4_mean = batch + 2 * MAX_WIDTH; fo.bn4_var = batch + 3 * MAX_WIDTH;
                PR_TRY(launch_adain_fold(fo, s));
                mp.phase = 3; mp.h_in = h2; mp.h_in_width = d.W2pad; mp.h_out = nullptr;
                if (launch) PR_TRY(launch_mlp(mp, max_tiles, false, &m, s));
                return PR_OK;
            }
            if (outs[t] && outs[t]->normalised_samples)
                PR_CHECK_HIP(hipMemcpyAsync(outs[t]->normalised_samples + k, stat_count, sizeof(int32_t),
                                            hipMemcpyDeviceToDevice, s));
            if (save && m.has_bender) {
                // Hutchinson divergence of the displacement field (train mode with a graph; zeros without a probe)
                const size_t cap_rows = (size_t)c.frames * c.rays * P;
                float* div = reinterpret_cast<float*>(ws + sv.div);      // (zeroed by the fill at the top of the type)
                // probes: explicit, or generated for training calls with a graph (the reference draws them whenever it
                // trains with a graph, object_composer.py:597)
                NoiseRef probes = make_noise(noise.divergence[k], c, NOISE_DIVERGENCE, t, k);
                if (!(c.flags & PR_FLAG_TRAIN_BN)) probes.generate = 0;
                if (probes.ptr || probes.generate) {
                    DivergenceParams dp;
                    memset(&dp, 0, sizeof(dp));
                    dp.total = totals + k; dp.max_rows = (int)cap_rows;
                    dp.rec_flat = rec_flat; dp.row_flags = mp.row_flags; dp.rec_pos = rec_pos;
                    dp.noise = probes; dp.positions = P;
                    dp.bin = mp.save_bin; dp.bin_pad = d.bin_pad; dp.benc = d.benc; dp.b_octaves = m.bender_octaves;
                    dp.bacts = mp.save_bact; dp.bact_stride = mp.save_bact_stride; dp.BW = d.BW; dp.BWpad = d.BWpad;
                    dp.b_count = m.bender_count; dp.b_skip = m.bender_skip; dp.bin_real = d.bin;
                    dp.layers = m.bender; dp.out_head = m.bender_out; dp.braw = mp.save_braw;
                    bbox_split(m, dp.lo, dp.hi, nullptr);
                    dp.canonical = (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0;
                    dp.t0 = reinterpret_cast<float*>(ws + plan.div_t0);
                    dp.ta = reinterpret_cast<float*>(ws + plan.div_ta);
                    dp.tb = reinterpret_cast<float*>(ws + plan.div_tb);
                    dp.div = div;
                    PR_TRY(launch_divergence(dp, s));
                }
            }
            return PR_OK;
        };
        for (int pass = (place_grouped && t == 0) ? 0 : 1; pass < 2; ++pass) {
        // pass 0 (grouped coarse placement only): collect the placement / compaction jobs of every object and launch them;
        // pass 1: everything else per object
        if (pass == 1 && place_grouped && t == 0) PR_TRY(launch_placement_group(place_jobs, fill_jobs, total_ptrs, K, s));
        for (int k = 0; k < K; ++k) {
            const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
            const float* packed = static_cast<const float*>(t ? objs[k].packed_fine : objs[k].packed_coarse);
            ModelDims d;
            PackedLayout l;
            PR_TRY(compute_dims(m, &d));
            PR_TRY(compute_layout(m, d, &l));
            const int P = m.positions;
            if (pass == 1) total_positions += P;
            int32_t* block_sums = block_sums_all + (size_t)k * plan.nblocks256;
            int32_t* block_offsets = block_offsets_all + (size_t)k * plan.nblocks256;
            const bool placed = place_grouped && t == 0;     // pass 0 has placed and compacted this object
            float* t_arr = reinterpret_cast<float*>(ws + tp.t[k]);
            float* sigma = reinterpret_cast<float*>(ws + tp.sigma[k]);
            int32_t* slot = reinterpret_cast<int32_t*>(ws + tp.slot[k]);
            float* dispmag = m.has_bender ? reinterpret_cast<float*>(ws + tp.dispmag[k]) : nullptr;
            float* feat = reinterpret_cast<float*>(ws + tp.feat[k]);
            float* adain = reinterpret_cast<float*>(ws + tp.adain[k]);
            const bool save = (c.flags & PR_FLAG_SAVE_FOR_BACKWARD) != 0;
            const SavedPlan& sv = tp.saved[k];
            if (save) {
                rec_pos = reinterpret_cast<float*>(ws + sv.rec_pos);
                rec_flat = reinterpret_cast<int32_t*>(ws + sv.rec_flat);
            } else if (grouped) {
                rec_pos = reinterpret_cast<float*>(ws + plan.rec_pos_k[k]);
                rec_flat = reinterpret_cast<int32_t*>(ws + plan.rec_flat_k[k]);
            }

            // ---- sample placement --------------------------------------------------------------
            if (t == 0) {
                PlaceParams pp;
                memset(&pp, 0, sizeof(pp));
                pp.frames = c.frames; pp.rays = c.rays; pp.positions = P; pp.objects = K; pp.object_index = k;
                pp.ray_origins = c.ray_origins; pp.ray_directions = c.ray_directions; pp.w2o = c.w2o;
                pp.in_scene = c.object_in_scene;
                bbox_split(m, pp.lo, pp.hi, nullptr);
                pp.z_near_min = m.z_near_min; pp.z_far_max = m.z_far_max; pp.empty_alpha = m.empty_space_alpha;
                pp.linspace = c.linspace_coarse[k];
                pp.jitter = perturb_noise(c.noise_coarse.jitter[k], c, NOISE_JITTER, 0, k);
                pp.t = t_arr; pp.sigma = sigma; pp.dispmag = dispmag; pp.block_sums = block_sums;
                if (pass == 0) place_jobs[k] = pp;
                else if (!placed) PR_TRY(launch_place_coarse(pp, s));
            } else {
                const TypePlan& cp = plan.type[0];
                ResampleParams rp;
                memset(&rp, 0, sizeof(rp));
                rp.frames = c.frames; rp.rays = c.rays; rp.objects = K; rp.object_index = k;
                rp.pc = objs[k].coarse.positions; rp.pf = c.positions_fine[k];
                rp.ray_origins = c.ray_origins; rp.ray_directions = c.ray_directions; rp.w2o = c.w2o;
                rp.in_scene = c.object_in_scene;
                bbox_split(m, rp.lo, rp.hi, nullptr);
                rp.empty_alpha = m.empty_space_alpha;
                rp.t_coarse = reinterpret_cast<const float*>(ws + cp.t[k]);
                rp.sigma_coarse = reinterpret_cast<const float*>(ws + cp.sigma[k]);
                rp.alpha_noise = perturb_noise(c.noise_coarse.alpha[k], c, NOISE_ALPHA, 0, k);
                rp.u_fixed = c.linspace_fine[k];
                rp.u_random = perturb_noise(c.noise_coarse.pdf[k], c, NOISE_PDF, 0, k);
                rp.t_fine = t_arr; rp.sigma_fine = sigma; rp.dispmag_fine = dispmag; rp.block_sums = block_sums;
                PR_TRY(launch_resample(rp, s));
            }
            if (pass == 1 && !placed) PR_TRY(launch_scan(block_sums, block_offsets, totals + k, plan.nblocks256, s));

            // ---- compaction --------------------------------------------------------------------
            FillParams fp;
            memset(&fp, 0, sizeof(fp));
            fp.frames = c.frames; fp.rays = c.rays; fp.positions = P; fp.objects = K; fp.object_index = k;
            fp.ray_origins = c.ray_origins; fp.ray_directions = c.ray_directions; fp.w2o = c.w2o;
            fp.in_scene = c.object_in_scene;
            bbox_split(m, fp.lo, fp.hi, nullptr);
            fp.t = t_arr; fp.block_offsets = block_offsets; fp.rec_pos = rec_pos; fp.rec_flat = rec_flat; fp.slot = slot;
            if (pass == 0) {
                fill_jobs[k] = fp;
                total_ptrs[k] = totals + k;
                continue;
            }
            if (!placed) PR_TRY(launch_fill(fp, s));

            // ---- style affine + BatchNorm fold; fused MLP ---------------------------------------
            FoldParams fo;
            memset(&fo, 0, sizeof(fo));
            fo.frames = c.frames; fo.objects = K; fo.object_index = k;
            fo.style = c.style; fo.S = m.style_features;
            fo.affine1 = m.affine1; fo.bn1_mean = m.bn1_mean; fo.bn1_var = m.bn1_var;
            fo.affine4 = m.affine4; fo.bn4_mean = m.bn4_mean; fo.bn4_var = m.bn4_var;
            fo.eps = m.bn_eps;
            fo.W = d.W; fo.Wpad = d.Wpad; fo.W2 = d.W2; fo.W2pad = d.W2pad;
            fo.table = adain; fo.row_floats = adain_row_floats(d);

            MlpParams mp;
            memset(&mp, 0, sizeof(mp));
            // training calls with PR_FLAG_SPLIT_BACKWARD: phase 1 of the grouped launches reads the bf16-triple packings
            const bool split3 = train_grouped && (c.flags & PR_FLAG_SPLIT_BACKWARD) && (c.flags & PR_FLAG_SAVE_FOR_BACKWARD);
            PR_TRY(build_mlp_layers(m, d, l, packed, &mp, split3));
            mp.rec_pos = rec_pos; mp.rec_flat = rec_flat; mp.total = totals + k;
            mp.samples_per_frame = c.rays * P; mp.positions = P; mp.rays = c.rays;
            mp.canonical = (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0;
            bbox_split(m, mp.lo, mp.hi, mp.size);
            mp.empty_alpha = m.empty_space_alpha;
            mp.in_scene = c.object_in_scene + k; mp.in_scene_stride = K;
            mp.ray_directions = c.ray_directions; mp.ray_origins = c.ray_origins;
            mp.w2o = c.w2o + (size_t)k * 12; mp.w2o_stride = K * 12;
            mp.deformation = c.deformation + (size_t)k * m.deformation_features;
            mp.deformation_stride = K * m.deformation_features;
            mp.adain = adain; mp.adain_stride = fo.row_floats;
            mp.sigma = sigma; mp.dispmag = dispmag; mp.feat = feat;
            if (gate) {
                mp.gate = 1;
                mp.pend_act = reinterpret_cast<float*>(ws + plan.pend_act);
                mp.pend_meta = reinterpret_cast<int32_t*>(ws + plan.pend_meta);
                mp.head_count = head_counts + k;
            }
            mp.tile_counter = head_counts + PR_MAX_OBJECTS + k;
            if (outs[t] && outs[t]->sample_delta[k]) {
                PR_TRY(launch_zero_fill(outs[t]->sample_delta[k], sizeof(float) * 3 * (size_t)c.frames * c.rays * P, s));
                if (m.has_bender) mp.delta_dense = outs[t]->sample_delta[k];
            }
            const size_t cap = (size_t)c.frames * c.rays * P;
            const int max_tiles = (int)cap;   // rows: every launcher derives its own tile count
            if (!(c.flags & (PR_FLAG_TRAIN_BN | PR_FLAG_SAVE_FOR_BACKWARD))) {
                if (grouped) {
                    fold_jobs[k] = fo;       // folded and evaluated together once every object of the type is prepared
                    jobs[k] = mp;
                    job_rows[k] = max_tiles;
                    continue;
                }
                PR_TRY(launch_adain_fold(fo, s));
                if (c.precision != PR_PRECISION_FP32 && !naive)
                    PR_TRY(launch_mlp_split(mp, max_tiles, c.precision == PR_PRECISION_F16 ? 1 : 3, s));
                else
                    PR_TRY(launch_mlp(mp, max_tiles, naive, &m, s));
            } else {
                // BatchNorm in training mode: the batch statistics of the first (second) AdaIN layer are
                // a reduction over every evaluated sample of this object call, between two matmuls.
                // Differentiable eval-mode calls (SAVE without TRAIN_BN) use the same phases - they leave the pre-BatchNorm
                // activations h1 / h2 behind for the backward pass - with the running statistics frozen.
                const bool frozen = !(c.flags & PR_FLAG_TRAIN_BN);
                PR_REQUIRE(!naive, "the scalar debugging kernel has no train-mode BatchNorm");
                double* stats = reinterpret_cast<double*>(ws + sv.stats);               // (zeroed by the fill above)
                int32_t* stat_count = reinterpret_cast<int32_t*>(ws + sv.stat_count);
                float* batch = reinterpret_cast<float*>(ws + (save ? sv.batch : plan.batch_stats));
                float* h1 = reinterpret_cast<float*>(ws + (save ? sv.h1 : plan.h1));
                float* h2 = reinterpret_cast<float*>(ws + (save ? sv.h2 : plan.h2));
                mp.row_flags = reinterpret_cast<int32_t*>(ws + (save ? sv.row_flags : plan.row_flags));
                mp.stat_count = stat_count;
                if (save) {
                    const size_t cap_rows = (size_t)c.frames * c.rays * P;
                    mp.save_enc = reinterpret_cast<float*>(ws + sv.enc);
                    mp.save_act = reinterpret_cast<float*>(ws + sv.act);
                    mp.save_act_stride = cap_rows * d.Wpad;
                    mp.save_bits = reinterpret_cast<unsigned char*>(ws + sv.bits);
                    mp.save_bits_stride = relu_bits_bytes(cap_rows, d.Wpad);
                    if (m.has_bender) {
                        mp.save_bin = reinterpret_cast<float*>(ws + sv.bin);
                        mp.save_bact = reinterpret_cast<float*>(ws + sv.bact);
                        mp.save_bact_stride = cap_rows * d.BWpad;
                        mp.save_bbits = reinterpret_cast<unsigned char*>(ws + sv.bbits);
                        mp.save_bbits_stride = relu_bits_bytes(cap_rows, d.BWpad);
                        mp.save_braw = reinterpret_cast<float*>(ws + sv.braw);
                        mp.save_delta = reinterpret_cast<float*>(ws + sv.delta);
                    }
                }
                mp.phase = 1; mp.h_out = h1; mp.h_out_width = d.Wpad; mp.stats = stats;
                TrainJob& J = train_jobs[k];
                J.mp = mp; J.fo = fo; J.d = d; J.m = &m; J.sv = &sv;
                J.stats = stats; J.stat_count = stat_count; J.batch = batch; J.h1 = h1; J.h2 = h2; J.rec_pos = rec_pos; J.rec_flat = rec_flat;
                J.k = k; J.P = P; J.max_tiles = max_tiles; J.frozen = frozen; J.save = save;
                if (train_grouped) {
                    jobs[k] = mp;
                    job_rows[k] = max_tiles;
                } else {
                    PR_TRY(launch_mlp(mp, max_tiles, false, &m, s));
                    for (int stage = 1; stage <= 3; ++stage) PR_TRY(finish_object(J, stage, true));
                }
            }
        }
        }   // pass
        if (train_grouped) {
            PR_TRY(launch_mlp_group(jobs, job_rows, K, s));                  // phase 1 of every object
            static thread_local BnFoldJobs bj;
            for (int stage = 1; stage <= 2; ++stage) {                       // statistics + fold of all objects, then their next phase
                for (int k = 0; k < K; ++k) {
                    TrainJob& J = train_jobs[k];
                    const pr_object_model_t& m = *J.m;
                    BnFoldJob& b = bj.job[k];
                    memset(&b, 0, sizeof(b));
                    b.count = J.stat_count; b.momentum = 0.1f; b.frozen = J.frozen ? 1 : 0;
                    b.style = c.style + (size_t)k * m.style_features; b.style_stride = K * m.style_features; b.S = m.style_features;
                    b.frames = c.frames; b.eps = m.bn_eps;
                    b.table = const_cast<float*>(J.mp.adain); b.row_floats = J.mp.adain_stride;
                    if (stage == 1) {
                        b.stats = J.stats; b.width = J.d.W; b.width_pad = J.d.Wpad;
                        b.running_mean = m.bn1_mean; b.running_var = m.bn1_var; b.num_batches_tracked = (long long*)m.bn1_batches;
                        b.batch_mean = J.batch; b.batch_var = J.batch + MAX_WIDTH;
                        b.affine = m.affine1; b.g_off = 0; b.b_off = J.d.Wpad;
                        J.mp.split3 = 0;     // (the head phases read fp32 fragments)
                        J.mp.phase = 2; J.mp.h_in = J.h1; J.mp.h_in_width = J.d.Wpad; J.mp.h_out = J.h2; J.mp.h_out_width = J.d.W2pad;
                        J.mp.stats = J.stats + 2 * MAX_WIDTH;
                    } else {
                        b.stats = J.stats + 2 * MAX_WIDTH; b.width = J.d.W2; b.width_pad = J.d.W2pad;
                        b.running_mean = m.bn4_mean; b.running_var = m.bn4_var; b.num_batches_tracked = (long long*)m.bn4_batches;
                        b.batch_mean = J.batch + 2 * MAX_WIDTH; b.batch_var = J.batch + 3 * MAX_WIDTH;
                        b.affine = m.affine4; b.g_off = 2 * J.d.Wpad; b.b_off = 2 * J.d.Wpad + J.d.W2pad;
                        b.normalised_out = (outs[t] && outs[t]->normalised_samples) ? outs[t]->normalised_samples + k : nullptr;
                        J.mp.phase = 3; J.mp.h_in = J.h2; J.mp.h_in_width = J.d.W2pad; J.mp.h_out = nullptr;
                    }
                    PR_REQUIRE(b.affine.weight && b.affine.bias && b.running_mean && b.running_var, "AdaIN parameters missing");
                    jobs[k] = J.mp;
                }
                PR_TRY(launch_bn_fold_group(bj, K, s));
                PR_TRY(launch_mlp_group(jobs, job_rows, K, s));
            }
            // Hutchinson divergence of the displacement fields (train mode with a graph): the ray benders of the call as one launch
            static thread_local DivChainJob dj[PR_MAX_OBJECTS];
            long div_rows[PR_MAX_OBJECTS];
            int div_jobs = 0;
            for (int k = 0; k < K; ++k) {
                TrainJob& J = train_jobs[k];
                const pr_object_model_t& m = *J.m;
                if (!(J.save && m.has_bender)) continue;
                NoiseRef probes = make_noise(noise.divergence[k], c, NOISE_DIVERGENCE, t, k);
                if (!(c.flags & PR_FLAG_TRAIN_BN)) probes.generate = 0;
                if (!(probes.ptr || probes.generate)) continue;
                if (!div_chain_supported(J.d.BWpad, J.d.bin_pad)) {
                    PR_TRY(finish_object(J, 3, false));          // (unusually wide bender: one product per layer)
                    continue;
                }
                PackedLayout l;
                PR_TRY(compute_layout(m, J.d, &l));
                const float* packed = static_cast<const float*>(t ? objs[k].packed_fine : objs[k].packed_coarse);
                const size_t cap_rows = (size_t)c.frames * c.rays * J.P;
                DivChainJob& q = dj[div_jobs];
                memset(&q, 0, sizeof(q));
                q.total = totals + k; q.rec_flat = J.rec_flat; q.row_flags = J.mp.row_flags; q.rec_pos = J.rec_pos;
                q.noise = probes; q.positions = J.P;
                q.bin = J.mp.save_bin; q.bin_pad = J.d.bin_pad; q.benc = J.d.benc; q.b_octaves = m.bender_octaves;
                q.bbits = J.mp.save_bbits; q.bbits_stride = J.mp.save_bbits_stride;
                q.BW = J.d.BW; q.BWpad = J.d.BWpad; q.b_count = m.bender_count; q.b_skip = m.bender_skip;
                for (int j = 0; j < m.bender_count; ++j) q.seg0[j] = Seg{packed + l.b_seg_off[j][0], (j == 0 ? J.d.bin_pad : J.d.BWpad) / 8, 0};
                q.seg1 = Seg{packed + l.b_seg_off[m.bender_skip][1], J.d.bin_pad / 8, 0};
                q.w_out = packed + l.b_out_off;
                q.braw = J.mp.save_braw;
                bbox_split(m, q.lo, q.hi, nullptr);
                q.canonical = (c.flags & PR_FLAG_CANONICAL_POSE) ? 1 : 0;
                q.div = reinterpret_cast<float*>(ws + J.sv->div);
                q.tile_counter = head_counts + 2 * PR_MAX_OBJECTS + k;
                div_rows[div_jobs] = (long)cap_rows;
                ++div_jobs;
            }
            if (div_jobs) PR_TRY(launch_div_chain_group(dj, div_rows, div_jobs, s));
        }

        if (grouped) {
            PR_TRY(launch_adain_fold_group(fold_jobs, K, s));
            if (c.precision != PR_PRECISION_FP32)
                PR_TRY(launch_mlp_split_group(jobs, job_rows, K, c.precision == PR_PRECISION_F16 ? 1 : 3, s));
            else
                PR_TRY(launch_mlp_group(jobs, job_rows, K, s));
        }

        // ---- compositing ------------------------------------------------------------------------
        const pr_outputs_t* out = outs[t];
        CompositeParams cp;
        memset(&cp, 0, sizeof(cp));
        cp.frames = c.frames; cp.rays = c.rays; cp.objects = K; cp.static_objects = c.static_objects;
        cp.F = objs[0].coarse.output_features;
        cp.fix_overlaps = (c.flags & PR_FLAG_FIX_OVERLAPS) ? 1 : 0;
        cp.sigmoid = (c.flags & PR_FLAG_SIGMOID_FEATURES) ? 1 : 0;
        cp.total_positions = total_positions;
        int ss = 64;
        while (ss < total_positions) ss <<= 1;
        cp.sort_size = ss;
        cp.ray_directions = c.ray_directions;
        cp.noise_global = perturb_noise(noise.integrate_global, c, NOISE_INTEGRATE_GLOBAL, t, 0);
        for (int k = 0; k < K; ++k) {
            const pr_object_model_t& m = t ? objs[k].fine : objs[k].coarse;
            PR_REQUIRE(m.output_features == cp.F, "all objects must share output_features");
            CompositeObject& o = cp.obj[k];
            o.t = reinterpret_cast<const float*>(ws + tp.t[k]);
            o.sigma = reinterpret_cast<const float*>(ws + tp.sigma[k]);
            o.slot = reinterpret_cast<const int32_t*>(ws + tp.slot[k]);
            o.dispmag = m.has_bender ? reinterpret_cast<const float*>(ws + tp.dispmag[k]) : nullptr;
            o.divergence = ((c.flags & PR_FLAG_SAVE_FOR_BACKWARD) && m.has_bender)
                               ? reinterpret_cast<const float*>(ws + tp.saved[k].div) : nullptr;
            if (o.divergence) cp.any_divergence = 1;
            o.feat = reinterpret_cast<const float*>(ws + tp.feat[k]);
            o.noise = perturb_noise(noise.integrate[k], c, NOISE_INTEGRATE, t, k);
            o.positions = m.positions;
            if (out) o.out = out->object[k];
        }
        if (out) {
            cp.global = out->global;
            cp.decoder = out->decoder;
        }
        PR_TRY(launch_composite(cp, s));

        // ---- optional exports -------------------------------------------------------------------
        if (out) {
            for (int k = 0; k < K; ++k) {
                const size_t cap = (size_t)c.frames * c.rays * tp.positions[k];
                if (out->sample_t[k])
                    PR_CHECK_HIP(hipMemcpyAsync(out->sample_t[k], ws + tp.t[k], sizeof(float) * cap, hipMemcpyDeviceToDevice, s));
                if (out->sample_sigma[k])
                    PR_CHECK_HIP(hipMemcpyAsync(out->sample_sigma[k], ws + tp.sigma[k], sizeof(float) * cap, hipMemcpyDeviceToDevice, s));
                if (out->sample_slot[k])
                    PR_CHECK_HIP(hipMemcpyAsync(out->sample_slot[k], ws + tp.slot[k], sizeof(int32_t) * cap, hipMemcpyDeviceToDevice, s));
            }
            if (out->evaluated_samples)
                PR_CHECK_HIP(hipMemcpyAsync(out->evaluated_samples, totals, sizeof(int32_t) * K, hipMemcpyDeviceToDevice, s));
            if (out->head_samples)   // without the gate every evaluated sample goes through the feature head
                PR_CHECK_HIP(hipMemcpyAsync(out->head_samples, gate ? head_counts : totals, sizeof(int32_t) * K,
                                            hipMemcpyDeviceToDevice, s));
        }
    }
    return PR_OK;
}

}  // namespace pr

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int pr_workspace_size(const pr_call_t* call, const pr_object_t* objects, size_t* bytes) {
    PR_REQUIRE(call && objects && bytes, "pr_workspace_size: NULL argument");
    PR_TRY(pr::validate_call(*call, objects));
    pr::Plan plan;
    PR_TRY(pr::make_plan(*call, objects, &plan));
    *bytes = plan.bytes;
    return PR_OK;
}

extern "C" int pr_render_forward(const pr_call_t* call, const pr_object_t* objects, const pr_outputs_t* coarse,
                                 const pr_outputs_t* fine, void* workspace, size_t workspace_bytes, void* stream) {
    PR_REQUIRE(call && objects && workspace, "pr_render_forward: NULL argument");
    PR_TRY(pr::validate_call(*call, objects));
    pr::Plan plan;
    PR_TRY(pr::make_plan(*call, objects, &plan));
    if (workspace_bytes < plan.bytes) {
        pr::set_error("workspace too small: %zu bytes given, %zu needed", workspace_bytes, plan.bytes);
        return PR_ERR_WORKSPACE;
    }
    PR_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const pr_outputs_t* outs[2] = {coarse, fine};
    return pr::render(*call, objects, outs, static_cast<char*>(workspace), plan, (hipStream_t)stream);
}

namespace pr {
__global__ __launch_bounds__(256) void k_noise_fill(NoiseRef n, int normal, long count, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256)
        out[i] = normal ? noise_normal(n, i, 1, 0) : noise_uniform(n, i, 1, 0);
}
}  // namespace pr

extern "C" int pr_noise_fill(uint64_t seed, int32_t kind, int32_t type, int32_t object, int64_t count, float* out, void* stream) {
    PR_REQUIRE(kind >= 0 && kind <= 5 && (type == 0 || type == 1) && object >= 0 && object < PR_MAX_OBJECTS, "pr_noise_fill: bad stream");
    if (count <= 0) return PR_OK;
    PR_REQUIRE(out != nullptr, "pr_noise_fill: NULL output");
    pr_call_t c;
    memset(&c, 0, sizeof(c));
    c.flags = PR_FLAG_DEVICE_NOISE | PR_FLAG_PERTURB;
    c.noise_seed = seed;
    c.rays = 1;
    pr::NoiseRef n = pr::make_noise(nullptr, c, kind, type, object);
    n.rays = n.total_rays = 1;      // element i is "ray i, element 0 of 1": the flat tensor index
    const int normal = (kind == pr::NOISE_JITTER || kind == pr::NOISE_PDF) ? 0 : 1;
    const long blocks = (count + 255) / 256;
    hipLaunchKernelGGL(pr::k_noise_fill, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream, n, normal,
                       (long)count, out);
    PR_LAUNCH_CHECK();
    return PR_OK;
}

extern "C" int pr_profile_enable(int enable) {
    pr::g_profile_on.store(enable != 0, std::memory_order_relaxed);
    return PR_OK;
}

extern "C" int pr_profile_collect(double* milliseconds, int32_t* launches) {
    PR_REQUIRE(milliseconds && launches, "pr_profile_collect: NULL argument");
    std::lock_guard<std::mutex> lock(pr::g_profile_mutex);
    for (int c = 0; c < PR_PROFILE_CATEGORIES; ++c) {
        milliseconds[c] = 0.0;
        launches[c] = 0;
    }
    for (auto& r : pr::g_profile) {
        PR_CHECK_HIP(hipEventSynchronize(r.stop));
        float ms = 0.f;
        PR_CHECK_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        if (r.category >= 0 && r.category < PR_PROFILE_CATEGORIES) {
            milliseconds[r.category] += ms;
            launches[r.category] += 1;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    pr::g_profile.clear();
    return PR_OK;
}

// Node census of a recorded HIP graph (child graphs included): what EnvironmentModel.frame_replay asks before it trusts a
// recording that holds launches of modules this library does not own (see include/playrender.h).
namespace pr {
static int census_of(hipGraph_t graph, int32_t* counts, int depth) {
    size_t n = 0;
    PR_CHECK_HIP(hipGraphGetNodes(graph, nullptr, &n));
    if (n == 0) return PR_OK;
    std::vector<hipGraphNode_t> nodes(n);
    PR_CHECK_HIP(hipGraphGetNodes(graph, nodes.data(), &n));
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        PR_CHECK_HIP(hipGraphNodeGetType(nodes[i], &type));
        counts[0] += 1;
        if (type == hipGraphNodeTypeKernel) counts[1] += 1;
        else if (type == hipGraphNodeTypeMemset) counts[2] += 1;
        else if (type == hipGraphNodeTypeMemcpy) counts[3] += 1;
        else if (type == hipGraphNodeTypeGraph && depth < 8) {
            hipGraph_t child = nullptr;
            PR_CHECK_HIP(hipGraphChildGraphNodeGetGraph(nodes[i], &child));
            const int status = census_of(child, counts, depth + 1);
            if (status != PR_OK) return status;
        }
    }
    return PR_OK;
}
}  // namespace pr

extern "C" int pr_graph_node_census(void* graph, int32_t* counts) {
    PR_REQUIRE(graph && counts, "pr_graph_node_census: NULL pointer");
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    return pr::census_of((hipGraph_t)graph, counts, 0);
}

extern "C" int pr_abi_version(void) { return PR_ABI_VERSION; }

extern "C" const char* pr_last_error(void) { return pr::g_error; }

extern "C" int pr_device_info(int32_t* compute_units, int32_t* lds_bytes, char* arch_name, size_t arch_name_len) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        pr::set_error("no HIP device visible");
        return PR_ERR_NO_DEVICE;
    }
    int dev = 0;
    PR_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    if (arch_name && arch_name_len) {
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return PR_OK;
}
