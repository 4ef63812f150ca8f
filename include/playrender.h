/*
 * playrender.h - C ABI of libplayrender.so, the MI355X (gfx950) volumetric renderer.
 *
 * The reference (willi-menapace/PlayableEnvironments) is pure Python/PyTorch: the interface this
 * library replaces is the tensor-in / dict-of-tensors-out call
 *     ObjectComposer.forward(ray_origins, ray_directions, focal_normals, transformation_matrix_w2o,
 *                            style, deformation, object_in_scene, perturb, ...)
 * (model/object_composer.py:786-892) plus the ray set-up of
 * EnvironmentModel.forward_from_scene_encoding (model/environment_model.py:1080-1112).  The Python
 * package `playableenvironments_amd` keeps those signatures and marshals raw device pointers into
 * the entry points below through ctypes (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions
 *   - every function returns 0 on success, a negative pr_status on failure; pr_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - all pointers are DEVICE pointers unless a parameter says "host"; the library never
 *     allocates or frees device memory, never synchronises the device and enqueues all work on
 *     the hipStream_t passed in (void* so that the header needs no HIP include);
 *   - all floating-point data is IEEE fp32, C-contiguous, in the layouts written next to each field;
 *   - "N" = frames (all leading dims of the reference tensors flattened), "R" = rays per frame,
 *     "K" = object instances, "P_k" = samples per ray of object k, "F" = output features.
 */
#ifndef PLAYRENDER_H
#define PLAYRENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PR_ABI_VERSION 5
#define PR_MAX_OBJECTS 8
#define PR_MAX_LAYERS 12
#define PR_MAX_OCTAVES 16

typedef enum pr_status {
    PR_OK = 0,
    PR_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    PR_ERR_WORKSPACE = -2,    /* workspace too small */
    PR_ERR_HIP = -3,          /* a HIP runtime call failed */
    PR_ERR_NO_DEVICE = -4     /* no gfx950 device visible */
} pr_status;

/* flags of pr_call_t.flags */
#define PR_FLAG_PERTURB        1u   /* stratified jitter + alpha noise; noise tensors must be supplied */
#define PR_FLAG_CANONICAL_POSE 2u   /* zero the ray-bender displacements (canonical_pose=True) */
#define PR_FLAG_FIX_OVERLAPS   4u   /* config["model"]["fix_object_overlaps"] */
#define PR_FLAG_NAIVE_MLP      8u   /* debugging: scalar one-thread-per-sample MLP kernel instead of MFMA */
#define PR_FLAG_TRAIN_BN       16u  /* module.training: the AdaIN BatchNorm1d layers normalise with batch statistics of
                                       the evaluated samples of each object call and update running_mean / running_var /
                                       num_batches_tracked in place (model/layers/adain.py:47,58) */

#define PR_FLAG_SAVE_FOR_BACKWARD 32u /* keep every intermediate pr_render_backward needs inside the workspace; the
                                       workspace must stay untouched until the backward call.  With PR_FLAG_TRAIN_BN
                                       the backward pass differentiates the batch statistics (training); without it
                                       the running statistics are constants (a differentiable eval-mode call, e.g.
                                       test-time optimisation of poses or style codes).  PR_PRECISION_FP32 only. */

#define PR_FLAG_GATE_HEAD 64u        /* sigma-gated feature head: samples whose raw density is <= 0 have alpha = 1 - exp(-relu(sigma) dt)
                                       = 0 exactly, so their 192-channel feature rows never reach a compositing sum
                                       (model/object_composer.py:180-214); with this flag the feature head (3 of the 11
                                       matrix products of a sample) runs on the other samples only.  Results are
                                       bit-identical.  Honoured for evaluation calls only - ignored with PR_FLAG_PERTURB
                                       or any integrate-noise pointer (noise is added to the density before the ReLU),
                                       PR_FLAG_TRAIN_BN (batch statistics need every row), PR_FLAG_SAVE_FOR_BACKWARD and
                                       PR_FLAG_NAIVE_MLP. */

#define PR_FLAG_DEVICE_NOISE 128u    /* with PR_FLAG_PERTURB (and for the Hutchinson probes of training calls): every noise tensor whose
                                       pointer in pr_noise_t is NULL is GENERATED inside the kernels that consume it - a
                                       counter-based generator (Philox4x32-10) keyed by pr_call_t.noise_seed and the tensor's
                                       stream id, indexed by the element index, so pr_render_backward regenerates exactly
                                       the forward pass's values and no (N,R,P) noise tensor is materialised (the reference
                                       draws them with torch.rand / torch.randn: utils/lib_3d/ray_helper.py:1275,1380,
                                       model/object_composer.py:553,597,751).  pr_noise_fill writes the same values to a
                                       tensor (tests replay them through the explicit path and the oracle). */

#define PR_FLAG_SIGMOID_FEATURES 512u /* config["model"]["apply_activation"]: the raw features of every sample go through a sigmoid before
                                        they are composited (object_composer.py:548-549, :573-574; RGB-output models).  Applied
                                        where the compositing kernel reads a feature row; samples without a row (outside the box:
                                        raw feature 0) composite sigmoid(0) = 0.5 wherever their weight is not zero. */
#define PR_FLAG_SPLIT_BACKWARD 1024u /* differentiable calls (with PR_FLAG_SAVE_FOR_BACKWARD, on the forward AND the backward call; ABI 5):
                                       the matrix products of pr_render_backward in split precision, fp32 accumulation, every operand an
                                       fp16 pair (x = hi + lo, three fp16 MFMAs per product): the backward chains on the gradient tile
                                       times a power of two chosen per 64-row tile and the weights times 2^8, the weight gradients on
                                       16-row half slabs - gradient rows times a power of two alpha, activation rows times C / alpha, C a
                                       running power of two per work item (all scalings exact; a build with -DPR_TNALL_F16=0 keeps the
                                       round-4 form of the weight gradients, three bf16 terms per operand and six bf16 MFMAs).  On the forward call the flag
                                       selects fp16-pair products for phase 1 of a train-mode forward pass too.  The call
                                       stays PR_PRECISION_FP32 (fp32-packed weights, which carry both split forms as well).  Products
                                       that have no split kernel run the exact fp32 one. */
#define PR_FLAG_DIVERGENCE_GRAD 256u /* pr_backward_workspace_size / pr_render_backward: gradients of integrated_divergence are
                                       given (pr_entry_grads_t.integrated_divergence); the backward pass then differentiates the
                                       Hutchinson estimate e^T (d delta / dx) e through the ray bender (the reference's double
                                       backward, object_composer.py:582-601 with create_graph=True): one more pass over the
                                       bender with the probe tangents in place of the activations.  Sizes the tangent stack
                                       inside the backward workspace; ignored by pr_render_forward. */

/* One nn.Linear in the reference layout: weight (out_features, in_features) row-major, bias (out) or NULL. */
typedef struct pr_linear_t {
    const float* weight;
    const float* bias;
    int32_t out_features;
    int32_t in_features;
} pr_linear_t;

/*
 * One object model = RayBendingStyleNerfModel (model/nerf_models/ray_bending_style_nerf_model.py:12)
 * with its raw parameter pointers (the nn.Parameter storages, zero copy).
 */
typedef struct pr_object_model_t {
    int32_t kind;                    /* 0 = AdaInStyleNerfModel, 1 = SkyboxAdaInStyleNerfModelV3 */
    int32_t has_bender;              /* 1 = PositionalRayBender, 0 = ZeroedRayBender */
    int32_t positions;               /* samples per ray evaluated by THIS model: P_coarse, or P_coarse + P_fine */
    int32_t style_features;          /* S */
    int32_t deformation_features;    /* D */
    int32_t output_features;         /* F */
    int32_t layers_width;            /* backbone width (256) */
    int32_t backbone_count;          /* 8 */
    int32_t skip_layer_idx;          /* 4 */
    int32_t octaves;                 /* NeRF positional-encoder octaves (10); append_original is required */
    int32_t bender_width;            /* 128 */
    int32_t bender_count;            /* 6 */
    int32_t bender_skip;             /* 3 */
    int32_t bender_octaves;          /* 6 */
    float bender_octave_weights[PR_MAX_OCTAVES]; /* annealing weights evaluated on the host from current_step
                                                    (model/annealable_positional_encoder.py:59-63) */
    float bbox[6];                   /* x_lo, x_hi, y_lo, y_hi, z_lo, z_hi */
    float empty_space_alpha;
    float z_near_min;
    float z_far_max;
    float bn_eps;                    /* 1e-5 */
    /* nerf_model.* */
    pr_linear_t backbone[PR_MAX_LAYERS];
    pr_linear_t alpha_head;          /* weight == NULL for the skybox (sigma == 10) */
    pr_linear_t head0;               /* features_head.0, no bias */
    pr_linear_t affine1;             /* features_head.1.affine_transform (2*W, S) */
    float* bn1_mean;                 /* features_head.1.ada_in.normalization.running_mean (W); written with PR_FLAG_TRAIN_BN */
    float* bn1_var;
    int64_t* bn1_batches;            /* ...num_batches_tracked (int64 scalar) or NULL */
    pr_linear_t head3;               /* features_head.3, no bias (W/2, W) */
    pr_linear_t affine4;             /* features_head.4.affine_transform (W, S) */
    float* bn4_mean;
    float* bn4_var;
    int64_t* bn4_batches;
    pr_linear_t head6;               /* features_head.6 (F, W/2) */
    /* ray_bender.* (ignored when has_bender == 0) */
    pr_linear_t bender[PR_MAX_LAYERS];
    pr_linear_t bender_out;          /* output_head (3, BW), no bias */
} pr_object_model_t;

/* Bytes of the MFMA-fragment-ordered copy of one model's weights (see DESIGN.md "packed weights"). */
int pr_packed_size(const pr_object_model_t* model, size_t* bytes);

/* Gathers the raw parameters into `packed` (device, pr_packed_size bytes, 256-B aligned).  Must be
 * re-run whenever parameter VALUES change; cheap (one pass over ~2.9 MB).
 * precision: PR_PRECISION_FP32 = fp32 MFMA fragments (exact fp32 arithmetic); PR_PRECISION_F16X3 = every weight
 * as an fp16 pair (hi, lo = w - hi) for the split kernel, which evaluates
 * a*w ~ a_hi*w_hi + a_hi*w_lo + a_lo*w_hi with three fp16 MFMAs and fp32 accumulation
 * (~22 significant bits).  Both layouts have the same size.  PR_PRECISION_F16 = the F16X3 layout (the same bytes)
 * evaluated with the a_hi*w_hi product only: plain fp16 operands (11 significant bits each), fp32 accumulation, one
 * MFMA per step instead of three - the throughput tier for interactive play, ~1e-3 relative error on the rendered
 * features (tests/test_gpu.py holds it to >= 40 dB PSNR against the oracle); not for parity checks. */
#define PR_PRECISION_FP32  0
#define PR_PRECISION_F16X3 1
#define PR_PRECISION_F16   2
int pr_pack_model(const pr_object_model_t* model, int32_t precision, void* packed, size_t packed_bytes, void* stream);
/* The same for several models (models[i] -> packed[i] at precisions[i]) in as few launches as their jobs allow: a training step
 * re-packs every model after every optimiser step. */
int pr_pack_models(int32_t count, const pr_object_model_t* const* models, const int32_t* precisions, void* const* packed,
                   const size_t* packed_bytes, void* stream);

/* One object instance of a call: its (shared) coarse / fine models and their packed copies. */
typedef struct pr_object_t {
    pr_object_model_t coarse;
    const void* packed_coarse;
    pr_object_model_t fine;          /* used only when pr_call_t.use_fine != 0 */
    const void* packed_fine;
} pr_object_t;

/* Optional explicit noise of one model type, NULL = not perturbed (required with PR_FLAG_PERTURB). */
typedef struct pr_noise_t {
    const float* jitter[PR_MAX_OBJECTS];      /* coarse only: U[0,1) (N,R,P_k)  ray_helper.py:1275 */
    const float* alpha[PR_MAX_OBJECTS];       /* coarse only: N(0,1) (N,R,P_k)  object_composer.py:553 */
    const float* pdf[PR_MAX_OBJECTS];         /* coarse only: U[0,1) (N,R,Pf_k) ray_helper.py:1380 */
    const float* integrate[PR_MAX_OBJECTS];   /* N(0,1) (N,R,P_k) object_composer.py:880 -> :751 */
    const float* integrate_global;            /* N(0,1) (N,R,sum P_k), applied AFTER the sort, :886 */
    const float* divergence[PR_MAX_OBJECTS];  /* N(0,1) (N,R,P_k,3): Hutchinson probe of compute_approximate_divergence
                                                 (object_composer.py:582-601); read only with PR_FLAG_SAVE_FOR_BACKWARD
                                                 (the reference returns zeros unless it trains with a graph), objects
                                                 with a ray bender only; NULL = divergence left at zero */
} pr_noise_t;

/* Result fields of ObjectComposer.integrate (model/object_composer.py:774-782); any pointer may be NULL. */
typedef struct pr_entry_t {
    float* integrated_features;               /* (N,R,F) */
    float* opacity;                           /* (N,R) */
    float* weights;                           /* (N,R,P) ; global: (N,R,sum P_k) in sorted order */
    float* depth;                             /* (N,R) */
    float* disparity;                         /* (N,R)  NaN where opacity == 0, as the reference */
    float* integrated_displacements_magnitude;/* (N,R) */
    float* integrated_divergence;             /* (N,R)  zeros (eval) */
} pr_entry_t;

/* Optional emission of the global entry's integrated features in the layout the reference's CNN decoder consumes
 * (model/environment_model_multiresolution_backpropagated_decoder.py:67-106, environment_model_backpropagated_autoencoder.py
 * :128-168): the rays of a call are the concatenation of `groups` row-major grids (the strided grids of a full frame, or
 * the strided patches of a training call, smallest stride first); group i owns the feature channels
 * [channel_begin[i], channel_end[i]) and receives them as a channels-first map.  Replaces split_strided_patch_ray_samples /
 * fold_strided_tensors + split_features_by_layer + permute (two passes over the feature map) by the compositing kernel's
 * own stores. */
#define PR_MAX_DECODER_GROUPS 4
typedef struct pr_decoder_layout_t {
    int32_t groups;                               /* 0 = off */
    int32_t rays[PR_MAX_DECODER_GROUPS];          /* rays of each group; their sum must be R */
    int32_t width[PR_MAX_DECODER_GROUPS];         /* grid width of each group (rays[i] % width[i] == 0) */
    int32_t channel_begin[PR_MAX_DECODER_GROUPS];
    int32_t channel_end[PR_MAX_DECODER_GROUPS];
    float* map[PR_MAX_DECODER_GROUPS];            /* (N, channel_end - channel_begin, rays / width, width) */
} pr_decoder_layout_t;

typedef struct pr_outputs_t {
    pr_entry_t object[PR_MAX_OBJECTS];
    pr_entry_t global;
    /* optional exports of intermediate per-sample state, for stage-level parity tests */
    float* sample_t[PR_MAX_OBJECTS];          /* (N,R,P_k) sample depths */
    float* sample_sigma[PR_MAX_OBJECTS];      /* (N,R,P_k) raw sigma incl. empty_space_alpha fill */
    int32_t* sample_slot[PR_MAX_OBJECTS];     /* (N,R,P_k) compact row of the sample, -1 = outside the box */
    int32_t* evaluated_samples;               /* (K) number of samples sent through the MLP */
    int32_t* normalised_samples;              /* (K) PR_FLAG_TRAIN_BN: samples that entered the batch statistics */
    float* sample_delta[PR_MAX_OBJECTS];      /* (N,R,P_k,3) ray-bender displacement of every sample (zeros outside the
                                                 box / without a bender), input of pr_expected_positions */
    int32_t* head_samples;                    /* (K) number of samples sent through the feature head: evaluated_samples
                                                 unless PR_FLAG_GATE_HEAD skipped the ones with density <= 0 */
    pr_decoder_layout_t decoder;              /* global.integrated_features additionally in decoder layout (groups = 0: off) */
} pr_outputs_t;

typedef struct pr_call_t {
    int32_t frames;                  /* N */
    int32_t rays;                    /* R */
    int32_t objects;                 /* K */
    int32_t static_objects;          /* first `static_objects` instances are static (overlap fix) */
    int32_t use_fine;                /* 1 = hierarchical pass with the fine models (all objects) */
    uint32_t flags;
    int32_t precision;               /* PR_PRECISION_*: must match the precision the objects' weights were packed with */
    int32_t reserved_;
    const float* ray_origins;        /* (N,3) world frame */
    const float* ray_directions;     /* (N,R,3) world frame, not normalised */
    const float* w2o;                /* (N,K,3,4) top three rows of transformation_matrix_w2o */
    const float* style;              /* (N,K,S) */
    const float* deformation;        /* (N,K,D) */
    const uint8_t* object_in_scene;  /* (N,K) */
    const float* linspace_coarse[PR_MAX_OBJECTS]; /* torch.linspace(0,1,P_k) evaluated by the host (P_k) */
    const float* linspace_fine[PR_MAX_OBJECTS];   /* torch.linspace(0,1,Pf_k) */
    int32_t positions_fine[PR_MAX_OBJECTS];       /* Pf_k (resampled positions, use_fine only) */
    pr_noise_t noise_coarse;
    pr_noise_t noise_fine;           /* only .integrate / .integrate_global / .divergence are read */
    uint64_t noise_seed;             /* PR_FLAG_DEVICE_NOISE: seed of the call's generated noise */
    int32_t noise_ray_offset;        /* PR_FLAG_DEVICE_NOISE, calls that are a ray range [offset, offset + R) of a larger */
    int32_t noise_total_rays;        /* render of noise_total_rays rays per frame (0 = this call is the whole render) */
    const uint64_t* noise_seed_device; /* PR_FLAG_DEVICE_NOISE: NULL, or the seed as ONE device word that the kernels read when they run
                                        (noise_seed is then ignored): a call recorded into a HIP graph draws fresh noise on
                                        every replay if the word is rewritten between replays.  pr_render_backward must find the
                                        value its forward call saw. */
} pr_call_t;

/* Workspace bytes pr_render_forward needs for this call (host computation, no device work). */
int pr_workspace_size(const pr_call_t* call, const pr_object_t* objects, size_t* bytes);

/*
 * The renderer: sample placement -> AABB cull + compaction -> fused MLP -> (hierarchical resampling
 * -> fused MLP) -> per-object and cross-object compositing.  `coarse` receives the result of the
 * coarse models, `fine` (may be NULL unless use_fine) the hierarchical pass.
 */
int pr_render_forward(const pr_call_t* call, const pr_object_t* objects,
                      const pr_outputs_t* coarse, const pr_outputs_t* fine,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward pass of pr_render_forward (what torch.autograd does for the reference's op graph when
 * training/trainer_backpropagated_autoencoder.py:349 calls total_loss.backward()).  The forward call must
 * have run with PR_FLAG_SAVE_FOR_BACKWARD (with or without PR_FLAG_TRAIN_BN) on the same `call`, `objects` and
 * `forward_workspace`, which must not have been touched since.  In hierarchical (use_fine) calls the resampled
 * depths are constants (the reference detaches them, ray_helper.py:1340): the fine pass differentiates through the
 * coarse depths it merged in and through the sample positions, the coarse pass through its own results only.
 *
 * Incoming gradients (d loss / d output field, same shapes as pr_entry_t; NULL = zero):
 */
typedef struct pr_entry_grads_t {
    const float* integrated_features;                 /* (N,R,F) */
    const float* opacity;                             /* (N,R) */
    const float* depth;                               /* (N,R) */
    const float* integrated_displacements_magnitude;  /* (N,R): flows into the displacements only, the weights are
                                                         detached there (object_composer.py:772) */
    const float* weights;                             /* (N,R,P) per object, (N,R,sum P) in merged order for the global
                                                         entry: consumers of the compositing weights themselves
                                                         (compute_expected_positions, object_composer.py:603-622) */
    const float* integrated_divergence;               /* (N,R); read with PR_FLAG_DIVERGENCE_GRAD only: flows into the Hutchinson
                                                         estimate (ray bender weights, sample positions), the alphas are
                                                         detached there (object_composer.py:768-769) */
} pr_entry_grads_t;

typedef struct pr_output_grads_t {
    pr_entry_grads_t object[PR_MAX_OBJECTS];
    pr_entry_grads_t global;
    /* gradients of the per-sample exports of pr_outputs_t (NULL = zero): the sample depths and the ray bender's
       displacement vectors, which forward_expected_positions averages (object_composer.py:603-722) */
    const float* sample_t[PR_MAX_OBJECTS];            /* (N,R,P) */
    const float* sample_delta[PR_MAX_OBJECTS];        /* (N,R,P,3); objects with a ray bender */
} pr_output_grads_t;

/* Gradient buffers of one object model, shaped like the parameters (nn.Linear layout).  The library ACCUMULATES
 * (+=) into them: the caller zero-initialises; object instances that share a model pass the same pointers.
 * Any pointer may be NULL (gradient not wanted). */
typedef struct pr_linear_grad_t {
    float* weight;
    float* bias;
} pr_linear_grad_t;

typedef struct pr_model_grads_t {
    pr_linear_grad_t backbone[PR_MAX_LAYERS];
    pr_linear_grad_t alpha_head;
    pr_linear_grad_t head0;
    pr_linear_grad_t affine1;
    pr_linear_grad_t head3;
    pr_linear_grad_t affine4;
    pr_linear_grad_t head6;
    pr_linear_grad_t bender[PR_MAX_LAYERS];
    pr_linear_grad_t bender_out;
} pr_model_grads_t;

typedef struct pr_input_grads_t {
    float* w2o;              /* (N,K,3,4) accumulated; or NULL */
    float* style;            /* (N,K,S)  accumulated; or NULL */
    float* deformation;      /* (N,K,D)  accumulated; or NULL */
    pr_model_grads_t model[PR_MAX_OBJECTS];   /* per object instance (coarse models) */
    pr_model_grads_t model_fine[PR_MAX_OBJECTS]; /* fine models (use_fine calls only) */
    /* gradients of the camera rays (NULL = not wanted): what learnable camera parameters are trained through
       (model/layers/camera_parameters_storage.py; the reference's rays are torch tensors with a graph,
       utils/lib_3d/ray_helper.py:15-52, 1203-1227).  Sample positions x = o + d t, the slab-test depths, the skybox input
       and the sample spacings dt |d| all depend on them. */
    float* ray_origins;      /* (N,3)   accumulated; or NULL */
    float* ray_directions;   /* (N,R,3) accumulated; or NULL */
} pr_input_grads_t;

/* Streams: the call is ordered on `stream` like every other entry point - work enqueued on `stream` before the call precedes
 * it, work enqueued after the call follows all of it.  Inside, calls with several objects run the objects on two lanes: `stream`
 * and one internal stream per device (created on the first such call, kept for the life of the process), forked from `stream`
 * after the compositing backward and joined back into it before the call returns (events, no host synchronisation).  Both
 * workspaces must stay untouched until work enqueued on `stream` after the call would run. */
int pr_backward_workspace_size(const pr_call_t* call, const pr_object_t* objects, size_t* bytes);
int pr_render_backward(const pr_call_t* call, const pr_object_t* objects, const pr_output_grads_t* grads_coarse,
                       const pr_output_grads_t* grads_fine /* NULL unless use_fine */, const pr_input_grads_t* out,
                       void* forward_workspace, size_t forward_workspace_bytes,
                       void* backward_workspace, size_t backward_workspace_bytes, void* stream);

/*
 * Camera rays (RayHelper.create_camera_rays + pixel selection + transform_rays, ray_helper.py:15-52,
 * :433-482, :1203-1227): for frame n and ray r with pixel (rows[r], cols[r]),
 *   d_cam = ((col - W/2)/f_n, -(row - H/2)/f_n, -1),  d_world = R_n d_cam,  o_world = t_n.
 * c2w (N,3,4); focals (N) already multiplied by focal_length_multiplier; rows/cols int32 (R), shared
 * by all frames, or (N,R) - one pixel list per frame, as the random samplers of the reference produce
 * (sample_rays_strided_patch :236, sample_rays_weighted :611, sample_rays :730) - when
 * per_frame_pixels != 0.
 */
int pr_camera_rays(int32_t frames, int32_t rays, int32_t height, int32_t width, int32_t per_frame_pixels,
                   const float* c2w, const float* focals, const int32_t* rows, const int32_t* cols,
                   float* ray_origins, float* ray_directions, float* focal_normals, void* stream);

/*
 * Strided patch pixels (RayHelper.sample_rays_strided_patch, utils/lib_3d/ray_helper.py:236-431): per frame one random patch
 * centre drawn from the bounding-box weight image (object k adds weights[k] / area_k over its pixel-aligned box), clamped into
 * the image and aligned to the grid of the largest stride; then for every stride s_i (ascending) a p_i x p_i pixel grid with
 * p_i = patch_size * s_0 / s_i, row-major, strides concatenated: rows / cols (N, sum p_i^2) int32.  boxes (N,4,K) normalised
 * [left, top, right, bottom]; weights (K); u (N) the uniform draws in [0, 1), one per frame.
 */
int pr_patch_pixels(int32_t frames, int32_t objects, int32_t height, int32_t width, int32_t patch_size, int32_t stride_count,
                    const int32_t* strides /* host */, const float* boxes, const float* weights, const float* u,
                    int32_t* rows, int32_t* cols, void* stream);

/*
 * Pose matrices: (rotation (x, y, z Euler angles, radians), translation) -> the 4x4 transform [R t; 0 1] with R = Ry (Rx Rz)
 * (Transformations3D.homogeneous_rotation_translation, utils/lib_3d/transformations_3d.py:69-96) and its inverse
 * [R^T  -R^T t; 0 1] (the reference calls torch.inverse on it: environment_model.py:221, :1078).  rotations, translations
 * (count,3); matrices, inverses (count,4,4).  Used for the cameras (c2w / w2c) and the objects (o2w / w2o) of a call.
 */
int pr_pose_matrices(int32_t count, const float* rotations, const float* translations, float* matrices, float* inverses,
                     void* stream);
/* Its backward pass: g_matrices / g_inverses (count,4,4) or NULL -> g_rotations, g_translations (count,3), written. */
int pr_pose_matrices_backward(int32_t count, const float* rotations, const float* translations, const float* g_matrices,
                              const float* g_inverses, float* g_rotations, float* g_translations, void* stream);

/* Scene set-up of an evaluation call in ONE launch - replaces, for tensors without a graph, the span of
 * EnvironmentModel.forward_from_scene_encoding between the scene encoding and the composer call (reference:
 * model/environment_model.py:1066-1112: Transformations3D.homogeneous_rotation_translation + torch.inverse for cameras and objects,
 * compute_object_bounding_boxes :234-327, compute_object_axes_projection :329-404) and the permutes that bring the reference's
 * (..., 3 | S | D, K) scene tensors into the renderer's layouts.  Arithmetic = pr_pose_matrices + pr_project_points, bit for bit.
 * frames = product of the leading dims incl. observations; renderer frame n = frame * cameras + camera. */
typedef struct {
    int32_t frames, cameras, objects, box_points_per_object, style_features, deformation_features, height, width;
    float focal_multiplier;          /* config["data"]["focal_length_multiplier"] */
    float upsample_factor;           /* boxes (and the rays) use focals * multiplier * upsample_factor */
    int32_t axes_with_upsampled_focals; /* 0: the axes use focals * multiplier (forward_from_scene_encoding), 1: the upsampled ones */
    const float* camera_rotations;   /* (frames, cameras, 3) */
    const float* camera_translations;
    const float* focals;             /* (frames, cameras) */
    const float* object_rotations;   /* (frames, 3, objects): the reference's trailing-object layout */
    const float* object_translations;
    const float* style;              /* (frames, S, objects) */
    const float* deformation;        /* (frames, D, objects) */
    const uint8_t* object_in_scene;  /* (frames, objects) bool */
    const float* box_points;         /* (objects, box_points_per_object, 3) object-frame corner + edge points */
    const float* axes_points;        /* (objects, 4, 3) origin + unit axes */
    float* boxes;                    /* out (frames, cameras, 4, objects) [left, top, right, bottom], clamped to [0, 1] */
    float* projected_points;         /* out (frames, cameras, box_points_per_object, 2, objects), clamped */
    float* axes;                     /* out (frames, cameras, 4, 2, objects), not clamped */
    float* camera34;                 /* out (frames * cameras, 3, 4): the c2w rows pr_camera_rays takes */
    float* render_focals;            /* out (frames * cameras) */
    float* w2o34;                    /* out (frames * cameras, objects, 3, 4): pr_call_t.w2o */
    float* style_nks;                /* out (frames * cameras, objects, S): pr_call_t.style */
    float* deformation_nkd;          /* out (frames * cameras, objects, D): pr_call_t.deformation */
    uint8_t* present;                /* out (frames * cameras, objects): pr_call_t.object_in_scene */
} pr_scene_setup_t;
int pr_scene_setup(const pr_scene_setup_t* setup, void* stream);
/*
 * Backward of pr_scene_setup's renderer inputs, one launch (a TRAINING call through the fused scene set-up): the gradients
 * pr_render_backward leaves in the renderer's layouts - g_w2o34 (frames x cameras, objects, 3, 4), g_style_nks (.., objects,
 * S), g_deformation_nkd (.., objects, D); any may be NULL = zero - are summed over the cameras of a frame and taken to the
 * scene tensors' layouts: g_rotations / g_translations (frames, 3, objects) through the pose matrices' backward (w2o is the
 * rigid inverse of [R t; 0 1], R = Ry (Rx Rz): /root/reference/utils/lib_3d/transformations_3d.py:69-96 and torch.inverse at
 * /root/reference/model/environment_model.py:221), g_style (frames, S, objects), g_deformation (frames, D, objects).  Outputs
 * that are NULL are skipped (rotations and translations come together).  What torch.autograd does for the reference through
 * ~150 small tensor ops (model/environment_model.py:206-232).
 */
int pr_scene_setup_backward(int32_t frames, int32_t cameras, int32_t objects, int32_t style_features, int32_t deformation_features,
                            const float* object_rotations, const float* object_translations, const float* g_w2o34,
                            const float* g_style_nks, const float* g_deformation_nkd, float* g_rotations, float* g_translations,
                            float* g_style, float* g_deformation, void* stream);

/*
 * Projects object-frame points into the cameras of their frame (EnvironmentModel.compute_object_bounding_boxes /
 * compute_object_axes_projection, model/environment_model.py:234-404).  points (K,P,3); o2w (F,K,4,4); w2c (F,C,4,4);
 * focals (F,C) -> projected (F,C,P,2,K): image coordinates normalised to [0,1] ((v + size/2) / size, x right, y down).
 * boxes (F,C,4,K) [left, top, right, bottom] or NULL: with boxes, points behind the camera do not bound the box and both
 * outputs are clamped to [0,1]; without, the projections are left as they are (the axes variant).
 */
int pr_project_points(int32_t frames, int32_t cameras, int32_t objects, int32_t points_per_object, const float* points,
                      const float* o2w, const float* w2c, const float* focals, int32_t height, int32_t width,
                      float* projected, float* boxes, void* stream);

/*
 * ObjectComposer.compute_expected_positions (model/object_composer.py:603-622) for object `object_index` of a call:
 *   expected[n][r] = sum_i w_i (o + d t_i + delta_i) / (sum_i w_i + 1e-8)      (object frame)
 * with o, d the w2o-transformed ray (RayHelper.transform_rays), t / weights (N,R,P) as produced by pr_render_forward
 * (sample_t export, object entry weights) and delta (N,R,P,3) or NULL.  expected: (N,R,3).
 */
int pr_expected_positions(int32_t frames, int32_t rays, int32_t objects, int32_t object_index, int32_t positions,
                          const float* ray_origins, const float* ray_directions, const float* w2o, const float* t,
                          const float* weights, const float* delta, float* expected, void* stream);

/*
 * The values PR_FLAG_DEVICE_NOISE generates for one noise tensor, written out: element i of the tensor with `kind`
 * (0 stratified jitter U[0,1), 1 coarse density noise N(0,1), 2 inverse-CDF positions U[0,1), 3 per-object density noise
 * N(0,1), 4 merged-list density noise N(0,1), 5 Hutchinson probes N(0,1)) of model type `type` (0 coarse / 1 fine) and
 * object `object`.  count elements, fp32.
 */
int pr_noise_fill(uint64_t seed, int32_t kind, int32_t type, int32_t object, int64_t count, float* out, void* stream);

/*
 * Region-of-interest max pooling: the crop the reference's object encoders and pose estimators take from the
 * observations before their small ResNets - torchvision.ops.roi_pool(observations, boxes, input_size) at
 * model/object_encoder_v4.py:121, model/object_encoder_v5.py:121 and model/object_parameters_encoder_v4.py:131
 * (torchvision 0.9.1, a dependency outside the reference tree; the operator's published definition is restated in
 * csrc/roi_pool.hip and oracle/roi_pool_oracle.py).  input (N,C,H,W); boxes (K,5) = [image index, x1, y1, x2, y2] in
 * input pixels times spatial_scale; output (K,C,ph,pw); argmax (K,C,ph,pw) int32 flat h*W + w of the selected element,
 * -1 for an empty bin, or NULL.  The backward call ACCUMULATES grad_output into grad_input (N,C,H,W) at the argmax
 * positions (the caller zero-initialises).
 */
int pr_roi_pool_forward(int32_t images, int32_t channels, int32_t height, int32_t width, const float* input,
                        int32_t rois, const float* boxes, int32_t pooled_height, int32_t pooled_width,
                        float spatial_scale, float* output, int32_t* argmax, void* stream);
int pr_roi_pool_backward(int32_t images, int32_t channels, int32_t height, int32_t width, int32_t rois,
                         const float* boxes, int32_t pooled_height, int32_t pooled_width, const float* grad_output,
                         const int32_t* argmax, float* grad_input, void* stream);

/*
 * Kernel timing for bench.py: while enabled, the launches of the dominant kernels are bracketed by
 * hipEventRecord on the launch stream.  Categories: 0 = fused MLP (k_mlp_mfma / k_mlp_split / k_mlp_head),
 * 1 = forward compositing (k_composite), 2 = backward dX products (k_gemm_nn), 3 = backward dW products
 * (k_gemm_tn + its reduction), 4 = backward compositing (k_composite_bwd); the rest are reserved.
 * pr_profile_collect synchronises the recorded events, returns the summed milliseconds and launch
 * counts per category (host arrays of PR_PROFILE_CATEGORIES) and clears the list.
 */
#define PR_PROFILE_CATEGORIES 8
int pr_profile_enable(int enable);
int pr_profile_collect(double* milliseconds, int32_t* launches);

/*
 * Measures the fp32 matrix-core rate this device sustains: every CU runs 8 waves of dependent
 * v_mfma_f32_32x32x2_f32 chains (2 accumulators per wave, the occupancy of the renderer's MLP kernel)
 * for `iterations` x 8 MFMAs; returns TFLOP/s from HIP events.  SYNCHRONISES the stream (probe only).
 * random_operands != 0 feeds full-range pseudo-random mantissas that change every iteration (the
 * chip sustains a lower matrix rate on such data than on constant operands).
 */
int pr_probe_mfma_f32(int32_t iterations, int32_t random_operands, double* tflops, double* milliseconds, void* stream);
/* Same for the fp16 matrix cores with the split-precision kernel's issue pattern (4 accumulators, 6
 * v_mfma_f32_32x32x16_f16 per step); TFLOP/s counts hardware fp16 FLOPs (3 per emulated fp32 FLOP). */
int pr_probe_mfma_f16(int32_t iterations, int32_t random_operands, double* tflops, double* milliseconds, void* stream);

/*
 * Adam / AdamW update of ONE flat fp32 tensor in place - the arena every trainable parameter of the renderer is a view of
 * (parallel.flatten_parameters).  Replaces, for that tensor, the optimiser step of the reference's trainers
 * (torch.optim.Adam built at /root/reference/training/trainer.py:62-75 and stepped at :207-216): torch's fused multi-tensor
 * kernel deals one block per 65 536 elements (32 blocks for the 2.1 M parameters of the minecraft renderers: 0.10 ms);
 * this launch covers the tensor with one thread per four elements.  Arithmetic and update order of torch.optim.Adam
 * (amsgrad = False): see csrc/optim.hip.  `step` = the step count AFTER this update (>= 1), by value; or `step_device`
 * != NULL: a device float holding the count BEFORE the update, incremented on the stream first (optimisers recorded into a
 * HIP graph).  grad_scale / found_inf: optional device floats of torch.amp.GradScaler (NULL: plain update).
 */
int pr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int32_t decoupled_weight_decay, int32_t maximize, double step, float* step_device,
                 const float* grad_scale, const float* found_inf, void* stream);

/*
 * Node census of a recorded HIP graph: counts[0] = all nodes, [1] = kernel nodes, [2] = memset nodes, [3] = memcpy nodes
 * (child graphs included).  `graph` is a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).  Host-only: no
 * launch, no synchronisation.  Why it exists: on ROCm 7.0.2 with the runtime's AQL packet capture the MEMSET nodes of a
 * replayed graph stop executing once the host has synchronised between replays (torch's multi-block reductions zero their
 * semaphores with one).  This library never records a memset (every zero fill is a kernel), but the automatic evaluation-frame
 * recording of the host side (EnvironmentModel.frame_replay, the drop-in for model/environment_model.py:581-651) also records
 * whatever encoder / decoder modules the caller injected - a recording with memset nodes is only replayed when the runtime
 * switch DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is set, otherwise the call stays eager.
 */
int pr_graph_node_census(void* graph, int32_t* counts);

/* Library / device introspection. */
int pr_abi_version(void);
const char* pr_last_error(void);
int pr_device_info(int32_t* compute_units, int32_t* lds_bytes, char* arch_name, size_t arch_name_len);

#ifdef __cplusplus
}
#endif
#endif /* PLAYRENDER_H */
