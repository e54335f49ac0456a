/*
 * vitvs.h — C ABI of the MI355X-native ViT-VS hot path (libvitvs_hip.so).
 *
 * The reference (begbaj/ViT-VS) has no FFI: the hot path sits behind Python bound methods.
 * Each entry point below names the reference interface it replaces (paths under
 * /root/reference/catkin_ws/ibvs/src unless noted).  The Python binding a maintainer adds is
 * shown in INTEGRATION.md (ctypes), and vit-vs_amd/_lib.py is exactly that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures (streams travel as void*).
 *   - every function returns a status: 0 ok, > 0 a servo status (enum below), < 0 an error;
 *     vitvs_last_error() gives the message.  Nothing throws across the boundary.
 *   - "_dev" functions take DEVICE pointers and enqueue work on the given hipStream_t (void*,
 *     NULL = the null stream) without synchronising; outputs are valid once the stream has reached
 *     the end of the call's work.  The un-suffixed forms take HOST pointers, copy, run and wait.
 *   - inputs are borrowed for the duration of the call; outputs go to caller-allocated buffers;
 *     the handle owns the device weights and workspaces.  One in-flight call per handle; a handle
 *     is bound to the HIP device that was current when it was created: every entry point that takes
 *     a handle runs on that device (and restores the caller's current device before returning), so
 *     handles of several GPUs may be driven from one thread.
 *   - images are RGB uint8, HWC, already resized to img_size x img_size (reference:
 *     vitvs_v2.py:474-475 PIL resize happens before the path; dinov2_extractor.py:177-191);
 *     vitvs_resize_frames_dev does that resize on the device, bit-identically to PIL, and after
 *     vitvs_set_frame_size the path takes camera-resolution frames and resizes while it builds its patch rows.
 *   - depth is the sensor's uint16 millimetre image, 0 = invalid (reference:
 *     realsense_gazebo_plugin/src/RealSensePlugin.cpp:250-262, consumed at vitvs_v2.py:566-586).
 */
#ifndef VITVS_H
#define VITVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITVS_ABI_VERSION 2   /* 2: num_pairs became a per-call argument of the velocity / servo entry points */
#define VITVS_API __attribute__((visibility("default")))

typedef struct vitvs_handle vitvs_handle;

/* operand type of the GEMMs and of attention (accumulation is always fp32, the residual stream, LayerNorm statistics,
 * descriptors and the correspondence fp32, the control law fp64): F32 = parity mode, BF16 = throughput mode, F16 = the
 * other 16-bit mode (BASELINE.json configs[4] names fp16; 11-bit significand, same speed as bf16) */
/* VITVS_F16X2 ("split-f16"): every GEMM / attention operand is kept as an fp16 hi / lo pair (22 significant bits) and every
 * contraction runs as hi.hi + hi.lo + lo.hi on the f16 matrix cores with fp32 accumulation: the reference's fp32 arithmetic
 * (dinov2_extractor.py:245-263: no autocast anywhere) to fp32 rounding, at 16-bit matrix rate instead of the fp32 matrix
 * pipe's 1/16 of it.  The parity mode at servo rate; same results class as VITVS_F32 (bit-exact arg-max tables on the
 * fixtures, tokens to 2e-5). */
enum vitvs_precision { VITVS_F32 = 0, VITVS_BF16 = 1, VITVS_F16 = 2, VITVS_F16X2 = 3 };

/* Servo status (reference error convention):
 *   NO_CORRESPONDENCE  find_correspondences_batch returned (None, None, None)   vitvs_v2.py:155, 500-505
 *   TOO_FEW            < 4 matches: calculate_uv returns all-zero features       vitvs_v2.py:539-541
 *   NO_DEPTH           no depth image: ibvs() returns early                      vitvs_v2.py:616-619 */
enum vitvs_status { VITVS_OK = 0, VITVS_NO_CORRESPONDENCE = 1, VITVS_TOO_FEW = 2, VITVS_NO_DEPTH = 3 };

/* How the servo features are drawn from the mutual-nearest-neighbour candidates
 * (reference: torch.randperm subset, vitvs_v2.py:134-141; see DESIGN.md "Selection"). */
enum vitvs_select {
    VITVS_SELECT_EXPLICIT = 0, /* caller passes the chosen token ids of the desired frame            */
    VITVS_SELECT_ORDER = 1,    /* caller passes a visiting order (permutation of 0..T-1); the first  */
                               /* num_pairs candidates met in that order are used                    */
    VITVS_SELECT_DENSE = 2     /* every candidate, ascending token id (no zero padding)              */
};

typedef struct vitvs_config {
    int32_t abi_version;  /* VITVS_ABI_VERSION */
    /* extractor geometry (reference: ViTExtractor.__init__, dinov2_extractor.py:25-55) */
    int32_t img_size;     /* S */
    int32_t patch;        /* p */
    int32_t stride;       /* patch-embed stride (== patch unless the stride hack is used, :122-144) */
    int32_t dim;          /* D, multiple of 128 */
    int32_t heads;        /* H, D / H must be 64 */
    int32_t blocks;       /* blocks to run = layer + 1 (descriptor = output of blocks[layer], :226-229) */
    int32_t layerscale;   /* 1: DINOv2 layout with ls1/ls2 gammas (dino_patch/block.py:73,85) */
    float mean[3];        /* Normalize constants (dinov2_extractor.py:49-50) */
    float std[3];
    float ln_eps;         /* 1e-6 */
    int32_t precision;    /* vitvs_precision of the ViT GEMMs/attention; the correspondence and the law are fp32/fp64 */
    int32_t binned;       /* 1: 3x3 log-bin descriptors (use_feature_binning, dinov2_extractor.py:265-311).  The velocity calls take
                           * the similarities of the 9D-wide descriptors as a 3x3 stencil over the D-wide Gram of the raw tokens
                           * (same values to fp32 rounding; one T x T fp32 workspace per pair); the extract_* calls return the
                           * concatenated descriptors */
    /* control law (reference: config.yaml:1-17; vitvs_v2.py:278-295) */
    int32_t num_pairs;    /* default feature-pair count of a call that passes num_pairs <= 0 */
    int32_t u_max, v_max; /* camera resolution; also the depth image size */
    double lambda;
    /* capacity */
    int32_t max_pairs;    /* frame pairs per call (B) */
    int32_t max_rows;     /* feature pairs per frame pair the law may use (>= num_pairs; >= T enables DENSE) */
} vitvs_config;

/* --- lifetime --------------------------------------------------------------------------------
 * Replaces ViTExtractor(model_type, stride, model=...) + model.to(device) (dinov2_extractor.py:25-55). */
VITVS_API int vitvs_create(const vitvs_config* cfg, vitvs_handle** out);
VITVS_API void vitvs_destroy(vitvs_handle* h);
VITVS_API const char* vitvs_last_error(const vitvs_handle* h); /* h may be NULL: last creation error */
VITVS_API int vitvs_abi_version(void);

/* --- weights ---------------------------------------------------------------------------------
 * Replaces model.load_state_dict (dinov2_extractor.py:79-82).  `name` is the DINO/timm/DINOv2
 * state-dict key (patch_embed.proj.weight, cls_token, pos_embed — already resampled to the token
 * grid, shape [1+T][D] —, blocks.{i}.norm1.weight, ... , blocks.{i}.ls2.gamma); `data` is fp32 on
 * the host.  vitvs_weights_ready returns 0 when every tensor the forward reads has been set. */
VITVS_API int vitvs_set_tensor(vitvs_handle* h, const char* name, const float* data, int64_t numel);
VITVS_API int vitvs_weights_ready(const vitvs_handle* h);

/* --- the hot path ----------------------------------------------------------------------------
 * compute_velocity(I_cur, I_des, Z, K) -> v_c : Controller.detect_features() + Controller.ibvs()
 * up to the raw (pre-EMA) twist (vitvs_v2.py:464-523, 588-622).
 *   n_pairs          frame pairs in this call (<= max_pairs)
 *   I_cur, I_des     uint8 [n_pairs][S][S][3]; if des_shared != 0, I_des is ONE image used for all
 *                    pairs (rotation compensation, vitvs_v2.py:1151-1189); I_des == NULL: the goal cached by
 *                    vitvs_set_goal[_dev] (below)
 *   Z_mm             uint16 [n_pairs][v_max][u_max], or NULL -> status NO_DEPTH
 *   K                double [n_pairs][4] = fx, fy, cx, cy
 *   select_mode      vitvs_select; `selection` int32: EXPLICIT [n_pairs][num_pairs] token ids with
 *                    n_selected[n_pairs] counts; ORDER [n_pairs][T]; DENSE ignored (NULL)
 *   num_pairs        feature pairs of the control law for THIS call (the reference's Controller.num_pairs, which its
 *                    callers change between calls: 24 in the servo loop, 48 in find_and_set_best_pose,
 *                    vitvs_v2.py:1151-1189); 1 .. cfg.max_rows, or <= 0 for cfg.num_pairs
 *   v_c              double [n_pairs][6] = vx, vy, vz, wx, wy, wz (camera optical frame)
 *   status           int32 [n_pairs] vitvs_status
 * The return value is < 0 on error, else 0 (per-pair statuses are in `status`). */
VITVS_API int vitvs_compute_velocity_dev(vitvs_handle* h, int32_t n_pairs, const uint8_t* I_cur, const uint8_t* I_des,
                               int32_t des_shared, const uint16_t* Z_mm, const double* K, int32_t select_mode,
                               const int32_t* selection, const int32_t* n_selected, int32_t num_pairs, double* v_c,
                               int32_t* status, void* stream);
VITVS_API int vitvs_compute_velocity(vitvs_handle* h, int32_t n_pairs, const uint8_t* I_cur, const uint8_t* I_des,
                           int32_t des_shared, const uint16_t* Z_mm, const double* K, int32_t select_mode,
                           const int32_t* selection, const int32_t* n_selected, int32_t num_pairs, double* v_c,
                           int32_t* status);
/* The host-pointer form is the reference's seam as it is called (numpy arrays in, a numpy twist out: vitvs_v2.py:464-523,
 * 588-632).  Per call the buffers are copied into a block of pinned, device-visible host memory the handle owns (plain memcpy):
 * the frames go on to device memory in one short copy launch each on the update's own stream, the depth image is copied after the
 * forward has been enqueued — only the pixels the law can read, the tokens' patch centres — and read in place by the law's
 * kernel, the twist and the status are written into the pinned block by that kernel and the detail block (vitvs_last_details:
 * everything but `selected` and `L`) follows in one copy launch: one polled wait, no copy-engine command.  Option "reuse_goal_frames" (vitvs_set_option, 0 / 1, default 0): while I_des repeats the
 * previous call's ADDRESS (and frame count and geometry) the goal frames already staged in device memory are forwarded again as
 * they are — for callers that keep the goal image in a buffer they never write to (a servo loop's goal image is fixed:
 * vitvs_v2.py:264); the goal's tokens are still recomputed on every update, like the reference does.
 *
 * vitvs_reselect: the reference draws its features on the HOST between the correspondence and the law
 * (find_correspondences_batch: sort + torch.randperm, vitvs_v2.py:127-141).  After a host-pointer velocity call the tables are
 * in host memory (vitvs_last_details: nn_1, nn_2, sim_1); the caller draws, and this entry point evaluates the law for that
 * selection on what the call left in the handle (arg-max keys on the device, depth image and intrinsics in the pinned block):
 * one launch, no forward, no staging.  selection / n_selected / num_pairs as in vitvs_compute_velocity; error -5 unless the
 * handle's last velocity call was a host-pointer one. */
VITVS_API int vitvs_reselect(vitvs_handle* h, int32_t select_mode, const int32_t* selection, const int32_t* n_selected,
                             int32_t num_pairs, double* v_c, int32_t* status);

/* --- a goal image that does not change between updates ---------------------------------------------
 * The reference recomputes the goal image's tokens on every update (vitvs_v2.py:482-487), and so does the call above
 * whenever it is given I_des (the benchmarked form).  A servo loop keeps one goal image for a whole run: these two entry
 * points forward n_goal goal frames ONCE (uint8 [n_goal][S][S][3]; n_goal = the later calls' n_pairs, or 1 for des_shared
 * calls) and keep their descriptors in the handle; a later vitvs_compute_velocity[_dev] with I_des == NULL then forwards
 * only the current frames.  The cache is dropped by any call that forwards frames of its own choice through the handle
 * (a velocity call WITH I_des, vitvs_forward_tokens_dev, vitvs_extract_*): a NULL I_des without a matching cached goal is
 * error -5.  Results equal the recomputing call's up to the summation order of the GEMMs (the row count differs). */
VITVS_API int vitvs_set_goal_dev(vitvs_handle* h, int32_t n_goal, const uint8_t* I_des, void* stream);
VITVS_API int vitvs_set_goal(vitvs_handle* h, int32_t n_goal, const uint8_t* I_des);

/* --- in front of it: camera frame -> extractor input -----------------------------------------------
 * goal_image.resize((S, S)) / latest_pil_image.resize((S, S)) (vitvs_v2.py:474-475; PIL default filter BICUBIC,
 * antialiased when shrinking).  frames uint8 [n][in_h][in_w][3] -> out uint8 [n][S][S][3], bit-identical to Pillow's
 * 8-bit resample (Resample.c).  The first call with a new (in_h, in_w) builds the coefficient tables on the host and
 * uploads them (one synchronisation); later calls only enqueue one launch. */
VITVS_API int vitvs_resize_frames_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t in_h, int32_t in_w,
                            uint8_t* out, void* stream);
/* The same resize INSIDE the path ("fused into the patch-embed load"): declares that every frame argument
 * handed to this handle from now on (I_cur, I_des, frames of every entry point above and below) is a camera frame
 * uint8 [in_h][in_w][3].  The launch that builds the patch rows then computes each pixel of the img_size x img_size image
 * PIL would have produced (same integer arithmetic as vitvs_resize_frames_dev, bit-identical) from the camera frame: no
 * resized image in memory, no extra launch.  (0, 0) or (img_size, img_size) restores frames at img_size x img_size.
 * Synchronises the device and rebuilds the tables when the geometry changes; cheap when it does not.  -3: the frame is too
 * large for the kernel's 64 KB of intermediate rows (several thousand pixels high) — use vitvs_resize_frames_dev. */
VITVS_API int vitvs_set_frame_size(vitvs_handle* h, int32_t in_h, int32_t in_w);

/* --- the seams inside it (same split as the reference's callables) ------------------------------
 * ViTExtractor.extract_descriptors(batch, layer, 'token', bin) (dinov2_extractor.py:313-337):
 * frames uint8 [n][S][S][3] -> desc fp32 [n][T][D'] (D' = D, or 9D when cfg.binned); raw, un-normalised. */
VITVS_API int vitvs_extract_descriptors_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, float* desc,
                                  void* stream);
/* The same call with facet = 'query' | 'key' | 'value' (bin = False, include_cls = False; dinov2_extractor.py:193-217,
 * 326-334): q / k / v of blocks[layer] for the patch tokens, fp32 [n][T][D] with descriptor index d * H + h.
 * facet: 0 query, 1 key, 2 value.  In bf16 mode the values carry the qkv GEMM's bf16 output rounding. */
VITVS_API int vitvs_extract_facet_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t facet, float* desc,
                            void* stream);
/* The extractor's whole descriptor surface, extract_descriptors(batch, layer, facet, bin, include_cls)
 * (dinov2_extractor.py:313-337): facet 0 query, 1 key, 2 value, 3 token; bin != 0: the 3x3 log-bin of that facet (:265-311),
 * desc fp32 [n][T][9 D]; include_cls != 0: the cls row is kept, desc [n][1 + T][D]; neither: [n][T][D].  bin together with
 * include_cls is refused like the reference's assertion (error -5).  Independent of cfg.binned (which selects what the
 * velocity path correlates).  Raw, un-normalised values, like vitvs_extract_descriptors_dev. */
VITVS_API int vitvs_extract_descriptors_ex_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t facet, int32_t bin,
                                     int32_t include_cls, float* desc, void* stream);
/* ViTExtractor.extract_saliency_maps(batch) (dinov2_extractor.py:339-353; the 'attn' facet, :230-231): the class token's
 * attention over the patch tokens in blocks[layer] — softmax over all 1 + T keys, patch columns kept — averaged over the
 * heads head_idxs (host array; the reference uses [0, 2, 4, 5] and supports dino_vits8 only) and min-max normalised per
 * image (the reference's broadcast of the [B] extremes is only well-formed for a batch of one; every image gets its own):
 * saliency fp32 [n][T] in [0, 1].  The handle must have been created with layer = the block wanted (the reference hooks
 * block 11). */
VITVS_API int vitvs_extract_saliency_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t n_heads,
                               const int32_t* head_idxs, float* saliency, void* stream);
/* Residual stream after block `cfg.blocks - 1`, fp32 [n][1+T][D] (what the forward hook captures,
 * dinov2_extractor.py:198-199), for parity tests. */
VITVS_API int vitvs_forward_tokens_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, float* tokens, void* stream);

/* find_correspondences_batch's similarity + argmax stage (vitvs_v2.py:78-81) on caller descriptors:
 * desc1 (desired), desc2 (current) fp32 [T][Dp], Dp a multiple of 32 -> nn_1, nn_2 int32 [T], sim_1 fp32 [T].
 * Optional S_out fp32 [T][T] receives the full similarity matrix (NULL to skip). */
VITVS_API int vitvs_correspond_dev(vitvs_handle* h, int32_t T, int32_t Dp, const float* desc1, const float* desc2,
                         int32_t* nn_1, int32_t* nn_2, float* sim_1, float* S_out, void* stream);

/* The control law on given nearest-neighbour tables (vitvs_v2.py:105-155 filter/selection, :511-553,
 * :566-586, :613-659): nn_1, nn_2 int32 [T], sim_1 fp32 [T] for ONE pair. */
VITVS_API int vitvs_servo_from_nn_dev(vitvs_handle* h, int32_t T, const int32_t* nn_1, const int32_t* nn_2, const float* sim_1,
                            const uint16_t* Z_mm, const double* K, int32_t select_mode, const int32_t* selection,
                            int32_t n_selected, int32_t num_pairs, double* v_c, int32_t* status, void* stream);

/* --- introspection of the last compute_velocity / servo call (device -> host copies, synchronising).
 * What detect_features() returns besides v_c (vitvs_v2.py:523) and what the parity tests check.
 *   nn_1, nn_2 int32 [n_pairs][T]; sim_1 fp32 [n_pairs][T]
 *   info int32 [n_pairs][8]: n_mutual, n_feature_rows, same_image, n_matched, svd_sweeps, L_rows, 0, 0
 *   selected int32 [n_pairs][max_rows] token ids of the desired frame (-1 = zero-padded row)
 *   s_uv int32 [n_pairs][max_rows][4] = u*, v*, u, v ; feat double [n_pairs][max_rows][4] = Z, x, y, sim
 *        (feat[..][3] over the first n_matched rows is the reference's sim_selected_12, vitvs_v2.py:523, 1167-1174)
 *   L double [n_pairs][7][2*max_rows] column-major: 6 columns of L_e then e.
 * A call's law uses n_feature_rows = info[1] feature pairs (num_pairs, or every candidate for DENSE); rows of
 * `selected` / `s_uv` / `feat` from n_feature_rows on, and rows of `L` from 2 * n_feature_rows on, are returned as
 * -1 / 0 / 0 / 0 whatever an earlier, larger call left in the workspace.
 * Any pointer may be NULL.  After a host-pointer velocity call (vitvs_compute_velocity, vitvs_reselect) everything but
 * `selected` and `L` is served from host memory (the handle's pinned block) without a device call. */
VITVS_API int vitvs_last_details(vitvs_handle* h, int32_t n_pairs, int32_t* nn_1, int32_t* nn_2, float* sim_1, int32_t* info,
                       int32_t* selected, int32_t* s_uv, double* feat, double* L);

/* --- several updates in flight ------------------------------------------------------------------
 * One update at one frame pair is a chain of 86 dependent launches; each pays the device's launch-to-launch floor and its own
 * ramp, so the chain leaves most of the chip idle most of the time.  Updates that do not depend on each other (several
 * cameras / control loops sharing the GPU, or a frame stream run as a pipeline) overlap when they are enqueued through
 * DIFFERENT handles on DIFFERENT streams: one call in flight per handle, any number of handles (vit-vs_amd/pipeline.py is
 * that arrangement; measured: profiles/r03_notes.md section 5).  Per-handle options for it:
 *   "graph_replay" 0 / 1   velocity calls replay a hipGraph captured per argument tuple — up to 32 tuples per handle, least
 *                          recently used evicted — (host cost ~50 us per update
 *                          instead of ~370 us of launch calls, so ONE host thread keeps several streams busy; on a single
 *                          stream plain launches are ~2 % faster, hence the default 0, or the VITVS_GRAPH environment variable
 *                          at creation).  The reference has no counterpart (one torch call chain per update, vitvs_v2.py:464-523).
 *   "in_flight"    n >= 1  a hint: this handle's updates run beside n - 1 others.  From 2 on the one-round GEMM launches
 *                          use 4-wave workgroups (half the LDS: two launches of different queues share a CU), the narrow
 *                          layers two K slices and a grid order that keeps a weight tile in one XCD's L2, and the
 *                          long-sequence attention whole query blocks (no key ranges to merge), the many-row layers
 *                          256 x 256 tiles wherever they divide (fewest operand bytes per FLOP instead of launch balance)
 *                          and at most two K slices.
 *   "reuse_goal_frames" 0 / 1  host-pointer calls: see vitvs_compute_velocity above.
 * Returns 0, or -5 for an unknown name / a value out of range. */
VITVS_API int vitvs_set_option(vitvs_handle* h, const char* name, int64_t value);
/* The handles of such an arrangement run ONE network: `h` (created with the same network, input geometry and precision, no
 * tensors uploaded) borrows the device weights of `src` instead of holding a copy — one set of weights stays resident in
 * the Infinity Cache for all queues (a copy per handle: 4 x 172 MB cycle through its 256 MB).  `src` must own its weights and
 * have all of them (vitvs_weights_ready); uploads go to `src` only (vitvs_set_tensor on `h` is error -5).  Ownership is shared:
 * the device memory is released when the LAST handle holding it is destroyed, so `src` and `h` may be destroyed in any order
 * (a borrower whose lender is gone keeps working; nothing can upload to those weights any more).  A borrower may borrow
 * again, from another owner: the call drains the device and drops the updates captured over the previous weights. */
VITVS_API int vitvs_share_weights(vitvs_handle* h, const vitvs_handle* src);

/* --- measurement hooks (bench.py roofline leg) --------------------------------------------------
 * With timing enabled every kernel of the path is dispatched with a HIP event pair that the dispatch
 * itself stamps with its begin / end times (hipExtLaunchKernelGGL on the launch stream; hipGraph replay
 * is bypassed).  vitvs_timing_collect synchronises and returns, per kernel class, the summed kernel
 * milliseconds and the number of launches since the last collect; class names come from vitvs_timing_class_name(0 .. vitvs_timing_classes()-1). */
VITVS_API int vitvs_timing_enable(vitvs_handle* h, int32_t on);
VITVS_API int vitvs_timing_classes(void);
VITVS_API const char* vitvs_timing_class_name(int32_t cls);
VITVS_API int vitvs_timing_collect(vitvs_handle* h, int32_t n_classes, double* total_ms, int32_t* launches);

/* Token count T and descriptor width D' for this handle. */
VITVS_API int vitvs_tokens(const vitvs_handle* h);
VITVS_API int vitvs_desc_dim(const vitvs_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* VITVS_H */
