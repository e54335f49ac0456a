// Internal launch interface between the C-ABI layer (api.cpp) and the HIP kernels.
// All pointers are device pointers; every launch is asynchronous on `stream`.
#pragma once
#include <atomic>
#include <vector>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace vitvs {

int fail_hip(hipError_t e, const char* what, const char* file, int line);

#define VITVS_HIP_CHECK(expr)                                                          \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return ::vitvs::fail_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

// Kernel timing: when api.hip's Span arms `g_launch_timing`, the NEXT launch goes through
// hipExtLaunchKernelGGL, which stamps the two events with the dispatch's own begin / end times (what
// rocprofv3 --kernel-trace reports), so hipEventElapsedTime(start, stop) is that kernel's duration.
struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchTiming g_launch_timing;

template <typename F, typename... Args>
inline void launch(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (g_launch_timing.start) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, g_launch_timing.start, g_launch_timing.stop, 0,
                              args...);
        g_launch_timing = LaunchTiming{};
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    }
}

// Updates the calling handle expects to run beside its own (vitvs_set_option "in_flight"; set by api.hip around a handle's
// launches, 1 otherwise).  From 2 on the one-round GEMM launches of gemm.hip use 4-wave workgroups (plan_tiles).
extern thread_local int g_updates_in_flight;

// The HIP device the calling thread's current entry point runs on (set by the C-ABI layer's DeviceScope; api.hip).
extern thread_local int g_current_device;
inline int current_device() {
    if (g_current_device < 0) (void)hipGetDevice(&g_current_device);
    return g_current_device;
}
// Kernels that need more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute, and the attribute is kept per
// (function, DEVICE): `done` holds one bit per device for one kernel instantiation.
inline int raise_lds_limit(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    const int dev = current_device();
    if (dev < 0 || dev >= 64) return -1;
    if ((done.load(std::memory_order_relaxed) >> dev) & 1ull) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return 0;
}

// operand type of the GEMMs / attention; fp32 accumulate.  PREC_X2 = split-f16 (common.h hx2): every operand is an fp16
// hi / lo pair, rows are [hi of 32 columns | lo of the same 32] per 64 fp16, three f16 MFMAs per k-step: fp32-class results
enum Precision : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_X2 = 3 };

inline size_t elem_size(Precision p) { return (p == PREC_F32 || p == PREC_X2) ? 4 : 2; }   // bytes per LOGICAL element
inline bool plain16(Precision p) { return p == PREC_BF16 || p == PREC_F16; }

// ---- gemm.hip ------------------------------------------------------------------------------
// out[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]); A, W, out in precision p; gelu: erf GELU.
// wexp (PREC_X2 only, 0 .. 31): W holds the weights times 2^wexp (so that their lo halves are normal fp16 numbers); the
// kernel multiplies the sums by 2^-wexp before the epilogue.
int launch_linear(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                  int gelu, hipStream_t stream, int wexp = 0);
// the same two operators restricted to the tiles of gemm.hip (no hand-over to gemm_big.hip)
int launch_linear_128(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int gelu,
                      int splits, bool partial, hipStream_t stream);
int launch_linear_classic(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                          int gelu, hipStream_t stream, int wexp = 0);
int launch_linear_partial_classic(Precision p, const void* A, const void* W, float* part, int M, int N, int K, int splits,
                                  hipStream_t stream, int wexp = 0);
// x[m][n] += ls[n] * (sum_k A[m][k] W[n][k] + bias[n]); x fp32 residual stream, ls may be null.
int launch_linear_residual(Precision p, const void* A, const void* W, const float* bias, const float* ls, float* x,
                           int M, int N, int K, hipStream_t stream, int wexp = 0);
// Split-K form for the narrow (N = D) layers: part[z][m][n] = sum over K slice z of A[m][k] W[n][k], fp32,
// z < splits (splitk_slices picks the count); finished by launch_residual_ln.
int splitk_slices(Precision p, int M, int N, int K);
int launch_linear_partial(Precision p, const void* A, const void* W, float* part, int M, int N, int K, int splits,
                          hipStream_t stream, int wexp = 0);
// x[img*(T+1) + 1 + t][n] = sum_k Ape[img*T + t][k] Wpe[n][k] + bias[n] + pos[1 + t][n]
int launch_patch_embed(Precision p, const void* Ape, const void* Wpe, const float* bias, const float* pos, float* x,
                       int n_img, int T, int D, int Kp, hipStream_t stream, int wexp = 0);

// ---- gemm_big.hip: 256-row tiles for many-row problems (16-bit operands) --------------------------------
// big_tile_width: 0 = not applicable (use the tiles of gemm.hip), else the tile code to pass on: the column width 256, 192 or 128 of a
// 256-row tile, or 1192 = 192 rows x 128 columns.
int big_tile_width(Precision p, int M, int N, int K, int splits, bool partial);
// (BM, BN, KG) of the kernel launch_linear (partial = false, splits = 1) / launch_linear_partial picks; KG = 0: 256-row tiles
int linear_tile_plan(Precision p, int M, int N, int K, int splits, bool partial, int out[3]);
// partial = false: out[m][n] = act(sum + bias[n]) in precision p; partial = true: out = fp32 part[z][m][n], z < splits.
int launch_linear_big(Precision p, int bn, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                      int splits, int gelu, bool partial, hipStream_t stream, int wexp = 0);

// ---- elementwise.hip -----------------------------------------------------------------------
struct PatchifyArgs {
    const uint8_t* des;   // [n_des][S][S][3] RGB u8
    const uint8_t* cur;   // [n_cur][S][S][3]
    int n_des, n_cur;
    int S, patch, stride, grid, Kp, D;
    float mean[3], std[3];
    const float* cls;     // [D]
    const float* pos;     // [1+T][D]
};
// Camera-resolution frames ([in_h][in_w][3] instead of [S][S][3]): Pillow's 8-bit bicubic resample tables of
// resize_coefficients() (resize.hip) for both axes, on the device.  rows = the most camera rows one patch's pixels draw on.
struct ResizeArgs {
    const int *xb, *xk, *yb, *yk;   // bounds [S][2] (first input sample, count) and fixed-point taps [S][ks] per axis
    int in_h, in_w, ksx, ksy, rows;
};
constexpr int kResizePrecisionBits = 32 - 8 - 2;   // Pillow Resample.c: PRECISION_BITS
__device__ __forceinline__ int resize_clip8(int v) { return min(max(v >> kResizePrecisionBits, 0), 255); }
// Ape[(img*T + t)][k] = ((u8/255) - mean_c)/std_c for k = c*p*p + py*p + px (zero for k >= 3p²);
// x[img*(T+1)][:] = cls + pos[0].  rs != nullptr: des / cur are camera frames and u8 is the pixel PIL's
// Image.resize((S, S)) would produce (vitvs_v2.py:474-475), computed while the row is built.
int launch_patchify(Precision p, const PatchifyArgs& a, const ResizeArgs* rs, void* Ape, float* x, hipStream_t stream);
// out[m][:] = LayerNorm(x[m][:]) * gamma + beta, out in precision p.
int launch_layernorm(Precision p, const float* x, const float* gamma, const float* beta, void* out, int M, int D,
                     float eps, hipStream_t stream);
// x[m][:] += ls * (sum_z part[z][m][:] + bias)  (fixed slice order), then, if gamma != null,
// out[m][:] = LayerNorm(x[m][:]) * gamma + beta in precision p.  One pass over the row.
// desc (optional, only with gamma == null, i.e. after the last block): the same launch also writes the plain
// L2-normalised descriptors dn[img][t][:] = x[img][1+t][:] / max(norm, 1e-8) (rows are [img][1+T] tokens, cls
// first) and clears zero_count 64-bit words of zero_a / zero_b (the Gram kernel's atomicMax targets).
struct DescOut {
    float* dn = nullptr;          // may be null when sq is given
    float* sq = nullptr;          // optional: sq[img * T + t] = |x[img][1 + t][:]|^2 (the stencil form of the binned Gram, correspond.hip)
    unsigned long long* zero_a = nullptr;
    unsigned long long* zero_b = nullptr;
    int T = 0;
    int zero_count = 0;
};
int launch_residual_ln(Precision p, float* x, const float* part, int splits, const float* bias, const float* ls,
                       const float* gamma, const float* beta, void* out, int M, int D, float eps, hipStream_t stream,
                       const DescOut* desc = nullptr);
// Finishes a split-K patch embedding (partial rows [n_img][T], launch_linear_partial) and applies block 0's norm1:
//   x[img][0][:] = cls + pos[0];  x[img][1+t][:] = pos[1+t] + sum_z part[z][img*T+t][:] + bias;  out = LayerNorm(x).
int launch_embed_ln(Precision p, float* x, const float* part, int splits, const float* bias, const float* pos,
                    const float* cls, const float* gamma, const float* beta, void* out, int n_img, int T, int D, float eps,
                    hipStream_t stream);
// Descriptors for the correspondence stage, fp32, L2-normalised with max(|x|,1e-8):
//   plain : dn[img][t][D]   = x[img][1+t][:] / max(norm, eps)
//   binned: dn[img][t][9D]  = 3x3 replicate-clamped neighbourhood concat, then normalised.
// raw (optional, may be null): the un-normalised descriptor in the same layout.
// zero_a / zero_b (may be null with zero_count 0): zero_count 64-bit words of each are cleared by the
// same launch (the Gram kernel's atomicMax targets), saving two memset nodes per update.
int launch_descriptors(const float* x, float* dn, float* raw, float* sqnorm_ws, int n_img, int T, int grid, int D,
                       int binned, unsigned long long* zero_a, unsigned long long* zero_b, int zero_count,
                       hipStream_t stream);

// dst[0 .. bytes) = src[0 .. bytes), 16 bytes per lane (both buffers hold a multiple of 16 bytes): the host-pointer entry point's
// in-stream copy of caller frames from the handle's pinned staging memory (device-visible host memory) to device memory —
// one short launch on the update's own stream instead of a copy-engine command and its queue hops.
int launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t stream);

// Experiment hook (tools/l2_warm_probe.py): every XCD's private L2 reads all `bytes` of `p` (256 workgroups,
// workgroup L on XCD L % 8 walks the 32nd share L / 8 of the buffer with 16-byte loads), so that a launch that follows finds its
// operand in L2 instead of the Infinity Cache.  share_xcds != 0: XCD x touches only the x-th eighth.
int launch_touch(const void* p, size_t bytes, int share_xcds, hipStream_t stream);

// dst[r][:] = src[r][:] / max(||src[r]||, 1e-8) for fp32 rows of width Dp (caller descriptors).
int launch_normalize_rows(const float* src, float* dst, int rows, int Dp, hipStream_t stream);

// ---- attention.hip -------------------------------------------------------------------------
// out[img*N + q][h*64 + d] = softmax_k(q.k * 64^-0.5) v ; qkv [n_img*N][3*D] in precision p.
// Long sequences (>= 512 tokens) may cut the keys of a query block into ranges merged inside the launch; that needs a
// workspace: `state` (attention_workspace_floats floats) and `tickets` (attention_ticket_count ints, ZERO before the first
// launch; every launch leaves them zero).  ws = nullptr: a per-device workspace shared by all callers without one (the
// operator hook), grown on demand.
struct AttnWorkspace {
    float* state = nullptr;
    int* tickets = nullptr;
};
size_t attention_workspace_floats(int n_img, int N, int H);
size_t attention_ticket_count(int n_img, int N, int H);
// q_prescaled (16-bit precisions only): the q third of qkv already carries hd^-0.5 * log2(e) (kAttnQScale) — the handle folds it
// into the q rows of attn.qkv.weight / bias at upload, in fp32 before the one rounding to 16 bits, so the scale costs the
// forward neither an instruction nor a rounding; raw q (the operator hook) is scaled inside the kernels.
constexpr float kAttnQScale = 0.125f * 1.44269504088896340736f;
int launch_attention(Precision p, const void* qkv, void* out, int n_img, int N, int H, hipStream_t stream,
                     const AttnWorkspace* ws = nullptr, bool q_prescaled = false);

// ---- correspond.hip ------------------------------------------------------------------------
// For pair b: S = dn[a_img(b)] . dn[b_img(b)]^T (T x T, fp32); row_best[b][i] / col_best[b][j] receive
// the packed (max similarity, first index) keys (common.h pack_best).  Buffers must be zeroed first.
int launch_gram_argmax(const float* dn, int T, int Dp, int n_pairs, int des_shared,
                       unsigned long long* row_best, unsigned long long* col_best, hipStream_t stream);
// Optional dense similarity matrix (tests / debugging): S[b][i][j].
// 16-bit modes from 1024 tokens on: hi / lo fp16 split of the descriptors into `dh` (gram_split_elems fp16 elements), then the
// Gram on the f16 matrix cores (correspond.hip)
size_t gram_split_elems(int n_frames, int T, int Dp);
int launch_split_desc(const float* dn, void* dh, int T, int Dp, int n_pairs, int des_shared, hipStream_t stream);
int launch_gram_argmax_split(const void* dh, int T, int Dp, int n_pairs, int des_shared, unsigned long long* row_best,
                             unsigned long long* col_best, hipStream_t stream);
int launch_gram_dense(const float* dn, int T, int Dp, int n_pairs, int des_shared, float* S, hipStream_t stream);
// Binned descriptors without building them (correspond.hip header): G[b][i][j] = raw dot products of the patch tokens in the
// residual stream x ([frames][1 + T][D] fp32, desired frames first), then the 3 x 3 "diagonal" stencil over G, normalised by the
// binned descriptors' norms (sq = |t|^2 per token: DescOut::sq of the forward's last launch), with the fused arg-max into row_best / col_best.
int launch_gram_raw_tokens(const float* x, int T, int D, int n_pairs, int des_shared, float* G, hipStream_t stream);
int launch_gram_stencil_argmax(const float* G, const float* sq, int T, int grid, int n_pairs, int des_shared,
                               unsigned long long* row_best, unsigned long long* col_best, hipStream_t stream);

// packed keys <-> (nn_1, sim_1, nn_2) tables for the standalone correspondence / servo entry points
int launch_decode_best(const unsigned long long* row_best, const unsigned long long* col_best, int T, int32_t* nn1,
                       int32_t* nn2, float* sim1, hipStream_t stream);
int launch_encode_best(const int32_t* nn1, const int32_t* nn2, const float* sim1, int T, unsigned long long* row_best,
                       unsigned long long* col_best, hipStream_t stream);

// ---- servo.hip -----------------------------------------------------------------------------
enum SelectMode : int { SEL_EXPLICIT = 0, SEL_PRIORITY = 1, SEL_DENSE = 2 };
enum Status : int { ST_OK = 0, ST_NO_CORRESPONDENCE = 1, ST_TOO_FEW = 2, ST_NO_DEPTH = 3 };

struct ServoArgs {
    int n_pairs, T, grid;
    int num_pairs;            // K rows pairs of the control law (reference num_pairs)
    int mode;                 // SelectMode
    int input_size;           // S
    int u_max, v_max;
    int depth_h, depth_w;
    float scale_f, half_f;    // fp32 patch-centre arithmetic (reference: vitvs_v2.py:511-513)
    double scale_x, scale_y;  // camera / ViT resolution ratios (vitvs_v2.py:544-545)
    const double* K;          // device [n_pairs][4] fx, fy, cx, cy
    double lambda;
    const unsigned long long* row_best;  // [n_pairs][T]
    const unsigned long long* col_best;  // [n_pairs][T]
    const uint16_t* depth;    // [n_pairs][depth_h][depth_w] mm, may be null -> ST_NO_DEPTH
    const int32_t* selection; // EXPLICIT: [n_pairs][sel_stride] token ids; PRIORITY: [n_pairs][T] priorities
    const int32_t* n_selected;// EXPLICIT: [n_pairs] count of ids (<= num_pairs)
    int sel_stride;
    // outputs
    double* v_c;              // [n_pairs][6]
    int32_t* status;          // [n_pairs]
    int32_t* nn1;             // [n_pairs][T]
    int32_t* nn2;             // [n_pairs][T]
    float* sim1;              // [n_pairs][T]
    int32_t* info;            // [n_pairs][8]: n_mutual, n_rows_pairs, same_image, n_matched, sweeps, ...
    int32_t* sel_out;         // [n_pairs][max_rows] selected token ids (image 1)
    int32_t* s_uv;            // [n_pairs][max_rows][4]: u*, v*, u, v
    double* feat;             // [n_pairs][max_rows][4]: Z, x, y, sim
    double* L_ws;             // [n_pairs][7][rows_cap] workspace (L columns + e), also an output for tests
    int max_rows;             // capacity in feature pairs (>= num_pairs; >= T for DENSE)
};
int launch_servo(const ServoArgs& a, hipStream_t stream);

// out[n_img][T][D] fp32, index d*H + h <- which-th (0 q, 1 k, 2 v) third of qkv[n_img*(1+T)][3][H][64], cls dropped
// q_unscale: factor that undoes a pre-scaled q third (1 / kAttnQScale for which == 0 in the 16-bit modes, else 1)
// keep_cls: 0 -> out [n_img][T][D]; 1 -> out [n_img][1 + T][D] (the cls row first)
int launch_facet(Precision p, const void* qkv, float* out, int n_img, int T, int H, int which, float q_unscale, int keep_cls,
                 hipStream_t stream);
// Saliency maps of the extractor (dinov2_extractor.py:339-353): out fp32 [n_img][T], class-token attention of the chosen heads
// of the block whose qkv is given, averaged and min-max normalised per image.
int launch_saliency(Precision p, const void* qkv, float* out, int n_img, int T, int H, const int* head_idx, int n_heads,
                    bool q_prescaled, hipStream_t stream);
// Pillow-exact bicubic resize of n RGB uint8 frames [in_h][in_w][3] -> [out][out][3] (resize.hip).  The tables come from
// resize_coefficients (host, double precision, Pillow's expressions): bounds [out][2] = (first tap, taps), coefficients
// [out][ksize] in 22-bit fixed point; x tables for the width, y tables for the height.
int resize_coefficients(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& coeffs);
int launch_resize_bicubic(const uint8_t* src, uint8_t* dst, int n, int in_h, int in_w, int out, const int* xb, const int* xk,
                          int ksx, const int* yb, const int* yk, int ksy, hipStream_t stream);

}  // namespace vitvs
