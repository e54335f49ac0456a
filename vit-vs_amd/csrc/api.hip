// C ABI of libvitvs_hip.so (include/vitvs.h, include/vitvs_ops.h): handle, weights, the
// compute_velocity path and its seams.  Host-side orchestration only; all arithmetic is in the
// kernels (gemm.hip, attention.hip, elementwise.hip, correspond.hip, servo.hip).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vitvs.h"
#include "../../include/vitvs_ops.h"
#include "kernels.h"

namespace vitvs {

static thread_local std::string g_last_error;
thread_local LaunchTiming g_launch_timing;
thread_local int g_current_device = -1;
thread_local int g_updates_in_flight = 1;
static thread_local int g_op_wexp = 0;   // vitvs_op_weight_exponent: the 2^e the f16x2 weights of the operator hooks carry

int fail_hip(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    g_last_error = buf;
    return -100 - (int)e;
}

static inline Precision to_prec(int32_t p) {
    return p == VITVS_F32 ? PREC_F32 : (p == VITVS_F16 ? PREC_F16 : (p == VITVS_F16X2 ? PREC_X2 : PREC_BF16));
}

static inline uint16_t f32_to_f16_host(float f) {   // round to nearest even, as the device's v_cvt_f16_f32
    const _Float16 hval = (_Float16)f;
    uint16_t bits;
    memcpy(&bits, &hval, 2);
    return bits;
}

static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Block {
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
    float *qkvb = nullptr, *projb = nullptr, *fc1b = nullptr, *fc2b = nullptr, *ls1 = nullptr, *ls2 = nullptr;
    void *qkvw = nullptr, *projw = nullptr, *fc1w = nullptr, *fc2w = nullptr;
    int qkve = 0, proje = 0, fc1e = 0, fc2e = 0;   // f16x2: each matrix is stored times 2^e (upload_matrix)
};

// Device memory of one set of weights.  Held through a shared_ptr by the handle that uploaded it AND by every handle that
// borrows it (vitvs_share_weights): the memory is freed when the LAST holder is destroyed, in whatever order the handles go,
// so a borrower's kernels and captured graphs never read freed weights.
struct WeightStore {
    int device = 0;
    std::vector<void*> allocs;
    ~WeightStore() {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != device) (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (prev != device && prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace vitvs

using namespace vitvs;

#include <stddef.h>
static_assert(sizeof(vitvs_config) == 96 && offsetof(vitvs_config, lambda) == 80, "vitvs_config layout is part of the ABI");

struct vitvs_handle {
    vitvs_config cfg;
    Precision prec;
    int device = 0;
    int grid = 0, T = 0, N = 0, Kp = 0, Dp = 0, hidden = 0, n_img_max = 0;
    std::string err;
    std::vector<void*> allocs;                      // workspaces, outputs, staging: this handle's own
    std::shared_ptr<WeightStore> wstore;            // the weights: shared with the handles that borrow them
    std::map<std::string, bool> have;
    int desc_keys = -1;   // >= 0 while a velocity update runs: the forward's last launch emits the descriptors and clears this many keys
    bool ready = false;       // cached result of vitvs_weights_ready (reset by vitvs_set_tensor)
    // Pillow-exact resize tables: `rs` of the last camera frame size seen by vitvs_resize_frames_dev, `fr` of the frame
    // geometry declared with vitvs_set_frame_size (fr.in_h == 0: frames arrive at img_size x img_size)
    ResizeArgs rs{}, fr{};
    size_t staged_frame_bytes = 0;   // capacity per frame of st_cur / st_des
    // weights
    std::vector<Block> blk;
    void* pe_w = nullptr;
    int pe_e = 0;
    float *pe_b = nullptr, *cls = nullptr, *pos = nullptr;
    // activations
    void *Ape = nullptr, *xn = nullptr, *qkv = nullptr, *attn = nullptr, *hid = nullptr;
    float *x = nullptr, *dn = nullptr, *sq = nullptr, *part = nullptr;  // part: split-K partial sums [8][M][D]
    unsigned short* dh = nullptr;   // fp16 hi / lo split of dn for the many-token Gram of the 16-bit modes (null: fp32 Gram)
    float* gram_ws = nullptr;       // binned descriptors: raw token Gram [max_pairs][T][T] for the stencil form (correspond.hip; null: the 9 D-wide Gram)
    AttnWorkspace attn_ws;    // key-split states / tickets of the long-sequence attention, sized for every image count <= n_img_max
    size_t dn_elems = 0;
    unsigned long long *row_best = nullptr, *col_best = nullptr;
    size_t best_elems = 0;
    // servo outputs / details
    int32_t *nn1 = nullptr, *nn2 = nullptr, *info = nullptr, *sel_out = nullptr, *s_uv = nullptr;
    float* sim1 = nullptr;
    double *feat = nullptr, *Lws = nullptr;
    int last_pairs = 0, last_T = 0;
    // device copies of the frames a host-pointer call hands over (filled from the pinned block, HostStage below), and the
    // graph replays' own copy of the selection
    uint8_t *st_cur = nullptr, *st_des = nullptr;
    int32_t *st_sel = nullptr, *st_nsel = nullptr;
    // per-kernel-class timing (HIP events on the launch stream), see vitvs_timing_*
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;      // class of event pair i (events 2i, 2i+1)
    size_t ev_used = 0;
    // captured hipGraphs of compute_velocity_dev (opt-in, VITVS_GRAPH=1), keyed on the argument tuple
    struct GraphEntry {
        std::vector<uintptr_t> key;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t last_use = 0;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    bool use_graphs = false;
    bool borrowed = false;    // weights belong to another handle (vitvs_share_weights): never uploaded to, never freed here
    int in_flight = 1;        // vitvs_set_option "in_flight": updates expected to run beside this handle's (tile plan hint)
    int goal_frames = 0;      // goal frames whose tokens / descriptors are cached in rows [0, goal_frames) (vitvs_set_goal_dev)
    // Host-pointer entry points (vitvs_compute_velocity, vitvs_set_goal): ONE block of pinned, device-visible host memory,
    // allocated on first use.  Caller buffers are copied into it with memcpy; the frames then reach device memory through a
    // short copy launch on the update's own stream (launch_copy16), the intrinsics / visiting order / depth image are read by
    // the law's kernel in place (it touches <= max_rows depth pixels), and v_c / status / the feature rows are written back
    // into it by the device — no copy-engine command, no pageable transfer, one wait at the end.
    struct HostStage {
        unsigned char* base = nullptr;
        size_t frame_cap = 0;     // bytes per staged frame
        uint8_t *cur = nullptr, *des = nullptr;
        uint16_t* depth = nullptr;
        double *K = nullptr, *vc = nullptr;
        int32_t *sel = nullptr, *nsel = nullptr, *status = nullptr;
        unsigned char* det = nullptr;   // image of the device's detail block (info | s_uv | feat) of the last host-pointer call
    } hs;
    hipStream_t host_stream = nullptr;
    bool details_pinned = false;        // hs.det holds the last call's detail block (vitvs_last_details serves it from there)
    struct HostTables { int n_pairs = 0, T = 0; bool have_depth = false; } host_tables;   // what vitvs_reselect may build on
    unsigned char* det_block = nullptr; // device copy of the detail block (detail_pointers), one allocation
    size_t det_bytes = 0;
    std::vector<int32_t> depth_sites;   // linear pixel index of every token's patch centre that lies inside the depth image (the only
                                        // pixels the law reads: servo.hip token_pixel, restated on the host at creation)
    bool reuse_goal = false;            // option "reuse_goal_frames": I_des of a host-pointer call is not staged again while its address repeats
    const void* staged_des = nullptr;   // host address, frame count and geometry of the goal frames in st_des
    size_t staged_des_bytes = 0;
};

namespace {

// The detail block — what vitvs_last_details hands out except `selected` and L — is one allocation laid out
// info [P][8] i32 | s_uv [P][R][4] i32 | feat [P][R][4] f64 | nn_1 [P][T] i32 | nn_2 [P][T] i32 | sim_1 [P][T] f32, so that a host-pointer
// call can hand the reference's return values (s_uv*, s_uv, the selected similarities, and the tables its host-side draw needs)
// to the host with ONE copy launch into the handle's pinned block (same layout) behind the law's kernel.
void detail_pointers(vitvs_handle* h, unsigned char* base) {
    const size_t P = h->cfg.max_pairs, R = h->cfg.max_rows;
    h->info = reinterpret_cast<int32_t*>(base);
    h->s_uv = reinterpret_cast<int32_t*>(base + P * 32);
    h->feat = reinterpret_cast<double*>(base + P * 32 + P * R * 16);
    h->nn1 = reinterpret_cast<int32_t*>(base + P * 32 + P * R * 48);
    h->nn2 = h->nn1 + h->best_elems;
    h->sim1 = reinterpret_cast<float*>(h->nn2 + h->best_elems);
}

int set_err(vitvs_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_last_error = msg;
    return code;
}

// Every entry point that takes a handle runs on the handle's device, whatever device the calling thread had current
// (include/vitvs.h: "a handle is bound to the HIP device that was current when it was created"); the caller's device is
// restored on return.  The pointer-only operator hooks (vitvs_op_*) run on the caller's current device.
struct DeviceScope {
    int prev = -1, prev_hint = 1;
    bool switched = false;
    explicit DeviceScope(const vitvs_handle* h) {
        prev_hint = g_updates_in_flight;
        (void)hipGetDevice(&prev);
        const int want = h ? h->device : prev;
        if (want != prev) switched = hipSetDevice(want) == hipSuccess;
        g_current_device = switched ? want : prev;
        if (h) g_updates_in_flight = h->in_flight;    // the tile plan of this handle's launches (kernels.h); the pointer-only
                                                      // operator hooks keep the thread's own hint (vitvs_op_plan_in_flight)
    }
    ~DeviceScope() {
        if (switched) { (void)hipSetDevice(prev); g_current_device = prev; }
        g_updates_in_flight = prev_hint;
    }
};

template <typename T>
int dev_alloc_into(std::vector<void*>& owner, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + 256);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc", __FILE__, __LINE__);
    owner.push_back(p);
    (void)hipMemset(p, 0, count * sizeof(T) + 256);   // detail rows a call does not write read as zeros, not as stale memory
    *out = reinterpret_cast<T*>(p);
    return 0;
}
template <typename T>
int dev_alloc(vitvs_handle* h, T** out, size_t count) { return dev_alloc_into(h->allocs, out, count); }

// frees one of the handle's own blocks (not a weight) and forgets it
void dev_free(vitvs_handle* h, void* p) {
    if (!p) return;
    h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), p), h->allocs.end());
    (void)hipFree(p);
}

int upload_f32(vitvs_handle* h, float** dst, const float* src, size_t n) {
    if (!*dst) {
        int rc = dev_alloc_into(h->wstore->allocs, dst, n);
        if (rc) return rc;
    }
    VITVS_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// matrix [rows][cols] fp32 host -> device in the handle's precision, row stride `ld` (zero padded).
// f16x2 (PREC_X2): every element becomes an fp16 pair hi = fp16(w 2^e), lo = fp16(w 2^e - hi), rows laid out as [hi of 32
// columns | lo of the same 32] per 64 fp16 (csrc/common.h); e is the power of two that puts the matrix's largest magnitude in
// [2^12, 2^13), so the lo halves of all but negligible weights are NORMAL fp16 numbers (22 significant bits) and nothing
// overflows; the GEMM multiplies its sums by 2^-e (*wexp, 0 .. 31).
int upload_matrix(vitvs_handle* h, void** dst, const float* src, size_t rows, size_t cols, size_t ld, int* wexp = nullptr) {
    const size_t es = elem_size(h->prec);
    if (!*dst) {
        unsigned char* p = nullptr;
        int rc = dev_alloc_into(h->wstore->allocs, &p, rows * ld * es);
        if (rc) return rc;
        *dst = p;
    }
    std::vector<unsigned char> tmp(rows * ld * es, 0);
    if (h->prec == PREC_X2) {
        if (ld % 32 != 0) return set_err(h, -6, "f16x2 rows are multiples of 32 columns");
        float amax = 0.f;
        for (size_t i = 0; i < rows * cols; ++i) amax = std::max(amax, fabsf(src[i]));
        int e = 0;
        if (amax > 0.f && std::isfinite(amax)) {
            int ex = 0;
            (void)frexpf(amax, &ex);                       // amax = f 2^ex, f in [0.5, 1)
            e = std::min(31, std::max(0, 13 - ex));        // amax 2^e in [2^12, 2^13)
        }
        if (wexp) *wexp = e;
        const float sc = ldexpf(1.0f, e);
        uint16_t* d = reinterpret_cast<uint16_t*>(tmp.data());
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const float v = src[r * cols + c] * sc;    // exact (power of two)
                const float vc = std::min(65504.0f, std::max(-65504.0f, v));
                const _Float16 hi = (_Float16)vc;
                const size_t at = r * 2 * ld + ((c >> 5) << 6) + (c & 31);
                d[at] = f32_to_f16_host((float)hi);
                d[at + 32] = f32_to_f16_host(v - (float)hi);
            }
        VITVS_HIP_CHECK(hipMemcpy(*dst, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
        return 0;
    }
    for (size_t r = 0; r < rows; ++r) {
        if (h->prec == PREC_F32) {
            memcpy(tmp.data() + r * ld * 4, src + r * cols, cols * 4);
        } else {
            uint16_t* d = reinterpret_cast<uint16_t*>(tmp.data()) + r * ld;
            if (h->prec == PREC_F16) for (size_t c = 0; c < cols; ++c) d[c] = f32_to_f16_host(src[r * cols + c]);
            else for (size_t c = 0; c < cols; ++c) d[c] = f32_to_bf16_host(src[r * cols + c]);
        }
    }
    VITVS_HIP_CHECK(hipMemcpy(*dst, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    return 0;
}

int check_cfg(const vitvs_config* c, std::string& why) {
    if (c->abi_version != VITVS_ABI_VERSION) { why = "abi_version mismatch"; return -1; }
    if (c->dim <= 0 || c->dim % 128 != 0) { why = "dim must be a positive multiple of 128"; return -1; }
    if (c->heads <= 0 || c->dim != c->heads * 64) { why = "dim / heads must be 64"; return -1; }
    if (c->patch <= 0 || c->stride <= 0 || c->img_size < c->patch) { why = "bad patch/stride/img_size"; return -1; }
    if ((c->img_size - c->patch) % c->stride != 0) { why = "img_size - patch must be a multiple of stride"; return -1; }
    if (c->blocks <= 0) { why = "blocks must be >= 1"; return -1; }
    if (c->precision != VITVS_F32 && c->precision != VITVS_BF16 && c->precision != VITVS_F16 && c->precision != VITVS_F16X2) { why = "unknown precision"; return -1; }
    if (c->max_pairs <= 0 || c->num_pairs <= 0 || c->max_rows < c->num_pairs) { why = "bad capacity"; return -1; }
    if (c->u_max <= 0 || c->v_max <= 0) { why = "bad camera resolution"; return -1; }
    if (c->dim != 128 && c->dim != 256 && c->dim != 384 && c->dim != 768 && c->dim != 1024) {
        why = "dim must be one of 128, 256, 384, 768, 1024";
        return -1;
    }
    return 0;
}

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Captured updates hold addresses (weights, resize tables) and a tile plan: whoever changes one of those drops them, after
// the device has drained (a replay may still be running).
void drop_graphs(vitvs_handle* h) {
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
}

enum KernelClass : int {
    KC_PATCHIFY = 0, KC_PATCH_EMBED, KC_LAYERNORM, KC_QKV, KC_ATTENTION, KC_PROJ, KC_FC1, KC_FC2, KC_DESCRIPTORS,
    KC_GRAM, KC_SERVO, KC_RESIDUAL_LN, KC_GRAM_STENCIL, KC_COUNT
};
const char* const kClassNames[KC_COUNT] = {"patchify", "patch_embed", "layernorm", "qkv", "attention", "proj",
                                           "fc1", "fc2", "descriptors", "gram_argmax", "servo",
                                           "residual_ln", "gram_stencil"};

// When timing is enabled, arms the launch helper (kernels.h) so that the next kernel launched inside the span
// is dispatched with an event pair stamped with its own begin / end times.
struct Span {
    vitvs_handle* h;
    bool armed = false;
    Span(vitvs_handle* h_, int cls, hipStream_t) : h(h_) {
        if (!h->timing) return;
        if (h->ev_used + 2 > h->ev_pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            h->ev_pool.push_back(a);
            h->ev_pool.push_back(b);
        }
        g_launch_timing.start = h->ev_pool[h->ev_used];
        g_launch_timing.stop = h->ev_pool[h->ev_used + 1];
        h->ev_used += 2;
        h->ev_class.push_back(cls);
        armed = true;
    }
    ~Span() {
        if (armed && g_launch_timing.start) {   // nothing was launched inside the span: drop the pair
            g_launch_timing = LaunchTiming{};
            h->ev_used -= 2;
            h->ev_class.pop_back();
        }
    }
};

// Bytes of one frame as the caller hands it over: img_size x img_size x 3, or the declared camera geometry.
static size_t frame_bytes(const vitvs_handle* h) {
    return h->fr.in_h ? (size_t)h->fr.in_h * h->fr.in_w * 3 : (size_t)h->cfg.img_size * h->cfg.img_size * 3;
}

// One launch chain's view of the workspaces (a contiguous range of images).
// Plain descriptors of the default forward are produced by the forward's own last launch.
// (binned descriptors in their stencil form need only the tokens' squared norms: the same launch writes those instead)
static bool desc_in_forward(const vitvs_handle* h) { return h->cfg.binned ? h->gram_ws != nullptr : h->Dp == h->cfg.dim; }

struct ChainCtx {
    int cnt = 0, M = 0;
    float* x = nullptr;
    unsigned char *xn = nullptr, *qkv = nullptr, *attn = nullptr, *hid = nullptr, *Ape = nullptr;
    float* part = nullptr;
    PatchifyArgs pa;
    const ResizeArgs* rs = nullptr;   // camera-resolution frames (vitvs_set_frame_size): resize inside the patch-row build
    bool want_desc = false;   // the last residual_ln also writes the plain descriptors (launch_residual_ln)
    DescOut desc;
};

// The forward, operator by operator, on one stream.  (The two FRAMES of one update as two chains — on two streams, or in
// lockstep on one stream with hipExtAnyOrderLaunch — were measured in round 1 and gained nothing: half-size launches cost
// nearly what full-size ones do, gfx9 ignores the any-order flag, and the kernel-trace timelines that showed the queues
// taking turns were the profiler's own serialisation (profiles/r03_notes.md section 5).  What does overlap is whole,
// independent UPDATES on separate handles and queues: include/vitvs.h "several updates in flight", vit-vs_amd/pipeline.py.)
int forward_lockstep(vitvs_handle* h, ChainCtx* cx, int n, hipStream_t st) {
    const vitvs_config& c = h->cfg;
    const int D = c.dim;
    int rc = 0;
    for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_PATCHIFY, st);
        rc = launch_patchify(h->prec, cx[k].pa, cx[k].rs, cx[k].Ape, cx[k].x, st); }
    // Patch embedding as a split-K GEMM (more workgroups than its 84 output tiles), finished together with
    // cls / pos_embed and block 0's norm1 by one residual_ln-style launch.
    // Block i: qkv -> attention -> proj (split-K partials) -> [residual + norm2] -> fc1+GELU ->
    // fc2 (split-K partials) -> [residual + norm1 of block i+1].
    for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_PATCH_EMBED, st);
        rc = launch_linear_partial(h->prec, cx[k].Ape, h->pe_w, cx[k].part, cx[k].cnt * h->T, D, h->Kp,
                                   splitk_slices(h->prec, cx[k].cnt * h->T, D, h->Kp), st, h->pe_e); }
    for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_LAYERNORM, st);
        rc = launch_embed_ln(h->prec, cx[k].x, cx[k].part, splitk_slices(h->prec, cx[k].cnt * h->T, D, h->Kp), h->pe_b, h->pos,
                             h->cls, h->blk[0].n1w, h->blk[0].n1b, cx[k].xn, cx[k].cnt, h->T, D, c.ln_eps, st); }
    for (int i = 0; i < c.blocks && !rc; ++i) {
        const Block& b = h->blk[i];
        const Block* nx = (i + 1 < c.blocks) ? &h->blk[i + 1] : nullptr;
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_QKV, st);
            rc = launch_linear(h->prec, cx[k].xn, b.qkvw, b.qkvb, cx[k].qkv, cx[k].M, 3 * D, D, 0, st, b.qkve); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_ATTENTION, st);
            rc = launch_attention(h->prec, cx[k].qkv, cx[k].attn, cx[k].cnt, h->N, c.heads, st, &h->attn_ws, plain16(h->prec)); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_PROJ, st);
            rc = launch_linear_partial(h->prec, cx[k].attn, b.projw, cx[k].part, cx[k].M, D, D,
                                       splitk_slices(h->prec, cx[k].M, D, D), st, b.proje); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_RESIDUAL_LN, st);
            rc = launch_residual_ln(h->prec, cx[k].x, cx[k].part, splitk_slices(h->prec, cx[k].M, D, D), b.projb, b.ls1,
                                    b.n2w, b.n2b, cx[k].xn, cx[k].M, D, c.ln_eps, st); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_FC1, st);
            rc = launch_linear(h->prec, cx[k].xn, b.fc1w, b.fc1b, cx[k].hid, cx[k].M, h->hidden, D, 1, st, b.fc1e); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_FC2, st);
            rc = launch_linear_partial(h->prec, cx[k].hid, b.fc2w, cx[k].part, cx[k].M, D, h->hidden,
                                       splitk_slices(h->prec, cx[k].M, D, h->hidden), st, b.fc2e); }
        for (int k = 0; k < n && !rc; ++k) { Span sp(h, KC_RESIDUAL_LN, st);
            rc = launch_residual_ln(h->prec, cx[k].x, cx[k].part, splitk_slices(h->prec, cx[k].M, D, h->hidden), b.fc2b,
                                    b.ls2, nx ? nx->n1w : nullptr, nx ? nx->n1b : nullptr, cx[k].xn, cx[k].M, D, c.ln_eps, st,
                                    (!nx && cx[k].want_desc) ? &cx[k].desc : nullptr); }
    }
    if (rc) return set_err(h, rc, "forward launch failed");
    return 0;
}

ChainCtx fill_ctx(vitvs_handle* h, int i0, int cnt, int n_des, const uint8_t* des, const uint8_t* cur, float* part) {
    const vitvs_config& c = h->cfg;
    const int D = c.dim;
    const size_t es = elem_size(h->prec);
    const size_t img_bytes = frame_bytes(h);
    const size_t row0 = (size_t)i0 * h->N;
    ChainCtx cx;
    cx.rs = h->fr.in_h ? &h->fr : nullptr;
    cx.cnt = cnt; cx.M = cnt * h->N; cx.part = part;
    cx.x = h->x + row0 * D;
    cx.xn = (unsigned char*)h->xn + row0 * D * es;
    cx.qkv = (unsigned char*)h->qkv + row0 * 3 * D * es;
    cx.attn = (unsigned char*)h->attn + row0 * D * es;
    cx.hid = (unsigned char*)h->hid + row0 * h->hidden * es;
    cx.Ape = (unsigned char*)h->Ape + (size_t)i0 * h->T * h->Kp * es;
    PatchifyArgs& pa = cx.pa;
    pa.n_des = std::max(0, std::min(i0 + cnt, n_des) - i0);
    pa.n_cur = cnt - pa.n_des;
    pa.des = des ? des + (size_t)std::min(i0, n_des) * img_bytes : nullptr;
    pa.cur = cur ? cur + (size_t)std::max(i0 - n_des, 0) * img_bytes : nullptr;
    pa.S = c.img_size; pa.patch = c.patch; pa.stride = c.stride; pa.grid = h->grid; pa.Kp = h->Kp; pa.D = D;
    for (int i = 0; i < 3; ++i) { pa.mean[i] = c.mean[i]; pa.std[i] = c.std[i]; }
    pa.cls = h->cls; pa.pos = h->pos;
    return cx;
}

// Forward of images [i0, i0 + cnt) of the call's image list (desired frames first, then current
// frames) on stream `st`: an independent chain of launches touching only those images' rows.
int forward_chain(vitvs_handle* h, int i0, int cnt, int n_des, const uint8_t* des, const uint8_t* cur, float* part,
                  hipStream_t st) {
    if (i0 == 0) h->goal_frames = 0;            // rows of a cached goal are about to be overwritten (vitvs_set_goal_dev re-arms)
    ChainCtx cx = fill_ctx(h, i0, cnt, n_des, des, cur, part);
    if (h->desc_keys >= 0 && desc_in_forward(h)) {
        cx.want_desc = true;
        if (h->cfg.binned) cx.desc.sq = h->sq + (size_t)i0 * h->T;
        else cx.desc.dn = h->dn + (size_t)i0 * h->T * h->Dp;
        cx.desc.zero_a = h->row_best; cx.desc.zero_b = h->col_best;
        cx.desc.T = h->T;
        cx.desc.zero_count = (i0 == 0 || h->goal_frames > 0) ? h->desc_keys : 0;   // the call's only chain clears the arg-max keys
    }
    return forward_lockstep(h, &cx, 1, st);
}

int forward(vitvs_handle* h, int n_des, const uint8_t* des, int n_cur, const uint8_t* cur, hipStream_t st) {
    const int n_img = n_des + n_cur;
    if (n_img <= 0 || n_img > h->n_img_max) return set_err(h, -3, "frame count exceeds the handle's capacity");
    if (vitvs_weights_ready(h) != 0) return set_err(h, -4, "weights not fully loaded: " + h->err);
    return forward_chain(h, 0, n_img, n_des, des, cur, h->part, st);
}

// num_pairs of one call: <= 0 means the handle's default (cfg.num_pairs); the reference changes it per call site
// (24 in the servo loop, 48 in the rotation search, vitvs_v2.py:1151-1189), so it is a per-call argument.
int call_num_pairs(const vitvs_handle* h, int32_t num_pairs) { return num_pairs > 0 ? num_pairs : h->cfg.num_pairs; }

int run_servo(vitvs_handle* h, int n_pairs, int T, const uint16_t* Z, const double* K, int mode, int num_pairs,
              const int32_t* selection, const int32_t* n_selected, double* v_c, int32_t* status, hipStream_t st) {
    const vitvs_config& c = h->cfg;
    if (num_pairs <= 0 || num_pairs > c.max_rows) return set_err(h, -5, "num_pairs must be in 1 .. max_rows");
    const int g = (int)floor(sqrt((double)T));  // reference: int(np.sqrt(T)), vitvs_v2.py:75
    if (g * g != T) return set_err(h, -5, "token count is not a square grid");
    if (mode < 0 || mode > 2) return set_err(h, -5, "unknown selection mode");
    if (mode != VITVS_SELECT_DENSE && !selection) return set_err(h, -5, "selection array required for this mode");
    if (mode == VITVS_SELECT_EXPLICIT && !n_selected) return set_err(h, -5, "n_selected required for EXPLICIT");
    if (mode == VITVS_SELECT_DENSE && c.max_rows < T) return set_err(h, -5, "DENSE selection needs max_rows >= T");
    ServoArgs a;
    memset(&a, 0, sizeof(a));
    a.n_pairs = n_pairs; a.T = T; a.grid = g; a.num_pairs = num_pairs; a.mode = mode;
    a.input_size = c.img_size; a.u_max = c.u_max; a.v_max = c.v_max; a.depth_h = c.v_max; a.depth_w = c.u_max;
    const double scale = (double)c.img_size / (double)g;                 // vitvs_v2.py:511
    a.scale_f = (float)scale; a.half_f = (float)(scale / 2.0);
    a.scale_x = (double)c.u_max / (double)c.img_size;                    // vitvs_v2.py:544
    a.scale_y = (double)c.v_max / (double)c.img_size;                    // vitvs_v2.py:545
    a.K = K; a.lambda = c.lambda;
    a.row_best = h->row_best; a.col_best = h->col_best; a.depth = Z;
    a.selection = selection; a.n_selected = n_selected;
    a.sel_stride = (mode == VITVS_SELECT_EXPLICIT) ? num_pairs : T;
    a.v_c = v_c; a.status = status; a.nn1 = h->nn1; a.nn2 = h->nn2; a.sim1 = h->sim1; a.info = h->info;
    a.sel_out = h->sel_out; a.s_uv = h->s_uv; a.feat = h->feat; a.L_ws = h->Lws; a.max_rows = c.max_rows;
    h->last_pairs = n_pairs; h->last_T = T;
    int rc = 0;
    { Span sp(h, KC_SERVO, st); rc = launch_servo(a, st); }
    if (rc) return set_err(h, rc, "servo launch failed (LDS budget or bad arguments)");
    return 0;
}

// The pinned staging block of the host-pointer entry points, sized for the handle's capacity and current frame geometry.
int ensure_host_stage(vitvs_handle* h) {
    const vitvs_config& c = h->cfg;
    if (h->hs.base && h->hs.frame_cap >= h->staged_frame_bytes) return 0;
    if (h->hs.base) {
        VITVS_HIP_CHECK(hipDeviceSynchronize());            // a launch may still read the previous block
        (void)hipHostFree(h->hs.base);
        h->hs = vitvs_handle::HostStage{};
        h->details_pinned = false;
        h->staged_des = nullptr;
    }
    const size_t P = c.max_pairs, fb = (h->staged_frame_bytes + 255) & ~(size_t)255;
    const size_t sel_cap = P * (size_t)(h->T > c.max_rows ? h->T : c.max_rows);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_cur = 0, o_des = o_cur + P * fb, o_depth = o_des + P * fb, o_K = o_depth + up(P * (size_t)c.u_max * c.v_max * 2),
                 o_vc = o_K + up(P * 32), o_sel = o_vc + up(P * 48), o_nsel = o_sel + up(sel_cap * 4), o_st = o_nsel + up(P * 4),
                 o_det = o_st + up(P * 4), total = o_det + up(h->det_bytes) + 256;
    void* p = nullptr;
    VITVS_HIP_CHECK(hipHostMalloc(&p, total, hipHostMallocDefault));
    memset(p, 0, total);
    unsigned char* b = static_cast<unsigned char*>(p);
    h->hs.base = b; h->hs.frame_cap = h->staged_frame_bytes;
    h->hs.cur = b + o_cur; h->hs.des = b + o_des; h->hs.depth = reinterpret_cast<uint16_t*>(b + o_depth);
    h->hs.K = reinterpret_cast<double*>(b + o_K); h->hs.vc = reinterpret_cast<double*>(b + o_vc);
    h->hs.sel = reinterpret_cast<int32_t*>(b + o_sel); h->hs.nsel = reinterpret_cast<int32_t*>(b + o_nsel);
    h->hs.status = reinterpret_cast<int32_t*>(b + o_st); h->hs.det = b + o_det;
    if (!h->host_stream) VITVS_HIP_CHECK(hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
    return 0;
}

// Wait for a stream the way a control loop wants it: poll (the update takes ~0.5 ms; a blocking wait adds its wake-up
// latency to every update), then hand over to the blocking wait if the device is far behind.
int wait_stream(hipStream_t st) {
    for (int i = 0; i < 200000; ++i) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery", __FILE__, __LINE__);
    }
    VITVS_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

}  // namespace

extern "C" {

int vitvs_abi_version(void) { return VITVS_ABI_VERSION; }

const char* vitvs_last_error(const vitvs_handle* h) {
    if (h && !h->err.empty()) return h->err.c_str();
    return g_last_error.c_str();
}

int vitvs_tokens(const vitvs_handle* h) { return h ? h->T : -1; }
int vitvs_desc_dim(const vitvs_handle* h) { return h ? h->Dp : -1; }

int vitvs_create(const vitvs_config* cfg, vitvs_handle** out) {
    if (!cfg || !out) return set_err(nullptr, -1, "null argument");
    std::string why;
    if (check_cfg(cfg, why)) return set_err(nullptr, -1, "bad config: " + why);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return set_err(nullptr, -2, "no HIP device available");
    vitvs_handle* h = new vitvs_handle();
    h->cfg = *cfg;
    h->prec = to_prec(cfg->precision);
    (void)hipGetDevice(&h->device);
    g_current_device = h->device;
    h->wstore = std::make_shared<WeightStore>();
    h->wstore->device = h->device;
    h->grid = 1 + (cfg->img_size - cfg->patch) / cfg->stride;
    h->T = h->grid * h->grid;
    h->N = h->T + 1;
    const int pk = 3 * cfg->patch * cfg->patch;
    h->Kp = (pk + 63) / 64 * 64;
    h->Dp = cfg->binned ? 9 * cfg->dim : cfg->dim;
    h->hidden = 4 * cfg->dim;
    h->n_img_max = 2 * cfg->max_pairs;
    h->blk.resize(cfg->blocks);
    {   // The depth pixels the law can read: the patch centre of each token in camera resolution — servo.hip token_pixel, the
        // same operations in the same precisions (fp32 centre, fp64 scale, round half to even; vitvs_v2.py:511-513, 544-549).
        // tests/test_gpu_path.py::test_host_pointer_entry_point_matches_device_entry_point holds the two restatements together.
        const int g = (int)floor(sqrt((double)h->T));
        if (g * g == h->T) {
            const double scale = (double)cfg->img_size / (double)g;
            const float scale_f = (float)scale, half_f = (float)(scale / 2.0);
            const double sx = (double)cfg->u_max / (double)cfg->img_size, sy = (double)cfg->v_max / (double)cfg->img_size;
            for (int tok = 0; tok < h->T; ++tok) {
                volatile float rm = (float)(tok / g) * scale_f, cm = (float)(tok % g) * scale_f;   // (separately rounded product)
                const float r = rm + half_f, cc = cm + half_f;
                const long u = (long)rint((double)cc * sx), v = (long)rint((double)r * sy);
                if (u >= 0 && u < cfg->u_max && v >= 0 && v < cfg->v_max) h->depth_sites.push_back((int32_t)(v * cfg->u_max + u));
            }
        }
    }
    // hipGraph replay of the update is opt-in (VITVS_GRAPH=1, read once per handle): with kernel arguments in device
    // memory (HIP_FORCE_DEV_KERNARG=1, set by the Python package before HIP initialises) plain stream launches measured
    // 2 % FASTER than replaying the captured graph; the graph's use is a caller whose host thread cannot spare the
    // ~0.35 ms of launch calls per update.
    const char* ng = getenv("VITVS_GRAPH");
    h->use_graphs = (ng && ng[0] == '1');
    const size_t M = (size_t)h->n_img_max * h->N, D = cfg->dim, es = elem_size(h->prec);
    int rc = 0;
    unsigned char* p8 = nullptr;
#define ALLOC_BYTES(field, bytes) \
    if (!rc) { rc = dev_alloc(h, &p8, (bytes)); h->field = reinterpret_cast<decltype(h->field)>(p8); }
    ALLOC_BYTES(Ape, (size_t)h->n_img_max * h->T * h->Kp * es);
    ALLOC_BYTES(xn, M * D * es);
    ALLOC_BYTES(qkv, M * 3 * D * es);
    ALLOC_BYTES(attn, M * D * es);
    ALLOC_BYTES(hid, M * h->hidden * es);
#undef ALLOC_BYTES
    if (!rc) rc = dev_alloc(h, &h->x, M * D);
    if (!rc) rc = dev_alloc(h, &h->part, (size_t)8 * M * D);   // at most 8 split-K slices (splitk_slices)
    {   // the key-split plan depends on the image count of a call: size for the largest need over 1 .. n_img_max, under the
        // plan of a handle that runs ALONE (in_flight 1: the divided plan, the only one that needs a workspace) — whatever hint
        // the calling thread carries (vitvs_op_plan_in_flight) and whatever `in_flight` the handle is given later
        const int thread_hint = g_updates_in_flight;
        g_updates_in_flight = 1;
        size_t f = 0, t = 0;
        for (int n = 1; n <= h->n_img_max; ++n) {
            f = std::max(f, attention_workspace_floats(n, h->N, cfg->heads));
            t = std::max(t, attention_ticket_count(n, h->N, cfg->heads));
        }
        g_updates_in_flight = thread_hint;
        if (!rc && f) rc = dev_alloc(h, &h->attn_ws.state, f);
        if (!rc && t) rc = dev_alloc(h, &h->attn_ws.tickets, t);   // dev_alloc zeroes: the tickets start at 0
    }
    h->dn_elems = (size_t)h->n_img_max * h->T * h->Dp;
    if (!rc) rc = dev_alloc(h, &h->dn, h->dn_elems);
    // binned descriptors: the velocity path takes the 9 D-wide Gram as a 3 x 3 stencil over the raw token Gram (correspond.hip),
    // which needs T x T floats per pair (3136 tokens: 39 MB); beyond 8 GiB in all it keeps the concatenated form
    if (!rc && cfg->binned && (size_t)cfg->max_pairs * h->T * h->T * 4 <= (8ull << 30))
        rc = dev_alloc(h, &h->gram_ws, (size_t)cfg->max_pairs * h->T * h->T);
    // 16-bit modes, >= 1024 tokens: the Gram runs on the f16 matrix cores from a hi / lo split of the descriptors (correspond.hip)
    if (!rc && !h->gram_ws && plain16(h->prec) && h->T >= 1024 && h->Dp % 64 == 0 &&
        gram_split_elems(h->n_img_max, h->T, h->Dp) * 2 < (1ull << 32))
        rc = dev_alloc(h, &h->dh, gram_split_elems(h->n_img_max, h->T, h->Dp));
    if (!rc) rc = dev_alloc(h, &h->sq, (size_t)h->n_img_max * h->T);

    h->best_elems = (size_t)cfg->max_pairs * h->T;
    if (!rc) rc = dev_alloc(h, &h->row_best, h->best_elems);
    if (!rc) rc = dev_alloc(h, &h->col_best, h->best_elems);
    const size_t P = cfg->max_pairs, R = cfg->max_rows;
    h->det_bytes = P * 32 + P * R * 16 + P * R * 32 + 3 * h->best_elems * 4;
    if (!rc) rc = dev_alloc(h, &h->det_block, h->det_bytes);
    if (!rc) detail_pointers(h, h->det_block);
    if (!rc) rc = dev_alloc(h, &h->sel_out, P * R);
    if (!rc) rc = dev_alloc(h, &h->Lws, P * 7 * 2 * R);
    const size_t img_bytes = (size_t)cfg->img_size * cfg->img_size * 3;
    h->staged_frame_bytes = img_bytes;
    if (!rc) rc = dev_alloc(h, &h->st_cur, P * img_bytes);
    if (!rc) rc = dev_alloc(h, &h->st_des, P * img_bytes);
    const size_t sel_cap = P * (size_t)(h->T > cfg->max_rows ? h->T : cfg->max_rows);
    if (!rc) rc = dev_alloc(h, &h->st_sel, sel_cap);
    if (!rc) rc = dev_alloc(h, &h->st_nsel, P);
    if (rc) {
        std::string msg = g_last_error;
        vitvs_destroy(h);
        return set_err(nullptr, rc, "allocation failed: " + msg);
    }
    *out = h;
    return 0;
}

void vitvs_destroy(vitvs_handle* h) {
    if (!h) return;
    DeviceScope dev(h);
    (void)hipDeviceSynchronize();               // nothing of this handle's is in flight when its graphs and memory go
    drop_graphs(h);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->hs.base) (void)hipHostFree(h->hs.base);
    if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
    delete h;
}

int vitvs_share_weights(vitvs_handle* h, const vitvs_handle* src) {
    if (!h || !src || h == src) return set_err(h, -1, "null argument");
    const vitvs_config &a = h->cfg, &b = src->cfg;
    if (h->device != src->device) return set_err(h, -5, "handles of different devices cannot share weights");
    if (a.img_size != b.img_size || a.patch != b.patch || a.stride != b.stride || a.dim != b.dim || a.heads != b.heads ||
        a.blocks != b.blocks || a.layerscale != b.layerscale || a.precision != b.precision)
        return set_err(h, -5, "weights are shared between handles of one network, input geometry and precision");
    if (!h->have.empty() && !h->borrowed) return set_err(h, -5, "this handle already holds weights of its own");
    if (src->borrowed) return set_err(h, -5, "share from the handle that owns the weights");
    if (vitvs_weights_ready(src) != 0) return set_err(h, -4, "the source handle's weights are not fully loaded");
    if (!h->graphs.empty()) {                   // borrowed before, from another owner: captured updates read THOSE weights
        DeviceScope dev(h);
        VITVS_HIP_CHECK(hipDeviceSynchronize());
        drop_graphs(h);
    }
    h->blk = src->blk;
    h->pe_w = src->pe_w; h->pe_e = src->pe_e; h->pe_b = src->pe_b; h->cls = src->cls; h->pos = src->pos;
    h->have = src->have;
    h->wstore = src->wstore;                    // shared ownership: the weights outlive whichever of the two is destroyed first
    h->ready = true;
    h->borrowed = true;
    return 0;
}

int vitvs_set_tensor(vitvs_handle* h, const char* name, const float* data, int64_t numel) {
    if (!h || !name || !data) return set_err(h, -1, "null argument");
    if (h->borrowed) return set_err(h, -5, "this handle borrows its weights (vitvs_share_weights): upload to their owner");
    DeviceScope dev(h);
    const vitvs_config& c = h->cfg;
    const size_t D = c.dim, H4 = h->hidden;
    const std::string nm(name);
    auto want = [&](size_t n) -> int {
        if ((size_t)numel != n) return set_err(h, -6, nm + ": expected " + std::to_string(n) + " elements, got " + std::to_string(numel));
        return 0;
    };
    int rc = 0;
    if (nm == "patch_embed.proj.weight") {
        const size_t pk = 3 * (size_t)c.patch * c.patch;
        if ((rc = want(D * pk))) return rc;
        rc = upload_matrix(h, &h->pe_w, data, D, pk, h->Kp, &h->pe_e);
    } else if (nm == "patch_embed.proj.bias") {
        if ((rc = want(D))) return rc;
        rc = upload_f32(h, &h->pe_b, data, D);
    } else if (nm == "cls_token") {
        if ((rc = want(D))) return rc;
        rc = upload_f32(h, &h->cls, data, D);
    } else if (nm == "pos_embed") {
        if ((rc = want((size_t)h->N * D))) return rc;
        rc = upload_f32(h, &h->pos, data, (size_t)h->N * D);
    } else if (nm.rfind("blocks.", 0) == 0) {
        const size_t dot = nm.find('.', 7);
        if (dot == std::string::npos) return set_err(h, -6, "unknown tensor " + nm);
        const int i = atoi(nm.substr(7, dot - 7).c_str());
        if (i >= c.blocks) return 0;  // blocks after the descriptor layer are never run
        if (i < 0) return set_err(h, -6, "unknown tensor " + nm);
        Block& b = h->blk[i];
        const std::string leaf = nm.substr(dot + 1);
        if (leaf == "norm1.weight") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.n1w, data, D); }
        else if (leaf == "norm1.bias") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.n1b, data, D); }
        else if (leaf == "norm2.weight") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.n2w, data, D); }
        else if (leaf == "norm2.bias") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.n2b, data, D); }
        else if (leaf == "attn.qkv.weight" || leaf == "attn.qkv.bias") {
            // 16-bit modes: the q rows carry hd^-0.5 * log2(e) (kernels.h kAttnQScale), folded in here in fp32, before the one
            // rounding of the weights to 16 bits: the attention kernels then find s * scale * log2(e) in their accumulators
            const bool w = leaf == "attn.qkv.weight";
            const size_t n = w ? 3 * D * D : 3 * D, nq = w ? D * D : D;
            if ((rc = want(n))) return rc;
            std::vector<float> scaled;
            const float* src = data;
            if (plain16(h->prec)) {
                scaled.assign(data, data + n);
                for (size_t i = 0; i < nq; ++i) scaled[i] *= kAttnQScale;
                src = scaled.data();
            }
            rc = w ? upload_matrix(h, &b.qkvw, src, 3 * D, D, D, &b.qkve) : upload_f32(h, &b.qkvb, src, 3 * D);
        }
        else if (leaf == "attn.proj.weight") { if ((rc = want(D * D))) return rc; rc = upload_matrix(h, &b.projw, data, D, D, D, &b.proje); }
        else if (leaf == "attn.proj.bias") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.projb, data, D); }
        else if (leaf == "mlp.fc1.weight") { if ((rc = want(H4 * D))) return rc; rc = upload_matrix(h, &b.fc1w, data, H4, D, D, &b.fc1e); }
        else if (leaf == "mlp.fc1.bias") { if ((rc = want(H4))) return rc; rc = upload_f32(h, &b.fc1b, data, H4); }
        else if (leaf == "mlp.fc2.weight") { if ((rc = want(D * H4))) return rc; rc = upload_matrix(h, &b.fc2w, data, D, H4, H4, &b.fc2e); }
        else if (leaf == "mlp.fc2.bias") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.fc2b, data, D); }
        else if (leaf == "ls1.gamma") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.ls1, data, D); }
        else if (leaf == "ls2.gamma") { if ((rc = want(D))) return rc; rc = upload_f32(h, &b.ls2, data, D); }
        else return set_err(h, -6, "unknown tensor " + nm);
    } else if (nm == "norm.weight" || nm == "norm.bias" || nm.rfind("head.", 0) == 0 || nm == "mask_token") {
        return 0;  // final norm / head are dead work for the descriptor
    } else {
        return set_err(h, -6, "unknown tensor " + nm);
    }
    if (rc == 0) {
        h->have[nm] = true;
        h->ready = false;
    }
    return rc;
}

int vitvs_weights_ready(const vitvs_handle* hc) {
    vitvs_handle* h = const_cast<vitvs_handle*>(hc);
    if (!h) return -1;
    if (h->ready) return 0;
    std::vector<std::string> need = {"patch_embed.proj.weight", "patch_embed.proj.bias", "cls_token", "pos_embed"};
    static const char* leaves[] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                                   "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                                   "mlp.fc2.weight", "mlp.fc2.bias"};
    for (int i = 0; i < h->cfg.blocks; ++i) {
        for (const char* l : leaves) need.push_back("blocks." + std::to_string(i) + "." + l);
        if (h->cfg.layerscale) {
            need.push_back("blocks." + std::to_string(i) + ".ls1.gamma");
            need.push_back("blocks." + std::to_string(i) + ".ls2.gamma");
        }
    }
    for (const auto& n : need)
        if (!h->have.count(n)) {
            h->err = "missing tensor " + n;
            return -4;
        }
    h->ready = true;
    return 0;
}

int vitvs_forward_tokens_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, float* tokens, void* stream) {
    if (!h || !frames || !tokens) return set_err(h, -1, "null argument");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    int rc = forward(h, n_frames, frames, 0, nullptr, st);
    if (rc) return rc;
    VITVS_HIP_CHECK(hipMemcpyAsync(tokens, h->x, (size_t)n_frames * h->N * h->cfg.dim * sizeof(float),
                                   hipMemcpyDeviceToDevice, st));
    return 0;
}

static int upload_table(vitvs_handle* h, const std::vector<int>& v, int** dev) {
    void* p = nullptr;
    VITVS_HIP_CHECK(hipMalloc(&p, v.size() * sizeof(int)));
    h->allocs.push_back(p);
    VITVS_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    *dev = static_cast<int*>(p);
    return 0;
}

// Pillow's tables for (in_h, in_w) -> img_size in `t` (a new resolution replaces the previous tables; synchronises once).
static int resize_tables(vitvs_handle* h, ResizeArgs& t, int in_h, int in_w) {
    if (in_h == t.in_h && in_w == t.in_w) return 0;
    for (const int** d : {&t.xb, &t.xk, &t.yb, &t.yk}) {
        if (*d) {
            VITVS_HIP_CHECK(hipDeviceSynchronize());   // a launch on ANY stream may still read them
            dev_free(h, (void*)*d);
            *d = nullptr;
        }
    }
    t = ResizeArgs{};
    if (in_h == 0 && in_w == 0) return 0;
    const int S = h->cfg.img_size, patch = h->cfg.patch;
    std::vector<int> xb, xk, yb, yk;
    const int ksx = resize_coefficients(in_w, S, xb, xk);
    const int ksy = resize_coefficients(in_h, S, yb, yk);
    // the patch-row build (patchify_resize_kernel) takes the camera rows of a patch from its first and last pixel row: both
    // bounds of Pillow's windows grow with the output row (they do: the centre does), checked here rather than assumed
    int rows = 0;
    for (int y = 0; y + 1 < S; ++y)
        if (yb[2 * y] > yb[2 * y + 2] || yb[2 * y] + yb[2 * y + 1] > yb[2 * y + 2] + yb[2 * y + 3]) return set_err(h, -5, "resize windows are not monotone");
    for (int y = 0; y + patch <= S; ++y) rows = std::max(rows, yb[2 * (y + patch - 1)] + yb[2 * (y + patch - 1) + 1] - yb[2 * y]);
    int *dxb = nullptr, *dxk = nullptr, *dyb = nullptr, *dyk = nullptr;
    if (upload_table(h, xb, &dxb) || upload_table(h, xk, &dxk) || upload_table(h, yb, &dyb) || upload_table(h, yk, &dyk))
        return set_err(h, -6, "resize table upload failed");
    t.xb = dxb; t.xk = dxk; t.yb = dyb; t.yk = dyk;
    t.in_h = in_h; t.in_w = in_w; t.ksx = ksx; t.ksy = ksy; t.rows = rows;
    return 0;
}

int vitvs_resize_frames_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t in_h, int32_t in_w,
                            uint8_t* out, void* stream) {
    if (!h || !frames || !out) return set_err(h, -1, "null argument");
    if (n_frames <= 0 || in_h <= 0 || in_w <= 0) return set_err(h, -5, "bad frame geometry");
    DeviceScope dev(h);
    if (int rc = resize_tables(h, h->rs, in_h, in_w)) return rc;
    const int rc = launch_resize_bicubic(frames, out, n_frames, in_h, in_w, h->cfg.img_size, h->rs.xb, h->rs.xk, h->rs.ksx,
                                         h->rs.yb, h->rs.yk, h->rs.ksy, as_stream(stream));
    if (rc) return set_err(h, rc, "resize launch failed");
    return 0;
}

int vitvs_set_frame_size(vitvs_handle* h, int32_t in_h, int32_t in_w) {
    if (!h) return set_err(h, -1, "null argument");
    if (in_h < 0 || in_w < 0 || (in_h == 0) != (in_w == 0)) return set_err(h, -5, "bad frame geometry");
    DeviceScope dev(h);
    if (in_h == h->cfg.img_size && in_w == h->cfg.img_size) in_h = in_w = 0;   // nothing to resize: the plain patch-row build
    if (in_h == h->fr.in_h && in_w == h->fr.in_w) return 0;
    // Everything the new geometry needs is built FIRST; the handle changes only once nothing can fail any more, so an error
    // (tables, the LDS limit of the fused resize, the staging allocation) leaves the previous geometry fully usable.
    ResizeArgs fresh{};
    if (int rc = resize_tables(h, fresh, in_h, in_w)) return rc;
    if (fresh.in_h && (size_t)fresh.rows * h->cfg.patch * 3 > 64 * 1024) {
        (void)resize_tables(h, fresh, 0, 0);
        return set_err(h, -3, "camera frame too large for the fused resize (use vitvs_resize_frames_dev)");
    }
    const size_t need = fresh.in_h ? (size_t)fresh.in_h * fresh.in_w * 3 : (size_t)h->cfg.img_size * h->cfg.img_size * 3;
    uint8_t *new_cur = nullptr, *new_des = nullptr;
    if (need > h->staged_frame_bytes) {             // host-buffer entry points stage whole frames
        const size_t P = (size_t)h->cfg.max_pairs;
        if (dev_alloc(h, &new_cur, P * need) || dev_alloc(h, &new_des, P * need)) {
            dev_free(h, new_cur);
            (void)resize_tables(h, fresh, 0, 0);
            return set_err(h, -6, "frame staging allocation failed (the previous frame geometry stays in place)");
        }
    }
    // commit.  Captured updates hold the previous tables' addresses: they go (a cached goal's tokens do not depend on the
    // geometry and stay)
    VITVS_HIP_CHECK(hipDeviceSynchronize());
    drop_graphs(h);
    (void)resize_tables(h, h->fr, 0, 0);            // frees the previous tables
    h->fr = fresh;
    h->staged_des = nullptr;                        // frames of another geometry
    if (new_cur) {
        dev_free(h, h->st_cur);
        dev_free(h, h->st_des);
        h->st_cur = new_cur;
        h->st_des = new_des;
        h->staged_frame_bytes = need;
    }
    return 0;
}

int vitvs_extract_descriptors_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, float* desc, void* stream) {
    if (!h || !frames || !desc) return set_err(h, -1, "null argument");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    int rc = forward(h, n_frames, frames, 0, nullptr, st);
    if (rc) return rc;
    rc = launch_descriptors(h->x, h->dn, desc, h->sq, n_frames, h->T, h->grid, h->cfg.dim, h->cfg.binned, nullptr, nullptr, 0, st);
    if (rc) return set_err(h, rc, "descriptor launch failed");
    return 0;
}

int vitvs_extract_facet_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t facet, float* desc,
                            void* stream) {
    if (facet < 0 || facet > 2) return set_err(h, -5, "facet must be 0 (query), 1 (key) or 2 (value)");
    return vitvs_extract_descriptors_ex_dev(h, n_frames, frames, facet, 0, 0, desc, stream);
}

int vitvs_extract_descriptors_ex_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t facet, int32_t bin,
                                     int32_t include_cls, float* desc, void* stream) {
    if (!h || !frames || !desc) return set_err(h, -1, "null argument");
    if (facet < 0 || facet > 3) return set_err(h, -5, "facet must be 0 (query), 1 (key), 2 (value) or 3 (token)");
    if (bin && include_cls)   // the reference's assertion (dinov2_extractor.py:330-331)
        return set_err(h, -5, "bin = True and include_cls = True are not supported together, set one of them False.");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    int rc = forward(h, n_frames, frames, 0, nullptr, st);   // the last block's qkv launch leaves its output in h->qkv
    if (rc) return rc;
    const int T = h->T, D = h->cfg.dim;
    const float* src = h->x;                                  // token facet: the residual stream itself, [n][1 + T][D]
    if (facet < 3) {
        // q / k / v of blocks[layer] in the reference's layout (index d * H + h), fp32, WITH the cls row, over the residual
        // stream's own buffer (the forward is done with it; any cached goal was dropped by the forward above)
        rc = launch_facet(h->prec, h->qkv, h->x, n_frames, T, h->cfg.heads, facet,
                          (facet == 0 && plain16(h->prec)) ? 1.0f / kAttnQScale : 1.0f, 1, st);   // the q rows carry the attention scale
        if (rc) return set_err(h, rc, "facet launch failed");
    }
    if (bin) {
        rc = launch_descriptors(src, nullptr, desc, h->sq, n_frames, T, h->grid, D, 1, nullptr, nullptr, 0, st);
        if (rc) return set_err(h, rc, "descriptor launch failed");
    } else if (include_cls) {
        VITVS_HIP_CHECK(hipMemcpyAsync(desc, src, (size_t)n_frames * (T + 1) * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        VITVS_HIP_CHECK(hipMemcpy2DAsync(desc, (size_t)T * D * sizeof(float), src + D, (size_t)(T + 1) * D * sizeof(float),
                                         (size_t)T * D * sizeof(float), n_frames, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

int vitvs_extract_saliency_dev(vitvs_handle* h, int32_t n_frames, const uint8_t* frames, int32_t n_heads, const int32_t* head_idxs,
                               float* saliency, void* stream) {
    if (!h || !frames || !head_idxs || !saliency) return set_err(h, -1, "null argument");
    if (n_heads <= 0 || n_heads > 16) return set_err(h, -5, "1 .. 16 heads");
    for (int i = 0; i < n_heads; ++i)
        if (head_idxs[i] < 0 || head_idxs[i] >= h->cfg.heads) return set_err(h, -5, "head index outside the model's heads");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    int rc = forward(h, n_frames, frames, 0, nullptr, st);   // the last block's qkv launch leaves its output in h->qkv
    if (rc) return rc;
    if (h->prec == PREC_X2) return set_err(h, -5, "saliency maps are not available in the f16x2 precision (use fp32)");
    rc = launch_saliency(h->prec, h->qkv, saliency, n_frames, h->T, h->cfg.heads, head_idxs, n_heads, plain16(h->prec), st);
    if (rc) return set_err(h, rc, rc == -3 ? "too many tokens for the saliency kernel's LDS rows" : "saliency launch failed");
    return 0;
}

int vitvs_correspond_dev(vitvs_handle* h, int32_t T, int32_t Dp, const float* desc1, const float* desc2, int32_t* nn_1,
                         int32_t* nn_2, float* sim_1, float* S_out, void* stream) {
    if (!h || !desc1 || !desc2 || !nn_1 || !nn_2 || !sim_1) return set_err(h, -1, "null argument");
    if (T <= 0 || Dp <= 0 || Dp % 32 != 0) return set_err(h, -5, "Dp must be a positive multiple of 32");
    if ((size_t)2 * T * Dp > h->dn_elems || (size_t)T > h->best_elems)
        return set_err(h, -3, "descriptors exceed the handle's workspace");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    h->goal_frames = 0;                         // the descriptor workspace is overwritten
    int rc = launch_normalize_rows(desc1, h->dn, T, Dp, st);
    if (!rc) rc = launch_normalize_rows(desc2, h->dn + (size_t)T * Dp, T, Dp, st);
    if (rc) return set_err(h, rc, "normalise launch failed");
    VITVS_HIP_CHECK(hipMemsetAsync(h->row_best, 0, (size_t)T * 8, st));
    VITVS_HIP_CHECK(hipMemsetAsync(h->col_best, 0, (size_t)T * 8, st));
    rc = launch_gram_argmax(h->dn, T, Dp, 1, 0, h->row_best, h->col_best, st);
    if (!rc) rc = launch_decode_best(h->row_best, h->col_best, T, nn_1, nn_2, sim_1, st);
    if (!rc && S_out) rc = launch_gram_dense(h->dn, T, Dp, 1, 0, S_out, st);
    if (rc) return set_err(h, rc, "correspondence launch failed");
    return 0;
}

int vitvs_servo_from_nn_dev(vitvs_handle* h, int32_t T, const int32_t* nn_1, const int32_t* nn_2, const float* sim_1,
                            const uint16_t* Z_mm, const double* K, int32_t select_mode, const int32_t* selection,
                            int32_t n_selected, int32_t num_pairs, double* v_c, int32_t* status, void* stream) {
    if (!h || !nn_1 || !nn_2 || !sim_1 || !K || !v_c || !status) return set_err(h, -1, "null argument");
    if ((size_t)T > h->best_elems) return set_err(h, -3, "T exceeds the handle's workspace");
    DeviceScope dev(h);
    hipStream_t st = as_stream(stream);
    h->details_pinned = false;
    int rc = launch_encode_best(nn_1, nn_2, sim_1, T, h->row_best, h->col_best, st);
    if (rc) return set_err(h, rc, "encode launch failed");
    VITVS_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->st_nsel), n_selected, 1, st));
    return run_servo(h, 1, T, Z_mm, K, select_mode, call_num_pairs(h, num_pairs), selection, h->st_nsel, v_c, status, st);
}

// One update = forward of the call's image list (desired frames first, then current frames) + the tail
// (descriptors when binned, Gram + arg-max, control law), all on the caller's stream.
struct UpdateArgs {
    int32_t n_pairs, des_shared, select_mode, num_pairs;
    const uint8_t *I_cur, *I_des;
    const uint16_t* Z_mm;
    const double* K;
    const int32_t *selection, *n_selected;
    double* v_c;
    int32_t* status;
    // host-pointer entry point: the caller's depth image is copied into the pinned staging block on the HOST, after the
    // forward's launches have been enqueued and before the law's launch (the only kernel that reads it): off the critical path
    // Only the pixels the law can ask for are copied: it looks the depth up at the patch CENTRE of a current-frame token
    // (vitvs_v2.py:511-553, 566-586), i.e. at one of T fixed sites of the image (late_sites: linear pixel indices, the handle's
    // depth_sites), so T 2-byte gathers stand for the 614 KB image.
    const uint16_t* late_src = nullptr;
    uint16_t* late_dst = nullptr;
    const int32_t* late_sites = nullptr;
    int late_count = 0, late_pairs = 0;
    size_t late_stride = 0;                     // pixels per depth image
};
static inline void late_inputs(const UpdateArgs& u) {
    if (!u.late_src) return;
    for (int b = 0; b < u.late_pairs; ++b) {
        const uint16_t* src = u.late_src + (size_t)b * u.late_stride;
        uint16_t* dst = u.late_dst + (size_t)b * u.late_stride;
        for (int i = 0; i < u.late_count; ++i) dst[u.late_sites[i]] = src[u.late_sites[i]];
    }
}

static int enqueue_update(vitvs_handle* h, const UpdateArgs& u, hipStream_t st) {
    const int n_des = u.des_shared ? 1 : u.n_pairs, n_img = n_des + u.n_pairs;
    h->desc_keys = u.n_pairs * h->T;
    // cached goal: only the current frames (images n_des .. n_img - 1 of the call's list) go through the network
    int rc = u.I_des ? forward_chain(h, 0, n_img, n_des, u.I_des, u.I_cur, h->part, st)
                     : forward_chain(h, n_des, u.n_pairs, n_des, nullptr, u.I_cur, h->part, st);
    h->desc_keys = -1;
    if (rc) return rc;
    if (h->cfg.binned && h->gram_ws) {
        // binned descriptors as a stencil over the raw token Gram: nothing 9 D wide is built or read (correspond.hip header)
        // (the tokens' squared norms came out of the forward's last launch, which also cleared the arg-max keys)
        { Span sp(h, KC_GRAM, st);
          rc = launch_gram_raw_tokens(h->x, h->T, h->cfg.dim, u.n_pairs, u.des_shared ? 1 : 0, h->gram_ws, st); }
        if (rc) return set_err(h, rc, "gram launch failed");
        { Span sp(h, KC_GRAM_STENCIL, st);
          rc = launch_gram_stencil_argmax(h->gram_ws, h->sq, h->T, h->grid, u.n_pairs, u.des_shared ? 1 : 0, h->row_best, h->col_best, st); }
        if (rc) return set_err(h, rc, "gram stencil launch failed");
        late_inputs(u);
        return run_servo(h, u.n_pairs, h->T, u.Z_mm, u.K, u.select_mode, u.num_pairs, u.selection, u.n_selected, u.v_c, u.status, st);
    }
    if (!desc_in_forward(h)) {
        Span sp(h, KC_DESCRIPTORS, st);
        rc = launch_descriptors(h->x, h->dn, nullptr, h->sq, n_img, h->T, h->grid, h->cfg.dim, h->cfg.binned,
                                h->row_best, h->col_best, u.n_pairs * h->T, st);
        if (rc) return set_err(h, rc, "descriptor launch failed");
    }
    if (h->dh) {
        rc = launch_split_desc(h->dn, h->dh, h->T, h->Dp, u.n_pairs, u.des_shared ? 1 : 0, st);
        if (rc) return set_err(h, rc, "descriptor split launch failed");
    }
    { Span sp(h, KC_GRAM, st);
      rc = h->dh ? launch_gram_argmax_split(h->dh, h->T, h->Dp, u.n_pairs, u.des_shared ? 1 : 0, h->row_best, h->col_best, st)
                 : launch_gram_argmax(h->dn, h->T, h->Dp, u.n_pairs, u.des_shared ? 1 : 0, h->row_best, h->col_best, st); }
    if (rc) return set_err(h, rc, "gram launch failed");
    late_inputs(u);
    return run_servo(h, u.n_pairs, h->T, u.Z_mm, u.K, u.select_mode, u.num_pairs, u.selection, u.n_selected, u.v_c,
                     u.status, st);
}

// VITVS_GRAPH=1: the update is captured once per argument tuple and replayed.  The selection array is the one argument
// a control loop changes every update (a fresh visiting order), so it is not part of the key: the graph reads the
// handle's own copy (st_sel / st_nsel), refreshed by two small device-to-device copies ahead of each replay.
static int replay_update(vitvs_handle* h, UpdateArgs u, hipStream_t st) {
    const size_t sel_elems = u.select_mode == VITVS_SELECT_EXPLICIT ? (size_t)u.n_pairs * u.num_pairs
                             : (u.select_mode == VITVS_SELECT_ORDER ? (size_t)u.n_pairs * h->T : 0);
    if (sel_elems && u.selection && u.selection != h->st_sel)
        VITVS_HIP_CHECK(hipMemcpyAsync(h->st_sel, u.selection, sel_elems * 4, hipMemcpyDefault, st));
    if (u.select_mode == VITVS_SELECT_EXPLICIT && u.n_selected && u.n_selected != h->st_nsel)
        VITVS_HIP_CHECK(hipMemcpyAsync(h->st_nsel, u.n_selected, (size_t)u.n_pairs * 4, hipMemcpyDefault, st));
    late_inputs(u);                             // a replay has no seam to do this later: before the launch
    if (u.selection) u.selection = h->st_sel;
    if (u.n_selected) u.n_selected = h->st_nsel;
    const std::vector<uintptr_t> key = {(uintptr_t)u.n_pairs, (uintptr_t)u.I_cur, (uintptr_t)u.I_des, (uintptr_t)u.des_shared,
                                        (uintptr_t)u.Z_mm, (uintptr_t)u.K, (uintptr_t)u.select_mode, (uintptr_t)u.num_pairs,
                                        (uintptr_t)(u.selection != nullptr), (uintptr_t)(u.n_selected != nullptr),
                                        (uintptr_t)u.v_c, (uintptr_t)u.status, (uintptr_t)h->fr.in_h, (uintptr_t)h->fr.in_w,
                                        (uintptr_t)h->info};
    vitvs_handle::GraphEntry* ge = nullptr;
    for (auto& g : h->graphs)
        if (g.key == key) ge = &g;
    if (!ge) {
        if (h->graphs.size() >= 32) {  // evict the least recently used entry (8 cameras sharing a pipeline slot = 8 keys)
            size_t victim = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i)
                if (h->graphs[i].last_use < h->graphs[victim].last_use) victim = i;
            if (h->graphs[victim].exec) (void)hipGraphExecDestroy(h->graphs[victim].exec);
            if (h->graphs[victim].graph) (void)hipGraphDestroy(h->graphs[victim].graph);
            h->graphs.erase(h->graphs.begin() + victim);
        }
        vitvs_handle::GraphEntry fresh;
        fresh.key = key;
        VITVS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        UpdateArgs cap = u;
        cap.late_src = nullptr;                 // (done above)
        const int rc = enqueue_update(h, cap, st);
        hipError_t e = hipStreamEndCapture(st, &fresh.graph);
        if (rc) {
            if (fresh.graph) (void)hipGraphDestroy(fresh.graph);
            return rc;
        }
        if (e != hipSuccess) return fail_hip(e, "hipStreamEndCapture", __FILE__, __LINE__);
        e = hipGraphInstantiate(&fresh.exec, fresh.graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGraphDestroy(fresh.graph);
            return fail_hip(e, "hipGraphInstantiate", __FILE__, __LINE__);
        }
        h->graphs.push_back(fresh);
        ge = &h->graphs.back();
    }
    ge->last_use = ++h->graph_clock;
    h->last_pairs = u.n_pairs; h->last_T = h->T;
    VITVS_HIP_CHECK(hipGraphLaunch(ge->exec, st));
    return 0;
}

int vitvs_set_goal_dev(vitvs_handle* h, int32_t n_goal, const uint8_t* I_des, void* stream) {
    if (!h || !I_des) return set_err(h, -1, "null argument");
    if (n_goal <= 0 || n_goal > h->cfg.max_pairs) return set_err(h, -3, "n_goal exceeds max_pairs");
    DeviceScope dev(h);
    if (vitvs_weights_ready(h) != 0) return set_err(h, -4, "weights not fully loaded: " + h->err);
    hipStream_t st = as_stream(stream);
    h->desc_keys = 0;                           // plain descriptors come out of the forward's last launch; no keys to clear
    int rc = forward_chain(h, 0, n_goal, n_goal, I_des, nullptr, h->part, st);
    h->desc_keys = -1;
    if (rc) return rc;
    h->goal_frames = n_goal;                    // (binned: the goal's token norms stay in h->sq, its token rows in h->x)
    return 0;
}

int vitvs_set_goal(vitvs_handle* h, int32_t n_goal, const uint8_t* I_des) {
    if (!h || !I_des) return set_err(h, -1, "null argument");
    if (n_goal <= 0 || n_goal > h->cfg.max_pairs) return set_err(h, -3, "n_goal exceeds max_pairs");
    DeviceScope dev(h);
    if (int rc = ensure_host_stage(h)) return rc;
    const size_t img = frame_bytes(h);
    memcpy(h->hs.des, I_des, n_goal * img);
    h->staged_des = nullptr;
    int rc = launch_copy16(h->hs.des, h->st_des, n_goal * img, h->host_stream);
    if (rc) return set_err(h, rc, "frame staging launch failed");
    rc = vitvs_set_goal_dev(h, n_goal, h->st_des, h->host_stream);
    if (rc) return rc;
    return wait_stream(h->host_stream);
}

// Validation + dispatch shared by the device-pointer and the host-pointer entry points (`u` carries the latter's late input).
static int velocity_update(vitvs_handle* h, UpdateArgs u, hipStream_t st) {
    if (!u.I_cur || !u.K || !u.v_c || !u.status) return set_err(h, -1, "null argument");   // I_des NULL: the cached goal
    if (u.n_pairs <= 0 || u.n_pairs > h->cfg.max_pairs) return set_err(h, -3, "n_pairs exceeds max_pairs");
    u.num_pairs = call_num_pairs(h, u.num_pairs);
    if (u.num_pairs > h->cfg.max_rows) return set_err(h, -5, "num_pairs exceeds max_rows");
    if (u.select_mode != VITVS_SELECT_DENSE && !u.selection) return set_err(h, -5, "selection array required for this mode");
    if (u.select_mode == VITVS_SELECT_EXPLICIT && !u.n_selected) return set_err(h, -5, "n_selected required for EXPLICIT");
    if (vitvs_weights_ready(h) != 0) return set_err(h, -4, "weights not fully loaded: " + h->err);
    // The goal cache is host-side state of the handle: it is checked and invalidated here, on every call, and never
    // inside the body that a hipGraph captures (a replay runs none of the body's host code).
    if (!u.I_des && h->goal_frames != (u.des_shared ? 1 : u.n_pairs))
        return set_err(h, -5, "I_des is NULL and no goal of this shape is cached (vitvs_set_goal_dev)");
    if (u.I_des) h->goal_frames = 0;            // the call forwards goal frames of its own over the cached rows
    h->details_pinned = false;                  // the detail block on the device is about to change
    h->host_tables = vitvs_handle::HostTables{};
    if (h->use_graphs && !h->timing && st != nullptr) return replay_update(h, u, st);
    return enqueue_update(h, u, st);
}

int vitvs_compute_velocity_dev(vitvs_handle* h, int32_t n_pairs, const uint8_t* I_cur, const uint8_t* I_des,
                               int32_t des_shared, const uint16_t* Z_mm, const double* K, int32_t select_mode,
                               const int32_t* selection, const int32_t* n_selected, int32_t num_pairs, double* v_c,
                               int32_t* status, void* stream) {
    if (!h) return set_err(h, -1, "null argument");
    DeviceScope dev(h);
    UpdateArgs u{n_pairs, des_shared, select_mode, num_pairs, I_cur, I_des, Z_mm, K, selection, n_selected, v_c, status};
    return velocity_update(h, u, as_stream(stream));
}

// The reference's seam as it is called (vitvs_v2.py:464-523, 588-632: numpy arrays in, a numpy twist out).  Per call: the
// frames, intrinsics and selection are copied into the handle's pinned block by memcpy; the frames go on to device memory
// in one short launch on the update's stream; the forward is enqueued; THEN the depth image is copied (host) — the law's
// kernel is the only reader, it is enqueued last and reads the <= max_rows pixels it needs in place; v_c, status and the
// feature rows come back through the pinned block; one polled wait.
int vitvs_compute_velocity(vitvs_handle* h, int32_t n_pairs, const uint8_t* I_cur, const uint8_t* I_des,
                           int32_t des_shared, const uint16_t* Z_mm, const double* K, int32_t select_mode,
                           const int32_t* selection, const int32_t* n_selected, int32_t num_pairs, double* v_c,
                           int32_t* status) {
    if (!h || !I_cur || !K || !v_c || !status) return set_err(h, -1, "null argument");   // I_des NULL: the cached goal
    if (n_pairs <= 0 || n_pairs > h->cfg.max_pairs) return set_err(h, -3, "n_pairs exceeds max_pairs");
    const vitvs_config& c = h->cfg;
    const int np = call_num_pairs(h, num_pairs);
    if (np > c.max_rows) return set_err(h, -5, "num_pairs exceeds max_rows");
    if (select_mode == VITVS_SELECT_EXPLICIT && (!selection || !n_selected)) return set_err(h, -5, "EXPLICIT selection needs ids and counts");
    if (select_mode == VITVS_SELECT_ORDER && !selection) return set_err(h, -5, "ORDER selection needs a visiting order");
    DeviceScope dev(h);
    if (int rc = ensure_host_stage(h)) return rc;
    vitvs_handle::HostStage& hs = h->hs;
    hipStream_t st = h->host_stream;
    const size_t img = frame_bytes(h), n_des = des_shared ? 1 : n_pairs;
    memcpy(hs.cur, I_cur, n_pairs * img);
    int rc = launch_copy16(hs.cur, h->st_cur, n_pairs * img, st);
    // option "reuse_goal_frames": a control loop's goal image does not change — while the caller passes the same address (and
    // count, and geometry) the goal frames staged by the previous call are still in device memory and are forwarded again as they are
    const bool goal_staged = h->reuse_goal && I_des && I_des == h->staged_des && n_des * img == h->staged_des_bytes;
    if (!rc && I_des && !goal_staged) {
        memcpy(hs.des, I_des, n_des * img);
        rc = launch_copy16(hs.des, h->st_des, n_des * img, st);
        h->staged_des = I_des;
        h->staged_des_bytes = n_des * img;
    }
    if (rc) return set_err(h, rc, "frame staging launch failed");
    memcpy(hs.K, K, (size_t)n_pairs * 4 * sizeof(double));
    if (select_mode == VITVS_SELECT_EXPLICIT) {
        memcpy(hs.sel, selection, (size_t)n_pairs * np * 4);
        memcpy(hs.nsel, n_selected, (size_t)n_pairs * 4);
    } else if (select_mode == VITVS_SELECT_ORDER) {
        memcpy(hs.sel, selection, (size_t)n_pairs * h->T * 4);
    }
    UpdateArgs u{n_pairs, des_shared, select_mode, np, h->st_cur, I_des ? h->st_des : nullptr, Z_mm ? hs.depth : nullptr, hs.K,
                 hs.sel, hs.nsel, hs.vc, hs.status};
    if (Z_mm) {
        u.late_src = Z_mm; u.late_dst = hs.depth; u.late_sites = h->depth_sites.data(); u.late_count = (int)h->depth_sites.size();
        u.late_pairs = n_pairs; u.late_stride = (size_t)c.u_max * c.v_max;
    }
    rc = velocity_update(h, u, st);
    if (rc) return rc;
    // the detail block of this call -> the pinned block, one copy launch of 16-byte stores (measured: the law's kernel writing
    // its ~20 small detail stores straight into host memory cost 60 us per update; one coalesced copy costs 3)
    rc = launch_copy16(h->det_block, hs.det, h->det_bytes, st);
    if (rc) return set_err(h, rc, "detail copy launch failed");
    if (int w = wait_stream(st)) return w;
    memcpy(v_c, hs.vc, (size_t)n_pairs * 6 * sizeof(double));
    memcpy(status, hs.status, (size_t)n_pairs * 4);
    h->details_pinned = true;
    h->host_tables = vitvs_handle::HostTables{n_pairs, h->T, Z_mm != nullptr};
    return 0;
}

// The reference draws its feature tokens on the HOST, between the correspondence and the law (find_correspondences_batch:
// sort + torch.randperm, vitvs_v2.py:127-141): a host-pointer call gives the tables (vitvs_last_details serves nn_1 / nn_2 /
// sim_1 of a host-pointer call from host memory), the caller draws, and this entry point runs the law again for that draw on
// what the call left in the handle — the arg-max keys on the device, the depth image and intrinsics in the pinned block: one
// short launch, no forward, no staging.
int vitvs_reselect(vitvs_handle* h, int32_t select_mode, const int32_t* selection, const int32_t* n_selected, int32_t num_pairs,
                   double* v_c, int32_t* status) {
    if (!h || !v_c || !status) return set_err(h, -1, "null argument");
    if (!h->details_pinned || h->host_tables.n_pairs <= 0 || !h->hs.base)
        return set_err(h, -5, "vitvs_reselect follows a host-pointer velocity call on the same handle (vitvs_compute_velocity)");
    const int n_pairs = h->host_tables.n_pairs, T = h->host_tables.T;
    const int np = call_num_pairs(h, num_pairs);
    if (np > h->cfg.max_rows) return set_err(h, -5, "num_pairs exceeds max_rows");
    if (select_mode == VITVS_SELECT_EXPLICIT && (!selection || !n_selected)) return set_err(h, -5, "EXPLICIT selection needs ids and counts");
    if (select_mode == VITVS_SELECT_ORDER && !selection) return set_err(h, -5, "ORDER selection needs a visiting order");
    DeviceScope dev(h);
    vitvs_handle::HostStage& hs = h->hs;
    if (select_mode == VITVS_SELECT_EXPLICIT) {
        memcpy(hs.sel, selection, (size_t)n_pairs * np * 4);
        memcpy(hs.nsel, n_selected, (size_t)n_pairs * 4);
    } else if (select_mode == VITVS_SELECT_ORDER) {
        memcpy(hs.sel, selection, (size_t)n_pairs * T * 4);
    }
    int rc = run_servo(h, n_pairs, T, h->host_tables.have_depth ? hs.depth : nullptr, hs.K, select_mode, np, hs.sel, hs.nsel, hs.vc,
                       hs.status, h->host_stream);
    if (rc) return rc;
    rc = launch_copy16(h->det_block, hs.det, h->det_bytes, h->host_stream);
    if (rc) return set_err(h, rc, "detail copy launch failed");
    if (int w = wait_stream(h->host_stream)) return w;
    memcpy(v_c, hs.vc, (size_t)n_pairs * 6 * sizeof(double));
    memcpy(status, hs.status, (size_t)n_pairs * 4);
    return 0;
}

int vitvs_last_details(vitvs_handle* h, int32_t n_pairs, int32_t* nn_1, int32_t* nn_2, float* sim_1, int32_t* info,
                       int32_t* selected, int32_t* s_uv, double* feat, double* L) {
    if (!h) return set_err(h, -1, "null argument");
    if (n_pairs <= 0 || n_pairs > h->last_pairs) return set_err(h, -3, "no such pairs in the last call");
    DeviceScope dev(h);
    const size_t T = h->last_T, R = h->cfg.max_rows, P = n_pairs, PM = h->cfg.max_pairs;
    // after a host-pointer call the feature rows are already in host memory (the handle's pinned block): a caller that asks
    // for those alone (the reference's detect_features return value: s_uv*, s_uv, the selected similarities) costs no HIP call
    const bool pinned = h->details_pinned;
    const unsigned char* pd = h->hs.det;
    if (!pinned || selected || L) VITVS_HIP_CHECK(hipDeviceSynchronize());
    const size_t o_nn = PM * 32 + PM * R * 48;              // nn_1 | nn_2 | sim_1 behind info | s_uv | feat (detail_pointers)
    if (nn_1) {
        if (pinned) memcpy(nn_1, pd + o_nn, P * T * 4);
        else VITVS_HIP_CHECK(hipMemcpy(nn_1, h->nn1, P * T * 4, hipMemcpyDeviceToHost));
    }
    if (nn_2) {
        if (pinned) memcpy(nn_2, pd + o_nn + h->best_elems * 4, P * T * 4);
        else VITVS_HIP_CHECK(hipMemcpy(nn_2, h->nn2, P * T * 4, hipMemcpyDeviceToHost));
    }
    if (sim_1) {
        if (pinned) memcpy(sim_1, pd + o_nn + 2 * h->best_elems * 4, P * T * 4);
        else VITVS_HIP_CHECK(hipMemcpy(sim_1, h->sim1, P * T * 4, hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> inf(P * 8);
    if (pinned) memcpy(inf.data(), pd, P * 8 * 4);
    else VITVS_HIP_CHECK(hipMemcpy(inf.data(), h->info, P * 8 * 4, hipMemcpyDeviceToHost));
    if (info) memcpy(info, inf.data(), P * 8 * 4);
    if (selected) VITVS_HIP_CHECK(hipMemcpy(selected, h->sel_out, P * R * 4, hipMemcpyDeviceToHost));
    if (s_uv) {
        if (pinned) memcpy(s_uv, pd + PM * 32, P * R * 16);
        else VITVS_HIP_CHECK(hipMemcpy(s_uv, h->s_uv, P * R * 4 * 4, hipMemcpyDeviceToHost));
    }
    if (feat) {
        if (pinned) memcpy(feat, pd + PM * 32 + PM * R * 16, P * R * 32);
        else VITVS_HIP_CHECK(hipMemcpy(feat, h->feat, P * R * 4 * 8, hipMemcpyDeviceToHost));
    }
    if (L) VITVS_HIP_CHECK(hipMemcpy(L, h->Lws, P * 7 * 2 * R * 8, hipMemcpyDeviceToHost));
    // The kernel writes the first n_feature_rows (info[1]) rows of a pair; the workspace rows behind them may still hold
    // an earlier, larger call's values.  The copies handed out are defined everywhere: selected = -1, everything else 0.
    for (size_t b = 0; b < P; ++b) {
        const size_t n = std::min<size_t>(R, (size_t)std::max(inf[b * 8 + 1], 0));
        for (size_t k = n; k < R; ++k) {
            if (selected) selected[b * R + k] = -1;
            if (s_uv) memset(s_uv + (b * R + k) * 4, 0, 16);
            if (feat) memset(feat + (b * R + k) * 4, 0, 32);
        }
        if (L)
            for (size_t c = 0; c < 7; ++c) memset(L + (b * 7 + c) * 2 * R + 2 * n, 0, (2 * R - 2 * n) * 8);
    }
    return 0;
}

int vitvs_set_option(vitvs_handle* h, const char* name, int64_t value) {
    if (!h || !name) return set_err(h, -1, "null argument");
    const std::string nm(name);
    if (nm == "graph_replay") {
        if (value != 0 && value != 1) return set_err(h, -5, "graph_replay takes 0 or 1");
        h->use_graphs = value == 1;
        return 0;
    }
    if (nm == "reuse_goal_frames") {
        if (value != 0 && value != 1) return set_err(h, -5, "reuse_goal_frames takes 0 or 1");
        h->reuse_goal = value == 1;
        h->staged_des = nullptr;
        return 0;
    }
    if (nm == "in_flight") {
        if (value < 1 || value > 64) return set_err(h, -5, "in_flight takes 1 .. 64");
        if ((int)value != h->in_flight) {       // captured updates hold the previous plan's launches
            DeviceScope dev(h);
            VITVS_HIP_CHECK(hipDeviceSynchronize());
            drop_graphs(h);
            h->in_flight = (int)value;
        }
        return 0;
    }
    return set_err(h, -5, "unknown option " + nm);
}

int vitvs_timing_enable(vitvs_handle* h, int32_t on) {
    if (!h) return set_err(h, -1, "null argument");
    DeviceScope dev(h);
    VITVS_HIP_CHECK(hipDeviceSynchronize());
    h->timing = on != 0;
    h->ev_used = 0;
    h->ev_class.clear();
    return 0;
}

int vitvs_timing_classes(void) { return KC_COUNT; }

const char* vitvs_timing_class_name(int32_t cls) { return (cls >= 0 && cls < KC_COUNT) ? kClassNames[cls] : ""; }

int vitvs_timing_collect(vitvs_handle* h, int32_t n_classes, double* total_ms, int32_t* launches) {
    if (!h || !total_ms || !launches || n_classes < KC_COUNT) return set_err(h, -1, "bad argument");
    DeviceScope dev(h);
    VITVS_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < n_classes; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
    for (size_t i = 0; i < h->ev_class.size(); ++i) {
        float ms = 0.f;
        VITVS_HIP_CHECK(hipEventElapsedTime(&ms, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
        total_ms[h->ev_class[i]] += ms;
        launches[h->ev_class[i]] += 1;
    }
    h->ev_used = 0;
    h->ev_class.clear();
    return 0;
}

// ---- include/vitvs_ops.h: single-operator entry points for the kernel-level parity tests ----
int vitvs_op_linear(int32_t precision, const void* A, const void* W, const float* bias, void* out, int32_t M,
                    int32_t N, int32_t K, int32_t gelu, void* stream) {
    DeviceScope dev(nullptr);
    return launch_linear(to_prec(precision), A, W, bias, out, M, N, K, gelu, as_stream(stream), g_op_wexp);
}
int vitvs_op_weight_exponent(int32_t e) {
    const int prev = g_op_wexp;
    if (e >= 0 && e <= 31) g_op_wexp = e;
    return prev;
}
int vitvs_op_linear_variant(int32_t precision, int32_t variant, const void* A, const void* W, const float* bias, void* out,
                            int32_t M, int32_t N, int32_t K, int32_t gelu, int32_t slices, void* stream) {
    DeviceScope dev(nullptr);
    const Precision p = to_prec(precision);
    hipStream_t st = as_stream(stream);
    if (variant == 0)
        return slices > 0 ? launch_linear_partial(p, A, W, (float*)out, M, N, K, slices, st, g_op_wexp)
                          : launch_linear(p, A, W, bias, out, M, N, K, gelu, st, g_op_wexp);
    if (variant == 1)
        return slices > 0 ? launch_linear_partial_classic(p, A, W, (float*)out, M, N, K, slices, st, g_op_wexp)
                          : launch_linear_classic(p, A, W, bias, out, M, N, K, gelu, st, g_op_wexp);
    if (variant == 2) return p == PREC_X2 ? -2 : launch_linear_128(p, A, W, bias, out, M, N, K, gelu, slices > 0 ? slices : 1, slices > 0, st);
    const int kk = (p == PREC_X2 ? 2 : 1) * K;     // elements per row of the big kernels' operands
    if (variant == 1256) return (N % 256 || kk % 64) ? -2 : launch_linear_big(p, 1256, A, W, bias, out, M, N, K, slices > 0 ? slices : 1, gelu, slices > 0, st, g_op_wexp);
    if (variant == 1192) return (N % 128 || kk % 64) ? -2 : launch_linear_big(p, 1192, A, W, bias, out, M, N, K, slices > 0 ? slices : 1, gelu, slices > 0, st, g_op_wexp);
    if ((variant != 256 && variant != 192 && variant != 128) || N % variant != 0 || kk % 64 != 0) return -2;
    return launch_linear_big(p, variant, A, W, bias, out, M, N, K, slices > 0 ? slices : 1, gelu, slices > 0, st, g_op_wexp);
}
int vitvs_op_linear_residual(int32_t precision, const void* A, const void* W, const float* bias, const float* ls,
                             float* x, int32_t M, int32_t N, int32_t K, void* stream) {
    DeviceScope dev(nullptr);
    return launch_linear_residual(to_prec(precision), A, W, bias, ls, x, M, N, K, as_stream(stream), g_op_wexp);
}
int vitvs_op_layernorm(int32_t precision, const float* x, const float* gamma, const float* beta, void* out, int32_t M,
                       int32_t D, float eps, void* stream) {
    DeviceScope dev(nullptr);
    return launch_layernorm(to_prec(precision), x, gamma, beta, out, M, D, eps, as_stream(stream));
}
int vitvs_op_attention(int32_t precision, const void* qkv, void* out, int32_t n_img, int32_t N, int32_t H,
                       void* stream) {
    DeviceScope dev(nullptr);
    return launch_attention(to_prec(precision), qkv, out, n_img, N, H, as_stream(stream));
}
int vitvs_op_attention_q(int32_t precision, const void* qkv, void* out, int32_t n_img, int32_t N, int32_t H,
                         int32_t q_prescaled, void* stream) {
    DeviceScope dev(nullptr);
    const Precision p = to_prec(precision);
    return launch_attention(p, qkv, out, n_img, N, H, as_stream(stream), nullptr, q_prescaled != 0 && plain16(p));
}
int vitvs_op_linear_tile(int32_t precision, int32_t M, int32_t N, int32_t K, int32_t slices, int32_t* tile) {
    if (!tile) return -1;
    int t[3] = {0, 0, 0};
    const int rc = linear_tile_plan(to_prec(precision), M, N, K, slices > 0 ? slices : 1, slices > 0, t);
    tile[0] = t[0]; tile[1] = t[1]; tile[2] = t[2];
    return rc;
}
int vitvs_op_touch(const void* p, int64_t bytes, int32_t share_xcds, void* stream) {
    DeviceScope dev(nullptr);
    return launch_touch(p, (size_t)bytes, share_xcds, as_stream(stream));
}
int vitvs_op_plan_in_flight(int32_t n) {
    const int prev = g_updates_in_flight;
    if (n >= 1) g_updates_in_flight = n;
    return prev;
}
int vitvs_op_splitk_slices(int32_t precision, int32_t M, int32_t N, int32_t K) {
    return splitk_slices(to_prec(precision), M, N, K);
}
int vitvs_op_linear_partial(int32_t precision, const void* A, const void* W, float* part, int32_t M, int32_t N,
                            int32_t K, int32_t slices, void* stream) {
    DeviceScope dev(nullptr);
    return launch_linear_partial(to_prec(precision), A, W, part, M, N, K, slices, as_stream(stream), g_op_wexp);
}
int vitvs_op_residual_ln(int32_t precision, float* x, const float* part, int32_t slices, const float* bias,
                         const float* ls, const float* gamma, const float* beta, void* out, int32_t M, int32_t D,
                         float eps, void* stream) {
    DeviceScope dev(nullptr);
    return launch_residual_ln(to_prec(precision), x, part, slices, bias, ls, gamma, beta, out, M, D,
                              eps, as_stream(stream));
}

}  // extern "C"
