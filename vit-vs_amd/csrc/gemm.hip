// Linear layers of the ViT forward on the gfx950 matrix cores (see gemm_core.h for the tile loop).
// Reference arithmetic being replaced (stock PyTorch ops issued by the reference):
//   qkv / proj Linear     dino_patch/attention.py:72, 79
//   fc1 -> GELU -> fc2     dino_patch/block.py:78-84 (Mlp, nn.GELU = erf form)
//   residual + LayerScale  dino_patch/block.py:90-96, 112-115
//   patch-embed Conv2d     dinov2_extractor.py:141, 259 (k = p, stride) as an im2col GEMM
#include <stdlib.h>

#include "gemm_core.h"
#include "kernels.h"
#include "probe.h"

namespace vitvs {

template <typename T, bool WT>
__device__ __forceinline__ void store4(T* dst, f32x4 v) {
    static_assert(!kSplit<T>, "f16x2 rows are addressed by logical column: store_x2");
    if constexpr (sizeof(T) == 4) {
        store_out<WT>(reinterpret_cast<float*>(dst), v);
    } else {
        const typename Vec16<T>::x4 h = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        store_out<WT>(dst, h);
    }
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// erf to ~1.5e-7 absolute (Abramowitz & Stegun 7.1.26): two orders below bf16 output rounding, a
// fraction of libm erff's instruction count.  Used only when the layer's output is bf16.
__device__ __forceinline__ float gelu_erf_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-x * x);
    return 0.5f * v * (1.0f + copysignf(e, v));
}

// Every epilogue is built inside the kernel from the same flat argument list (out, c0, c1, M, N, i0):
// flat leading kernel arguments are preloaded into SGPRs by the command processor
// (-amdgpu-kernarg-preload-count), a by-value struct would be fetched by the wave after it starts.
template <typename T>
struct EpiStore {
    T* out;
    const float* bias;
    int ldo;
    int gelu;
    __device__ __forceinline__ void set_slice(int) {}
    __device__ __forceinline__ static EpiStore make(void* out, const float* c0, const float*, int, int N, int i0) {
        return EpiStore{(T*)out, c0, N, i0};
    }
    __device__ __forceinline__ float4 column_terms(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    // staged epilogue (linear_kernel): the tile leaves through an LDS image as whole rows
    using Out = T;
    static constexpr bool STAGED = true;
    // the fast erf is two orders below the output rounding of the plain 16-bit types only; f16x2 keeps fp32-class outputs
    static constexpr bool FAST_GELU = sizeof(T) == 2 && !kSplit<T>;
    static constexpr int OUT_BYTES = kSplit<T> ? 4 : (int)sizeof(T);     // bytes per logical output column
    __device__ __forceinline__ f32x4 value(f32x4 v, float4 b) const {
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        if (gelu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = FAST_GELU ? gelu_erf_fast(v[i]) : gelu_erf(v[i]);
        }
        return v;
    }
    // first byte of logical column n0 (n0 % 32 == 0) of output row m
    __device__ __forceinline__ unsigned char* row_bytes(int m, int n0) const {
        return reinterpret_cast<unsigned char*>(out) + ((size_t)m * ldo + n0) * OUT_BYTES;
    }
    template <bool WT>
    __device__ __forceinline__ void store(int m, int n, f32x4 v, float4 b) const {
        v = value(v, b);
        if constexpr (kSplit<T>) store_x2<WT>(out + (size_t)m * ldo * 2, n, v);
        else store4<T, WT>(out + (size_t)m * ldo + n, v);
    }
};

struct EpiResidual {
    static constexpr bool STAGED = false;
    float* x;
    const float* bias;
    const float* ls;  // may be null
    int ld;
    __device__ __forceinline__ void set_slice(int) {}
    __device__ __forceinline__ static EpiResidual make(void* out, const float* c0, const float* c1, int, int N, int) {
        return EpiResidual{(float*)out, c0, c1, N};
    }
    __device__ __forceinline__ float4 column_terms(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    template <bool WT>
    __device__ __forceinline__ void store(int m, int n, f32x4 v, float4 b) const {
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        if (ls) {
            const float4 g = *reinterpret_cast<const float4*>(ls + n);
            v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
        }
        float4* p = reinterpret_cast<float4*>(x + (size_t)m * ld + n);
        float4 r = *p;
        r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
        *p = r;
    }
};

struct EpiPatch {
    static constexpr bool STAGED = false;
    float* x;
    const float* bias;
    const float* pos;
    int T, D;
    __device__ __forceinline__ void set_slice(int) {}
    __device__ __forceinline__ static EpiPatch make(void* out, const float* c0, const float* c1, int, int N, int i0) {
        return EpiPatch{(float*)out, c0, c1, i0, N};
    }
    __device__ __forceinline__ float4 column_terms(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
    template <bool WT>
    __device__ __forceinline__ void store(int m, int n, f32x4 v, float4 b) const {
        const int img = m / T, t = m - img * T;
        const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)(1 + t) * D + n);
        float4 r = make_float4(v[0] + b.x + pe.x, v[1] + b.y + pe.y, v[2] + b.z + pe.z, v[3] + b.w + pe.w);
        *reinterpret_cast<float4*>(x + ((size_t)img * (T + 1) + 1 + t) * D + n) = r;
    }
};

// Raw fp32 partial sums of one K slice (split-K): part[z][m][n]; bias / LayerScale / residual /
// LayerNorm are applied by residual_ln_kernel (elementwise.hip), which sums the slices in a fixed
// order, so the result does not depend on scheduling (no atomics).
struct EpiPartial {
    float* part;
    int M, N;
    __device__ __forceinline__ static EpiPartial make(void* out, const float*, const float*, int M, int N, int) {
        return EpiPartial{(float*)out, M, N};
    }
    __device__ __forceinline__ float4 column_terms(int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void set_slice(int z) { part += (size_t)z * M * N; }
    using Out = float;
    static constexpr bool STAGED = true;
    static constexpr int OUT_BYTES = 4;
    __device__ __forceinline__ f32x4 value(f32x4 v, float4) const { return v; }
    __device__ __forceinline__ unsigned char* row_bytes(int m, int n0) const {
        return reinterpret_cast<unsigned char*>(part + (size_t)m * N + n0);
    }
    template <bool WT>
    __device__ __forceinline__ void store(int m, int n, f32x4 v, float4) const {
        store4<float, WT>(part + (size_t)m * N + n, v);
    }
};

#ifdef VITVS_PROBE
// probe builds only (tools/gemm_probe.cpp): per-wave cycle-counter stamps of the kernel's phases (probe.h)
__device__ unsigned long long* g_gemm_probe;
extern "C" __attribute__((visibility("default"))) int vitvs_debug_set_gemm_probe(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_probe), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

#define VITVS_EPI_WT(BM) ((BM) < 128)
template <typename T, int BM, int BN, int KG, class Epi, int NS = 0>
__global__ __launch_bounds__(256 * KG) void linear_kernel(const T* __restrict__ A, const T* __restrict__ W, void* out,
                                                          const float* c0, const float* c1, int M, int N, int K,
                                                          int ks_i0) {
    // ks_i0 packs [31:24] K per split-K slice in k-tiles of 32 (fp32) / 64 (two-byte types; f16x2: 32 logical k) elements,
    // [23:19] e: f16x2 weights arrive multiplied by 2^e (api.hip upload_matrix), the sums leave multiplied by 2^-e,
    // [18] staged epilogue, [16] XCD map, [15:0] i0: one preloaded dword instead of gridDim (hidden kernel arguments the
    // wave would have to fetch) and an integer division.
    VITVS_IF_PROBE(unsigned long long ts_buf[8]; ts_buf[0] = __builtin_readcyclecounter();)
    unsigned long long* const ts = VITVS_PROBE_OR_NULL(ts_buf);   // per-phase stamps: null in the product build (folded away)
    using Tile = GemmTile<BM, BN, KG, NS>;
    Epi epi = Epi::make(out, c0, c1, M, N, ks_i0 & 0xffff);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned pk = (unsigned)ks_i0;
    const int kslice = (int)(pk >> 24) * (128 / (int)sizeof(T));
    int tx = blockIdx.x, ty = blockIdx.y, tz = blockIdx.z;
    if (pk & (1u << 16)) {
        // XCD map of the two-slice partial-sum launches (launch_one): a 1-D grid whose workgroup L runs on XCD L % 8; XCD x
        // takes K slice x / 4 and the column tiles of class x % 4, all row tiles of them.  A weight tile is then fetched by
        // ONE private L2 (every row tile that needs it is there) and an activation slice by four instead of eight.
        const int L = blockIdx.x, x = L & 7, r = L >> 3, ncq = (N / BN) >> 2;
        ty = r / ncq;
        tx = (x & 3) + 4 * (r - ty * ncq);
        tz = x >> 2;
    }
    epi.set_slice(tz);
    const int m0 = ty * BM, n0 = tx * BN;
    f32x4 acc[Tile::NT][Tile::MT];
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int kg = (KG == 2) ? k_group() : 0;
    // per-column epilogue terms (bias) are requested BEFORE the main loop: they are older than every tile
    // copy, so the counted waits are unaffected, and their memory round trip is off the epilogue's path
    float4 col[Tile::NT];
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni) col[ni] = epi.column_terms(min(n0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4), N - 4));
    gemm_mainloop<T, BM, BN, KG, NS>(A, W, K, K, M, N, m0, n0, tz * kslice, (tz + 1) * kslice, smem, acc, ts);
    VITVS_IF_PROBE(
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[Tile::NT - 1][Tile::MT - 1][3]) : "memory");
        ts[4] = __builtin_readcyclecounter();
    )
    if constexpr (kSplit<T>) {
        const float ws = __uint_as_float((127u - ((pk >> 19) & 31u)) << 23);   // 2^-e, exact
#pragma unroll
        for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] *= ws;
    }
    if constexpr (BM == 64 && Epi::STAGED) {
        if (pk & (1u << 18)) {
            // Staged epilogue: a lane of the MFMA layout owns 4 consecutive columns of one row, so a wave's store instruction
            // writes 16 rows x 32 bytes (16-bit output) — four partial lines per row over the tile.  Here the tile goes through an
            // LDS image (the ring is free: every wave is past its last k-tile) and leaves as 16-byte pieces of whole rows:
            // consecutive lanes write consecutive bytes, whole 128-byte lines.
            using Out = typename Epi::Out;
            constexpr int ROWB = BN * Epi::OUT_BYTES, PITCH = ROWB + 16, CH = ROWB / 16;
            // one k-group: the barrier ends the last k-tile's reads.  Two k-groups: the ring has been free since the barrier of
            // the accumulator swap (gemm_mainloop) unless the swap area itself lives in the ring; each group stages the tiles it owns.
            if constexpr (KG == 1 || Tile::SWAP_ALIAS) __syncthreads();
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < Tile::MT; ++mi) {
                    if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) continue;
                    const f32x4 v = epi.value(acc[ni][mi], col[ni]);
                    const int cn = wn * Tile::WN + ni * 16 + 4 * (lane >> 4);      // tile-local column of v[0]
                    unsigned char* dst = smem + (wm * Tile::WM + mi * 16 + (lane & 15)) * PITCH +
                                         (kSplit<Out> ? 2 * x2_index(cn) : cn * (int)sizeof(Out));
                    if constexpr (kSplit<Out>) {       // the image has the row layout of memory: [hi of 32 columns | lo of them]
                        const Split4 h = split4(v);
                        *reinterpret_cast<f16x4*>(dst) = h.hi;
                        *reinterpret_cast<f16x4*>(dst + 64) = h.lo;
                    } else if constexpr (sizeof(Out) == 4) *reinterpret_cast<f32x4*>(dst) = v;
                    else {
                        const typename Vec16<Out>::x4 h = {(Out)v[0], (Out)v[1], (Out)v[2], (Out)v[3]};
                        *reinterpret_cast<typename Vec16<Out>::x4*>(dst) = h;
                    }
                }
            __syncthreads();
#pragma unroll
            for (int idx = threadIdx.x; idx < BM * CH; idx += 256 * KG) {
                const int r = idx / CH, ch = idx - r * CH;
                if (m0 + r < M) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(smem + r * PITCH + ch * 16);
                    store_out16<true>(epi.row_bytes(m0 + r, n0) + ch * 16, d);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni) {
        const int n = n0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) {
            if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) continue;
            const int m = m0 + wm * Tile::WM + mi * 16 + (lane & 15);
            if (m < M && n < N) epi.template store<VITVS_EPI_WT(BM)>(m, n, acc[ni][mi], col[ni]);   // 64-row tiles: one-wave launches
        }
    }
    VITVS_IF_PROBE(
        ts[5] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[6] = __builtin_readcyclecounter();
        if (lane == 0 && g_gemm_probe) {
            const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            unsigned long long* dst = g_gemm_probe + ((size_t)wg * 8 + (threadIdx.x >> 6)) * 8;
            for (int i = 0; i < 7; ++i) dst[i] = ts[i];
        }
    )
}

static int k_tile(Precision p) { return plain16(p) ? 64 : 32; }   // logical k per 128-byte k-tile (f16x2: 32, hi and lo halves)

static bool shapes_ok(Precision p, int M, int N, int K) {
    const long long es = (long long)elem_size(p);   // operands are addressed with 32-bit byte offsets
    return M > 0 && N > 0 && K > 0 && (K % k_tile(p)) == 0 && (N % 64) == 0 && (long long)M * K * es < (1ll << 32) &&
           (long long)N * K * es < (1ll << 32);
}

// Tile plan.  In the one-frame-pair regime a launch cannot fill the chip, and a second wave of
// workgroups costs more than bigger tiles save (measured: 336 workgroups of 64x64 ran 2.4 us longer than
// 252), so: the column-tile width with the MOST workgroups that still fit the 256 CUs; if even the
// widest tile needs more than 256 workgroups the problem is large and the widest tile wins.  Two
// k-groups (8 waves, intra-workgroup split-K) whenever one workgroup per CU is all there is.
struct TilePlan {
    int bn, kg;
};
static TilePlan plan_tiles(int M, int N, int nk_per_slice, int splits, bool fixed64) {
    const long mt = (M + 63) / 64;
    int bn = 64;
    if (!fixed64) {
        long best = -1;
        for (int c : {128, 96, 64}) {
            if (N % c) continue;
            const long wgs = mt * (N / c);
            if (wgs <= 256 && wgs > best) { best = wgs; bn = c; }
        }
        if (best < 0) bn = (N % 128 == 0) ? 128 : 64;
        // (Measured in round 5 and not kept: beside other queues' launches a wider column tile for these one-round launches — fewer
        //  re-reads of the activation rows, fewer and fatter workgroups — 96 / 128 columns wherever they divide: bf16 3942 -> 3969 /
        //  3748 updates/s with three in flight, f16x2 2402 -> 2311 / 2067; again with four in flight and nothing on a fifth queue: bf16
        //  4818-4841 -> 4828-4846 / 4452-4457, f16x2 2593-2636 -> 2630-2658 / 2357: 96 inside the noise, 128 -8 %.)
    }
    const long wgs = mt * (N / bn) * splits;
    // Alone on the chip a one-round launch wants its serial k-loop short: two k-groups (8 waves, 112-144 KB of LDS, one
    // workgroup per CU).  Beside the launches of other queues (several updates in flight, vitvs_set_option) that footprint
    // keeps a second launch's workgroups off the CU until the first has left; 4-wave workgroups (64-80 KB) let two launches
    // share it: ViT-B/16 224², 3 updates in flight 3113 -> 3392 updates/s, but 2220 -> 2078 on one stream (same box,
    // profiles/r03_notes.md section 5), hence by the caller's hint and not by default.
    int kg = (g_updates_in_flight < 2 && wgs <= 256 && nk_per_slice >= 4 && nk_per_slice % 2 == 0) ? 2 : 1;
    return TilePlan{bn, kg};
}

struct EpiArgs {   // host image of the flat epilogue arguments
    void* out;
    const float* c0;
    const float* c1;
    int i0;
    int wexp = 0;   // f16x2: the weights carry 2^wexp (linear_kernel)
};

// The XCD map (linear_kernel) is for the partial-sum launches only (their i0 field is unused, bit 16 of the packed argument is free)
template <class Epi> constexpr bool xcd_mapped() { return false; }
template <> constexpr bool xcd_mapped<EpiPartial>() { return true; }
static bool want_xcd_map() {
    // Beside other queues' launches the narrow layers are bound by what their private L2s fetch, not by latency (fc2 at three
    // queues 4.48 -> 3.67 us per launch, 3932 -> 4072 updates/s); alone, XCD balance comes first (round 1: the map lost).
    // (A column-class map for the one-slice launches — qkv's 36 column tiles over 8 XCDs — lost: 3.13 -> 3.37 us.)
    return g_updates_in_flight >= 2;
}

static bool want_staged_epilogue(int kg) {
    // 4-wave workgroups: measured faster alone and side by side (qkv 6.00 -> 5.11 us alone, 3.07 -> 2.81 at three queues; fc1
    // 8.62 -> 7.16, 3.99 -> 3.39; 2 pairs +3 % on one stream and +10 % with three updates in flight; never slower)
    // 8-wave workgroups (two k-groups, the one-stream plan): +1.4 % updates/s on one stream (2202 -> 2233, three interleaved rounds)
    return true;
}

template <typename T, int BN, int KG, class Epi, int BM = 64, int NS = 0>
static int launch_one(const T* A, const T* W, int M, int N, int K, const EpiArgs& e, hipStream_t stream, int splits) {
    using Tile = GemmTile<BM, BN, KG, NS>;
    static std::atomic<unsigned long long> raised{0};   // > 64 KiB of dynamic LDS needs the opt-in attribute, per device
    if (raise_lds_limit(reinterpret_cast<const void*>(&linear_kernel<T, BM, BN, KG, Epi, NS>), Tile::LDS_BYTES, raised)) return -1;
    dim3 grid(N / BN, (M + BM - 1) / BM, splits);
    constexpr int KU = 128 / (int)sizeof(T);   // elements per k-tile (f16x2: K counts fp16, two per logical k)
    const int kslice = K / splits;
    if (kslice % KU != 0 || kslice / KU > 255 || splits > 15 || e.i0 < 0 || e.i0 > 0xffff || e.wexp < 0 || e.wexp > 31) return -2;
    unsigned xcd_map = 0;
    if (xcd_mapped<Epi>() && splits == 2 && (N / BN) % 4 == 0 && want_xcd_map()) {
        grid = dim3(grid.x * grid.y * 2, 1, 1);
        xcd_map = 1u << 16;
    }
    if (BM == 64 && want_staged_epilogue(KG)) xcd_map |= 1u << 18;
    launch(linear_kernel<T, BM, BN, KG, Epi, NS>, grid, dim3(Tile::THREADS), Tile::LDS_BYTES, stream, A, W, e.out, e.c0, e.c1,
           M, N, K, (int)(((unsigned)(kslice / KU) << 24) | ((unsigned)e.wexp << 19) | xcd_map | (unsigned)e.i0));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Ring depth of a one-k-group 64-row launch.  Up to one workgroup per CU the deep ring (4 stages) hides the operand latency
// of the only workgroup there is.  Beyond that the LDS footprint sets how many workgroups a CU holds at once, and the other
// resident workgroups hide the latency instead (bf16, us with 4 / 3 / 2 stages; weights rotating through 19 MB | 600 MB):
//   64 x 64  (64 / 48 / 32 KB: 2 / 3 / 5 per CU)  1576 x 768 x 768, 300 workgroups:  6.1 /  6.0 /  7.6 |  7.1 /  7.3 / 10.7
//                                                 3152 x 768 x 768, 600:            12.0 / 11.5 / 10.3 | 12.1 / 11.6 / 12.0
//                                                 2740 x 1024 x 1024, 688:          14.8 / 13.1 / 12.4 | 15.4 / 14.7 / 15.4
//   64 x 128 (96 / 72 / 48 KB: 1 / 2 / 3 per CU)   788 x 3072 x 768, 312:           20.3 / 12.2 / 11.4 | 20.7 / 13.0 / 15.6
//                                                  985 x 2304 x 768, 288:           18.4 / 11.1 / 11.3 | 18.8 / 11.9 / 13.7
// In the forward (kernel trace, same box): 8-pair proj 11.40 / 11.47 / 11.31 us, 2-pair fc1 19.8 / 13.5 / 12.2 us; end to end
// against 4 stages everywhere: 8 pairs +0.9 ... +1.3 % with 3 or 2 stages, 2 pairs +1.6 % (profiles/r03_notes.md).
static int ring_stages(int bn, long wgs) {
    if (wgs <= 256) return 0;                 // the tile's default
    if (bn == 128) return 2;
    return wgs <= 512 ? 3 : 2;
}

// Many-row problems (many frame pairs, 448² / 518² inputs): 128x128 tiles halve the LDS and L2 bytes per MFMA.
static bool big_problem(int M, int N, int splits) {
    // >= 256 tiles: with 2 workgroups per CU that is one round on every CU or more (measured: 294 tiles -12 %, 176 tiles
    // +37 % against 64x64 tiles)
    return splits == 1 && (N % 128) == 0 && (long)((M + 127) / 128) * (N / 128) >= 256;
}

template <typename T, class Epi>
static int launch_64wide(const T* A, const T* W, int M, int N, int K, const EpiArgs& epi, hipStream_t stream, int splits, int kg,
                         int ns) {
    if (kg == 2) return launch_one<T, 64, 2, Epi>(A, W, M, N, K, epi, stream, splits);
    if (ns == 3) return launch_one<T, 64, 1, Epi, 64, 3>(A, W, M, N, K, epi, stream, splits);
    if (ns == 2) return launch_one<T, 64, 1, Epi, 64, 2>(A, W, M, N, K, epi, stream, splits);
    return launch_one<T, 64, 1, Epi>(A, W, M, N, K, epi, stream, splits);
}

template <typename T, class Epi>
static int launch_tiles(const T* A, const T* W, int M, int N, int K, const EpiArgs& epi, hipStream_t stream,
                        int splits = 1, bool fixed64 = false) {
    const int bk = 128 / (int)sizeof(T);
    if (big_problem(M, N, splits)) return launch_one<T, 128, 1, Epi, 128>(A, W, M, N, K, epi, stream, splits);
    const TilePlan pl = plan_tiles(M, N, K / splits / bk, splits, fixed64);
    int ns = pl.kg == 1 ? ring_stages(pl.bn, (long)((M + 63) / 64) * (N / pl.bn) * splits) : 0;
    if (pl.bn == 128) {
        if (pl.kg == 2) return launch_one<T, 128, 2, Epi>(A, W, M, N, K, epi, stream, splits);
        if (ns == 2) return launch_one<T, 128, 1, Epi, 64, 2>(A, W, M, N, K, epi, stream, splits);
        if (ns == 3) return launch_one<T, 128, 1, Epi, 64, 3>(A, W, M, N, K, epi, stream, splits);
        return launch_one<T, 128, 1, Epi>(A, W, M, N, K, epi, stream, splits);
    }
    if (pl.bn == 96) {
        if (pl.kg == 2) return launch_one<T, 96, 2, Epi>(A, W, M, N, K, epi, stream, splits);
        return launch_one<T, 96, 1, Epi>(A, W, M, N, K, epi, stream, splits);
    }
    return launch_64wide<T, Epi>(A, W, M, N, K, epi, stream, splits, pl.kg, ns);
}

// narrow layers / patch embed: 64-wide tiles only (keeps the number of instantiations down)
template <typename T, class Epi>
static int launch_tiles64(const T* A, const T* W, int M, int N, int K, const EpiArgs& epi, hipStream_t stream,
                          int splits = 1) {
    const int bk = 128 / (int)sizeof(T);
    if (big_problem(M, N, splits)) return launch_one<T, 128, 1, Epi, 128>(A, W, M, N, K, epi, stream, splits);
    const TilePlan pl = plan_tiles(M, N, K / splits / bk, splits, true);
    const int ns = pl.kg == 1 ? ring_stages(64, (long)((M + 63) / 64) * (N / 64) * splits) : 0;
    return launch_64wide<T, Epi>(A, W, M, N, K, epi, stream, splits, pl.kg, ns);
}

int launch_linear(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                  int gelu, hipStream_t stream, int wexp) {
    if (!shapes_ok(p, M, N, K)) return -2;
    if (const int bn = big_tile_width(p, M, N, K, 1, false)) return launch_linear_big(p, bn, A, W, bias, out, M, N, K, 1, gelu, false, stream, wexp);
    return launch_linear_classic(p, A, W, bias, out, M, N, K, gelu, stream, wexp);
}

int launch_linear_classic(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                          int gelu, hipStream_t stream, int wexp) {
    if (!shapes_ok(p, M, N, K)) return -2;
    const EpiArgs e{out, bias, nullptr, gelu, p == PREC_X2 ? wexp : 0};
    if (p == PREC_X2) return launch_tiles<hx2, EpiStore<hx2>>((const hx2*)A, (const hx2*)W, M, N, 2 * K, e, stream);
    if (p == PREC_F32) return launch_tiles<float, EpiStore<float>>((const float*)A, (const float*)W, M, N, K, e, stream);
    if (p == PREC_F16) return launch_tiles<f16, EpiStore<f16>>((const f16*)A, (const f16*)W, M, N, K, e, stream);
    return launch_tiles<bf16, EpiStore<bf16>>((const bf16*)A, (const bf16*)W, M, N, K, e, stream);
}

// experiments (vitvs_op_linear_variant 2): the 128 x 128 tiles of this file whatever the shape
int launch_linear_128(Precision p, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int gelu,
                      int splits, bool partial, hipStream_t stream) {
    if (!shapes_ok(p, M, N, K) || N % 128 != 0 || !plain16(p)) return -2;
    if (partial) {
        const EpiArgs e{out, nullptr, nullptr, 0};
        if (p == PREC_F16) return launch_one<f16, 128, 1, EpiPartial, 128>((const f16*)A, (const f16*)W, M, N, K, e, stream, splits);
        return launch_one<bf16, 128, 1, EpiPartial, 128>((const bf16*)A, (const bf16*)W, M, N, K, e, stream, splits);
    }
    const EpiArgs e{out, bias, nullptr, gelu};
    if (p == PREC_F16) return launch_one<f16, 128, 1, EpiStore<f16>, 128>((const f16*)A, (const f16*)W, M, N, K, e, stream, 1);
    return launch_one<bf16, 128, 1, EpiStore<bf16>, 128>((const bf16*)A, (const bf16*)W, M, N, K, e, stream, 1);
}

int launch_linear_residual(Precision p, const void* A, const void* W, const float* bias, const float* ls, float* x,
                           int M, int N, int K, hipStream_t stream, int wexp) {
    if (!shapes_ok(p, M, N, K)) return -2;
    const EpiArgs e{x, bias, ls, 0, p == PREC_X2 ? wexp : 0};
    if (p == PREC_X2) return launch_tiles64<hx2, EpiResidual>((const hx2*)A, (const hx2*)W, M, N, 2 * K, e, stream);
    if (p == PREC_F32) return launch_tiles64<float, EpiResidual>((const float*)A, (const float*)W, M, N, K, e, stream);
    if (p == PREC_F16) return launch_tiles64<f16, EpiResidual>((const f16*)A, (const f16*)W, M, N, K, e, stream);
    return launch_tiles64<bf16, EpiResidual>((const bf16*)A, (const bf16*)W, M, N, K, e, stream);
}

int splitk_slices(Precision p, int M, int N, int K) {
    const int bk = k_tile(p);
    // Many rows, narrow layer (8 frame pairs or a 518² input through proj / fc2): the 256x128 tiles of gemm_big.hip fill
    // well under half of the chip (78 tiles at 3152 x 768), so K is cut into the most slices that still fit one workgroup
    // per CU and leave >= 8 k-tiles per slice (3152 x 768 x 3072: 35 us on the tiles below -> 3 slices of 234 tiles).
    if (p != PREC_F32 && M >= 1024 && N % 128 == 0) {
        const long t128 = (long)((M + 255) / 256) * (N / 128);
        if (t128 >= 96) return 1;
        int pick = 1;
        for (int c : {2, 3, 4}) {
            if (K % (c * 64) != 0 || K / c < 8 * 64) continue;
            if (t128 * c <= 256) pick = c;
        }
        // Beside other queues' launches (vitvs_set_option "in_flight"), from ~2000 rows on, a third slice costs more in partial sums
        // (residual_ln is at the HBM rate there and does not overlap with anything) than it gains in fill: three in flight, same
        // box, 8 / 6 pairs 9108 -> 9578 / 8640 -> 9105 updates/s with two; 4 pairs (1576 rows, 4 slices) 7610 -> 7248: not there.
        if (g_updates_in_flight >= 2 && pick > 2 && M >= 2048) pick = 2;
        if (pick > 1) return pick;
    }
    // The most K slices that still put at most one workgroup on every CU (each slice >= 4 k-tiles);
    // none if the tiles alone already cover the chip.
    const long tiles = (long)((M + 63) / 64) * (N / 64);
    int best = 1;
    for (int c : {2, 3, 4, 6, 8}) {
        if ((K % (c * bk)) != 0 || K / c < 4 * bk) continue;
        if (tiles * c <= 256) best = c;
    }
    // Beside other queues' launches (vitvs_set_option "in_flight") the chip is filled by THEIR workgroups: a third K slice
    // only adds partial sums for residual_ln to read (ViT-B/16 224², 4 in flight, same box: 3946 -> 4075 updates/s with 2; ONE slice:
    // 4301-4348 -> 4045-4091, round 5 — 84 workgroups with 48 k-tiles each are too long a chain; THREE slices again with four clean
    // queues: 4818-4841 -> 4420-4464).
    if (g_updates_in_flight >= 2 && best > 2) best = 2;
    return best;
}

int launch_linear_partial(Precision p, const void* A, const void* W, float* part, int M, int N, int K, int splits,
                          hipStream_t stream, int wexp) {
    if (!shapes_ok(p, M, N, K) || splits < 1 || (K % (splits * k_tile(p))) != 0) return -2;
    if (const int bn = big_tile_width(p, M, N, K, splits, true))
        return launch_linear_big(p, bn, A, W, nullptr, part, M, N, K, splits, 0, true, stream, wexp);
    return launch_linear_partial_classic(p, A, W, part, M, N, K, splits, stream, wexp);
}

int launch_linear_partial_classic(Precision p, const void* A, const void* W, float* part, int M, int N, int K, int splits,
                                  hipStream_t stream, int wexp) {
    if (!shapes_ok(p, M, N, K) || splits < 1 || (K % (splits * k_tile(p))) != 0) return -2;
    const EpiArgs e{part, nullptr, nullptr, 0, p == PREC_X2 ? wexp : 0};
    if (p == PREC_X2) return launch_tiles64<hx2, EpiPartial>((const hx2*)A, (const hx2*)W, M, N, 2 * K, e, stream, splits);
    if (p == PREC_F32) return launch_tiles64<float, EpiPartial>((const float*)A, (const float*)W, M, N, K, e, stream, splits);
    if (p == PREC_F16) return launch_tiles64<f16, EpiPartial>((const f16*)A, (const f16*)W, M, N, K, e, stream, splits);
    return launch_tiles64<bf16, EpiPartial>((const bf16*)A, (const bf16*)W, M, N, K, e, stream, splits);
}

int linear_tile_plan(Precision p, int M, int N, int K, int splits, bool partial, int out[3]) {
    if (!shapes_ok(p, M, N, K) || splits < 1 || (K % (splits * k_tile(p))) != 0 || (!partial && splits != 1)) return -2;
    if (const int bn = big_tile_width(p, M, N, K, splits, partial)) {
        out[0] = bn > 1000 ? 192 : 256; out[1] = bn > 1000 ? bn - 1000 - 64 * (bn == 1192) : bn; out[2] = 0;   // 1192 -> 128, 1256 -> 256
        return 0;
    }
    if (big_problem(M, N, splits)) { out[0] = 128; out[1] = 128; out[2] = 1; return 0; }
    const TilePlan pl = plan_tiles(M, N, K / splits / k_tile(p), splits, partial);
    out[0] = 64; out[1] = pl.bn; out[2] = pl.kg;
    return 0;
}

int launch_patch_embed(Precision p, const void* Ape, const void* Wpe, const float* bias, const float* pos, float* x,
                       int n_img, int T, int D, int Kp, hipStream_t stream, int wexp) {
    const int M = n_img * T;
    if (!shapes_ok(p, M, D, Kp)) return -2;
    const EpiArgs e{x, bias, pos, T, p == PREC_X2 ? wexp : 0};
    if (p == PREC_X2) return launch_tiles64<hx2, EpiPatch>((const hx2*)Ape, (const hx2*)Wpe, M, D, 2 * Kp, e, stream);
    if (p == PREC_F32) return launch_tiles64<float, EpiPatch>((const float*)Ape, (const float*)Wpe, M, D, Kp, e, stream);
    if (p == PREC_F16) return launch_tiles64<f16, EpiPatch>((const f16*)Ape, (const f16*)Wpe, M, D, Kp, e, stream);
    return launch_tiles64<bf16, EpiPatch>((const bf16*)Ape, (const bf16*)Wpe, M, D, Kp, e, stream);
}

}  // namespace vitvs
