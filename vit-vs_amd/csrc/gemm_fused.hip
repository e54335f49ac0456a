// Linear layers with the LayerNorm folded in, so that a transformer block is 5 launches
// (qkv, attention, proj, fc1, fc2) instead of 7: no split-K slices to reduce, no LayerNorm kernel.
//
// Reference arithmetic (dino_patch/block.py:90-96): x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x))).
// Algebra used:   norm(x) W^T = rstd * (x (gamma ⊙ W)^T) - rstd * mu * c1 + c2,
//                 c1[n] = sum_k gamma_k W[n][k],  c2[n] = sum_k beta_k W[n][k] + bias[n]
// so the consumer GEMM (qkv, fc1) multiplies the RAW residual stream by the gamma-folded weight and applies
// the row's (mu, rstd) in its epilogue (`linear_ln_kernel`).  The row moments come from the producer GEMM
// (patch-embed, proj, fc2): its epilogue adds the residual, writes x (fp32) and, in bf16 mode, the bf16 copy
// the next GEMM reads, and emits per-row partial moments (sum, M2) over each group of 16 columns
// (`EpiResidualStats`); the consumer merges the D/16 partials of a row with Chan's parallel-variance update in
// a fixed order.  Everything is deterministic (no atomics, no cross-workgroup reduction order).
#include "gemm_core.h"
#include "kernels.h"

namespace vitvs {

__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_fast_f(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-x * x);
    return 0.5f * v * (1.0f + copysignf(e, v));
}

template <typename T>
__device__ __forceinline__ void store4t(T* dst, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4t<float>(float* dst, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4t<bf16>(bf16* dst, float a, float b, float c, float d) {
    bf16x4 h = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
    *reinterpret_cast<bf16x4*>(dst) = h;
}

// Partial moments of the 16 columns [16p, 16p+16) of row m: the 4 lanes (lane, lane^16, lane^32, lane^48) of a
// wave hold 4 consecutive columns each.  Every lane of the wave must call this (shuffles).
__device__ __forceinline__ void write_moments16(float4 r, bool valid, float* stats, int m, int np, int p) {
    float s = (r.x + r.y) + (r.z + r.w);
    s = rows_sum(s);
    const float mean = s * 0.0625f;
    const float a = r.x - mean, b = r.y - mean, c = r.z - mean, d = r.w - mean;
    float q = (a * a + b * b) + (c * c + d * d);
    q = rows_sum(q);
    if (valid && (threadIdx.x & 48) == 0) *reinterpret_cast<float2*>(stats + ((size_t)m * np + p) * 2) = make_float2(s, q);
}

// ---- producers -------------------------------------------------------------------------------
template <typename T>
struct EpiResidualStats {
    float* x;          // [M][N] fp32 residual stream, updated in place
    T* xb;             // same values in the GEMM operand type (null when T = float: x itself is the operand)
    const float* bias;
    const float* ls;   // LayerScale, may be null
    float* stats;      // [M][N/16][2]
    int ld;
    __device__ __forceinline__ void apply(int m, int n, f32x4 v, bool valid) const {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        if (ls) {
            const float4 g = *reinterpret_cast<const float4*>(ls + n);
            v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            float4* px = reinterpret_cast<float4*>(x + (size_t)m * ld + n);
            r = *px;
            r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
            *px = r;
            if (xb) store4t<T>(xb + (size_t)m * ld + n, r.x, r.y, r.z, r.w);
        }
        write_moments16(r, valid, stats, m, ld >> 4, n >> 4);
    }
};

template <typename T>
struct EpiPatchStats {
    float* x;
    T* xb;
    const float* bias;
    const float* pos;
    float* stats;
    int Tn, D;
    __device__ __forceinline__ void apply(int m, int n, f32x4 v, bool valid) const {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        int row = 0;
        if (valid) {
            const int img = m / Tn, t = m - img * Tn;
            row = img * (Tn + 1) + 1 + t;
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)(1 + t) * D + n);
            r = make_float4(v[0] + b.x + pe.x, v[1] + b.y + pe.y, v[2] + b.z + pe.z, v[3] + b.w + pe.w);
            *reinterpret_cast<float4*>(x + (size_t)row * D + n) = r;
            if (xb) store4t<T>(xb + (size_t)row * D + n, r.x, r.y, r.z, r.w);
        }
        write_moments16(r, valid, stats, row, D >> 4, n >> 4);
    }
};

template <typename T, int BN, int KG, class Epi>
__global__ __launch_bounds__(256 * KG) void linear_stats_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                int M, int N, int K, Epi epi) {
    using Tile = GemmTile<64, BN, KG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
    f32x4 acc[Tile::NT][Tile::MT];
    gemm_mainloop<T, 64, BN, KG>(A, W, K, K, M, N, m0, n0, 0, K, smem, acc);
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int kg = (KG == 2) ? k_group() : 0;
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni) {
        const int n = n0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) {
            if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) continue;
            const int m = m0 + wm * Tile::WM + mi * 16 + (lane & 15);
            epi.apply(m, n, acc[ni][mi], m < M);
        }
    }
}

// ---- consumer: out = act(rstd*(x W'^T) - rstd*mu*c1 + c2) ----------------------------------------
struct LnArgs {
    const float* stats;  // [M][np][2] partial (sum, M2) over 16 columns each
    const float* c1;     // [N]
    const float* c2;     // [N]
    int np;              // D / 16
    float inv_d, eps;
};

template <typename T, int BN, int KG>
__global__ __launch_bounds__(256 * KG) void linear_ln_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                             T* __restrict__ out, int M, int N, int K, int gelu,
                                                             LnArgs ln) {
    using Tile = GemmTile<64, BN, KG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x;

    // Row moments: thread t < 256 owns a quarter (np/4 partials) of row t/4; the loads are issued now and
    // consumed after the main loop (they are older than every LDS-DMA copy, so the counted waits cover them).
    float4 sv[8];
    const int np8 = ln.np >> 3;  // float4 loads per thread: (np / 4 partials) * 2 floats / 4
    if (tid < 256) {
        const int r = min(m0 + (tid >> 2), M - 1);
        const float4* src = reinterpret_cast<const float4*>(ln.stats + ((size_t)r * ln.np + (tid & 3) * (ln.np >> 2)) * 2);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < np8) sv[i] = src[i];
    }

    f32x4 acc[Tile::NT][Tile::MT];
    gemm_mainloop<T, 64, BN, KG>(A, W, K, K, M, N, m0, n0, 0, K, smem, acc);

    // merge the partial moments (Chan et al.), fixed order: within the thread, then across the 4 quarter lanes
    float2* rowstat = reinterpret_cast<float2*>(smem);   // ring stage 0: idle once every wave is past the main loop
    if (KG == 1) __syncthreads();   // the ring tail may still be read by a slower wave's last tile
    if (tid < 256) {
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < np8) {
                const float ps[2] = {sv[i].x, sv[i].z}, pq[2] = {sv[i].y, sv[i].w};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float mb = ps[h] * 0.0625f;
                    const float delta = mb - mean, tot = cnt + 16.f;
                    mean += delta * (16.f / tot);
                    m2 += pq[h] + delta * delta * (cnt * 16.f / tot);
                    cnt = tot;
                }
            }
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {   // equal counts on both sides
            const float mean_o = __shfl_xor(mean, o, WAVE), m2_o = __shfl_xor(m2, o, WAVE);
            const float delta = mean_o - mean;
            mean = 0.5f * (mean + mean_o);
            m2 = m2 + m2_o + delta * delta * (cnt * 0.5f);
            cnt *= 2.f;
        }
        if ((tid & 3) == 0) rowstat[tid >> 2] = make_float2(mean, 1.0f / sqrtf(m2 * ln.inv_d + ln.eps));
    }
    __syncthreads();

    const int lane = tid & 63, wave = (tid >> 6) & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int kg = (KG == 2) ? k_group() : 0;
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni) {
        const int n = n0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4);
        const float4 c1 = *reinterpret_cast<const float4*>(ln.c1 + n);
        const float4 c2 = *reinterpret_cast<const float4*>(ln.c2 + n);
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) {
            if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) continue;
            const int ml = wm * Tile::WM + mi * 16 + (lane & 15);
            const int m = m0 + ml;
            const float2 st = rowstat[ml];
            const float rs = st.y, rm = st.y * st.x;
            float v0 = rs * acc[ni][mi][0] - rm * c1.x + c2.x;
            float v1 = rs * acc[ni][mi][1] - rm * c1.y + c2.y;
            float v2 = rs * acc[ni][mi][2] - rm * c1.z + c2.z;
            float v3 = rs * acc[ni][mi][3] - rm * c1.w + c2.w;
            if (gelu) {
                if (sizeof(T) == 2) {
                    v0 = gelu_erf_fast_f(v0); v1 = gelu_erf_fast_f(v1); v2 = gelu_erf_fast_f(v2); v3 = gelu_erf_fast_f(v3);
                } else {
                    v0 = gelu_erf_f(v0); v1 = gelu_erf_f(v1); v2 = gelu_erf_f(v2); v3 = gelu_erf_f(v3);
                }
            }
            if (m < M) store4t<T>(out + (size_t)m * N + n, v0, v1, v2, v3);
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------
static int ktile(Precision p) { return (p == PREC_F32) ? 32 : 64; }

template <typename K>
static int raise_lds(K kernel, int bytes, bool* done) {
    if (*done) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) !=
        hipSuccess)
        return -1;
    *done = true;
    return 0;
}

static int pick_kg(int M, int N, int bn, int nk) {
    const long wgs = (long)((M + 63) / 64) * (N / bn);
    return (wgs <= 256 && nk >= 4 && nk % 2 == 0) ? 2 : 1;
}

static int pick_bn(int M, int N) {  // same rule as plan_tiles() in gemm.hip
    const long mt = (M + 63) / 64;
    long best = -1;
    int bn = 64;
    for (int c : {128, 96, 64}) {
        if (N % c) continue;
        const long wgs = mt * (N / c);
        if (wgs <= 256 && wgs > best) { best = wgs; bn = c; }
    }
    if (best < 0) bn = (N % 128 == 0) ? 128 : 64;
    return bn;
}

template <typename T, int BN, int KG>
static int launch_ln_one(const T* A, const T* W, T* out, int M, int N, int K, int gelu, const LnArgs& ln,
                         hipStream_t stream) {
    using Tile = GemmTile<64, BN, KG>;
    static bool raised = false;
    if (raise_lds(&linear_ln_kernel<T, BN, KG>, Tile::LDS_BYTES, &raised)) return -1;
    dim3 grid(N / BN, (M + 63) / 64);
    launch(linear_ln_kernel<T, BN, KG>, grid, dim3(Tile::THREADS), Tile::LDS_BYTES, stream, A, W, out, M, N, K, gelu, ln);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
static int launch_ln_t(const T* A, const T* W, T* out, int M, int N, int K, int gelu, const LnArgs& ln,
                       hipStream_t stream) {
    const int bn = pick_bn(M, N), kg = pick_kg(M, N, bn, K / (128 / (int)sizeof(T)));
    if (bn == 128) return kg == 2 ? launch_ln_one<T, 128, 2>(A, W, out, M, N, K, gelu, ln, stream)
                                  : launch_ln_one<T, 128, 1>(A, W, out, M, N, K, gelu, ln, stream);
    if (bn == 96) return kg == 2 ? launch_ln_one<T, 96, 2>(A, W, out, M, N, K, gelu, ln, stream)
                                 : launch_ln_one<T, 96, 1>(A, W, out, M, N, K, gelu, ln, stream);
    return kg == 2 ? launch_ln_one<T, 64, 2>(A, W, out, M, N, K, gelu, ln, stream)
                   : launch_ln_one<T, 64, 1>(A, W, out, M, N, K, gelu, ln, stream);
}

int launch_linear_ln(Precision p, const void* A, const void* Wf, const float* c1, const float* c2, const float* stats,
                     int D, void* out, int M, int N, int K, int gelu, float eps, hipStream_t stream) {
    if (M <= 0 || N % 64 || K % ktile(p) || D % 128 || D > 1024 || K != D) return -2;
    LnArgs ln{stats, c1, c2, D / 16, 1.0f / (float)D, eps};
    if (p == PREC_F32) return launch_ln_t<float>((const float*)A, (const float*)Wf, (float*)out, M, N, K, gelu, ln, stream);
    return launch_ln_t<bf16>((const bf16*)A, (const bf16*)Wf, (bf16*)out, M, N, K, gelu, ln, stream);
}

template <typename T, int KG, class Epi>
static int launch_stats_one(const T* A, const T* W, int M, int N, int K, const Epi& epi, hipStream_t stream) {
    using Tile = GemmTile<64, 64, KG>;
    static bool raised = false;
    if (raise_lds(&linear_stats_kernel<T, 64, KG, Epi>, Tile::LDS_BYTES, &raised)) return -1;
    dim3 grid(N / 64, (M + 63) / 64);
    launch(linear_stats_kernel<T, 64, KG, Epi>, grid, dim3(Tile::THREADS), Tile::LDS_BYTES, stream, A, W, M, N, K, epi);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T, class Epi>
static int launch_stats_t(const T* A, const T* W, int M, int N, int K, const Epi& epi, hipStream_t stream) {
    const int kg = pick_kg(M, N, 64, K / (128 / (int)sizeof(T)));
    return kg == 2 ? launch_stats_one<T, 2, Epi>(A, W, M, N, K, epi, stream)
                   : launch_stats_one<T, 1, Epi>(A, W, M, N, K, epi, stream);
}

int launch_linear_residual_stats(Precision p, const void* A, const void* W, const float* bias, const float* ls, float* x,
                                 void* xb, float* stats, int M, int N, int K, hipStream_t stream) {
    if (M <= 0 || N % 128 || K % ktile(p)) return -2;
    if (p == PREC_F32) {
        EpiResidualStats<float> e{x, nullptr, bias, ls, stats, N};
        return launch_stats_t<float>((const float*)A, (const float*)W, M, N, K, e, stream);
    }
    EpiResidualStats<bf16> e{x, (bf16*)xb, bias, ls, stats, N};
    return launch_stats_t<bf16>((const bf16*)A, (const bf16*)W, M, N, K, e, stream);
}

int launch_patch_embed_stats(Precision p, const void* Ape, const void* Wpe, const float* bias, const float* pos, float* x,
                             void* xb, float* stats, int n_img, int T, int D, int Kp, hipStream_t stream) {
    const int M = n_img * T;
    if (M <= 0 || D % 128 || Kp % ktile(p)) return -2;
    if (p == PREC_F32) {
        EpiPatchStats<float> e{x, nullptr, bias, pos, stats, T, D};
        return launch_stats_t<float>((const float*)Ape, (const float*)Wpe, M, D, Kp, e, stream);
    }
    EpiPatchStats<bf16> e{x, (bf16*)xb, bias, pos, stats, T, D};
    return launch_stats_t<bf16>((const bf16*)Ape, (const bf16*)Wpe, M, D, Kp, e, stream);
}

}  // namespace vitvs
