// Multi-head self-attention (head dim 64) for the ViT blocks, flash-style: one workgroup owns
// 64 queries of one (image, head) and streams the keys/values in tiles of 64 through LDS with an
// online softmax, so the N x N score matrix never exists in memory (N = 197 ... 3137).
//
// Reference arithmetic being replaced: dino_patch/attention.py:70-80
//   q,k,v = qkv(x).reshape(B,N,3,H,hd) ; softmax(q k^T * hd^-0.5) v ; heads re-interleaved to [B,N,C].
//
// Orientation (both precisions): the score tile is computed TRANSPOSED, S^T = K Q^T, so that after
// the MFMA a lane holds 16 keys of ONE query (q = lane & 15): the softmax statistics are per-lane
// scalars, P stays in registers as the B operand of O^T = V^T P^T, and the O^T accumulator again has
// its query on the lane, so the running rescale is a per-lane multiply.  The MFMA k-slot -> key map
// is permuted the same way for P and V (any bijection is legal as long as both operands agree).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace vitvs {

// s_waitcnt vmcnt(PER * rem): the immediate must be a constant, so the (wave-uniform) count is dispatched.
template <int PER>
__device__ __forceinline__ void wait_copies(int rem) {
    switch (rem) {
        case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PER) : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PER) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * PER) : "memory"); break;
    }
}

constexpr float kScaleLog2e = 0.125f * 1.44269504088896340736f;  // hd^-0.5 * log2(e), hd = 64

// ------------------------------------------------------------------------------------ bf16
// RESIDENT: all keys/values of the (image, head) fit in LDS (N <= 512 in bf16): they are copied once by
// LDS-DMA (source-side swizzle for K, linear V), one wait, and the tile loop then runs without further
// global loads or barriers.  Otherwise tiles are streamed through a single 64-key stage.
template <bool RESIDENT>
__global__ __launch_bounds__(256) void attention_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                             int N, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ntiles_all = (N + 63) / 64;
    unsigned char* ldsK = smem;
    unsigned char* ldsV = smem + (RESIDENT ? ntiles_all : 1) * 64 * 128;
    unsigned char* ldsQ = smem + 2 * ntiles_all * 64 * 128;   // RESIDENT only: 64 query rows x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int img = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const size_t ld = (size_t)3 * D;
    const bf16* base = qkv + (size_t)img * N * ld + h * 64;
    const bf16* Kp = base + D;
    const bf16* Vp = base + 2 * D;

    const int q = q0 + 16 * wave + qi;
    const int qrow = min(q, N - 1);
    bf16x8 qf[2];
    if constexpr (!RESIDENT) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + (size_t)qrow * ld + 32 * s + 8 * g));
    }

    f32x4 acc_o[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) acc_o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int srow = tid >> 3, schunk = tid & 7;
    u32x4 rk[2], rv[2];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(kb + srow + 32 * i, N - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(Kp + (size_t)key * ld + schunk * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(Vp + (size_t)key * ld + schunk * 8);
        }
    };
    const int ntiles = ntiles_all;
    if constexpr (RESIDENT) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* gbl_ptr;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // 8 KiB per 64-key tile per operand = 8 copy instructions of 1 KiB (8 keys x 128 B) each; every
        // wave issues, tile by tile, 2 of the K and 2 of the V instructions (4 per tile), so that tile t
        // has landed when at most 4 * (ntiles - 1 - t) of the wave's copies are still outstanding.
        // Q goes through LDS as well (this wave's own 16 query rows, 2 copies): with every vector-memory
        // op an LDS-DMA the compiler inserts no vmcnt of its own, and the counted waits below hold.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r8 = wv * 2 + j;
            const int row = r8 * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const bf16* srcp = base + (size_t)min(q0 + row, N - 1) * ld + c * 8;
            __builtin_amdgcn_global_load_lds((gbl_ptr)srcp, (lds_ptr)(ldsQ + r8 * 1024), 16, 0, 0);
        }
        for (int tt = 0; tt < ntiles; ++tt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool isV = j >= 2;
                const int r8 = tt * 8 + wv * 2 + (j & 1);            // group of 8 keys
                const int row = (r8 & 7) * 8 + (lane >> 3);          // row inside its 64-key tile
                const int key = min(tt * 64 + row, N - 1);
                const int c = isV ? (lane & 7) : ((lane & 7) ^ ((row >> 1) & 7));
                const bf16* srcp = (isV ? Vp : Kp) + (size_t)key * ld + c * 8;
                unsigned char* dstp = (isV ? ldsV : ldsK) + r8 * 1024;
                __builtin_amdgcn_global_load_lds((gbl_ptr)srcp, (lds_ptr)dstp, 16, 0, 0);
            }
        }
    } else {
        gload(0);
    }
    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        if constexpr (!RESIDENT) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<u32x4*>(ldsK + tile128_off(srow + 32 * i, schunk)) = rk[i];
                *reinterpret_cast<u32x4*>(ldsV + (srow + 32 * i) * 128 + schunk * 16) = rv[i];
            }
            __syncthreads();
            if (t + 1 < ntiles) gload(kb + 64);
        }
        if constexpr (RESIDENT) {
            wait_copies<4>(ntiles - 1 - t);
            __builtin_amdgcn_s_barrier();
            if (t == 0) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    qf[s] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(ldsQ + tile128_off(16 * wave + qi, 4 * s + g)));
            }
        }
        const unsigned char* tK = ldsK + (RESIDENT ? t * 64 * 128 : 0);
        const unsigned char* tV = ldsV + (RESIDENT ? t * 64 * 128 : 0);

        // S^T tiles: acc_s[t4][r] = S[key = kb + 16*t4 + 4g + r][q]
        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kf = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(tK + tile128_off(16 * t4 + qi, 4 * s + g)));
                acc_s[t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc_s[t4], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + 16 * t4 + 4 * g + r;
                float x = acc_s[t4][r] * kScaleLog2e;
                x = (key < N) ? x : -INFINITY;
                acc_s[t4][r] = x;
                mloc = fmaxf(mloc, x);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, WAVE));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, WAVE));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(acc_s[t4][r] - m_new);
                acc_s[t4][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int td = 0; td < 4; ++td) acc_o[td] *= alpha;

        // O^T += V^T P^T ; k-slot (g, j) of step u  <->  key 32u + 16(j>>2) + 4g + (j&3)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16)acc_s[2 * u + (j >> 2)][j & 3];
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                // hardware-transposed LDS read: lane i of the 16-lane group gets column d0+i of 4 key rows
                const int k0 = 32 * u + 4 * g;
                const unsigned char* a0 = tV + (k0 + (qi >> 2)) * 128 + (16 * td + 4 * (qi & 3)) * 2;
                const unsigned char* a1 = a0 + 16 * 128;
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc_o[td] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v8), pf, acc_o[td], 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 16, WAVE);
    l_run += __shfl_xor(l_run, 32, WAVE);
    const float inv = 1.0f / l_run;
    if (q < N) {
        bf16* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * g;
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            bf16x4 o = {(bf16)(acc_o[td][0] * inv), (bf16)(acc_o[td][1] * inv), (bf16)(acc_o[td][2] * inv),
                        (bf16)(acc_o[td][3] * inv)};
            *reinterpret_cast<bf16x4*>(dst + 16 * td) = o;
        }
    }
}

// ------------------------------------------------------------------------------------ fp32
template <bool RESIDENT>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int N, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ntiles_all = (N + 63) / 64;
    unsigned char* ldsK = smem;
    unsigned char* ldsV = smem + (RESIDENT ? ntiles_all : 1) * 64 * 256;
    unsigned char* ldsQ = smem + 2 * ntiles_all * 64 * 256;   // RESIDENT only: 64 query rows x 256 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int img = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)img * N * ld + h * 64;
    const float* Kp = base + D;
    const float* Vp = base + 2 * D;

    const int q = q0 + 16 * wave + qi;
    const int qrow = min(q, N - 1);
    float4 qf[4];
    if constexpr (!RESIDENT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + 16 * c + 4 * g);
    }

    f32x4 acc_o[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) acc_o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int srow = tid >> 4, schunk = tid & 15;
    u32x4 rk[4], rv[4];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(kb + srow + 16 * i, N - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(Kp + (size_t)key * ld + schunk * 4);
            rv[i] = *reinterpret_cast<const u32x4*>(Vp + (size_t)key * ld + schunk * 4);
        }
    };
    const int ntiles = ntiles_all;
    if constexpr (RESIDENT) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* gbl_ptr;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // 16 KiB per 64-key tile per operand = 16 copy instructions of 1 KiB (4 keys x 256 B) each; every
        // wave issues, tile by tile, 4 of the K and 4 of the V instructions (8 per tile).
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // Q through LDS: this wave's own 16 query rows (see the bf16 kernel)
            const int r4 = wv * 4 + j;
            const int row = r4 * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            const float* srcp = base + (size_t)min(q0 + row, N - 1) * ld + c * 4;
            __builtin_amdgcn_global_load_lds((gbl_ptr)srcp, (lds_ptr)(ldsQ + r4 * 1024), 16, 0, 0);
        }
        for (int tt = 0; tt < ntiles; ++tt) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool isV = j >= 4;
                const int r4 = tt * 16 + wv * 4 + (j & 3);          // group of 4 keys
                const int row = (r4 & 15) * 4 + (lane >> 4);         // row inside its 64-key tile
                const int key = min(tt * 64 + row, N - 1);
                const int c = (lane & 15) ^ (row & 15);              // tile256_off swizzle, applied to the source
                const float* srcp = (isV ? Vp : Kp) + (size_t)key * ld + c * 4;
                unsigned char* dstp = (isV ? ldsV : ldsK) + r4 * 1024;
                __builtin_amdgcn_global_load_lds((gbl_ptr)srcp, (lds_ptr)dstp, 16, 0, 0);
            }
        }
    } else {
        gload(0);
    }
    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        if constexpr (!RESIDENT) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(ldsK + tile256_off(srow + 16 * i, schunk)) = rk[i];
                *reinterpret_cast<u32x4*>(ldsV + tile256_off(srow + 16 * i, schunk)) = rv[i];
            }
            __syncthreads();
            if (t + 1 < ntiles) gload(kb + 64);
        }
        if constexpr (RESIDENT) {
            wait_copies<8>(ntiles - 1 - t);
            __builtin_amdgcn_s_barrier();
            if (t == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    qf[c] = *reinterpret_cast<const float4*>(ldsQ + tile256_off(16 * wave + qi, 4 * c + g));
            }
        }
        const unsigned char* tK = ldsK + (RESIDENT ? t * 64 * 256 : 0);
        const unsigned char* tV = ldsV + (RESIDENT ? t * 64 * 256 : 0);

        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 kf = *reinterpret_cast<const float4*>(tK + tile256_off(16 * t4 + qi, 4 * c + g));
                acc_s[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, acc_s[t4], 0, 0, 0);
                acc_s[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, acc_s[t4], 0, 0, 0);
                acc_s[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, acc_s[t4], 0, 0, 0);
                acc_s[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, acc_s[t4], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + 16 * t4 + 4 * g + r;
                float x = acc_s[t4][r] * kScaleLog2e;
                x = (key < N) ? x : -INFINITY;
                acc_s[t4][r] = x;
                mloc = fmaxf(mloc, x);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, WAVE));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, WAVE));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(acc_s[t4][r] - m_new);
                acc_s[t4][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int td = 0; td < 4; ++td) acc_o[td] *= alpha;

        // O^T += V^T P^T ; MFMA step (t4, r): k-slot g  <->  key 16*t4 + 4g + r
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t4 + 4 * g + r;
#pragma unroll
                for (int td = 0; td < 4; ++td) {
                    const float v = *reinterpret_cast<const float*>(tV + tile256_off(key, 4 * td + (qi >> 2)) +
                                                                    (qi & 3) * 4);
                    acc_o[td] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, acc_s[t4][r], acc_o[td], 0, 0, 0);
                }
            }
    }
    l_run += __shfl_xor(l_run, 16, WAVE);
    l_run += __shfl_xor(l_run, 32, WAVE);
    const float inv = 1.0f / l_run;
    if (q < N) {
        float* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * g;
#pragma unroll
        for (int td = 0; td < 4; ++td)
            *reinterpret_cast<float4*>(dst + 16 * td) =
                make_float4(acc_o[td][0] * inv, acc_o[td][1] * inv, acc_o[td][2] * inv, acc_o[td][3] * inv);
    }
}

template <typename K>
static int launch_attn(K kernel, dim3 grid, size_t lds, hipStream_t stream, const void* qkv, void* out, int N, int D,
                       bool* raised) {
    if (lds > 64 * 1024 && !*raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                144 * 1024) != hipSuccess)
            return -1;
        *raised = true;
    }
    return 0;
}

int launch_attention(Precision p, const void* qkv, void* out, int n_img, int N, int H, hipStream_t stream) {
    if (n_img <= 0 || N <= 0 || H <= 0) return -2;
    const int D = H * 64;
    const int nt = (N + 63) / 64;
    dim3 grid(nt, H, n_img), block(256);
    static bool raised_f32 = false, raised_bf16 = false;
    // The resident (all K/V in LDS, counted waits) variant is kept for experiments: on MI355X it measured
    // 1 % SLOWER end to end than streaming at N = 197 (A/B in one process, profiles/r01_notes.md), because the
    // kernel is bound by the softmax VALU chain at one wave per SIMD, not by the K/V load latency.
    static const bool force_stream = [] { const char* e = getenv("VITVS_ATTN_RESIDENT"); return !(e && e[0] == '1'); }();
    if (p == PREC_F32) {
        const size_t res = (size_t)(2 * nt + 1) * 64 * 256;
        if (res <= 144 * 1024 && !force_stream) {
            if (launch_attn(attention_f32_kernel<true>, grid, res, stream, qkv, out, N, D, &raised_f32)) return -1;
            launch(attention_f32_kernel<true>, grid, block, res, stream, (const float*)qkv, (float*)out, N, D);
        } else {
            launch(attention_f32_kernel<false>, grid, block, 2 * 64 * 256, stream, (const float*)qkv, (float*)out, N, D);
        }
    } else {
        const size_t res = (size_t)(2 * nt + 1) * 64 * 128;
        if (res <= 144 * 1024 && !force_stream) {
            if (launch_attn(attention_bf16_kernel<true>, grid, res, stream, qkv, out, N, D, &raised_bf16)) return -1;
            launch(attention_bf16_kernel<true>, grid, block, res, stream, (const bf16*)qkv, (bf16*)out, N, D);
        } else {
            launch(attention_bf16_kernel<false>, grid, block, 2 * 64 * 128, stream, (const bf16*)qkv, (bf16*)out, N, D);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
