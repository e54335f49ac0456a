// LDS-tiled MFMA main loop shared by the linear-layer GEMMs and the token Gram kernel.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        (both operands K-contiguous, as nn.Linear stores W)
//
// One workgroup = 256 threads = 4 waves (2 along M x 2 along N) computes a BM x BN tile.
// Each k-tile is 128 bytes of K per row (32 f32 or 64 bf16), staged global -> registers ->
// LDS (tile128_off swizzle) with the next tile's global loads issued before the current
// tile's MFMAs.  The MFMA takes the W rows as its A operand and the activation rows as its
// B operand, so a lane ends up holding 4 consecutive n for one m (16-byte epilogue accesses).
//
//   f32 : v_mfma_f32_16x16x4_f32   (exact fp32 FMA chain, 4 per 16-byte chunk pair)
//   bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte chunk pair), fp32 accumulate
#pragma once
#include "common.h"

namespace vitvs {

template <typename T>
__device__ __forceinline__ f32x4 mma_chunk(f32x4 acc, u32x4 a, u32x4 b);

template <>
__device__ __forceinline__ f32x4 mma_chunk<float>(f32x4 acc, u32x4 a, u32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0]), __uint_as_float(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[1]), __uint_as_float(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[2]), __uint_as_float(b[2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[3]), __uint_as_float(b[3]), acc, 0, 0, 0);
    return acc;
}
template <>
__device__ __forceinline__ f32x4 mma_chunk<bf16>(f32x4 acc, u32x4 a, u32x4 b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                   0, 0);
}

template <int BM, int BN>
struct GemmTile {
    static constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
    static constexpr int MT = WM / 16, NT = WN / 16;    // 16x16 MFMA tiles per wave
    static constexpr int STAGE_BYTES = (BM + BN) * 128;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
};

// acc[ni][mi]: n = n0 + wn*WN + ni*16 + 4*(lane>>4) + reg,  m = m0 + wm*WM + mi*16 + (lane&15)
// Rows of A beyond m_rows-1 and rows of W beyond n_rows-1 are clamped (their results are garbage
// the caller must mask).  k range [k_begin, k_end) must be a multiple of the k-tile.
template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, const T* __restrict__ W, int lda, int ldw,
                                              int m_rows, int n_rows, int m0, int n0, int k_begin, int k_end,
                                              unsigned char* smem, f32x4 (&acc)[GemmTile<BM, BN>::NT][GemmTile<BM, BN>::MT]) {
    using Tile = GemmTile<BM, BN>;
    constexpr int EPC = Elem<T>::PER_CHUNK;
    constexpr int BK = 8 * EPC;
    constexpr int PA = BM / 32, PB = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int srow = tid >> 3, schunk = tid & 7;

    // per-thread staging sources: row (srow + 32 i), 16-byte chunk schunk of every k-tile
    size_t a_off[PA], w_off[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) a_off[i] = (size_t)min(m0 + srow + 32 * i, m_rows - 1) * lda + schunk * EPC;
#pragma unroll
    for (int i = 0; i < PB; ++i) w_off[i] = (size_t)min(n0 + srow + 32 * i, n_rows - 1) * ldw + schunk * EPC;
    int lds_a[PA], lds_w[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) lds_a[i] = tile128_off(srow + 32 * i, schunk);
#pragma unroll
    for (int i = 0; i < PB; ++i) lds_w[i] = BM * 128 + tile128_off(srow + 32 * i, schunk);

#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[PA], rb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(A + a_off[i] + k_begin);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const u32x4*>(W + w_off[i] + k_begin);
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(smem + lds_a[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(smem + lds_w[i]) = rb[i];
    __syncthreads();

    const int nk = (k_end - k_begin) / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            const int k0 = k_begin + (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(A + a_off[i] + k0);
#pragma unroll
            for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const u32x4*>(W + w_off[i] + k0);
        }
        const unsigned char* sa = smem + (kt & 1) * Tile::STAGE_BYTES;
        const unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 4 * s + (lane >> 4);
            u32x4 wf[Tile::NT], xf[Tile::MT];
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
                wf[ni] = *reinterpret_cast<const u32x4*>(sb + tile128_off(wn * Tile::WN + ni * 16 + (lane & 15), c));
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi)
                xf[mi] = *reinterpret_cast<const u32x4*>(sa + tile128_off(wm * Tile::WM + mi * 16 + (lane & 15), c));
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = mma_chunk<T>(acc[ni][mi], wf[ni], xf[mi]);
        }
        if (more) {
            unsigned char* dst = smem + ((kt + 1) & 1) * Tile::STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(dst + lds_a[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(dst + lds_w[i]) = rb[i];
        }
        __syncthreads();
    }
}

}  // namespace vitvs
