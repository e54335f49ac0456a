// LDS-tiled MFMA main loop shared by the linear-layer GEMMs and the token Gram kernel.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        (both operands K-contiguous, as nn.Linear stores W)
//
// One workgroup = 256 threads = 4 waves (2 along M x 2 along N) computes a BM x BN tile.
// Each k-tile is 128 bytes of K per row (32 f32 or 64 bf16).  At M = 394 rows (one frame pair) a
// launch has only ~100-500 workgroups of ~10 k-tiles each, so the loop is bound by the latency of
// the global loads, not by MFMA or bandwidth: the loads of k-tile kt+PREFETCH are issued (into
// registers) while tile kt is multiplied, i.e. PREFETCH tiles per workgroup are always in flight;
// a tile goes registers -> LDS (tile128_off swizzle, double-buffered) one iteration before use.
// The MFMA takes the W rows as its A operand and the activation rows as its B operand, so a lane
// ends up holding 4 consecutive n for one m (16-byte epilogue accesses).
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

constexpr int kPrefetch = 4;  // k-tiles in flight per workgroup (must be even: LDS stage = tile & 1)

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
    constexpr int P = kPrefetch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int srow = tid >> 3, schunk = tid & 7;

    // per-thread staging sources: row (srow + 32 i), 16-byte chunk schunk of every k-tile
    const T* a_src[PA];
    const T* w_src[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) a_src[i] = A + (size_t)min(m0 + srow + 32 * i, m_rows - 1) * lda + schunk * EPC + k_begin;
#pragma unroll
    for (int i = 0; i < PB; ++i) w_src[i] = W + (size_t)min(n0 + srow + 32 * i, n_rows - 1) * ldw + schunk * EPC + k_begin;
    int lds_a[PA], lds_w[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) lds_a[i] = tile128_off(srow + 32 * i, schunk);
#pragma unroll
    for (int i = 0; i < PB; ++i) lds_w[i] = BM * 128 + tile128_off(srow + 32 * i, schunk);

#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (k_end - k_begin) / BK;
    // Every vector-memory op and barrier below is unconditional (tile indices are clamped to the last
    // tile instead of branching): hipcc can then count the loads in flight and waits only for the
    // oldest register set before each LDS write (s_waitcnt vmcnt(N), N = loads of the younger sets),
    // whereas loads under a branch make it drain the whole queue (vmcnt(0)) every iteration.
    u32x4 ra[P][PA], rb[P][PB];   // register set p holds k-tiles p, p+P, p+2P, ...
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int k0 = min(p, nk - 1) * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[p][i] = *reinterpret_cast<const u32x4*>(a_src[i] + k0);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[p][i] = *reinterpret_cast<const u32x4*>(w_src[i] + k0);
    }
    // One barrier per k-tile: the LDS write of tile kt (stage kt & 1) comes first, then the barrier,
    // then the MFMAs on that stage.  A wave can only reach the write of tile kt + 1 (other stage... the
    // stage tile kt - 1 used) after barrier kt, which every wave passes only after finishing tile kt - 1.
    for (int kt0 = 0; kt0 < nk; kt0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int kt = kt0 + p;
            unsigned char* stage = smem + (p & 1) * Tile::STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(stage + lds_a[i]) = ra[p][i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(stage + lds_w[i]) = rb[p][i];
            __syncthreads();
            {   // set p is free again: refill it with tile kt + P (clamped)
                const int k0 = min(kt + P, nk - 1) * BK;
#pragma unroll
                for (int i = 0; i < PA; ++i) ra[p][i] = *reinterpret_cast<const u32x4*>(a_src[i] + k0);
#pragma unroll
                for (int i = 0; i < PB; ++i) rb[p][i] = *reinterpret_cast<const u32x4*>(w_src[i] + k0);
            }
            if (kt < nk) {  // wave-uniform; contains LDS reads and MFMAs only
                const unsigned char* sa = stage;
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
            }
        }
    }
}

}  // namespace vitvs

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the main loop: the global -> LDS copies are `global_load_lds_dwordx4`
// (no VGPR destination), written into a ring of kStages LDS stages, and the waits are counted by
// hand (`s_waitcnt vmcnt(N)` + raw `s_barrier`), so kStages - 1 k-tiles stay in flight across every
// barrier.  The LDS image is the same tile128_off swizzle as above; because an LDS-DMA instruction
// writes its 64 x 16 bytes linearly (8 rows x 128 B), the swizzle is applied to the per-lane SOURCE
// chunk instead (lane l of the instruction covering rows 8g..8g+7 loads chunk (l&7) ^ ((row>>1)&7)).
namespace vitvs {

constexpr int kStages = 4;

template <int BM, int BN>
struct DmaTile {
    static constexpr int ROWS = BM + BN;
    static constexpr int STAGE_BYTES = ROWS * 128;
    static constexpr int LDS_BYTES = kStages * STAGE_BYTES;
    static constexpr int L = ROWS / 32;   // LDS-DMA instructions per wave per k-tile
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_mainloop_dma(const T* __restrict__ A, const T* __restrict__ W, int lda, int ldw,
                                                  int m_rows, int n_rows, int m0, int n0, int k_begin, int k_end,
                                                  unsigned char* smem,
                                                  f32x4 (&acc)[GemmTile<BM, BN>::NT][GemmTile<BM, BN>::MT]) {
    using Tile = GemmTile<BM, BN>;
    using Dma = DmaTile<BM, BN>;
    constexpr int EPC = Elem<T>::PER_CHUNK;
    constexpr int BK = 8 * EPC;
    constexpr int L = Dma::L;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    // per-lane source of each of this wave's L copy instructions (advances by BK elements per k-tile)
    const T* src[L];
    int dst_off[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int g8 = wave * L + j;             // group of 8 rows of the combined (A rows, then W rows) tile
        const int row = g8 * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dst_off[j] = g8 * 1024;
        if (g8 * 8 < BM)
            src[j] = A + (size_t)min(m0 + row, m_rows - 1) * lda + k_begin + c * EPC;
        else
            src[j] = W + (size_t)min(n0 + row - BM, n_rows - 1) * ldw + k_begin + c * EPC;
    }
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (k_end - k_begin) / BK;
    auto issue = [&](int kt) {
        unsigned char* stage = smem + (kt & (kStages - 1)) * Dma::STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < L; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(src[j] + (size_t)kt * BK), (lds_ptr)(stage + dst_off[j]), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < kStages - 1; ++p)
        if (p < nk) issue(p);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most min(2, nk-1-kt) younger tiles (L copies each) are outstanding
        const int younger = nk - 1 - kt;
        if (younger >= 2) wait_vmcnt<2 * L>();
        else if (younger == 1) wait_vmcnt<L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // every wave has finished reading stage (kt-1) % kStages: refill it with tile kt + kStages - 1
        if (kt + kStages - 1 < nk) issue(kt + kStages - 1);
        const unsigned char* sa = smem + (kt & (kStages - 1)) * Dma::STAGE_BYTES;
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
    }
}

}  // namespace vitvs
