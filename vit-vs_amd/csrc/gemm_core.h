// LDS-tiled MFMA main loop shared by the linear-layer GEMMs and the token Gram kernel.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        (both operands K-contiguous, as nn.Linear stores W)
//
// One workgroup computes a BM x BN tile with KG "k-groups" of 4 waves (2 along M x 2 along N each):
// with KG = 2 the two groups split the workgroup's K range in halves (intra-workgroup split-K) and
// the second group's accumulators are added through LDS at the end.  At one frame pair (M = 394
// rows) a launch has only ~250 workgroups of ~12 k-tiles, so the loop is bound by the LATENCY of its
// global loads, not by MFMA rate or bandwidth; everything here is about keeping loads in flight:
//   * k-tiles are 128 bytes of K per row (32 f32 or 64 bf16), copied global -> LDS by LDS-DMA
//     (`global_load_lds_dwordx4`, no VGPR staging) into a ring of NST stages per k-group;
//   * waits are counted by hand (`s_waitcnt vmcnt(N)` + raw `s_barrier`), so NST - 1 k-tiles per
//     k-group stay in flight across every barrier (hipcc would drain the queue at each barrier);
//   * an LDS-DMA instruction writes its 64 x 16 bytes linearly (8 rows x 128 B), so the bank-conflict
//     swizzle of the LDS image (tile128_off) is applied to the per-lane SOURCE chunk instead.
// The MFMA takes the W rows as its A operand and the activation rows as its B operand, so a lane
// ends up holding 4 consecutive n for one m (16-byte epilogue accesses).
//
//   f32 : v_mfma_f32_16x16x4_f32   (exact fp32 FMA chain, 4 per 16-byte chunk pair)
//   bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte chunk pair), fp32 accumulate; fp16: v_mfma_f32_16x16x32_f16, the same
//   f16x2 (hx2, common.h): rows hold [hi of 32 k | lo of the same 32 k] per 128 bytes; three v_mfma_f32_16x16x32_f16 per
//         k-tile and accumulator (hi.hi + hi.lo + lo.hi): fp32-class sums at 16-bit MFMA rate, operand bytes as fp32
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
__device__ __forceinline__ f32x4 mma_chunk<f16>(f32x4 acc, u32x4 a, u32x4 b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma_chunk<bf16>(f32x4 acc, u32x4 a, u32x4 b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                   0, 0);
}

template <int BM, int BN, int KG = 1, int NS = 0>
struct GemmTile {
    static constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
    static constexpr int MT = WM / 16, NT = WN / 16;    // 16x16 MFMA tiles per wave
    static constexpr int ROWS = BM + BN;
    static constexpr int L = ROWS / 32;                 // LDS-DMA instructions per wave per k-tile
    // ring stages per k-group.  One-wave launches (64-row tiles, one workgroup per CU) want tiles in flight: 4 stages, 3
    // with two k-groups (a 4-stage ring there measured slower).  The 128-row tiles serve many-row problems that run
    // several waves of workgroups: there occupancy wins — 2 stages = 64 KB (128x128) or 48 KB (128x64), i.e. 2-3
    // workgroups per CU, measured -24 ... -27 % on the 6274-row fc1 / qkv GEMMs against 4 stages (1 workgroup per CU).
    // NS > 0 overrides the depth: the launchers of gemm.hip pick a shallower ring for 64-row launches of more than one
    // workgroup per CU, where the LDS footprint decides how many are resident (ring_stages() there has the measurements).
    static constexpr int NST = NS ? NS : ((BM >= 128) ? 2 : ((KG == 1) ? 4 : 3));
    static constexpr int STAGE_BYTES = ROWS * 128;
    static constexpr int GROUP_BYTES = NST * STAGE_BYTES;
    static constexpr int RING_BYTES = KG * GROUP_BYTES;
    // two k-groups swap half of their accumulators through a region of its own (not the rings: no barrier is
    // needed before the swap's writes while other waves still read their last stage)
    static constexpr int SWAP_SLOTS = (NT * MT + 1) / 2;            // accumulator tiles a wave hands over
    static constexpr int SWAP_BYTES = (KG == 2) ? SWAP_SLOTS * 16 * 512 : 0;
    static constexpr bool SWAP_ALIAS = RING_BYTES + SWAP_BYTES > 160 * 1024;   // no room: reuse the rings (one more barrier)
    static constexpr int SWAP_OFFSET = SWAP_ALIAS ? 0 : RING_BYTES;
    static constexpr int LDS_BYTES = SWAP_ALIAS ? (RING_BYTES > SWAP_BYTES ? RING_BYTES : SWAP_BYTES) : RING_BYTES + SWAP_BYTES;
    static constexpr int THREADS = 256 * KG;
    static_assert(ROWS % 32 == 0 && WN % 16 == 0 && WM % 16 == 0, "tile shape");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// With two k-groups, accumulator tile (ni, mi) of a wave is finished (summed + epilogue) by this k-group: the first
// half of the wave's NT x MT tiles in (ni, mi) order belongs to group 0 (an odd NT, e.g. the 64x96 tile, splits evenly).
template <int NT, int MT>
__device__ __forceinline__ int tile_owner(int ni, int mi) { return (2 * (ni * MT + mi) >= NT * MT) ? 1 : 0; }
__device__ __forceinline__ int k_group() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 8); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// acc[ni][mi]: n = n0 + wn*WN + ni*16 + 4*(lane>>4) + reg,  m = m0 + wm*WM + mi*16 + (lane&15)
// Rows of A beyond m_rows-1 and rows of W beyond n_rows-1 are clamped (their results are garbage the
// caller must mask).  k range [k_begin, k_end) must be a multiple of KG k-tiles.  On return the
// accumulators hold the full sums for the column tiles the wave's k-group owns (tile_owner); with
// KG = 1 that is every tile.
template <typename T, int BM, int BN, int KG, int NS = 0>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, const T* __restrict__ W, int lda, int ldw,
                                              int m_rows, int n_rows, int m0, int n0, int k_begin, int k_end,
                                              unsigned char* smem,
                                              f32x4 (&acc)[GemmTile<BM, BN, KG>::NT][GemmTile<BM, BN, KG>::MT],
                                              unsigned long long* ts = nullptr) {
    // ts: cycle-counter stamps of the phases (probe builds, tools/gemm_probe.cpp); null in production (folded away)
    using Tile = GemmTile<BM, BN, KG, NS>;
    constexpr int EPC = Elem<T>::PER_CHUNK;
    constexpr int BK = 8 * EPC;
    constexpr int L = Tile::L, NST = Tile::NST;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave_all >> 2, wave = wave_all & 3;
    const int wm = wave & 1, wn = wave >> 1;
    unsigned char* ring = smem + kg * Tile::GROUP_BYTES;
    const int nk = (k_end - k_begin) / (BK * KG);        // k-tiles per k-group
    const int kg_begin = k_begin + kg * nk * BK;

    // per-lane source of each of this wave's L copy instructions: a wave-uniform base (A or W) plus a 32-bit
    // byte offset that advances by 128 bytes per k-tile (operands are < 4 GiB; checked by the launchers)
    const unsigned char* base[L];
    unsigned off[L];
    int dst_off[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int g8 = wave * L + j;             // group of 8 rows of the combined (A rows, then W rows) tile
        const int row = g8 * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dst_off[j] = g8 * 1024;
        const bool is_a = g8 * 8 < BM;           // wave-uniform
        base[j] = reinterpret_cast<const unsigned char*>(is_a ? A : W);
        const int r = is_a ? min(m0 + row, m_rows - 1) : min(n0 + row - BM, n_rows - 1);
        // 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate): rows < 2^24, row pitch in bytes < 2^24
        off[j] = __umul24((unsigned)r, (unsigned)(is_a ? lda : ldw) * (unsigned)sizeof(T)) + (unsigned)kg_begin * (unsigned)sizeof(T) + c * 16;
    }
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int kt, int stage) {
        unsigned char* dst = ring + stage * Tile::STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < L; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(base[j] + (off[j] + (unsigned)kt * 128u)), (lds_ptr)(dst + dst_off[j]), 16,
                                             0, 0);
    };
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < nk) issue(p, p);

    if (ts) ts[1] = __builtin_readcyclecounter();
    int stage = 0;                                        // kt % NST
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most min(NST - 2, nk - 1 - kt) younger tiles (L copies each) are outstanding
        const int younger = min(NST - 2, nk - 1 - kt);
        if (younger >= 2) wait_vmcnt<2 * L>();
        else if (younger == 1) wait_vmcnt<L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (ts && kt == 0) ts[2] = __builtin_readcyclecounter();
        // every wave has finished reading the stage tile kt - 1 used: refill it with tile kt + NST - 1
        if (kt + NST - 1 < nk) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
        const unsigned char* sa = ring + stage * Tile::STAGE_BYTES;
        const unsigned char* sb = sa + BM * 128;
        if constexpr (kSplit<T>) {
            // f16x2: chunks 0-3 of the 128-byte k-tile hold the hi halves of 32 k, chunks 4-7 the lo halves (common.h).  One
            // k-step of 32: lo.hi + hi.lo first (the small terms), then hi.hi; the dropped lo.lo is < 2^-22 of the product.
            // Term-outer order: the three MFMAs into one accumulator are NT x MT - 1 independent MFMAs apart.
            const int g = lane >> 4;
            u32x4 wh[Tile::NT], wl[Tile::NT], xh[Tile::MT], xl[Tile::MT];
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni) {
                const int r = wn * Tile::WN + ni * 16 + (lane & 15);
                wh[ni] = *reinterpret_cast<const u32x4*>(sb + tile128_off(r, g));
                wl[ni] = *reinterpret_cast<const u32x4*>(sb + tile128_off(r, 4 + g));
            }
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi) {
                const int r = wm * Tile::WM + mi * 16 + (lane & 15);
                xh[mi] = *reinterpret_cast<const u32x4*>(sa + tile128_off(r, g));
                xl[mi] = *reinterpret_cast<const u32x4*>(sa + tile128_off(r, 4 + g));
            }
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = mma_chunk<f16>(acc[ni][mi], wl[ni], xh[mi]);
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = mma_chunk<f16>(acc[ni][mi], wh[ni], xl[mi]);
#pragma unroll
            for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < Tile::MT; ++mi) acc[ni][mi] = mma_chunk<f16>(acc[ni][mi], wh[ni], xh[mi]);
        } else {
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
                    for (int mi = 0; mi < Tile::MT; ++mi) {
                        acc[ni][mi] = mma_chunk<T>(acc[ni][mi], wf[ni], xf[mi]);
                    }
            }
        }
        stage = (stage + 1 == NST) ? 0 : stage + 1;
    }
    if (ts) {
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[Tile::NT - 1][Tile::MT - 1][3]) : "memory");
        ts[3] = __builtin_readcyclecounter();
    }
    if constexpr (KG == 2) {
        // The two k-groups swap halves through LDS: each group ends up with the full sums of the column
        // tiles it owns (tile_owner) and runs the epilogue for those only, so the epilogue work is shared by
        // all 8 waves.  One barrier when the swap area is not part of the rings.
        if constexpr (Tile::SWAP_ALIAS) __syncthreads();   // every wave is done reading the rings
        f32x4* buf = reinterpret_cast<f32x4*>(smem + Tile::SWAP_OFFSET) + (wave_all * Tile::SWAP_SLOTS) * 64 + lane;
        f32x4* peer = reinterpret_cast<f32x4*>(smem + Tile::SWAP_OFFSET) + ((wave_all ^ 4) * Tile::SWAP_SLOTS) * 64 + lane;
        int slot = 0;
#pragma unroll
        for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi)
                if (tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) buf[(slot++) * 64] = acc[ni][mi];
        __syncthreads();
        slot = 0;
#pragma unroll
        for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi)
                if (tile_owner<Tile::NT, Tile::MT>(ni, mi) == kg) acc[ni][mi] += peer[(slot++) * 64];
    }
}

}  // namespace vitvs
