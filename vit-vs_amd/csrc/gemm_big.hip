// Many-row linear layers (several frame pairs per call, 448² / 518² inputs: M = 2.7k ... 6.3k rows) on 256-row tiles.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        (both operands K-contiguous, 16-bit operands, fp32 accumulate)
//
// The 128x128 / 4-wave tile of gemm_core.h is bound by the CU's LDS fill path at these sizes (one 128-byte
// row of operand per 128 MFMA-lanes of work: profiles/r01_notes.md).  This kernel doubles the tile edge (half the
// operand bytes per FLOP) and runs the 8 waves of a workgroup as two groups that alternate between a LOAD segment (LDS
// fragment reads + LDS-DMA issue) and an MFMA segment, one barrier apart, so that each SIMD's matrix pipe always has
// one of its two waves in an MFMA segment.
//
// Tile: 8 waves as WGM x WGN, each wave MT x NT MFMA tiles of 16x16 (v_mfma_f32_16x16x32_{bf16,f16}); BM = WGM*MT*16,
// BN = WGN*NT*16; k-tile = 64 elements = 128 bytes per row.  A k-tile is held as four "half slots": A0 / A1 = the rows
// every wave reads for the first / second half of its m-tiles, B0 / B1 likewise for its n-tiles.  A k-tile's 4 phases
// multiply the quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0) of every wave's tile.  LDS = 2 stages x 4 slots.
//
// Pipeline (per wave; LA / LB = LDS-DMA instructions per wave per A / B slot):
//   * slot read in phase q is refilled in phase q + 2 (every wave has passed two barriers since its reads of q were
//     issued, and one since they returned), with the k-tile two ahead:
//       phase (j,0): B1(j+1)   (j,1): A1(j+1)   (j,2): A0(j+2)   (j,3): B0(j+2)
//   * a slot is read no earlier than two barriers after every wave's counted wait that retires its copies: the wait at
//     the end of each LOAD segment leaves the 2 LA + 2 LB youngest copies in flight (4 half slots = one k-tile of
//     prefetch distance, ~4 phases of latency budget per copy); the last two k-tiles use the reduced counts derived in
//     `k_tile()` below.
//   * waves 4-7 run one barrier behind waves 0-3 (one extra barrier before their loop, one after for waves 0-3).
// The swizzle of the LDS image (tile128_off) is applied to the per-lane SOURCE chunk (LDS-DMA writes linearly).
// Epilogue: the accumulators go through a wave-private LDS image (all k-tile slots are dead by then) so that every
// store instruction writes whole 128-byte (16-bit output) or 256-byte (fp32 partial sums) row segments.
#include <algorithm>

#include "gemm_core.h"
#include "kernels.h"
#include "probe.h"

namespace vitvs {

template <int WGM_, int WGN_, int MT_, int NT_>
struct BigTile {
    static constexpr int WGM = WGM_, WGN = WGN_, MT = MT_, NT = NT_;
    static constexpr int BM = WGM * MT * 16, BN = WGN * NT * 16;
    static constexpr int AH = (MT / 2) * 16, BH = (NT / 2) * 16;       // rows of one wave in a half slot
    static constexpr int A_ROWS = WGM * AH, B_ROWS = WGN * BH;        // rows of a half slot
    // LDS-DMA instructions per wave per slot (8 waves x 8 rows each).  A slot whose 8-row groups do not divide by the 8 waves
    // (96 rows = 12 groups: the 192-column tile) is rounded up: the surplus instructions of the last round copy groups that
    // another wave copies too (the same bytes to the same LDS address), so that every wave counts the same vmcnt
    static constexpr int GA = A_ROWS / 8, GB = B_ROWS / 8, LA = (GA + 7) / 8, LB = (GB + 7) / 8;
    static constexpr int A_SLOT = A_ROWS * 128, B_SLOT = B_ROWS * 128;
    static constexpr int STAGE = 2 * A_SLOT + 2 * B_SLOT;
    static constexpr int RING = 2 * STAGE;
    static constexpr int WAVE_REGION = RING / 8;                      // epilogue image of one wave
    static constexpr int BIAS_OFFSET = RING;                          // the tile's BN bias values (fp32) behind the ring ...
    // ... for the epilogues that add a bias, and only where the 512 bytes do not push the footprint over half of the CU's
    // 160 KB (the 192 x 128 tile's ring is exactly 80 KB: with the bias area two such workgroups — two queues, a neighbouring
    // launch — could no longer share a CU; that tile reads its column terms from memory at the start of the epilogue instead)
    static constexpr bool bias_in_lds(bool has_bias) { return has_bias && !(RING <= 80 * 1024 && RING + BN * 4 > 80 * 1024); }
    static constexpr int lds_bytes(bool has_bias) { return RING + (bias_in_lds(has_bias) ? BN * 4 : 0); }
    static_assert(WGM * WGN == 8 && MT % 2 == 0 && NT % 2 == 0, "8 waves, even tile counts");
    static_assert(A_ROWS % 8 == 0 && B_ROWS % 8 == 0 && BH % 8 == 0 && AH % 8 == 0, "half slots are made of 8-row copies");
    static_assert(RING + BN * 4 <= 160 * 1024, "LDS budget");
};

// 8-row group of a half slot that copy j of wave `wave` fills: G groups, L = ceil(G / 8) copies per wave.  Whole rounds keep
// a wave's copies adjacent (wave * L + j); a partial last round wraps its surplus onto the round's own first groups.
template <int G, int L>
__device__ __forceinline__ int copy_group(int wave, int j) {
    if constexpr (G == 8 * L) return wave * L + j;
    else {
        const int g = 8 * j + wave;                          // round j, one group per wave
        return g < G ? g : g - (8 * L - G);                  // (96 rows: round 1 holds groups 8 .. 11, copied by waves 0-3 and 4-7)
    }
}

// epilogue policies: what is added / applied to the fp32 sums and what is stored
template <typename T>
struct BigStore {      // out[m][n] = act(sum + bias[n]) in the operand type
    typedef T Out;
    T* out;
    const float* bias;
    int ldo, gelu;
    static constexpr bool HAS_BIAS = true;
    __device__ __forceinline__ f32x4 apply(f32x4 v, float4 b) const {
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        if (gelu) {
            if constexpr (kSplit<T>) {      // f16x2 keeps fp32-class outputs: libm erff, as linear_kernel's fp32 / f16x2 epilogues
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752440f));
                return v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // erf to ~1.5e-7 absolute (Abramowitz & Stegun 7.1.26), as linear_kernel's 16-bit epilogue
                const float x = fabsf(v[i]) * 0.70710678118654752440f;
                const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
                const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
                const float e = 1.0f - poly * __expf(-x * x);
                v[i] = 0.5f * v[i] * (1.0f + copysignf(e, v[i]));
            }
        }
        return v;
    }
};
struct BigPartial {    // part[z][m][n] = raw fp32 sums of k slice z
    typedef float Out;
    float* out;
    int ldo;
    static constexpr bool HAS_BIAS = false;
    __device__ __forceinline__ f32x4 apply(f32x4 v, float4) const { return v; }
};

#ifdef VITVS_PROBE
// probe builds only (tools/big_ops fixed): what a launch's fixed cost is made of.  bit 0: the epilogue runs but its global stores are
// not issued (the output burst); per workgroup, the cycle counter at entry, after the prologue's first landing and at the end of
// the k-loop goes to g_big_probe (3 words per workgroup) when it is set.
__device__ int g_big_probe_flags;
__device__ unsigned long long* g_big_probe;
extern "C" __attribute__((visibility("default"))) int vitvs_debug_set_big_probe(int flags, void* stamps) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_big_probe_flags), &flags, sizeof(flags)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_big_probe), &stamps, sizeof(stamps)) == hipSuccess ? 0 : -1;
}
#endif

template <typename T, class Tile, class Epi>
__global__ __launch_bounds__(512, 2) void linear_big_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                           typename Epi::Out* __restrict__ out, const float* __restrict__ bias,
                                                           int M, int N, int K, int flags, int slots) {
    // flags: [31:24] k-tiles per split-K slice, [23:16] column tiles, [15:8] slices, [5:1] e (f16x2: the weights carry 2^e, the sums
    // leave multiplied by 2^-e), [0] gelu; slots: [15:0] workgroups in
    // the grid (a multiple of 8, at most one per CU), [31:16] XCD map (below).  PERSISTENT workgroups: the slots / 8
    // workgroups of an XCD walk that XCD's tile list with stride slots / 8; a tile's output stores drain while the
    // workgroup already requests the next tile's operands.
    constexpr int MT = Tile::MT, NT = Tile::NT, WGN = Tile::WGN;
    constexpr int LA = Tile::LA, LB = Tile::LB, FULL = 2 * LA + 2 * LB;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned fl = (unsigned)flags;
    const int nk = (int)(fl >> 24), nx = (int)((fl >> 16) & 255), nz = (int)((fl >> 8) & 255);
    const int ny = (M + Tile::BM - 1) / Tile::BM;
    // Tile list of this XCD (workgroup i runs on XCD i % 8).  `xmap` = 0: the x-th eighth of the (slice, row tile, column
    // tile) list, column fastest — balanced to one tile, used whenever every CU has at most one tile.  `xmap` = XR in
    // {1, 2, 4, 8} (several tiles per CU): the XCDs form an XR x (8 / XR) grid over (row-slices, column tiles) and each owns
    // one block, so that what its CUs stream concurrently is a few A panels and a few W panels that stay in its 4 MB L2
    // (6274 x 3072 on 256x128 tiles: every XCD re-read all 4.7 MB of W on each of its three passes, 108 MB fetched for
    // 14 MB of operands; as a 4 x 2 grid: 7 row panels + 12 column panels per XCD).
    const int tiles = nx * ny * nz;
    const int xcd = (int)(blockIdx.x & 7), local = (int)(blockIdx.x >> 3);
    const int stride = (slots & 0xffff) >> 3, xmap = slots >> 16;
    int blk_r0 = 0, blk_c0 = 0, blk_cols = nx, blk_tiles;
    if (xmap == 0) {
        const int per = (tiles + 7) >> 3;
        blk_r0 = xcd * per;                                   // (used as a linear offset in this mode)
        blk_tiles = min(per, tiles - xcd * per);
    } else {
        const int xc_n = 8 / xmap, xr_i = xcd / xc_n, xc_i = xcd - xr_i * xc_n, R = ny * nz;
        blk_r0 = (R * xr_i) / xmap;
        blk_c0 = (nx * xc_i) / xc_n;
        blk_cols = (nx * (xc_i + 1)) / xc_n - blk_c0;
        blk_tiles = ((R * (xr_i + 1)) / xmap - blk_r0) * blk_cols;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave - wr * WGN;     // waves 0 .. 3 and 4 .. 7 are the two staggered groups
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(A);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);
    bool first_tile = true;
    VITVS_IF_PROBE(
        const unsigned long long probe_t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long probe_t1 = 0, probe_t2 = 0;
        const bool probe_nostore = (g_big_probe_flags & 1) != 0;
    )
  for (int it = local; it < blk_tiles; it += stride) {
    int tz, ty, tx;
    if (xmap == 0) {
        const int lin = blk_r0 + it;
        tz = lin / (nx * ny);
        const int rem = lin - tz * nx * ny;
        ty = rem / nx;
        tx = rem - ty * nx;
    } else {
        const int rr = blk_r0 + it / blk_cols;
        tx = blk_c0 + (it - (it / blk_cols) * blk_cols);
        tz = rr / ny;
        ty = rr - tz * ny;
    }
    const int m0 = ty * Tile::BM, n0 = tx * Tile::BN, k0 = tz * nk * 64;
    if (!first_tile) {
        // every wave has read its epilogue image back (its stores are issued): the ring may be overwritten
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    first_tile = false;

    // ---- LDS-DMA sources: per slot kind (A0, A1, B0, B1) and copy j, a 32-bit byte offset from A / W
    unsigned offA[2][LA], offB[2][LB];
    {
        const int r8 = lane >> 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int rs = copy_group<Tile::GA, LA>(wave, j) * 8 + r8;        // row inside the slot
                const int w_ = rs / Tile::AH, ml = rs - w_ * Tile::AH;
                const int row = min(m0 + w_ * (MT * 16) + h * Tile::AH + ml, M - 1);
                const int c = (lane & 7) ^ ((rs >> 1) & 7);
                offA[h][j] = (unsigned)row * (unsigned)K * (unsigned)sizeof(T) + (unsigned)k0 * (unsigned)sizeof(T) + c * 16;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int rs = copy_group<Tile::GB, LB>(wave, j) * 8 + r8;
                const int w_ = rs / Tile::BH, nl = rs - w_ * Tile::BH;
                const int row = n0 + w_ * (NT * 16) + h * Tile::BH + nl;          // N is a multiple of BN
                const int c = (lane & 7) ^ ((rs >> 1) & 7);
                offB[h][j] = (unsigned)row * (unsigned)K * (unsigned)sizeof(T) + (unsigned)k0 * (unsigned)sizeof(T) + c * 16;
            }
        }
    }
    auto slot = [&](int stage, int kind) -> unsigned char* {   // kind: 0 A0, 1 A1, 2 B0, 3 B1
        return smem + stage * Tile::STAGE + (kind < 2 ? kind * Tile::A_SLOT : 2 * Tile::A_SLOT + (kind - 2) * Tile::B_SLOT);
    };
    auto issue_a = [&](int h, int kt) {
        unsigned char* dst = slot(kt & 1, h);
#pragma unroll
        for (int j = 0; j < LA; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(Ab + (offA[h][j] + (unsigned)kt * 128u)),
                                             (lds_ptr)(dst + copy_group<Tile::GA, LA>(wave, j) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int h, int kt) {
        unsigned char* dst = slot(kt & 1, 2 + h);
#pragma unroll
        for (int j = 0; j < LB; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(Wb + (offB[h][j] + (unsigned)kt * 128u)),
                                             (lds_ptr)(dst + copy_group<Tile::GB, LB>(wave, j) * 1024), 16, 0, 0);
    };

    Epi epi;
    epi.out = out;
    epi.ldo = N;
    if constexpr (!__is_same(Epi, BigPartial)) { epi.bias = bias; epi.gelu = (int)(fl & 1u); }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the tile's bias values go to LDS behind the ring.  Requested BEFORE the prologue's copies: vector-memory operations
    // retire in order, so the prologue's counted wait below (all but the FULL youngest copies) covers this load without
    // waiting for anything it would not wait for anyway — where the epilogue used to start with a dependent global load of
    // its column terms (1 - 1.3 us per tile between the last k-tile and the first output store: profiles/r04_notes.md section 4).
    // Inline asm: for an ordinary load hipcc would drain every LDS-DMA copy in flight at the load's first use.
    // The asm's output is defined, as far as hipcc knows, the moment it is issued, while the data lands at the counted wait
    // below: EVERY lane issues the load (address clamped into the tile's columns) into an early-clobber output, so there is no
    // conditional definition whose merge with an initial value (a register copy before the data has landed) hipcc could place
    // between the two; the value is next named behind the wait.
    constexpr bool BIAS_LDS = Tile::bias_in_lds(Epi::HAS_BIAS);
    u32x4 bias_raw;
    const bool bias_lane = BIAS_LDS && wave == 0 && lane < Tile::BN / 4;
    if constexpr (BIAS_LDS) {
        const float* bsrc = bias + n0 + 4 * min(lane, Tile::BN / 4 - 1);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bias_raw) : "v"(bsrc) : "memory");
    }
    // ---- prologue: k-tile 0 whole, k-tile 1's first halves, in the steady-state order
    issue_a(0, 0); issue_b(0, 0); issue_b(1, 0); issue_a(1, 0);
    issue_a(0, 1); issue_b(0, 1);
    wait_vmcnt<FULL>();                                  // A0(0), B0(0) have landed (this wave's copies)
    if constexpr (BIAS_LDS) {
        asm volatile("" : "+v"(bias_raw));               // (the wait above retired the load: it is older than every copy)
        if (bias_lane) *reinterpret_cast<u32x4*>(smem + Tile::BIAS_OFFSET + 16 * lane) = bias_raw;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in LDS before the barriers that follow
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                        // ... and every other wave's
    if (wave >= 4) __builtin_amdgcn_s_barrier();         // stagger: the second group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    VITVS_IF_PROBE(if (probe_t1 == 0) probe_t1 = __builtin_amdgcn_s_memrealtime();)

    // fragment registers: the current A half, both B halves
    u32x4 xa[MT / 2][2], wb[2][NT / 2][2];
    const int a_row = wr * Tile::AH + l15, b_row = wc * Tile::BH + l15;
    auto read_a = [&](int stage, int h) {
        const unsigned char* s = slot(stage, h);
#pragma unroll
        for (int mi = 0; mi < MT / 2; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xa[mi][ks] = *reinterpret_cast<const u32x4*>(s + tile128_off(a_row + mi * 16, 4 * ks + g));
    };
    auto read_b = [&](int stage, int h) {
        const unsigned char* s = slot(stage, 2 + h);
#pragma unroll
        for (int ni = 0; ni < NT / 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wb[h][ni][ks] = *reinterpret_cast<const u32x4*>(s + tile128_off(b_row + ni * 16, 4 * ks + g));
    };
    auto mfma_quadrant = [&](int ha, int hb) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (kSplit<T>) {
            // f16x2: fragment 0 = the hi halves of the k-tile's 32 k, fragment 1 = the lo halves (common.h): lo.hi + hi.lo, then
            // hi.hi; term-outer so that the three MFMAs into one accumulator are a quadrant's other tiles apart
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int ni = 0; ni < NT / 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MT / 2; ++mi)
                        acc[hb * (NT / 2) + ni][ha * (MT / 2) + mi] =
                            mma_chunk<f16>(acc[hb * (NT / 2) + ni][ha * (MT / 2) + mi], wb[hb][ni][term == 0 ? 1 : 0], xa[mi][term == 1 ? 1 : 0]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < NT / 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MT / 2; ++mi)
                        acc[hb * (NT / 2) + ni][ha * (MT / 2) + mi] =
                            mma_chunk<T>(acc[hb * (NT / 2) + ni][ha * (MT / 2) + mi], wb[hb][ni][ks], xa[mi][ks]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // one phase = LOAD segment | barrier | MFMA segment | barrier
#define VITVS_BIG_PHASE(READS, ISSUE, WAITN, HA, HB)                 \
    do {                                                              \
        READS;                                                        \
        ISSUE;                                                        \
        wait_vmcnt<WAITN>();                                          \
        __builtin_amdgcn_sched_barrier(0);                            \
        __builtin_amdgcn_s_barrier();                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        mfma_quadrant(HA, HB);                                        \
        __builtin_amdgcn_sched_barrier(0);                            \
        __builtin_amdgcn_s_barrier();                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)

    // k_tile(j): counts of copies that may stay in flight at each phase's wait (see the header):
    //   phase 0 retires B1(j): younger copies A1(j) + k-tile j+1's A0, B0, B1            -> FULL, or LA when j + 1 == nk
    //   phase 1 retires A1(j): younger copies k-tile j+1's A0, B0, B1, A1                -> FULL, or 0
    //   phase 3 retires A0(j+1), B0(j+1): younger B1(j+1), A1(j+1), A0(j+2), B0(j+2)     -> FULL, or LA + LB when j + 2 == nk
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        const int st = kt & 1;
        VITVS_BIG_PHASE((read_a(st, 0), read_b(st, 0)), issue_b(1, kt + 1), FULL, 0, 0);
        VITVS_BIG_PHASE(read_b(st, 1), issue_a(1, kt + 1), FULL, 0, 1);
        VITVS_BIG_PHASE(read_a(st, 1), issue_a(0, kt + 2), FULL, 1, 1);
        VITVS_BIG_PHASE((void)0, issue_b(0, kt + 2), FULL, 1, 0);
    }
    {   // kt == nk - 2: nothing left to request in phases 2 and 3
        const int st = kt & 1;
        VITVS_BIG_PHASE((read_a(st, 0), read_b(st, 0)), issue_b(1, kt + 1), FULL, 0, 0);
        VITVS_BIG_PHASE(read_b(st, 1), issue_a(1, kt + 1), FULL, 0, 1);
        VITVS_BIG_PHASE(read_a(st, 1), (void)0, FULL, 1, 1);
        VITVS_BIG_PHASE((void)0, (void)0, LA + LB, 1, 0);
        ++kt;
    }
    {   // kt == nk - 1
        const int st = kt & 1;
        VITVS_BIG_PHASE((read_a(st, 0), read_b(st, 0)), (void)0, LA, 0, 0);
        VITVS_BIG_PHASE(read_b(st, 1), (void)0, 0, 0, 1);
        VITVS_BIG_PHASE(read_a(st, 1), (void)0, 0, 1, 1);
        VITVS_BIG_PHASE((void)0, (void)0, 0, 1, 0);
    }
#undef VITVS_BIG_PHASE
    if (wave < 4) __builtin_amdgcn_s_barrier();          // the first group's matching extra barrier
    __builtin_amdgcn_sched_barrier(0);
    VITVS_IF_PROBE(probe_t2 = __builtin_amdgcn_s_memrealtime();)
    // every wave has passed its last LDS read and every copy has landed (vmcnt 0 above): the ring is free

    // ---- epilogue through a wave-private LDS image, then whole-row stores
    // (lane-dependent addresses are rebuilt from an opaque copy of the lane id: hoisted out of the tile loop they
    // would be spilled around the k-loop, and a spill reload's vmcnt(0) serialises the LDS-DMA prologue)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15e = lane_e & 15, ge = lane_e >> 4;
    typedef typename Epi::Out O;
    constexpr int ES = kSplit<O> ? 4 : (int)sizeof(O);             // bytes per logical output column (f16x2: an fp16 hi / lo pair)
    if constexpr (kSplit<T>) {
        const float wsc = __uint_as_float((127u - ((fl >> 1) & 31u)) << 23);   // 2^-e, exact
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[ni][mi] *= wsc;
    }
    // the wave's NT column tiles leave in groups of NTG whose row segment is a power-of-two number of 16-byte chunks
    // (NT = 4: one group of 128 / 256 bytes per row; NT = 6, the 192-column tile: three groups of 64 / 128 bytes)
    constexpr int NTG = (NT % 4 == 0) ? 4 : 2, NCG = NT / NTG;
    constexpr int ROW_BYTES = NTG * 16 * ES;
    constexpr int CHUNKS = ROW_BYTES / 16;                         // 16-byte chunks per row (a power of two)
    constexpr int PASS_MT = (MT * 16 * ROW_BYTES <= Tile::WAVE_REGION) ? MT : MT / 2;
    static_assert(PASS_MT * 16 * ROW_BYTES <= Tile::WAVE_REGION, "epilogue image does not fit");
    static_assert((CHUNKS & (CHUNKS - 1)) == 0 && CHUNKS <= 64 && NT % NTG == 0, "row chunks");
    // per-column terms (bias) from the LDS copy the prologue left behind the ring (registers the k-loop cannot spare)
    float4 col[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        if constexpr (BIAS_LDS) col[ni] = *reinterpret_cast<const float4*>(smem + Tile::BIAS_OFFSET + 4 * (wc * (NT * 16) + ni * 16 + 4 * ge));
        else if constexpr (Epi::HAS_BIAS) col[ni] = *reinterpret_cast<const float4*>(bias + n0 + wc * (NT * 16) + ni * 16 + 4 * ge);   // (every copy has landed: an ordinary load)
        else col[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    unsigned char* img = smem + wave * Tile::WAVE_REGION;
    unsigned char* obase = reinterpret_cast<unsigned char*>(out) + (size_t)tz * M * N * ES;
    const int wm0 = m0 + wr * (MT * 16), wn0 = n0 + wc * (NT * 16);
    constexpr int ROWS_PER_INST = 64 / CHUNKS;                     // rows one 64-lane 16-byte access covers
#pragma unroll
    for (int pass = 0; pass < MT / PASS_MT; ++pass) {
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
        for (int nl = 0; nl < NTG; ++nl)
#pragma unroll
            for (int mp = 0; mp < PASS_MT; ++mp) {
                const int mi = pass * PASS_MT + mp, ni = cg * NTG + nl;
                const f32x4 v = epi.apply(acc[ni][mi], col[ni]);
                const int row = mp * 16 + l15e;
                if constexpr (kSplit<O>) {
                    // the image row has the layout of memory: per 32 columns [hi | lo] (128 bytes = 8 chunks); this lane's 4
                    // columns nl * 16 + 4 ge .. + 3 are 8 bytes of a hi chunk and 8 bytes of the lo chunk 4 further on
                    const int c = nl * 16 + 4 * ge;
                    const int chunk = ((c >> 5) << 3) + ((c & 31) >> 3);
                    const Split4 h = split4(v);
                    *reinterpret_cast<f16x4*>(img + row * ROW_BYTES + ((chunk ^ (row & (CHUNKS - 1))) << 4) + (ge & 1) * 8) = h.hi;
                    *reinterpret_cast<f16x4*>(img + row * ROW_BYTES + (((chunk + 4) ^ (row & (CHUNKS - 1))) << 4) + (ge & 1) * 8) = h.lo;
                } else if constexpr (ES == 4) {
                    const int chunk = nl * 4 + ge;
                    *reinterpret_cast<f32x4*>(img + row * ROW_BYTES + ((chunk ^ (row & (CHUNKS - 1))) << 4)) = v;
                } else {
                    const int chunk = nl * 2 + (ge >> 1);
                    const typename Vec16<O>::x4 h = {(O)v[0], (O)v[1], (O)v[2], (O)v[3]};
                    *reinterpret_cast<typename Vec16<O>::x4*>(img + row * ROW_BYTES + ((chunk ^ (row & (CHUNKS - 1))) << 4) + (ge & 1) * 8) = h;
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the image is this wave's own: no barrier
        __builtin_amdgcn_sched_barrier(0);
        // whole row segments back from the image, 8 reads in flight before their stores
        // (the batch must divide the row-instruction count: 96 rows x 128 bytes are 12 instructions, read 6 + 6, not 8 + 8)
        constexpr int NI = PASS_MT * 16 / ROWS_PER_INST;
        constexpr int BATCH = NI < 8 ? NI : (NI % 8 == 0 ? 8 : (NI % 6 == 0 ? 6 : (NI % 4 == 0 ? 4 : (NI % 2 == 0 ? 2 : 1))));
        static_assert(NI % BATCH == 0 && (PASS_MT * 16) % ROWS_PER_INST == 0, "epilogue batches cover the image exactly");
#pragma unroll
        for (int b0 = 0; b0 < NI; b0 += BATCH) {
            u32x4 rowv[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int row = (b0 + i) * ROWS_PER_INST + lane_e / CHUNKS, chunk = lane_e & (CHUNKS - 1);
                rowv[i] = *reinterpret_cast<const u32x4*>(img + row * ROW_BYTES + ((chunk ^ (row & (CHUNKS - 1))) << 4));
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int row = (b0 + i) * ROWS_PER_INST + lane_e / CHUNKS, chunk = lane_e & (CHUNKS - 1);
                const int m = wm0 + pass * PASS_MT * 16 + row;
                // write-through (sc1): the rows leave for memory as they are stored instead of waiting, dirty in this XCD's
                // L2, for the write-back at the kernel boundary (measured -3 ... -7 % on these launches: all of a tile's
                // output is produced at its very end, so there is nothing for a write-back cache to merge)
                VITVS_IF_PROBE(if (probe_nostore) { asm volatile("" :: "v"(rowv[i])); continue; })
                if (m < M)
                    store_out16<true>(obase + ((size_t)m * N + wn0 + cg * NTG * 16) * ES + chunk * 16, rowv[i]);
            }
        }
        if (pass + 1 < MT / PASS_MT || cg + 1 < NCG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }   // next tile of this workgroup
    VITVS_IF_PROBE(
        if (g_big_probe && tid == 0) {
            unsigned long long* dst = g_big_probe + 4 * (size_t)blockIdx.x;
            dst[0] = probe_t0; dst[1] = probe_t1; dst[2] = probe_t2; dst[3] = __builtin_amdgcn_s_memrealtime();
        }
    )
}


// ---------------------------------------------------------------------------------------------------- launch side
typedef BigTile<2, 4, 8, 4> Tile256x256;
typedef BigTile<4, 2, 4, 4> Tile256x128;
typedef BigTile<4, 2, 4, 6> Tile256x192;
typedef BigTile<2, 4, 6, 2> Tile192x128;      // selected as width code 1192 (192 rows x 128 columns)
typedef BigTile<2, 4, 6, 4> Tile192x256;      // width code 1256 (192 rows x 256 columns)

template <typename T, class Tile, class Epi>
static int launch_big_one(const T* A, const T* W, typename Epi::Out* out, const float* bias, int M, int N, int K, int splits,
                          int gelu, hipStream_t stream, int wexp = 0) {
    // (f16x2: K counts fp16 per row, two per logical k; a k-tile is 32 logical k)
    static std::atomic<unsigned long long> raised{0};
    constexpr int LDS_BYTES = Tile::lds_bytes(Epi::HAS_BIAS);
    if (raise_lds_limit(reinterpret_cast<const void*>(&linear_big_kernel<T, Tile, Epi>), LDS_BYTES, raised)) return -1;
    const int nx = N / Tile::BN, ny = (M + Tile::BM - 1) / Tile::BM, nk = K / splits / 64;
    if (N % Tile::BN || K % (splits * 64) || nk < 2 || nk > 255 || nx > 255 || splits > 255 || wexp < 0 || wexp > 31) return -2;
    const long tiles = (long)nx * ny * splits;
    const int slots = (int)std::min<long>(8 * ((tiles + 7) / 8), 256);       // one workgroup per CU at most
    // XCD map (see the kernel): blocks of an XR x (8 / XR) XCD grid when CUs walk several tiles; among the grids whose
    // largest block needs no more passes than the balanced 1-D split, the one with the fewest operand bytes per XCD
    int xmap = 0;
    if (tiles > 256) {
        const long R = (long)ny * splits, wgs = slots / 8;
        const long passes_1d = ((tiles + 7) / 8 + wgs - 1) / wgs;
        double best = 1e30;
        for (int xr : {1, 2, 4, 8}) {
            const long xc = 8 / xr, rows = (R + xr - 1) / xr, cols = (nx + xc - 1) / xc;
            if ((rows * cols + wgs - 1) / wgs > passes_1d) continue;
            const double bytes = (double)rows * Tile::BM + (double)cols * Tile::BN * std::max<long>(1, (rows + ny - 1) / ny);
            if (bytes < best) { best = bytes; xmap = xr; }
        }
    }
    launch(linear_big_kernel<T, Tile, Epi>, dim3((unsigned)slots), dim3(512), LDS_BYTES, stream, A, W, out, bias, M, N, K,
           (int)(((unsigned)nk << 24) | ((unsigned)nx << 16) | ((unsigned)splits << 8) | ((unsigned)wexp << 1) | (unsigned)(gelu & 1)),
           (int)((unsigned)slots | ((unsigned)xmap << 16)));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Which tile family for a many-row problem.  Measured on MI355X (tools/big_ops [mid], random bf16 data, profiles/r02_big_ops*.txt):
// in its k-loop the 256x256 tile sustains ~1.5 PFLOP/s chip-wide, so what decides is how many tiles the busiest CU walks
// (workgroups are persistent, one per CU) and the fixed cost per tile (operand latency at the start, the output burst at the
// end: ~8 us at 6274 x 2304) — the rule below is that model with the relative tile costs measured.
// Returns 0 (use gemm.hip), the column width 256, 192 or 128 of a 256-row tile, or 1192 / 1256 for the 192 x 128 / 192 x 256 tile.
int big_tile_width(Precision p, int M, int N, int K, int splits, bool partial) {
    if (p == PREC_F32 || splits < 1) return 0;
    // f16x2 runs the same kernels on rows of 2 K fp16 (a k-tile = 32 logical k, three MFMAs per k-step): the tile rule below is
    // about rows, columns and rounds of workgroups, which do not change; only the k-tile count does
    const int kt = (p == PREC_X2 ? 2 : 1) * K;
    if (kt % (splits * 64) || kt / splits < 128 || kt / splits / 64 > 255) return 0;
    // The 64-row tiles of gemm.hip keep the layers they cover in ONE round of <= 256 workgroups (788 x 2304: 7.9 us there,
    // 11.9 us on 256 x 128 tiles); where they need a second round the 256-row tiles win from 64 tiles up (985 x 2304: 18.4 vs
    // 11.9 us).  Between the tile families of this file the busiest CU's share decides (below).
    const long mt = (M + 63) / 64;
    for (int c : {128, 96, 64}) {
        if (partial && c != 64) continue;                         // the partial-sum kernels of gemm.hip are 64 wide
        if (N % c == 0 && mt * (N / c) * splits <= 256) return 0;
    }
    // A little over one workgroup per CU, the 64 x 128 tiles of gemm.hip on their 2-stage ring (48 KB: three resident per CU, so
    // still one round) are ahead of every tile of this file: 788 x 3072 x 768 (312 workgroups) 11.4-12.5 us against 13.3 on
    // 192 x 128 tiles, 985 x 2304 (288) 11.3 / 11.8; end to end +1.6 % at 2 frame pairs.  From ~340 workgroups on the forward
    // does not gain (3 pairs, qkv 1182 x 2304: -0.4 % end to end although 11.7 / 11.8 us alone), so the rule stops at 320.
    if (!partial && splits == 1 && N % 128 == 0 && mt * (N / 128) <= 320) return 0;
    const long ny = (M + 255) / 256;
    if (N % 128 != 0 || N / 128 > 255) return 0;
    const long t128 = ny * (N / 128) * splits;
    if (t128 < (partial ? 96 : 64)) return 0;
    // relative cost of one tile (k-loop share + the same fixed cost): 256 x 256 = 1, 256 x 192 = 0.85, 256 x 128 = 0.62
    // (3152 x 3072: 156 tiles of 256 x 256 25.1 us, 208 of 256 x 192 21.4 us, 312 of 256 x 128 29.6 us;
    //  6274 x 3072: 300 -> 51.6, 400 -> 41.9, 600 -> 44.6 us; 2740 x 3072 x 1024: 132 -> 23.9, 176 -> 20.6, 264 -> 30.3 us)
    auto rounds = [](long tiles) { return (double)((tiles + 255) / 256); };
    int best = 128;
    double cost = 0.62 * rounds(t128);
    if (N % 192 == 0 && 0.85 * rounds(ny * (N / 192) * splits) < cost) { best = 192; cost = 0.85 * rounds(ny * (N / 192) * splits); }
    if (N % 256 == 0 && rounds(ny * (N / 256) * splits) < cost) { best = 256; cost = rounds(ny * (N / 256) * splits); }
    // 192-row x 128-column tiles (code 1192): more, smaller tiles for the narrow layers — 6274 x 768 x 3072: 198 tiles 34.8 us
    // against 150 of 256 x 128 37.4 us; 2740 x 1024 x 4096 in 2 K slices: 240 tiles 25.7 us against 176 -> 27.8 us; a tile costs
    // 0.58 (its waves' 96 x 32 sub-tiles read more LDS per FLOP than 64 x 64 ones, so it only pays while it stays in one round
    // where 256 x 128 leaves CUs idle: 3152 x 768 x 3072 in 3 slices, 306 tiles, 27.5 us against 234 -> 19.3 us)
    const long t1192 = (long)((M + 191) / 192) * (N / 128) * splits;
    // (also ahead on wide layers while in one round: 788 .. 1576 x 3072: 13.2 .. 14.1 us against 14.8 .. 15.7; two rounds from
    //  1970 rows on: 24.8 against 16.2 us)
    if (0.58 * rounds(t1192) < cost) { best = 1192; cost = 0.58 * rounds(t1192); }
    // 192 x 256 (code 1256), priced 0.9: 2740 x 4096 x 1024 (N is not a multiple of 192): 240 tiles 27.6 us against 176 of
    // 256 x 256 -> 29.8 us; level with 256 x 192 where that applies (3152 x 3072: 21.7 / 21.6 us)
    if (N % 256 == 0 && 0.9 * rounds((long)((M + 191) / 192) * (N / 256) * splits) < cost) best = 1256;
    // Everything above balances ONE launch over the chip.  Beside other queues' launches (vitvs_set_option "in_flight") the
    // other queues' workgroups fill what a launch leaves idle, and the tile with the fewest operand bytes per FLOP wins: three
    // updates in flight, same box, 256 x 256 for every layer it divides: 8 / 6 / 4 pairs 8607 -> 9249 / 7980 -> 8730 / 7385 -> 7760
    // updates/s, ViT-B/8 448² 438 -> 460, ViT-L/14 518² 731 -> 793 (3 pairs, 1182 rows = 4.6 row tiles: 6787 -> 6664, hence the bound).
    if (g_updates_in_flight >= 2 && N % 256 == 0 && M >= 1536) best = 256;
    return best;
}

template <typename T>
static int launch_big_t(int bn, const T* A, const T* W, const float* bias, void* out, int M, int N, int K, int splits, int gelu,
                        bool partial, hipStream_t stream, int wexp = 0) {
    if (bn == 1256) {
        if (partial) return launch_big_one<T, Tile192x256, BigPartial>(A, W, (float*)out, nullptr, M, N, K, splits, 0, stream, wexp);
        return launch_big_one<T, Tile192x256, BigStore<T>>(A, W, (T*)out, bias, M, N, K, 1, gelu, stream, wexp);
    }
    if (bn == 1192) {
        if (partial) return launch_big_one<T, Tile192x128, BigPartial>(A, W, (float*)out, nullptr, M, N, K, splits, 0, stream, wexp);
        return launch_big_one<T, Tile192x128, BigStore<T>>(A, W, (T*)out, bias, M, N, K, 1, gelu, stream, wexp);
    }
    if (partial) {
        if (bn == 256) return launch_big_one<T, Tile256x256, BigPartial>(A, W, (float*)out, nullptr, M, N, K, splits, 0, stream, wexp);
        if (bn == 192) return launch_big_one<T, Tile256x192, BigPartial>(A, W, (float*)out, nullptr, M, N, K, splits, 0, stream, wexp);
        return launch_big_one<T, Tile256x128, BigPartial>(A, W, (float*)out, nullptr, M, N, K, splits, 0, stream, wexp);
    }
    if (bn == 256) return launch_big_one<T, Tile256x256, BigStore<T>>(A, W, (T*)out, bias, M, N, K, 1, gelu, stream, wexp);
    if (bn == 192) return launch_big_one<T, Tile256x192, BigStore<T>>(A, W, (T*)out, bias, M, N, K, 1, gelu, stream, wexp);
    return launch_big_one<T, Tile256x128, BigStore<T>>(A, W, (T*)out, bias, M, N, K, 1, gelu, stream, wexp);
}

int launch_linear_big(Precision p, int bn, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                      int splits, int gelu, bool partial, hipStream_t stream, int wexp) {
    if ((long long)M * K * (long long)elem_size(p) >= (1ll << 32) || (long long)N * K * (long long)elem_size(p) >= (1ll << 32)) return -2;   // 32-bit operand offsets
    if (p == PREC_X2) return launch_big_t<hx2>(bn, (const hx2*)A, (const hx2*)W, bias, out, M, N, 2 * K, splits, gelu, partial, stream, wexp);
    if (p == PREC_BF16) return launch_big_t<bf16>(bn, (const bf16*)A, (const bf16*)W, bias, out, M, N, K, splits, gelu, partial, stream);
    if (p == PREC_F16) return launch_big_t<f16>(bn, (const f16*)A, (const f16*)W, bias, out, M, N, K, splits, gelu, partial, stream);
    return -2;
}

}  // namespace vitvs
