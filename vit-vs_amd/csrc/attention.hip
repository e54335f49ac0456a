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
//
// bf16 kernel, KS = 2 / 4 ("key split"): at one frame pair there are only 96 workgroups of 4 key tiles, and
// the tile loop is a serial chain (MFMA -> softmax VALU -> MFMA) at one wave per SIMD.  With KS key groups a
// workgroup has 4 KS waves: group k takes the k-th share of the key tiles (own LDS stage each), and the
// partial (max, sum, O) states are merged through LDS at the end in a fixed order — 1/KS of the chain
// length, KS waves per SIMD.  Long sequences (many workgroups) use KS = 1.  KS = 2 is what is launched:
// KS = 4 (16 waves, one key tile per group at 197 tokens) was measured 1 % slower end to end.
// (A variant that kept all K/V of a head resident in LDS, filled by LDS-DMA with counted waits, was
// measured 1 % slower end to end than this streaming form and was removed: profiles/r01_notes.md.)
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace vitvs {

constexpr float kScaleLog2e = 0.125f * 1.44269504088896340736f;  // hd^-0.5 * log2(e), hd = 64

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32; exp2(-inf) = 0
__device__ __forceinline__ void wait_vmcnt4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }

// ------------------------------------------------------------------------------------ bf16
template <typename HT, int KS>
__global__ __launch_bounds__(256 * KS) void attention_16_kernel(const HT* __restrict__ qkv, HT* __restrict__ out,
                                                                int N, int D) {
    typedef typename Vec16<HT>::x8 hx8;
    typedef typename Vec16<HT>::x4 hx4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = tid >> 6, kgp = wave_all >> 2, wave = wave_all & 3, tl = tid & 255;
    unsigned char* ldsK = smem + kgp * (2 * 64 * 128);
    unsigned char* ldsV = ldsK + 64 * 128;
    const int qi = lane & 15, g = lane >> 4;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u8* vtr = (lds_u8*)ldsV + (4 * g + (qi >> 2)) * 128 + 8 * (qi & 3);   // this lane's corner of a transposed V read
    const int img = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const size_t ld = (size_t)3 * D;
    const HT* base = qkv + (size_t)img * N * ld + h * 64;
    const HT* Kp = base + D;
    const HT* Vp = base + 2 * D;

    const int q = q0 + 16 * wave + qi;
    const int qrow = min(q, N - 1);
    hx8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = __builtin_bit_cast(hx8, *reinterpret_cast<const u32x4*>(base + (size_t)qrow * ld + 32 * s + 8 * g));

    f32x4 acc_o[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) acc_o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int srow = tl >> 3, schunk = tl & 7;
    u32x4 rk[2], rv[2];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(kb + srow + 32 * i, N - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(Kp + (size_t)key * ld + schunk * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(Vp + (size_t)key * ld + schunk * 8);
        }
    };
    const int ntiles = (N + 63) / 64;
    const int per_group = (ntiles + KS - 1) / KS;
    const int first = kgp * per_group;
    gload(min(first, ntiles - 1) * 64);
    for (int tt = 0; tt < per_group; ++tt) {
        const int t = first + tt;
        const int kb = t * 64;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(ldsK + tile128_off(srow + 32 * i, schunk)) = rk[i];
            *reinterpret_cast<u32x4*>(ldsV + (srow + 32 * i) * 128 + schunk * 16) = rv[i];
        }
        __syncthreads();
        if (tt + 1 < per_group) gload(min(t + 1, ntiles - 1) * 64);
        if (t >= ntiles) continue;   // wave-uniform: this key group has run out of tiles (barriers above still taken)

        // S^T tiles: acc_s[t4][r] = S[key = kb + 16*t4 + 4g + r][q]
        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const hx8 kf = __builtin_bit_cast(
                    hx8, *reinterpret_cast<const u32x4*>(ldsK + tile128_off(16 * t4 + qi, 4 * s + g)));
                acc_s[t4] = mfma16(kf, qf[s], acc_s[t4]);
            }
        }
        // softmax on the raw scores: keys beyond N are masked only in the tile that has any (wave-uniform), the
        // hd^-0.5 * log2(e) scale rides in the exp2 argument's FMA, and the running O^T is rescaled only when some
        // query's maximum moved (most tiles after the first few leave it alone)
        if (kb + 64 > N) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + 16 * t4 + 4 * g + r >= N) acc_s[t4][r] = -INFINITY;
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, acc_s[t4][r]);
        mloc = rows_max(mloc) * kScaleLog2e;
        const float m_new = fmaxf(m_run, mloc);
        const float neg_m = -m_new;
        float psum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(__builtin_fmaf(acc_s[t4][r], kScaleLog2e, neg_m));
                acc_s[t4][r] = p;
                psum += p;
            }
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {
            const float alpha = fast_exp2(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int td = 0; td < 4; ++td) acc_o[td] *= alpha;
            m_run = m_new;
        }
        l_run += psum;

        // O^T += V^T P^T ; k-slot (g, j) of step u  <->  key 32u + 16(j>>2) + 4g + (j&3)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            hx8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (HT)acc_s[2 * u + (j >> 2)][j & 3];
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                // hardware-transposed LDS read: lane i of the 16-lane group gets column d0+i of 4 key rows;
                // one lane-dependent LDS address (vtr), everything else is an immediate offset
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vtr + (32 * u) * 128 + td * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vtr + (32 * u + 16) * 128 + td * 32));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc_o[td] = mfma16(__builtin_bit_cast(hx8, v8), pf, acc_o[td]);
            }
        }
    }
    if constexpr (KS >= 2) {
        // merge the key groups' online-softmax states: groups 1 .. KS-1 -> LDS -> group 0 (fixed order);
        // 5 x 16 bytes per lane (O^T accumulators, then max and sum), lane-major
        __syncthreads();
        f32x4* buf = reinterpret_cast<f32x4*>(smem) + (wave * 64 + lane);   // [KS-1][5][256] x 16 B
        if (kgp >= 1) {
            f32x4* mine = buf + (kgp - 1) * 5 * 256;
#pragma unroll
            for (int td = 0; td < 4; ++td) mine[td * 256] = acc_o[td];
            mine[4 * 256] = f32x4{m_run, l_run, 0.f, 0.f};
        }
        __syncthreads();
        if (kgp >= 1) return;
#pragma unroll
        for (int o = 0; o < KS - 1; ++o) {
            const f32x4* other = buf + o * 5 * 256;
            const f32x4 ml = other[4 * 256];
            const float m_tot = fmaxf(m_run, ml[0]);
            const float wa = fast_exp2(m_run - m_tot), wb = fast_exp2(ml[0] - m_tot);
            m_run = m_tot;
            l_run = l_run * wa + ml[1] * wb;
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                const f32x4 ob = other[td * 256];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_o[td][r] = acc_o[td][r] * wa + ob[r] * wb;
            }
        }
    }
    l_run = rows_sum(l_run);
    const float inv = 1.0f / l_run;
    if (q < N) {
        HT* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * g;
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            const hx4 o = {(HT)(acc_o[td][0] * inv), (HT)(acc_o[td][1] * inv), (HT)(acc_o[td][2] * inv), (HT)(acc_o[td][3] * inv)};
            store_out<(KS >= 2)>(dst + 16 * td, o);   // key-split variants only run on small grids
        }
    }
}

// ------------------------------------------------------------------------------------ 16-bit, long sequences
// N >= 512 (448² / 518² inputs: 3137 / 1370 tokens).  The 64-query kernel above is bound by the K / V traffic into LDS
// (every 64 queries re-read the (image, head)'s whole K and V: 0.94 GB per launch at 3137 tokens) and by exposed LDS
// latency (each MFMA waits for the fragment read issued just before it).  Here:
//   * one workgroup = 128 queries, wave w = queries 32 w .. 32 w + 31 as two 16-query tiles that share every K and V
//     fragment read (half the LDS traffic per MFMA, half the K / V traffic per query);
//   * K / V tiles of 64 keys arrive by LDS-DMA into a 3-stage ring with counted waits, one barrier per tile, two tiles
//     in flight across it (no register staging, no ds_write pass);
//   * per tile a wave issues its 8 K fragment reads up front, runs 16 score MFMAs, the two softmaxes, and 16 PV MFMAs
//     whose V fragments (hardware-transposed reads) are shared by both query tiles;
//   * 1-D grid, XCD-aware item order: the query blocks of an (image, head) run on the XCD whose L2 already holds its
//     K and V.
// Same arithmetic as attention_16_kernel (raw-score maximum, scale folded into the exp2 FMA, exact rescale only when a
// maximum moved), so the two agree to rounding.
template <typename HT>
__global__ __launch_bounds__(256, 2) void attention_16_long_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int N,
                                                                   int D, int n_img) {
    typedef typename Vec16<HT>::x8 hx8;
    typedef typename Vec16<HT>::x4 hx4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 2 * 64 * 128;                      // K tile then V tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int H = D >> 6, nqb = (N + 127) >> 7, items = n_img * H * nqb, per = (items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= items) return;
    const int pair = item / nqb;
    const int img = pair / H, h = pair - img * H, q0 = (item - pair * nqb) * 128 + 32 * wave;
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned row_bytes = 6u * (unsigned)D;
    const unsigned head_off = (unsigned)(img * N) * row_bytes + (unsigned)h * 128u;
    const unsigned k_off = head_off + 2u * (unsigned)D, v_off = head_off + 4u * (unsigned)D;

    // this wave's four LDS-DMA copies per key tile: rows 16 wave .. 16 wave + 15 of the K image (swizzled source
    // chunk) and of the V image (linear), 8 rows per copy
    const int r8 = lane >> 3;
    unsigned koff[2], voff[2];
    int krow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        krow[c] = 16 * wave + 8 * c + r8;
        koff[c] = k_off + 16u * (unsigned)((lane & 7) ^ ((krow[c] >> 1) & 7));
        voff[c] = v_off + 16u * (unsigned)(lane & 7);
    }
    const int ntiles = (N + 63) >> 6;
    auto issue = [&](int t) {
        unsigned char* dst = smem + (t % 3) * STAGE + (16 * wave) * 128;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned ro = (unsigned)min(64 * t + krow[c], N - 1) * row_bytes;
            __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (koff[c] + ro)), (lds_ptr)(dst + c * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (voff[c] + ro)), (lds_ptr)(dst + 64 * 128 + c * 1024), 16, 0, 0);
        }
    };
    issue(0);
    if (ntiles > 1) issue(1);
    // Q fragments by ordinary loads AFTER the first copies, and consumed (empty asm) before the loop: hipcc places its
    // wait for an ordinary load at the first use, and inside the tile loop that wait would be vmcnt(0) on every
    // iteration, draining the LDS-DMA ring; here it is one wait in the prologue, which tile 0 needs anyway.
    u32x4 qraw[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned o = head_off + (unsigned)min(q0 + 16 * j + qi, N - 1) * row_bytes + 16u * (unsigned)g;
#pragma unroll
        for (int s = 0; s < 2; ++s) qraw[j][s] = *reinterpret_cast<const u32x4*>(qb + (o + 64u * s));
    }
    asm volatile("" : "+v"(qraw[0][0]), "+v"(qraw[0][1]), "+v"(qraw[1][0]), "+v"(qraw[1][1]));
    hx8 qf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[j][s] = __builtin_bit_cast(hx8, qraw[j][s]);

    f32x4 acc_o[2][4];
    float m_run[2], l_run[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        m_run[j] = -INFINITY;
        l_run[j] = 0.f;
#pragma unroll
        for (int td = 0; td < 4; ++td) acc_o[j][td] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) wait_vmcnt4();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // tile t has landed for every wave; everyone is done with tile t - 1
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntiles) issue(t + 2);
        const unsigned char* ldsK = smem + (t % 3) * STAGE;
        const lds_u8* vtr = (const lds_u8*)(ldsK + 64 * 128) + (4 * g + (qi >> 2)) * 128 + 8 * (qi & 3);
        const int kb = t * 64;
        hx8 kf[4][2];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                kf[t4][s] = __builtin_bit_cast(hx8, *reinterpret_cast<const u32x4*>(ldsK + tile128_off(16 * t4 + qi, 4 * s + g)));
        f32x4 acc_s[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                acc_s[j][t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) acc_s[j][t4] = mfma16(kf[t4][s], qf[j][s], acc_s[j][t4]);
            }
        if (kb + 64 > N) {                              // keys beyond N: only the last tile has any (wave-uniform)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb + 16 * t4 + 4 * g + r >= N) acc_s[j][t4][r] = -INFINITY;
        }
        // V fragments (hardware-transposed reads), requested now and consumed after the softmaxes.  Inline asm: for the
        // ds_read_tr16 builtin hipcc waits vmcnt(0) first (it cannot tell the LDS-DMA copies in flight apart from the
        // image being read), which would drain the ring on every tile; the reads' completion is waited for by hand.
        s16x4 vlo[2][4], vhi[2][4];
        {
            const unsigned va = (unsigned)(size_t)vtr;
#define VITVS_TR(dst, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(OFF))
            VITVS_TR(vlo[0][0], 0);    VITVS_TR(vhi[0][0], 2048);      VITVS_TR(vlo[0][1], 32);   VITVS_TR(vhi[0][1], 2048 + 32);
            VITVS_TR(vlo[0][2], 64);   VITVS_TR(vhi[0][2], 2048 + 64); VITVS_TR(vlo[0][3], 96);   VITVS_TR(vhi[0][3], 2048 + 96);
            VITVS_TR(vlo[1][0], 4096); VITVS_TR(vhi[1][0], 6144);      VITVS_TR(vlo[1][1], 4128); VITVS_TR(vhi[1][1], 6144 + 32);
            VITVS_TR(vlo[1][2], 4160); VITVS_TR(vhi[1][2], 6144 + 64); VITVS_TR(vlo[1][3], 4192); VITVS_TR(vhi[1][3], 6144 + 96);
#undef VITVS_TR
        }
        hx8 pf[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mloc = fmaxf(fmaxf(acc_s[j][0][0], acc_s[j][0][1]), fmaxf(acc_s[j][0][2], acc_s[j][0][3]));
#pragma unroll
            for (int t4 = 1; t4 < 4; ++t4)
                mloc = fmaxf(mloc, fmaxf(fmaxf(acc_s[j][t4][0], acc_s[j][t4][1]), fmaxf(acc_s[j][t4][2], acc_s[j][t4][3])));
            mloc = rows_max(mloc) * kScaleLog2e;
            const float m_new = fmaxf(m_run[j], mloc);
            const float neg_m = -m_new;
            float psum = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(__builtin_fmaf(acc_s[j][t4][r], kScaleLog2e, neg_m));
                    acc_s[j][t4][r] = p;
                    psum += p;
                }
            if (__builtin_amdgcn_ballot_w64(m_new != m_run[j]) != 0ull) {
                const float alpha = fast_exp2(m_run[j] - m_new);
                l_run[j] *= alpha;
#pragma unroll
                for (int td = 0; td < 4; ++td) acc_o[j][td] *= alpha;
                m_run[j] = m_new;
            }
            l_run[j] += psum;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[j][u][jj] = (HT)acc_s[j][2 * u + (jj >> 2)][jj & 3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]), "+v"(vlo[0][2]), "+v"(vhi[0][2]),
                       "+v"(vlo[0][3]), "+v"(vhi[0][3]), "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]),
                       "+v"(vlo[1][2]), "+v"(vhi[1][2]), "+v"(vlo[1][3]), "+v"(vhi[1][3]));
        __builtin_amdgcn_sched_barrier(0);
        // O^T += V^T P^T for both query tiles; k-slot (g, jj) of step u  <->  key 32u + 16(jj>>2) + 4g + (jj&3)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                const s16x4 lo = vlo[u][td], hi = vhi[u][td];
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int j = 0; j < 2; ++j) acc_o[j][td] = mfma16(__builtin_bit_cast(hx8, v8), pf[j][u], acc_o[j][td]);
            }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float inv = 1.0f / rows_sum(l_run[j]);
        const int q = q0 + 16 * j + qi;
        if (q < N) {
            HT* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * g;
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                const hx4 o = {(HT)(acc_o[j][td][0] * inv), (HT)(acc_o[j][td][1] * inv), (HT)(acc_o[j][td][2] * inv),
                               (HT)(acc_o[j][td][3] * inv)};
                store_out<false>(dst + 16 * td, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ bf16, short sequences
// N <= 256 (the 197 tokens of a 224² frame): one workgroup = 16 queries of one (image, head), wave w = key tile w.
// Every wave requests its whole input at entry in one memory round trip — Q and its 64 keys straight into MFMA
// operand registers, its 64 values into a wave-private LDS slab by LDS-DMA (read back transposed) — and runs the
// score -> softmax -> PV chain for ONE tile; the four (max, sum, O) states are merged in parallel, wave w
// finishing head dims 16w .. 16w+15.  No barrier before the merge, a quarter of the 64-query kernel's serial chain.
// Grid: 1-D, 8 * ceil(items / 8) workgroups, item = (image, head, query block).  Workgroup id i runs on XCD i % 8, and
// every XCD has its own L2: XCD x takes the x-th eighth of the items in (image, head)-major order, so the K / V slab of
// an (image, head) — fresh from the qkv launch, i.e. read from memory — is fetched by one XCD (two at a boundary)
// instead of by all eight.
template <typename HT>
__global__ __launch_bounds__(256) void attention_16_short_kernel(const HT* __restrict__ qkv, HT* __restrict__ out,
                                                                 int N, int D, int n_img) {
    typedef typename Vec16<HT>::x8 hx8;
    typedef typename Vec16<HT>::x4 hx4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int H = D >> 6, nqb = (N + 15) >> 4, items = n_img * H * nqb, per = (items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= items) return;
    const int pair = item / nqb;
    const int img = pair / H, h = pair - img * H, q0 = (item - pair * nqb) * 16;
    // Addresses: wave-uniform base pointer + 32-bit byte offsets built from 24-bit multiplies (the kernel is a
    // chain of latencies: ~150 instructions of 64-bit address arithmetic before the first request cost 0.4 us).
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned row_bytes = 6u * (unsigned)D;                                 // 3D bf16 per token row
    const unsigned head_off = (unsigned)(img * N) * row_bytes + (unsigned)h * 128u;   // q of this (image, head)
    const unsigned k_off = head_off + 2u * (unsigned)D, v_off = head_off + 4u * (unsigned)D;
    unsigned char* ldsV = smem + wave * 8192;                                  // [64 keys][128 B], this wave's tile
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u8* vtr = (lds_u8*)ldsV + (4 * g + (qi >> 2)) * 128 + 8 * (qi & 3);   // this lane's corner of a transposed V read
    f32x4* mbuf = reinterpret_cast<f32x4*>(smem + 4 * 8192);                   // [4 waves][5][64 lanes] x 16 B
    const int kb = wave * 64;
    const bool active = kb < N;                                                // wave-uniform
    const int q = q0 + qi;
    const unsigned lane_col = 16u * (unsigned)g;
    hx8 qf[2], kf[4][2];
    {
        const unsigned o = head_off + __umul24((unsigned)min(q, N - 1), row_bytes) + lane_col;
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = __builtin_bit_cast(hx8, *reinterpret_cast<const u32x4*>(qb + (o + 64u * s)));
    }
    if (active) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const unsigned o = k_off + __umul24((unsigned)min(kb + 16 * t4 + qi, N - 1), row_bytes) + lane_col;
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[t4][s] = __builtin_bit_cast(hx8, *reinterpret_cast<const u32x4*>(qb + (o + 64u * s)));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned o = v_off + __umul24((unsigned)min(kb + 8 * j + (lane >> 3), N - 1), row_bytes) + 16u * (lane & 7);
            __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + o), (lds_ptr)(ldsV + j * 1024), 16, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // every request is issued before anything waits on one

    f32x4 acc_o[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) acc_o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    if (active) {
        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) acc_s[t4] = mfma16(kf[t4][s], qf[s], acc_s[t4]);
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
        m_run = rows_max(mloc);               // the tile's first key is valid (kb < N), so this is finite
        float psum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(acc_s[t4][r] - m_run);
                acc_s[t4][r] = p;
                psum += p;
            }
        l_run = rows_sum(psum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's V slab has landed (its own LDS-DMA copies)
        // O^T = V^T P^T ; k-slot (g, j) of step u  <->  key 32u + 16(j>>2) + 4g + (j&3)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            hx8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (HT)acc_s[2 * u + (j >> 2)][j & 3];
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vtr + (32 * u) * 128 + td * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vtr + (32 * u + 16) * 128 + td * 32));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc_o[td] = mfma16(__builtin_bit_cast(hx8, v8), pf, acc_o[td]);
            }
        }
    }
    // parallel merge: every wave publishes its state, wave w finishes head dims 16w .. 16w+15
    f32x4* mine = mbuf + wave * 5 * 64 + lane;
#pragma unroll
    for (int td = 0; td < 4; ++td) mine[td * 64] = acc_o[td];
    mine[4 * 64] = f32x4{m_run, l_run, 0.f, 0.f};
    __syncthreads();
    float m_k[4], l_k[4];
    float m_tot = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 ml = mbuf[(k * 5 + 4) * 64 + lane];
        m_k[k] = ml[0];
        l_k[k] = ml[1];
        m_tot = fmaxf(m_tot, m_k[k]);
    }
    float l_tot = 0.f;
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = fast_exp2(m_k[k] - m_tot);          // 0 for a wave without keys (m = -inf)
        l_tot += l_k[k] * wk;
        const f32x4 ok = mbuf[(k * 5 + wave) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += ok[r] * wk;
    }
    const float inv = 1.0f / l_tot;
    if (q < N) {
        HT* dst = out + ((size_t)img * N + q) * D + h * 64 + 16 * wave + 4 * g;
        const hx4 ob = {(HT)(o[0] * inv), (HT)(o[1] * inv), (HT)(o[2] * inv), (HT)(o[3] * inv)};
        store_out<true>(dst, ob);
    }
}

// ------------------------------------------------------------------------------------ fp32
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int N, int D) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 64 * 256];
    unsigned char* ldsK = smem;
    unsigned char* ldsV = smem + 64 * 256;
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
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + 16 * c + 4 * g);

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
    const int ntiles = (N + 63) / 64;
    gload(0);
    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(ldsK + tile256_off(srow + 16 * i, schunk)) = rk[i];
            *reinterpret_cast<u32x4*>(ldsV + tile256_off(srow + 16 * i, schunk)) = rv[i];
        }
        __syncthreads();
        if (t + 1 < ntiles) gload(kb + 64);

        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 kf = *reinterpret_cast<const float4*>(ldsK + tile256_off(16 * t4 + qi, 4 * c + g));
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
        mloc = rows_max(mloc);
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
                    const float v = *reinterpret_cast<const float*>(ldsV + tile256_off(key, 4 * td + (qi >> 2)) +
                                                                    (qi & 3) * 4);
                    acc_o[td] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, acc_s[t4][r], acc_o[td], 0, 0, 0);
                }
            }
    }
    l_run = rows_sum(l_run);
    const float inv = 1.0f / l_run;
    if (q < N) {
        float* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * g;
#pragma unroll
        for (int td = 0; td < 4; ++td)
            *reinterpret_cast<float4*>(dst + 16 * td) =
                make_float4(acc_o[td][0] * inv, acc_o[td][1] * inv, acc_o[td][2] * inv, acc_o[td][3] * inv);
    }
}

template <typename HT>
static void launch_attention_16(const HT* qkv, HT* out, int n_img, int N, int H, hipStream_t stream) {
    const int D = H * 64;
    const int nt = (N + 63) / 64;
    dim3 grid(nt, H, n_img);
    if (N <= 256 && (long)((N + 15) / 16) * H * n_img <= 640 && (long)n_img * N * 6 * D < (1l << 32)) {
        const int items = ((N + 15) / 16) * H * n_img;
        launch(attention_16_short_kernel<HT>, dim3(8 * ((items + 7) / 8)), dim3(256), 4 * 8192 + 4 * 5 * 64 * 16, stream, qkv,
               out, N, D, n_img);
    } else if (N >= 512 && (long)n_img * N * 6 * D < (1l << 32)) {
        const int items = ((N + 127) / 128) * H * n_img;
        launch(attention_16_long_kernel<HT>, dim3(8 * ((items + 7) / 8)), dim3(256), 3 * 2 * 64 * 128, stream, qkv, out, N, D, n_img);
    } else if ((long)nt * H * n_img <= 256 && nt >= 2) {
        launch((attention_16_kernel<HT, 2>), grid, dim3(512), 2 * 2 * 64 * 128, stream, qkv, out, N, D);
    } else {
        launch((attention_16_kernel<HT, 1>), grid, dim3(256), 2 * 64 * 128, stream, qkv, out, N, D);
    }
}

int launch_attention(Precision p, const void* qkv, void* out, int n_img, int N, int H, hipStream_t stream) {
    if (n_img <= 0 || N <= 0 || H <= 0) return -2;
    const int D = H * 64;
    const int nt = (N + 63) / 64;
    dim3 grid(nt, H, n_img);
    if (p == PREC_F32) {
        launch(attention_f32_kernel, grid, dim3(256), 0, stream, (const float*)qkv, (float*)out, N, D);
    } else if (p == PREC_F16) {
        launch_attention_16<f16>((const f16*)qkv, (f16*)out, n_img, N, H, stream);
    } else {
        launch_attention_16<bf16>((const bf16*)qkv, (bf16*)out, n_img, N, H, stream);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
