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

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "probe.h"

namespace vitvs {

constexpr float kScaleLog2e = kAttnQScale;                        // hd^-0.5 * log2(e), hd = 64 (kernels.h)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32; exp2(-inf) = 0
__device__ __forceinline__ void wait_vmcnt4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
// lo halves of two weights whose hi halves are packed in `hi`: f16(p - hi), the subtraction exact in fp32.  One mixed-precision fma per
// value (fp16 source widened inside the instruction, fp16 result written to its half of the destination) instead of widen + subtract +
// narrow: the vector pipe is the long kernel's longer resource.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ inline f16x2 split_lo_pair(f16x2 hi, float p0, float p1) {
    const unsigned h = __builtin_bit_cast(unsigned, hi);
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(p0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "v"(p1));
    return __builtin_bit_cast(f16x2, d);
}
// max of three as ONE instruction: for fmaxf() on MFMA results hipcc first emits a canonicalising v_max_f32 x, x, x per
// operand (32 extra vector instructions per key tile in a loop that is bound by vector issue)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max of the 16 scores a lane holds for one query tile: 8 instructions
__device__ __forceinline__ float max16(const f32x4 (&s)[4]) {
    float m = max3(s[0][0], s[0][1], s[0][2]);
    m = max3(m, s[0][3], s[1][0]);
    m = max3(m, s[1][1], s[1][2]);
    m = max3(m, s[1][3], s[2][0]);
    m = max3(m, s[2][1], s[2][2]);
    m = max3(m, s[2][3], s[3][0]);
    m = max3(m, s[3][1], s[3][2]);
    return max3(m, s[3][3], s[3][3]);
}

// ------------------------------------------------------------------------------------ bf16
template <typename HT, int KS>
__global__ __launch_bounds__(256 * KS) void attention_16_kernel(const HT* __restrict__ qkv, HT* __restrict__ out,
                                                                int N, int D, float sc) {
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
        mloc = rows_max(mloc) * sc;
        const float m_new = fmaxf(m_run, mloc);
        const float neg_m = -m_new;
        float psum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(__builtin_fmaf(acc_s[t4][r], sc, neg_m));
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
// (every 64 queries re-read the (image, head)'s whole K and V: 0.94 GB per launch at 3137 tokens), by exposed LDS
// latency (each MFMA waits for the fragment read issued just before it) and by vector-instruction issue.  Here:
//   * one workgroup = 128 queries, wave w = queries 32 w .. 32 w + 31 on the 32x32x16 MFMA (half the MFMA issue slots of
//     the 16x16x32 form for the same work, one softmax state per lane, one cross-lane step per reduction);
//   * K / V tiles of 64 keys arrive by LDS-DMA into a 3-stage ring with counted waits, one barrier per tile, two tiles
//     in flight across it (no register staging, no ds_write pass);
//   * S^T = K Q^T per 32-key block leaves a lane with 16 scores of ONE query (column = lane & 31); the accumulator
//     registers 8s .. 8s+7, converted pairwise, ARE the B operand of k-step s of O^T += V^T P^T (register index -> key map
//     16s + 8(j>>2) + 4(lane>>5) + (j&3)), and the V fragments are fetched with hardware-transposed reads in that key order;
//   * 1-D grid, XCD-aware item order: the query blocks of an (image, head) run on the XCD whose L2 already holds its
//     K and V.
// Same arithmetic as attention_16_kernel (raw-score maximum, scale folded into the exp2 FMA, exact rescale only when a
// maximum moved), so the two agree to rounding.
// Where the time goes (probe build, tools/big_ops attn, 3137 tokens, 2.0 GHz in-kernel clock; cycles per key tile of a
// wave, 2-3 waves per SIMD): DMA wait 56, barrier 145, K reads + 8 score MFMAs 780, softmax 1500, V reads + 8 PV MFMAs 330.
// The softmax is ~145 vector instructions per tile (32 each of fma / exp2 / add, 16 max3, 16 cvt_pk) at ~4 issue cycles
// each per wave: with head dimension 64 a score costs 2 MFMA-k-steps but the same vector work as at 128, so the kernel
// is bound by vector issue, not by the matrix pipe (profiles/r02_notes.md has the variants that did not move it:
// 16x16x32 tiles, staggered workgroup starts, fewer address instructions).
#ifdef VITVS_PROBE
// probe builds only (tools/big_ops probe): per-wave cycle sums of the tile loop's parts + realtime span (probe.h)
__device__ unsigned long long* g_attn_probe;
static int g_attn_lds_bytes = 3 * 2 * 64 * 128;   // dynamic LDS per workgroup of the long kernel: limits how many share a CU (48 KB: 3, 64 KB: 2, 100 KB: 1)
extern "C" __attribute__((visibility("default"))) int vitvs_debug_set_attn_probe(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_probe), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int vitvs_debug_set_attn_lds(int bytes) {
    g_attn_lds_bytes = bytes < 3 * 2 * 64 * 128 ? 3 * 2 * 64 * 128 : bytes;
    return 0;
}
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Work division over key tiles.  The unit of work is one 64-key tile of one 128-query block; a launch has W = items x nt of them
// (items = images x heads x query blocks, nt = key tiles per sequence), in item-major order.  Workgroup g walks the contiguous
// range [g per, (g + 1) per) of that list; `per` comes from attention_plan() below (whole items, halves or quarters of an item:
// the fewest ranges per item that minimise rounds x (tiles + hand-off)).  A range may cross item boundaries: it is cut into
// segments, one per item it touches.  A segment that covers its item's whole key range finishes like a plain flash kernel.
// Any other segment leaves its un-normalised state (O, running maximum, running sum) in `ws`, draws a ticket for its item, and
// the workgroup that draws the item's last ticket merges the item's segments in range order (fixed order: bit-reproducible)
// and writes the output; nobody waits for anybody, so the workgroups need not be co-resident, and nothing depends on placement.
//
// The hand-off has NO fence on either side: it is the "sc1 loads in place of the acquire" form of MI355X_MICROARCH.md
// (Workgroup dispatch ... Valid forms), first row of its table (one lane of each storing workgroup adds to ONE unsharded
// counter; the workgroup whose add came last, told by the value its add returned, reads).  Its four conditions, and where
// the code meets each:
//   (1) EVERY load of the handed-off bytes is a global_ sc1 load to registers: the nine `global_load_dwordx4 ... sc1` of the
//       merge loop ("state_of" / og[]); the merging workgroup's own state never goes through memory (registers);
//   (2) the producer stored every one of those bytes sc1: the nine store_out<true> (16-byte write-through stores) of the
//       "a partial segment" block, per wave;
//   (3) every storing wave drains its stores (`s_waitcnt vmcnt(0)`) and the ONE signalling lane (tid 0) adds to the item's
//       ticket behind the workgroup barrier that follows those waits (`__syncthreads()` then the relaxed agent-scope
//       `__hip_atomic_fetch_add`);
//   (4) the table's row: hipMalloc memory (the handle's workspace), 16-byte sc1 stores and loads; the wave that added last
//       learns it from the add's return value, publishes it through an LDS word, and the other waves load only behind the
//       workgroup barrier that follows (`*flag = drawn; __syncthreads()`).
// An agent acquire by every merging wave instead (buffer_inv sc1 per wave, three workgroups per CU) cost ~2.5 us per merging
// workgroup; a release fence per producing workgroup (an L2 write-back each) made the launch 13 % slower than not dividing at
// all.  Soak: 30 000 updates through three queues, bit-identical (profiles/r03_soak_pipeline.txt).
// The ticket is reset by the last arriver: the array only has to be zero before the first launch.  per = nt is the undivided
// form: every segment is a whole item, `ws` is not touched.  A workgroup has at most two partial segments: slot 0 = the one that
// starts inside an item (its first), slot 1 = the one that starts an item and ends inside it (its last).
constexpr int kAttnStateFloats = 9 * 64 * 4;                // per wave: 8 x 16 bytes of accumulators + (maximum, sum), lane-major

// ONES (measured variant; launch_attention_16 has the outcome): the row sums l = sum_k P come out of the matrix
// pipe — a third 32-row block of V^T whose rows are all ones, i.e. one more MFMA per 16-key step with a constant A operand —
// instead of 32 vector adds per tile; the fast path's headroom test then takes the maximum of the shifted scores (16 v_max3)
// BEFORE the exponentials rather than the sum of the probabilities after them.
// (ONES needs 16 more accumulator registers: 187, over the 170 a wave may hold at three waves per SIMD — compiled for three it
// spills 300 bytes per lane, and spill traffic between the hand-counted vmcnt waits breaks them — so the variant is built for
// two workgroups per CU.)
template <typename HT, bool ONES = false>
__global__ __launch_bounds__(256, ONES ? 2 : 3) void attention_16_long_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int N,
                                                                   int D, int n_img, int per, int g_per_xcd, float qscale,
                                                                   float* __restrict__ ws, int* __restrict__ tickets) {
    typedef typename Vec16<HT>::x8 hx8;
    typedef typename Vec16<HT>::x4 hx4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 2 * 64 * 128;                      // K tile then V tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int H = D >> 6, nqb = (N + 127) >> 7, items = n_img * H * nqb;
    const int nt = (N + 63) >> 6, W = items * nt;
    // XCD x (workgroup ids x, x + 8, ...) walks the x-th eighth of the list: the query blocks of an (image, head) share its L2
    const int g = (int)(blockIdx.x & 7) * g_per_xcd + (int)(blockIdx.x >> 3);
    int w0 = g * per;
    const int w1 = min(w0 + per, W);
    if (w0 >= w1) return;
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned row_bytes = 6u * (unsigned)D;
    const unsigned tile_bytes = 64u * row_bytes;
    const int r8 = lane >> 3;
    // lane constants of the fragment reads
    const int k_sw = (r32 >> 1) & 7;                              // K image swizzle of this lane's key row (32 kb + r32)
    const int li = lane & 15, dh = (lane >> 4) & 1;              // transposed V read: lane li of a 16-lane group, dim half dh
    const int v_row = 4 * hh + (li >> 2);                        // key row inside a 16-key step (+ 8 for elements 4..7)
    const int v_sw = ((v_row >> 1) & 1) << 1;                    // window swizzle of that row (the 8-row step keeps bit 1)
    int* flag = reinterpret_cast<int*>(smem + 3 * STAGE);        // one word behind the ring (same LDS array)
    int slot = 0;                                                 // ring slot of the NEXT tile to be computed; runs on across segments
    VITVS_IF_PROBE(
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, sum[5] = {0, 0, 0, 0, 0}, tiles_done = 0, segs = 0, xchg = 0, tx0 = 0, tx1 = 0, slow_tiles = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_readcyclecounter();
    )
    while (w0 < w1) {
    const int item = w0 / nt, t_begin = w0 - item * nt, ntiles = min(nt - t_begin, w1 - w0);
    w0 += ntiles;
    const int pair = item / nqb;
    const int img = pair / H, h = pair - img * H, q0 = (item - pair * nqb) * 128 + 32 * wave;
    const unsigned head_off = (unsigned)(img * N) * row_bytes + (unsigned)h * 128u;
    const unsigned k_off = head_off + 2u * (unsigned)D, v_off = head_off + 4u * (unsigned)D;

    // this wave's four LDS-DMA copies per key tile: rows 16 wave .. 16 wave + 15 of the K image and of the V image, 8
    // rows per copy, swizzled on the SOURCE chunk (LDS-DMA writes linearly):
    //   K image: chunk c of row r at slot c ^ ((r >> 1) & 7)   (tile128_off: conflict-free ds_read_b128 of 32 rows)
    //   V image: chunk c of row r at slot c ^ (((r >> 1) & 1) << 2): the 4 rows x two 32-byte windows a half-wave of a
    //            transposed read touches then fall on 8 different 32-byte bank windows
    unsigned koff[2], voff[2];
    int krow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        krow[c] = 16 * wave + 8 * c + r8;
        koff[c] = k_off + 16u * (unsigned)((lane & 7) ^ ((krow[c] >> 1) & 7)) + (unsigned)krow[c] * row_bytes + (unsigned)t_begin * tile_bytes;
        voff[c] = v_off + 16u * (unsigned)((lane & 7) ^ (((krow[c] >> 1) & 1) << 2)) + (unsigned)krow[c] * row_bytes + (unsigned)t_begin * tile_bytes;
    }
    // Ring slots.  The ring index runs on across segments: when a wave starts a new segment every wave of the workgroup has
    // passed the barrier of the previous segment's LAST tile, i.e. is done with every slot but that tile's; the new
    // segment's tiles 0 and 1 go to the two other slots, and its tile 2 is issued behind the barrier of its tile 0.
    // source offsets advance by 64 rows per tile (one add per copy); only the sequence's last tile can reach past row
    // N - 1 and takes the clamped form
    auto issue = [&](int t, int sl) {                          // t: tile index inside this segment; sl: its ring slot
        unsigned char* dst = smem + sl * STAGE + (16 * wave) * 128;
        if (t_begin + t + 1 < nt) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + koff[c]), (lds_ptr)(dst + c * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + voff[c]), (lds_ptr)(dst + 64 * 128 + c * 1024), 16, 0, 0);
                koff[c] += tile_bytes;
                voff[c] += tile_bytes;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned back = __umul24((unsigned)max(64 * (t_begin + t) + krow[c] - (N - 1), 0), row_bytes);   // rows past the end -> row N - 1
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (koff[c] - back)), (lds_ptr)(dst + c * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (voff[c] - back)), (lds_ptr)(dst + 64 * 128 + c * 1024), 16, 0, 0);
            }
        }
    };
    const int slot1 = slot == 2 ? 0 : slot + 1;
    issue(0, slot);
    if (ntiles > 1) issue(1, slot1);
    // Q fragments by ordinary loads AFTER the first copies, and consumed (empty asm) before the loop: hipcc places its
    // wait for an ordinary load at the first use, and inside the tile loop that wait would be vmcnt(0) on every
    // iteration, draining the LDS-DMA ring; here it is one wait in the prologue, which tile 0 needs anyway.
    // B operand of S^T = K Q^T: lane (r32, hh) holds Q[query r32][dims 16 ks + 8 hh + 0..7]
    u32x4 qraw[4];
    {
        const unsigned o = head_off + (unsigned)min(q0 + r32, N - 1) * row_bytes + 16u * (unsigned)hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qraw[ks] = *reinterpret_cast<const u32x4*>(qb + (o + 32u * ks));
    }
    asm volatile("" : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]));
    hx8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(hx8, qraw[ks]);
    if (qscale != 1.0f) {                                       // raw q (the operator hook): scale here, one more 16-bit rounding
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (HT)((float)qf[ks][j] * qscale);
    }
    // The running maximum is subtracted INSIDE the score MFMAs: a fifth k-step multiplies a constant K column pair (1, 1)
    // with the query's (-m_hi, -m_lo), so the accumulators leave the matrix pipe as s - m, ready for exp2 (no per-score
    // FMA on the vector pipe, which is what bounds this loop).  m = m_hi + m_lo is exact in two 16-bit terms; every use of
    // the shift (the scores, the rescale of O and l) sees the same fp32 value m_t.
    hx8 qm, kone;
#pragma unroll
    for (int j = 0; j < 8; ++j) { qm[j] = (HT)0.f; kone[j] = (HT)((hh == 0 && j < 2) ? 1.f : 0.f); }
    float m_t = -INFINITY;                                       // the shift baked into qm (-inf: none yet, qm = 0)

    f32x16 acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
    float l_run = 0.f;
    f32x16 acc_l;                                                // ONES: every row = the query's running sum of P (over ALL keys)
    hx8 ones;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_l[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (HT)1.f;
    for (int t = 0; t < ntiles; ++t) {
        VITVS_STAMP(ts0);
        if (t + 1 < ntiles) wait_vmcnt4();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VITVS_STAMP(ts1);
        __builtin_amdgcn_s_barrier();                   // tile t has landed for every wave; everyone is done with tile t - 1
        __builtin_amdgcn_sched_barrier(0);
        VITVS_STAMP(ts2);
        const int slot2 = slot == 0 ? 2 : slot - 1;     // (slot + 2) % 3: the slot of tile t - 1, free behind the barrier
        if (t + 2 < ntiles) issue(t + 2, slot2);
        const unsigned char* ldsK = smem + slot * STAGE;
        slot = slot == 2 ? 0 : slot + 1;
        const int kb0 = (t_begin + t) * 64;
        // scores: acc_s[kb][i] = S[key 64 t + 32 kb + (i & 3) + 8 (i >> 2) + 4 hh][query r32] (log2 units) - m_t
        f32x16 acc_s[2];
        auto scores = [&](bool shifted) {
            hx8 kf[2][4];                               // all 8 K fragment reads in flight before the first MFMA
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    kf[kb][ks] = __builtin_bit_cast(hx8, *reinterpret_cast<const u32x4*>(ldsK + (32 * kb + r32) * 128 + (((2 * ks + hh) ^ k_sw) << 4)));
            __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks each read next to its MFMA: 8 exposed LDS latencies)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                acc_s[kb] = mfma32(kf[kb][0], qf[0], zero);   // C = 0 is an inline constant of the instruction: no register zeroing
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) acc_s[kb] = mfma32(kf[kb][ks], qf[ks], acc_s[kb]);
                if (shifted) acc_s[kb] = mfma32(kone, qm, acc_s[kb]);
            }
        };
        scores(true);
        VITVS_IF_PROBE(asm volatile("s_nop 0" ::"v"(acc_s[0][0]), "v"(acc_s[1][15]) : "memory");)
        VITVS_STAMP(ts3);
        // V fragments (hardware-transposed reads), requested now and consumed after the softmax.  Inline asm: for the
        // ds_read_tr16 builtin hipcc waits vmcnt(0) first (it cannot tell the LDS-DMA copies in flight apart from the
        // image being read), which would drain the ring on every tile; the reads' completion is waited for by hand.
        // vf[step][db][half]: A operand of k-step `step` (16 keys) for dim block db; element j <-> key 16 step + 8 (j >> 2)
        // + 4 hh + (j & 3), dim 32 db + 16 dh + li
        s16x4 vf[4][2][2];
        {
            const lds_u8* vbase = (const lds_u8*)(ldsK + 64 * 128) + v_row * 128 + 8 * (li & 3);
            const unsigned va0 = (unsigned)(size_t)vbase + 32u * (unsigned)((0 + dh) ^ v_sw);   // dim block 0
            const unsigned va1 = (unsigned)(size_t)vbase + 32u * (unsigned)((2 + dh) ^ v_sw);   // dim block 1
#define VITVS_TR(dst, va, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(OFF))
            VITVS_TR(vf[0][0][0], va0, 0);    VITVS_TR(vf[0][0][1], va0, 1024);        VITVS_TR(vf[0][1][0], va1, 0);    VITVS_TR(vf[0][1][1], va1, 1024);
            VITVS_TR(vf[1][0][0], va0, 2048); VITVS_TR(vf[1][0][1], va0, 2048 + 1024); VITVS_TR(vf[1][1][0], va1, 2048); VITVS_TR(vf[1][1][1], va1, 2048 + 1024);
            VITVS_TR(vf[2][0][0], va0, 4096); VITVS_TR(vf[2][0][1], va0, 4096 + 1024); VITVS_TR(vf[2][1][0], va1, 4096); VITVS_TR(vf[2][1][1], va1, 4096 + 1024);
            VITVS_TR(vf[3][0][0], va0, 6144); VITVS_TR(vf[3][0][1], va0, 6144 + 1024); VITVS_TR(vf[3][1][0], va1, 6144); VITVS_TR(vf[3][1][1], va1, 6144 + 1024);
#undef VITVS_TR
        }
        auto mask_tail = [&]() {                         // keys beyond N: only the sequence's last tile has any (wave-uniform)
            if (kb0 + 64 > N) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (kb0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) acc_s[kb][i] = -INFINITY;
            }
        };
        // Softmax of the lane's 32 scores (one query; the other half of its keys sits in lane ^ 32).  Fast path: the shift of
        // an EARLIER tile is still in place, p = exp2(s - m_t) straight from the accumulators, no maximum is formed; the
        // tile is accepted if every lane's sum of p stays <= 64 (so every p <= 64: scores may exceed the shift by up to 6
        // in log2 units; P in 16 bits and the fp32 sums have that headroom).  Otherwise, and on a segment's first tile, the
        // slow path: raw scores again, their exact maximum, O and l rescaled to the new shift exactly once.
        bool slow = t == 0;
        float psum = 0.f;
        if (!slow) {
            mask_tail();
            if constexpr (ONES) {
                // every p = exp2(s - m_t) stays <= 64 iff the largest shifted score is <= 6: tested before the exponentials
                float mx = max3(acc_s[0][0], acc_s[0][1], acc_s[0][2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) mx = max3(mx, acc_s[0][i], acc_s[0][i + 1]);
                mx = max3(mx, acc_s[0][15], acc_s[1][0]);
#pragma unroll
                for (int i = 1; i < 15; i += 2) mx = max3(mx, acc_s[1][i], acc_s[1][i + 1]);
                mx = fmaxf(mx, acc_s[1][15]);
                slow = __builtin_amdgcn_ballot_w64(!(mx <= 6.f)) != 0ull;        // (a NaN score also lands here)
                if (slow) scores(false);
                else {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc_s[kb][i] = fast_exp2(acc_s[kb][i]);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = fast_exp2(acc_s[kb][i]);
                        acc_s[kb][i] = p;
                        psum += p;
                    }
                slow = __builtin_amdgcn_ballot_w64(!(psum <= 64.f)) != 0ull;     // (an inf or NaN sum also lands here)
                if (slow) scores(false);                     // the accumulators hold p now: raw scores again (the K tile is still in LDS)
            }
        }
        VITVS_IF_PROBE(if (slow) ++slow_tiles;)
        if (slow) {
            if (t == 0 && m_t != -INFINITY) scores(false);   // (never: a segment starts with m_t = -inf, qm = 0, i.e. raw scores)
            mask_tail();
            float mloc = max3(acc_s[0][0], acc_s[0][1], acc_s[0][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mloc = max3(mloc, acc_s[0][i], acc_s[0][i + 1]);
            mloc = max3(mloc, acc_s[0][15], acc_s[1][0]);
#pragma unroll
            for (int i = 1; i < 15; i += 2) mloc = max3(mloc, acc_s[1][i], acc_s[1][i + 1]);
            mloc = fmaxf(mloc, acc_s[1][15]);
            mloc = fmaxf(mloc, lane_xor32(mloc));
            const float m_want = fmaxf(m_t, mloc);       // finite: a tile's first key is valid
            const HT m_hi = (HT)m_want;
            const HT m_lo = (HT)(m_want - (float)m_hi);
            const float m_new = (float)m_hi + (float)m_lo;   // the shift the MFMAs will subtract from now on
            const float alpha = fast_exp2(m_t - m_new);  // 0 on a first tile (m_t = -inf; O = l = 0 anyway)
            l_run *= alpha;
            if constexpr (ONES) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_l[i] *= alpha;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
            m_t = m_new;
            qm[0] = (HT)(hh == 0 ? -(float)m_hi : 0.f);
            qm[1] = (HT)(hh == 0 ? -(float)m_lo : 0.f);
            psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(acc_s[kb][i] - m_new);
                    acc_s[kb][i] = p;
                    if constexpr (!ONES) psum += p;
                }
        }
        if constexpr (!ONES) l_run += psum;
        hx8 pf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[st][j] = (HT)acc_s[st >> 1][8 * (st & 1) + j];
        VITVS_STAMP(ts4);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vf[0][0][0]), "+v"(vf[0][0][1]), "+v"(vf[0][1][0]), "+v"(vf[0][1][1]), "+v"(vf[1][0][0]), "+v"(vf[1][0][1]),
                       "+v"(vf[1][1][0]), "+v"(vf[1][1][1]), "+v"(vf[2][0][0]), "+v"(vf[2][0][1]), "+v"(vf[2][1][0]), "+v"(vf[2][1][1]),
                       "+v"(vf[3][0][0]), "+v"(vf[3][0][1]), "+v"(vf[3][1][0]), "+v"(vf[3][1][1]));
        __builtin_amdgcn_sched_barrier(0);
        // O^T[dim][query] += V^T P^T, 4 k-steps of 16 keys x 2 dim blocks
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const s16x4 lo = vf[st][db][0], hi = vf[st][db][1];
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc_o[db] = mfma32(__builtin_bit_cast(hx8, v8), pf[st], acc_o[db]);
            }
        if constexpr (ONES) {
#pragma unroll
            for (int st = 0; st < 4; ++st) acc_l = mfma32(ones, pf[st], acc_l);
        }
        VITVS_IF_PROBE(
            asm volatile("s_nop 0" ::"v"(acc_o[0][0]), "v"(acc_o[1][15]) : "memory");
            VITVS_STAMP(ts5);
            sum[0] += ts1 - ts0; sum[1] += ts2 - ts1; sum[2] += ts3 - ts2; sum[3] += ts4 - ts3; sum[4] += ts5 - ts4;
            ++tiles_done;
        )
    }
    if constexpr (ONES) l_run = acc_l[0];                      // (the contraction runs over the whole tile: both lane halves' keys)
    else l_run += lane_xor32(l_run);                           // both lane halves hold the query's whole sum
    VITVS_IF_PROBE(++segs;)
    VITVS_STAMP(tx0);
    if (ntiles != nt) {
        // a partial segment: leave its state, ws[2 g + slot][wave][group 0 .. 8][lane] x 16 bytes, written through (sc1)
        const int first_g = (item * nt) / per, last_g = (item * nt + nt - 1) / per;   // the workgroups that share this item
        float* mine = ws + ((size_t)(2 * g + (t_begin == 0 ? 1 : 0)) * 4 + wave) * kAttnStateFloats + 4 * lane;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                store_out<true>(mine + (4 * db + g4) * 256,
                                f32x4{acc_o[db][4 * g4], acc_o[db][4 * g4 + 1], acc_o[db][4 * g4 + 2], acc_o[db][4 * g4 + 3]});
        store_out<true>(mine + 8 * 256, f32x4{m_t, l_run, 0.f, 0.f});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores ...
        __syncthreads();                                       // ... before the workgroup's one ticket
        if (tid == 0) {
            const int drawn = __hip_atomic_fetch_add(tickets + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (drawn == last_g - first_g)
                __hip_atomic_store(tickets + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            *flag = drawn;
        }
        __syncthreads();
        const bool last = *flag == last_g - first_g;           // workgroup-uniform
        __syncthreads();                                       // everyone has read the flag before a later segment rewrites it
        VITVS_IF_PROBE(if (!last) { VITVS_STAMP(tx1); xchg += tx1 - tx0; })
        if (!last) continue;                                   // another workgroup finishes this item
        // The other segments' states are read with sc1 loads, every byte of them: written through by their producers (sc1
        // stores, drained by every storing wave before the workgroup's one ticket add), they are in memory, and an sc1 load
        // does not look in this CU's L1 — the only cache that could hold an older copy (the XCD's L2 was invalidated at
        // launch start and has not seen these lines since).  This is the CDNA4 guide's "sc1 loads in place of the acquire"
        // hand-off (one lane of each storing workgroup adds to one counter; the workgroup whose add came last reads, its
        // other waves behind the barrier above); an agent acquire by every wave instead cost ~2.5 us per merging workgroup
        // (buffer_inv per wave, three workgroups per CU).  Merged in range order, one pass, this workgroup's own state from
        // its registers: bit-reproducible.
        auto state_of = [&](int gg) { return ws + ((size_t)(2 * gg + (gg == first_g ? 1 : 0)) * 4 + wave) * kAttnStateFloats + 4 * lane; };
        float m_tot = -INFINITY, l_tot = 0.f;
        f32x16 o_tot[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_tot[db][i] = 0.f;
        for (int gg = first_g; gg <= last_g; ++gg) {
            const bool own = gg == g;                          // wave-uniform
            float mp = m_t, lp = l_run;
            u32x4 og[9];
            if (!own) {
                const float* st = state_of(gg);
#pragma unroll
                for (int g8 = 0; g8 < 9; ++g8)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(og[g8]) : "v"(st + g8 * 256) : "memory");
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(og[0]), "+v"(og[1]), "+v"(og[2]), "+v"(og[3]), "+v"(og[4]), "+v"(og[5]), "+v"(og[6]), "+v"(og[7]), "+v"(og[8])
                             :: "memory");
                mp = __uint_as_float(og[8][0]);
                lp = __uint_as_float(og[8][1]);
            } else {
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
                    for (int r = 0; r < 4; ++r) og[g8][r] = __float_as_uint(acc_o[g8 >> 2][4 * (g8 & 3) + r]);
            }
            const float m_new = fmaxf(m_tot, mp);
            const float w_old = fast_exp2(m_tot - m_new), w_seg = fast_exp2(mp - m_new);   // (first segment: w_old = exp2(-inf) = 0)
            m_tot = m_new;
            l_tot = l_tot * w_old + lp * w_seg;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o_tot[g8 >> 2][4 * (g8 & 3) + r] = o_tot[g8 >> 2][4 * (g8 & 3) + r] * w_old + __uint_as_float(og[g8][r]) * w_seg;
        }
        l_run = l_tot;
        acc_o[0] = o_tot[0];
        acc_o[1] = o_tot[1];
    }
    // acc_o[db][i] = O[query r32][dim 32 db + (i & 3) + 8 (i >> 2) + 4 hh] * l
    const float inv = 1.0f / l_run;
    const int q = q0 + r32;
    if (q < N) {
        HT* dst = out + ((size_t)img * N + q) * D + h * 64 + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const hx4 o = {(HT)(acc_o[db][4 * g4 + 0] * inv), (HT)(acc_o[db][4 * g4 + 1] * inv), (HT)(acc_o[db][4 * g4 + 2] * inv),
                               (HT)(acc_o[db][4 * g4 + 3] * inv)};
                store_out<false>(dst + 32 * db + 8 * g4, o);
            }
    }
    VITVS_IF_PROBE(VITVS_STAMP(tx1); xchg += tx1 - tx0;)
    }   // segments
    VITVS_IF_PROBE(
        if (lane == 0 && g_attn_probe) {
            unsigned long long* dst = g_attn_probe + ((size_t)blockIdx.x * 4 + wave) * 14;
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            dst[0] = sum[0]; dst[1] = sum[1]; dst[2] = sum[2]; dst[3] = sum[3]; dst[4] = sum[4];
            dst[5] = __builtin_readcyclecounter() - ct0; dst[6] = rt1 - rt0; dst[7] = tiles_done;
            dst[8] = rt0; dst[9] = rt1; dst[10] = xchg; dst[11] = segs; dst[12] = slow_tiles; dst[13] = 0;
        }
    )
}

// ------------------------------------------------------------------------------------ split-f16 (f16x2), long sequences
// The 128-query arrangement of attention_16_long_kernel for hi / lo operands (N > 256: the 485 / 1370 / 3137 tokens of 308² / 518² /
// 448² inputs): wave w = queries 32 w .. 32 w + 31 on the 32x32x16 MFMA, both contractions as lo.hi + hi.lo + hi.hi.  K / V tiles
// of 32 KEYS (a key's row is 256 bytes here: 32 keys x 256 B of K and as much of V = 16 KB per stage) by LDS-DMA into a THREE-stage
// ring with counted waits — two tiles in flight across every barrier, 48 KB of LDS — and at most 170 registers, so that THREE
// workgroups share a CU: the first version of this kernel (64-key tiles, two stages, every V fragment of a tile in registers: 214
// registers, two workgroups per CU) ran 2 x 3137 x 12 in 254 us, no faster than the generic kernel — its 600 workgroups needed a
// second round on 512 slots and a one-tile prefetch distance was exposed on every tile; this form 235 us, 219 us with the softmax's
// vector work trimmed (below).  Counters of that launch (profiles/r05_attention_x2_long_pmc.txt, r05_notes.md section 9): no LDS bank
// conflicts, LDS array busy 12 % of the time; the matrix pipe busy 47 % of the kernel's cycles averaged over the chip and ~59 % on the
// 88 CUs that hold three of the 600 workgroups (the others hold two and finish early); 49 % of a wave's life is issue stall behind
// the other waves of its SIMD, 18 % parked on a wait.  Priority raised around the MFMA clusters (s_setprio) and scalar instead of
// packed fp32 softmax arithmetic were measured on the same box and change nothing (+-1 %).
// Per wave and key tile: 12 score MFMAs + 12 PV MFMAs (three times the 16-bit kernel's per key) and ~110 vector instructions (16
// exponentials, the hi / lo split of P as one conversion and two mixed-precision fmas per pair), so the kernel keeps the plain online
// softmax (fp32 statistics on raw scores, hd^-0.5 log2 e inside the exponent's fma, P split in registers at 2^8) — the 16-bit
// kernel's shift-inside-the-MFMA and key-range hand-off buy nothing here.
//   K image: chunk c (16 bytes) of row r at slot c ^ (r & 15) (tile256_off), applied to the copy's SOURCE chunk (LDS-DMA writes linearly)
//   V image: 32-byte window w of row r at slot w ^ ((r & 3) << 1): the 16 (row, window) pieces one transposed read touches (8 rows x
//            2 windows) fall two on each of the eight 32-byte bank windows
// 1-D grid, XCD-aware item order (the query blocks of an (image, head) run on the XCD whose L2 already holds its K and V).
__global__ __launch_bounds__(256, 3) void attention_x2_long_kernel(const hx2* __restrict__ qkv, hx2* __restrict__ out, int N, int D,
                                                                   int n_img, int g_per_xcd) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 2 * 32 * 256;                       // K tile then V tile, 32 keys each
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int H = D >> 6, nqb = (N + 127) >> 7, items = n_img * H * nqb, nt = (N + 31) >> 5;
    const int item = (int)(blockIdx.x & 7) * g_per_xcd + (int)(blockIdx.x >> 3);
    if (item >= items) return;
    const int pair = item / nqb;
    const int img = pair / H, h = pair - img * H, q0 = (item - pair * nqb) * 128 + 32 * wave;
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned row_bytes = 12u * (unsigned)D;             // 3 D logical columns x (hi + lo)
    const unsigned tile_bytes = 32u * row_bytes;
    const unsigned head_off = (unsigned)(img * N) * row_bytes + (unsigned)h * 256u;
    const unsigned k_off = head_off + 4u * (unsigned)D, v_off = head_off + 8u * (unsigned)D;

    // this wave's LDS-DMA copies per tile: rows 8 wave .. 8 wave + 7 of the K image and of the V image, 4 rows per copy
    unsigned koff[2], voff[2];
    int crow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        crow[c] = 8 * wave + 4 * c + (lane >> 4);
        const int sl = lane & 15;
        koff[c] = k_off + (unsigned)crow[c] * row_bytes + 16u * (unsigned)(sl ^ (crow[c] & 15));
        voff[c] = v_off + (unsigned)crow[c] * row_bytes + 32u * (unsigned)((sl >> 1) ^ ((crow[c] & 3) << 1)) + 16u * (unsigned)(sl & 1);
    }
    auto issue = [&](int t, int stage) {
        unsigned char* dst = smem + stage * STAGE + (8 * wave) * 256;
        if (t + 1 < nt) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (koff[c] + (unsigned)t * tile_bytes)), (lds_ptr)(dst + c * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (voff[c] + (unsigned)t * tile_bytes)), (lds_ptr)(dst + 32 * 256 + c * 1024), 16, 0, 0);
            }
        } else {                                              // the sequence's last tile: rows past the end -> row N - 1
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned back = __umul24((unsigned)max(32 * t + crow[c] - (N - 1), 0), row_bytes);
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (koff[c] + (unsigned)t * tile_bytes - back)), (lds_ptr)(dst + c * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr)(qb + (voff[c] + (unsigned)t * tile_bytes - back)), (lds_ptr)(dst + 32 * 256 + c * 1024), 16, 0, 0);
            }
        }
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    // Q fragments (B operand of S^T = K Q^T): lane (r32, hh) holds Q[query r32][dims 16 ks + 8 hh .. + 7], hi and lo; ordinary loads
    // AFTER the first copies and consumed (empty asm) before the loop, so that hipcc's wait for them is one wait in the prologue
    // (inside the loop it would be vmcnt(0) on every tile: the ring drained)
    u32x4 qraw[8];
    {
        const unsigned o = head_off + (unsigned)min(q0 + r32, N - 1) * row_bytes + 16u * (unsigned)hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned at = o + 128u * (unsigned)(ks >> 1) + 32u * (unsigned)(ks & 1);   // dims 16 ks + 8 hh: block ks >> 1, 32 bytes per 16 dims
            qraw[2 * ks] = *reinterpret_cast<const u32x4*>(qb + at);
            qraw[2 * ks + 1] = *reinterpret_cast<const u32x4*>(qb + at + 64u);
        }
    }
    asm volatile("" : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]), "+v"(qraw[4]), "+v"(qraw[5]), "+v"(qraw[6]), "+v"(qraw[7]));
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qh[ks] = __builtin_bit_cast(f16x8, qraw[2 * ks]); ql[ks] = __builtin_bit_cast(f16x8, qraw[2 * ks + 1]); }

    // lane constants of the fragment reads
    const int li = lane & 15, dh = (lane >> 4) & 1;
    const int v_row = 4 * hh + (li >> 2);                     // key row inside a 16-key step (+ 8 for elements 4 .. 7)
    const int v_sw = (li >> 2) << 1;                          // window swizzle of that row: (row & 3) << 1 (+ 8 rows keeps it)
    f32x16 acc_o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int slot = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) wait_vmcnt4();                        // tile t has landed (this wave's copies): only tile t + 1's four are younger
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... every wave's; and everyone is done with tile t - 1
        __builtin_amdgcn_sched_barrier(0);
        const int slot2 = slot == 0 ? 2 : slot - 1;           // (slot + 2) % 3: the slot of tile t - 1, free behind the barrier
        if (t + 2 < nt) issue(t + 2, slot2);
        const unsigned char* ldsK = smem + slot * STAGE;
        slot = slot == 2 ? 0 : slot + 1;
        const int kb0 = t * 32;
        // scores: acc_s[i] = S[key 32 t + (i & 3) + 8 (i >> 2) + 4 hh][query r32]
        f32x16 acc_s;
        {
            f16x8 kh[4], kl[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 8 * (ks >> 1) + 2 * (ks & 1) + hh;     // 16-byte chunk of dims 16 ks + 8 hh (hi); the lo half is 4 chunks on
                kh[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ldsK + tile256_off(r32, c)));
                kl[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ldsK + tile256_off(r32, c + 4)));
            }
            __builtin_amdgcn_sched_barrier(0);                // all 8 fragment reads in flight before the first MFMA
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc_s = mfma32(kl[0], qh[0], zero);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) acc_s = mfma32(kl[ks], qh[ks], acc_s);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc_s = mfma32(kh[ks], ql[ks], acc_s);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc_s = mfma32(kh[ks], qh[ks], acc_s);
        }
        // V fragments (hardware-transposed reads), requested now and consumed after the softmax; inline asm for the reason given in
        // attention_16_long_kernel (for the builtin hipcc would first drain the LDS-DMA copies in flight).  vf[step][db][hi / lo][half]:
        // A operand of k-step `step` (16 keys) for dim block db; element j <-> key 16 step + 8 (j >> 2) + 4 hh + (j & 3), dim
        // 32 db + 16 dh + li
        s16x4 vf[2][2][2][2];
        {
            const lds_u8* vbase = (const lds_u8*)(ldsK + 32 * 256) + v_row * 256 + 8 * (li & 3);
#define VITVS_TRX(dst, va, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(OFF))
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const unsigned va = (unsigned)(size_t)vbase + 32u * (unsigned)((4 * db + 2 * pl + dh) ^ v_sw);   // hi window, lo two on
                    VITVS_TRX(vf[0][db][pl][0], va, 0);        VITVS_TRX(vf[0][db][pl][1], va, 2048);
                    VITVS_TRX(vf[1][db][pl][0], va, 4096);     VITVS_TRX(vf[1][db][pl][1], va, 4096 + 2048);
                }
#undef VITVS_TRX
        }
        // online softmax of the lane's 16 scores (one query; the other half of its keys sits in lane ^ 32).  The vector pipe is this
        // kernel's longer resource (section 9 of the notes), so: statistics on the RAW scores with hd^-0.5 log2 e folded into the
        // exponent's fma, packed fp32 arithmetic, the key mask behind a real branch (only the sequence's last tile has masked keys; the
        // empty asm keeps hipcc from turning the branch into 48 selects on every tile)
        if (kb0 + 32 > N) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kb0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) acc_s[i] = -INFINITY;
        }
        float mloc = fmaxf(acc_s[0], acc_s[1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mloc = fmaxf(fmaxf(mloc, acc_s[i]), acc_s[i + 1]);
        mloc = fmaxf(mloc, lane_xor32(mloc));
        const float m_new = fmaxf(m_run, mloc);               // finite: a tile's first key is valid
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {
            const float alpha = fast_exp2((m_run - m_new) * kScaleLog2e);   // 0 on the first tile (O = l = 0 anyway)
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_o[db][i] *= alpha;
            m_run = m_new;
        }
        const float shift = 8.0f - m_run * kScaleLog2e;       // p' = 2^8 p: the lo halves of all but negligible weights are normal fp16
        float psum = 0.f;                                     // ONE scalar chain on purpose: hipcc packs two chains into v_pk_add_f32, which costs more than its two halves beside MFMAs
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float p[2] = {fast_exp2(__builtin_fmaf(acc_s[8 * st + j], kScaleLog2e, shift)),
                                    fast_exp2(__builtin_fmaf(acc_s[8 * st + j + 1], kScaleLog2e, shift))};
                psum += p[0];
                psum += p[1];
                const f16x2 hi = {(f16)p[0], (f16)p[1]};
                const f16x2 lo = split_lo_pair(hi, p[0], p[1]);
                ph[st][j] = hi[0];
                ph[st][j + 1] = hi[1];
                pl[st][j] = lo[0];
                pl[st][j + 1] = lo[1];
            }
        l_run += psum;
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vf[0][0][0][0]), "+v"(vf[0][0][0][1]), "+v"(vf[0][0][1][0]), "+v"(vf[0][0][1][1]), "+v"(vf[0][1][0][0]), "+v"(vf[0][1][0][1]),
                       "+v"(vf[0][1][1][0]), "+v"(vf[0][1][1][1]), "+v"(vf[1][0][0][0]), "+v"(vf[1][0][0][1]), "+v"(vf[1][0][1][0]), "+v"(vf[1][0][1][1]),
                       "+v"(vf[1][1][0][0]), "+v"(vf[1][1][0][1]), "+v"(vf[1][1][1][0]), "+v"(vf[1][1][1][1]));
        __builtin_amdgcn_sched_barrier(0);
        // O^T[dim][query] += V^T P^T, 2 k-steps of 16 keys x 2 dim blocks x (lo.hi + hi.lo + hi.hi)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const s16x4 h0 = vf[st][db][0][0], h1 = vf[st][db][0][1], l0 = vf[st][db][1][0], l1 = vf[st][db][1][1];
                const f16x8 vh = __builtin_bit_cast(f16x8, (s16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
                const f16x8 vl = __builtin_bit_cast(f16x8, (s16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
                acc_o[db] = mfma32(vl, ph[st], acc_o[db]);
                acc_o[db] = mfma32(vh, pl[st], acc_o[db]);
                acc_o[db] = mfma32(vh, ph[st], acc_o[db]);
            }
    }
    l_run += lane_xor32(l_run);                               // both lane halves hold the query's whole sum
    // acc_o[db][i] = O[query r32][dim 32 db + (i & 3) + 8 (i >> 2) + 4 hh] * l
    const float inv = 1.0f / l_run;
    const int q = q0 + r32;
    if (q < N) {
        hx2* dst = out + ((size_t)img * N + q) * D * 2;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                store_x2<false>(dst, h * 64 + 32 * db + 8 * g4 + 4 * hh,
                                f32x4{acc_o[db][4 * g4] * inv, acc_o[db][4 * g4 + 1] * inv, acc_o[db][4 * g4 + 2] * inv, acc_o[db][4 * g4 + 3] * inv});
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
                                                                 int N, int D, int n_img, float sc) {
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
                float x = acc_s[t4][r] * sc;
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

// ------------------------------------------------------------------------------------ split-f16 (f16x2)
// The fp32-class mode on the f16 matrix cores (common.h hx2): q, k, v arrive as hi / lo pairs — a head's 64 dims are 256
// bytes, [hi 0-31 | lo 0-31 | hi 32-63 | lo 32-63] — and both contractions run as hi.hi + hi.lo + lo.hi (three
// v_mfma_f32_16x16x32_f16 per k-step, fp32 accumulate; the dropped lo.lo term is < 2^-22 of the product).  Same orientation
// and online softmax as attention_16_kernel (S^T = K Q^T, P in registers as the B operand of O^T += V^T P^T), fp32
// statistics; q is NOT pre-scaled (the scores leave the matrix pipe raw and take hd^-0.5 log2 e in fp32, like
// attention_f32_kernel).  P is split in registers: p' = exp2(s - m + 8) puts the probabilities at 2^8 so that the lo halves
// of all but negligible weights are normal fp16 numbers; the factor cancels in O / l.
// LDS per key group: K tile [64 keys][256 B] with the chunk swizzle of tile256_off, V tile [64][256 B] with its 32-byte
// windows XOR-swizzled by (key & 7), so the four key rows a hardware-transposed read touches lie in four windows.
template <int KS>
__global__ __launch_bounds__(256 * KS) void attention_x2_kernel(const hx2* __restrict__ qkv, hx2* __restrict__ out, int N, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = tid >> 6, kgp = wave_all >> 2, wave = wave_all & 3, tl = tid & 255;
    unsigned char* ldsK = smem + kgp * (2 * 64 * 256);
    unsigned char* ldsV = ldsK + 64 * 256;
    const int qi = lane & 15, g = lane >> 4;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int img = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const size_t ldb = (size_t)12 * D;                         // bytes per token row: 3 D logical columns x (hi + lo)
    const unsigned char* base = reinterpret_cast<const unsigned char*>(qkv) + (size_t)img * N * ldb + h * 256;
    const unsigned char* Kp = base + 4 * (size_t)D;
    const unsigned char* Vp = base + 8 * (size_t)D;

    const int q = q0 + 16 * wave + qi;
    const int qrow = min(q, N - 1);
    f16x8 qh[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qh[s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (size_t)qrow * ldb + (8 * s + g) * 16));
        ql[s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (size_t)qrow * ldb + (8 * s + 4 + g) * 16));
    }
    // this lane's corner of a transposed V read: key row 4g + (qi >> 2) of a 16-key group, 8 bytes at 8 (qi & 3) inside a
    // 32-byte window; window w of that row sits at slot w ^ rsw (rsw = the row's low three bits, the same for every group)
    const int rsw = 4 * (g & 1) + (qi >> 2);
    lds_u8* vtr = (lds_u8*)ldsV + (4 * g + (qi >> 2)) * 256 + 8 * (qi & 3);

    f32x4 acc_o[4];
#pragma unroll
    for (int td = 0; td < 4; ++td) acc_o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int srow = tl >> 4, schunk = tl & 15;
    u32x4 rk[4], rv[4];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(kb + srow + 16 * i, N - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(Kp + (size_t)key * ldb + schunk * 16);
            rv[i] = *reinterpret_cast<const u32x4*>(Vp + (size_t)key * ldb + schunk * 16);
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
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 16 * i;
            *reinterpret_cast<u32x4*>(ldsK + tile256_off(r, schunk)) = rk[i];
            *reinterpret_cast<u32x4*>(ldsV + r * 256 + ((((schunk >> 1) ^ (r & 7)) << 5) | ((schunk & 1) << 4))) = rv[i];
        }
        __syncthreads();
        if (tt + 1 < per_group) gload(min(t + 1, ntiles - 1) * 64);
        if (t >= ntiles) continue;   // wave-uniform: this key group has run out of tiles (barriers above still taken)

        // S^T tiles: acc_s[t4][r] = S[key = kb + 16*t4 + 4g + r][q]
        f32x4 acc_s[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 kh[4], kl[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                kh[t4] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ldsK + tile256_off(16 * t4 + qi, 8 * s + g)));
                kl[t4] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ldsK + tile256_off(16 * t4 + qi, 8 * s + 4 + g)));
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kl[t4], qh[s], acc_s[t4]);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kh[t4], ql[s], acc_s[t4]);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kh[t4], qh[s], acc_s[t4]);
        }
        // online softmax of the lane's 16 scores.  Trimmed for the vector pipe, which is this kernel's longer resource (profiles/
        // r05_notes.md section 9): statistics on the RAW scores with hd^-0.5 log2 e folded into the exponent's fma, packed fp32
        // arithmetic, the key mask and the rescale of O behind real branches (masked keys: the sequence's last tile only)
        if (kb + 64 > N) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + 16 * t4 + 4 * g + r >= N) acc_s[t4][r] = -INFINITY;
        }
        const float mloc = rows_max(max16(acc_s));
        const float m_new = fmaxf(m_run, mloc);               // finite: a tile's first key is valid
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {
            const float alpha = fast_exp2((m_run - m_new) * kScaleLog2e);   // 0 on a first tile (O = l = 0 anyway)
            l_run *= alpha;
#pragma unroll
            for (int td = 0; td < 4; ++td) acc_o[td] *= alpha;
            m_run = m_new;
        }
        const float shift = 8.0f - m_run * kScaleLog2e;       // p' = 2^8 p (header)
        float psum = 0.f;                                     // ONE scalar chain on purpose: hipcc packs two chains into v_pk_add_f32, which costs more than its two halves beside MFMAs
        f16x8 phs[2], pls[2];
        // k-slot (g, j) of step u  <->  key 32u + 16(j>>2) + 4g + (j&3)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x4 a = acc_s[2 * u + (j >> 2)];
                const float p[2] = {fast_exp2(__builtin_fmaf(a[j & 3], kScaleLog2e, shift)), fast_exp2(__builtin_fmaf(a[(j & 3) + 1], kScaleLog2e, shift))};
                psum += p[0];
                psum += p[1];
                const f16x2 hi = {(f16)p[0], (f16)p[1]};
                const f16x2 lo = split_lo_pair(hi, p[0], p[1]);
                phs[u][j] = hi[0];
                phs[u][j + 1] = hi[1];
                pls[u][j] = lo[0];
                pls[u][j + 1] = lo[1];
            }
        l_run += psum;

        // O^T += V^T P^T
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f16x8 ph = phs[u], pl = pls[u];
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                // head dims 16 td .. 16 td + 15: hi halves in window 4 (td >> 1) + (td & 1) of a key row, lo halves two windows on
                const int wh = 4 * (td >> 1) + (td & 1);
                lds_u8* ph0 = vtr + (32 * u) * 256 + ((wh ^ rsw) << 5);
                lds_u8* pl0 = vtr + (32 * u) * 256 + (((wh + 2) ^ rsw) << 5);
                const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ph0);
                const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ph0 + 16 * 256));
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pl0);
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl0 + 16 * 256));
                const f16x8 vh = __builtin_bit_cast(f16x8, (s16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
                const f16x8 vl = __builtin_bit_cast(f16x8, (s16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
                acc_o[td] = mfma16(vl, ph, acc_o[td]);
                acc_o[td] = mfma16(vh, pl, acc_o[td]);
                acc_o[td] = mfma16(vh, ph, acc_o[td]);
            }
        }
    }
    if constexpr (KS >= 2) {
        // merge the key groups' online-softmax states: groups 1 .. KS-1 -> LDS -> group 0 (fixed order)
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
            const float wa = fast_exp2((m_run - m_tot) * kScaleLog2e), wb = fast_exp2((ml[0] - m_tot) * kScaleLog2e);   // (m: raw scores)
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
        hx2* dst = out + ((size_t)img * N + q) * D * 2;
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            const f32x4 o = {acc_o[td][0] * inv, acc_o[td][1] * inv, acc_o[td][2] * inv, acc_o[td][3] * inv};
            if constexpr (KS >= 2) store_x2<true>(dst, h * 64 + 16 * td + 4 * g, o);   // key-split variants only run on small grids
            else store_x2<false>(dst, h * 64 + 16 * td + 4 * g, o);
        }
    }
}

// f16x2, short sequences (N <= 256: the 197 tokens of a 224² frame) — attention_16_short_kernel's arrangement for hi / lo
// operands: one workgroup = 16 queries of one (image, head), wave w = key tile w; every wave requests its whole input at
// entry in one round trip (Q and its 64 keys, both halves, straight into MFMA operand registers; its 64 values into a
// wave-private 16 KB slab by LDS-DMA, the window swizzle of attention_x2_kernel applied on the SOURCE side) and runs one
// score -> softmax -> PV chain; the four states are merged in parallel through the slabs themselves (a wave parks its state in
// its own slab once its PV reads are done: 64 KB of LDS per workgroup, two workgroups per CU), wave w finishing head dims
// 16 w .. 16 w + 15.  Same 1-D XCD-aware item order as the 16-bit kernel.
__global__ __launch_bounds__(256) void attention_x2_short_kernel(const hx2* __restrict__ qkv, hx2* __restrict__ out, int N, int D,
                                                                 int n_img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int H = D >> 6, nqb = (N + 15) >> 4, items = n_img * H * nqb, per = (items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= items) return;
    const int pair = item / nqb;
    const int img = pair / H, h = pair - img * H, q0 = (item - pair * nqb) * 16;
    const unsigned char* qb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned row_bytes = 12u * (unsigned)D;                                  // 3 D logical columns x (hi + lo)
    const unsigned head_off = (unsigned)(img * N) * row_bytes + (unsigned)h * 256u;
    const unsigned k_off = head_off + 4u * (unsigned)D, v_off = head_off + 8u * (unsigned)D;
    unsigned char* ldsV = smem + wave * 16384;                                     // [64 keys][256 B], this wave's tile
    const int rsw = 4 * (g & 1) + (qi >> 2);
    lds_u8* vtr = (lds_u8*)ldsV + (4 * g + (qi >> 2)) * 256 + 8 * (qi & 3);
    const int kb = wave * 64;
    const bool active = kb < N;                                                    // wave-uniform
    const int q = q0 + qi;
    f16x8 qh[2], ql[2], kh[4][2], kl[4][2];
    {
        const unsigned o = head_off + __umul24((unsigned)min(q, N - 1), row_bytes) + 16u * (unsigned)g;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qh[s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qb + (o + 128u * s)));
            ql[s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qb + (o + 128u * s + 64u)));
        }
    }
    if (active) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const unsigned o = k_off + __umul24((unsigned)min(kb + 16 * t4 + qi, N - 1), row_bytes) + 16u * (unsigned)g;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kh[t4][s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qb + (o + 128u * s)));
                kl[t4][s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(qb + (o + 128u * s + 64u)));
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // copy j fills key rows 4 j .. 4 j + 3 of the slab linearly; lane l lands in slot (l & 15) of row 4 j + (l >> 4),
            // which must hold the chunk whose window is that slot's window XOR the row's low bits
            const int r = 4 * j + (lane >> 4), sl = lane & 15;
            const int c = ((((sl >> 1) ^ (r & 7)) << 1) | (sl & 1));
            const unsigned o = v_off + __umul24((unsigned)min(kb + r, N - 1), row_bytes) + 16u * (unsigned)c;
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
        for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kl[t4][s], qh[s], acc_s[t4]);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kh[t4][s], ql[s], acc_s[t4]);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) acc_s[t4] = mfma16(kh[t4][s], qh[s], acc_s[t4]);
        }
        // softmax of the lane's 16 scores, trimmed as in attention_x2_kernel: statistics on the RAW scores (m_run in raw units, the
        // merge below applies hd^-0.5 log2 e), the scale folded into the exponent's fma, packed fp32 arithmetic, the key mask behind a
        // branch (only a sequence's last tile has masked keys)
        if (kb + 64 > N) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + 16 * t4 + 4 * g + r >= N) acc_s[t4][r] = -INFINITY;
        }
        m_run = rows_max(max16(acc_s));       // the tile's first key is valid (kb < N), so this is finite
        const float shift = 8.0f - m_run * kScaleLog2e;   // p' = 2^8 p: the lo halves of all but negligible weights are normal fp16 numbers
        float psum = 0.f;                                     // ONE scalar chain on purpose: hipcc packs two chains into v_pk_add_f32, which costs more than its two halves beside MFMAs
        f16x8 phs[2], pls[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x4 a = acc_s[2 * u + (j >> 2)];
                const float p[2] = {fast_exp2(__builtin_fmaf(a[j & 3], kScaleLog2e, shift)), fast_exp2(__builtin_fmaf(a[(j & 3) + 1], kScaleLog2e, shift))};
                psum += p[0];
                psum += p[1];
                const f16x2 hi = {(f16)p[0], (f16)p[1]};
                const f16x2 lo = split_lo_pair(hi, p[0], p[1]);
                phs[u][j] = hi[0];
                phs[u][j + 1] = hi[1];
                pls[u][j] = lo[0];
                pls[u][j + 1] = lo[1];
            }
        l_run = rows_sum(psum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's V slab has landed (its own LDS-DMA copies)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f16x8 ph = phs[u], pl = pls[u];
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                const int wh = 4 * (td >> 1) + (td & 1);
                lds_u8* ph0 = vtr + (32 * u) * 256 + ((wh ^ rsw) << 5);
                lds_u8* pl0 = vtr + (32 * u) * 256 + (((wh + 2) ^ rsw) << 5);
                const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ph0);
                const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ph0 + 16 * 256));
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pl0);
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pl0 + 16 * 256));
                const f16x8 vh = __builtin_bit_cast(f16x8, (s16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
                const f16x8 vl = __builtin_bit_cast(f16x8, (s16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
                acc_o[td] = mfma16(vl, ph, acc_o[td]);
                acc_o[td] = mfma16(vh, pl, acc_o[td]);
                acc_o[td] = mfma16(vh, ph, acc_o[td]);
            }
        }
    }
    // parallel merge through the slabs: wave w parks its state in its OWN slab (its PV reads are complete: the accumulators
    // that depend on them are its operands here), wave w finishes head dims 16 w .. 16 w + 15
    f32x4* mine = reinterpret_cast<f32x4*>(ldsV) + lane;
#pragma unroll
    for (int td = 0; td < 4; ++td) mine[td * 64] = acc_o[td];
    mine[4 * 64] = f32x4{m_run, l_run, 0.f, 0.f};
    __syncthreads();
    float m_k[4], l_k[4];
    float m_tot = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 ml = reinterpret_cast<const f32x4*>(smem + k * 16384)[4 * 64 + lane];
        m_k[k] = ml[0];
        l_k[k] = ml[1];
        m_tot = fmaxf(m_tot, m_k[k]);
    }
    float l_tot = 0.f;
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = fast_exp2((m_k[k] - m_tot) * kScaleLog2e);   // (m: raw scores) 0 for a wave without keys (m = -inf)
        l_tot += l_k[k] * wk;
        const f32x4 ok = reinterpret_cast<const f32x4*>(smem + k * 16384)[wave * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += ok[r] * wk;
    }
    const float inv = 1.0f / l_tot;
    if (q < N) store_x2<true>(out + ((size_t)img * N + q) * D * 2, h * 64 + 16 * wave + 4 * g, f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv});
}

// Work division of the long-sequence kernel: `per` key tiles per workgroup (see the kernel's header).  Measured on MI355X
// (profiles/r03_notes.md, attention): the three workgroups a CU holds do NOT share it evenly — issue arbitration is by age, the
// first-dispatched workgroup runs almost unimpeded — and three resident workgroups finish tiles only 1.3x faster than one, so an
// even cut of the list over 768 co-resident workgroups (every CU the same work, all ending together) gains nothing over whole
// items (104 vs 103 us at 2 x 3137 tokens): what pays is MORE units than slots, so that the hardware's dispatcher back-fills
// CUs as units finish and the hand-offs of one unit hide behind the tiles of its neighbours.  The plan is therefore the
// fewest ranges per item, of {1, 2, 4}, that minimises (rounds of 256 units) x (tiles per unit + 1.5 tiles of hand-off); a
// range keeps at least 4 key tiles.
struct AttnPlan { int per, groups; bool divided; };
static AttnPlan attention_plan(int n_img, int N, int H) {
    const int nt = (N + 63) / 64;
    const long items = (long)((N + 127) / 128) * H * n_img;
    const long W = items * nt;
    AttnPlan best{nt, (int)items, false};
    if (N < 512) return best;                                   // short sequences in batches: whole items (tools/big_ops attnmid)
    // Beside other queues' launches (vitvs_set_option "in_flight") the chip is filled by THEIR workgroups: whole items, no hand-off
    // (three updates in flight, same box: ViT-B/8 448² 419 -> 436 updates/s, ViT-L/14 518² 703 -> 732; the 64-query kernel of the
    // mid sizes keeps its two key groups: ViT-S/14 308² 4480 -> 4333 without them)
    if (g_updates_in_flight >= 2) return best;
    double best_cost = 1e30;
    for (int sp : {1, 2, 4}) {
        if (sp > 1 && nt / sp < 4) break;
        const int per = (nt + sp - 1) / sp;
        const long units = (W + per - 1) / per;
        const double cost = (double)((units + 255) / 256) * ((double)per + (sp > 1 ? 1.5 : 0.0));
        if (cost < best_cost - 1e-9) { best_cost = cost; best = AttnPlan{per, (int)units, sp > 1}; }   // ties: the fewer ranges
    }
    return best;
}
int attention_splits(int n_img, int N, int H) { return attention_plan(n_img, N, H).divided ? 2 : 1; }   // (reported by tools only)

size_t attention_workspace_floats(int n_img, int N, int H) {
    const AttnPlan pl = attention_plan(n_img, N, H);
    return pl.divided ? (size_t)2 * (8 * ((pl.groups + 7) / 8)) * 4 * kAttnStateFloats : 0;
}
size_t attention_ticket_count(int n_img, int N, int H) {
    return attention_plan(n_img, N, H).divided ? (size_t)((N + 127) / 128) * H * n_img : 0;
}

// The pointer-only operator hook (vitvs_op_attention) has no handle to own the key-split workspace: one per (device, stream),
// grown on demand.  A launch's states and tickets are private to its stream's workspace, so two streams (or threads) running
// split attention at the same time never share a slot; launches on ONE stream are ordered by the stream.  Growth allocates a
// new block and keeps the old one alive (a launch in flight, or a captured graph, may still hold its address) and is refused
// while the stream is capturing (an allocation + memset cannot be recorded).  Handles pre-size their own workspace at creation,
// so the product path never comes here.
static const AttnWorkspace* shared_attention_workspace(hipStream_t stream, size_t floats, size_t tickets) {
    struct Slot { int dev; hipStream_t stream; AttnWorkspace ws; size_t cap_f, cap_t; };
    static std::mutex mu;
    static std::vector<Slot*> slots;            // never freed: the hook is a test / tool entry point, the set of streams is small
    std::lock_guard<std::mutex> lock(mu);
    const int dev = current_device();
    if (dev < 0) return nullptr;
    Slot* sl = nullptr;
    for (Slot* c : slots)
        if (c->dev == dev && c->stream == stream) sl = c;
    if (!sl) { sl = new Slot{dev, stream, AttnWorkspace{}, 0, 0}; slots.push_back(sl); }
    if (floats <= sl->cap_f && tickets <= sl->cap_t) return &sl->ws;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    if (floats > sl->cap_f) {
        float* p = nullptr;
        if (hipMalloc((void**)&p, floats * sizeof(float)) != hipSuccess) return nullptr;
        sl->ws.state = p;                       // the previous block stays allocated
        sl->cap_f = floats;
    }
    if (tickets > sl->cap_t) {
        int* p = nullptr;
        if (hipMalloc((void**)&p, tickets * sizeof(int)) != hipSuccess || hipMemset(p, 0, tickets * sizeof(int)) != hipSuccess) return nullptr;
        (void)hipDeviceSynchronize();           // the zeros are in place before this stream's next launch draws a ticket
        sl->ws.tickets = p;
        sl->cap_t = tickets;
    }
    return &sl->ws;
}

template <typename HT>
static int launch_attention_16(const HT* qkv, HT* out, int n_img, int N, int H, hipStream_t stream, const AttnWorkspace* ws,
                               bool q_prescaled) {
    const int D = H * 64;
    const float sc = q_prescaled ? 1.0f : kScaleLog2e;         // what is left to apply to q . k (log2 units)
    const int nt = (N + 63) / 64;
    dim3 grid(nt, H, n_img);
    // (the 16-query kernel beyond 640 workgroups: 16 images x 197 tokens 16.2 us against 12.6 us on the 64-query kernel)
    if (N <= 256 && (long)((N + 15) / 16) * H * n_img <= 640 && (long)n_img * N * 6 * D < (1l << 32)) {
        const int items = ((N + 15) / 16) * H * n_img;
        launch(attention_16_short_kernel<HT>, dim3(8 * ((items + 7) / 8)), dim3(256), 4 * 8192 + 4 * 5 * 64 * 16, stream, qkv,
               out, N, D, n_img, sc);
    } else if ((N >= 512 || (N >= 128 && (long)nt * H * n_img > 256)) && (long)n_img * N * 6 * D < (1l << 32)) {
        // 128 queries per workgroup: from 512 tokens on, and for shorter sequences once the 64-query kernel no longer fits its
        // 512-thread one-workgroup-per-CU form (each of its workgroups re-reads its head's K / V): 16 x 197 x 12 heads
        // 12.7 -> 10.7 us, 6 x 485 x 6 heads 12.9 -> 10.9 us; equal at 6..12 x 197; below that bound the 64-query kernel wins
        // (5 x 197, the rotation search: 5.3 vs 7.2 us; 5 x 485 x 6: 8.0 vs 10.7 us)  [tools/big_ops attnmid]
        const AttnPlan pl = attention_plan(n_img, N, H);
        if (pl.divided) {
            if (!ws) ws = shared_attention_workspace(stream, attention_workspace_floats(n_img, N, H), attention_ticket_count(n_img, N, H));
            if (!ws || !ws->state || !ws->tickets) return -3;
        }
        int lds = 3 * 2 * 64 * 128 + 16;
        VITVS_IF_PROBE(
            lds = g_attn_lds_bytes + 16;
            if (lds > 64 * 1024) {
                static std::atomic<unsigned long long> raised{0};
                if (raise_lds_limit((const void*)attention_16_long_kernel<HT, false>, 160 * 1024, raised)) return -3;
            }
        )
        const int g_per_xcd = (pl.groups + 7) / 8;
        // Row sums on the matrix pipe (the ONES variant above): VITVS_ATTN_ONES=1 selects it for A/B runs; the measured outcome is
        // in profiles/r05_notes.md — the default is what measured faster.
        static const bool ones_variant = [] { const char* e = getenv("VITVS_ATTN_ONES"); return e && e[0] == '1'; }();
        if (ones_variant)
            launch((attention_16_long_kernel<HT, true>), dim3(8 * g_per_xcd), dim3(256), lds, stream, qkv, out, N, D, n_img, pl.per, g_per_xcd, sc,
                   pl.divided ? ws->state : nullptr, pl.divided ? ws->tickets : nullptr);
        else
            launch((attention_16_long_kernel<HT, false>), dim3(8 * g_per_xcd), dim3(256), lds, stream, qkv, out, N, D, n_img, pl.per, g_per_xcd, sc,
                   pl.divided ? ws->state : nullptr, pl.divided ? ws->tickets : nullptr);
    } else if ((long)nt * H * n_img <= 256 && nt >= 2) {
        launch((attention_16_kernel<HT, 2>), grid, dim3(512), 2 * 2 * 64 * 128, stream, qkv, out, N, D, sc);
    } else {
        launch((attention_16_kernel<HT, 1>), grid, dim3(256), 2 * 64 * 128, stream, qkv, out, N, D, sc);
    }
    return 0;
}

int launch_attention(Precision p, const void* qkv, void* out, int n_img, int N, int H, hipStream_t stream,
                     const AttnWorkspace* ws, bool q_prescaled) {
    if (n_img <= 0 || N <= 0 || H <= 0) return -2;
    const int D = H * 64;
    const int nt = (N + 63) / 64;
    dim3 grid(nt, H, n_img);
    int rc = 0;
    if (p == PREC_X2) {
        const int items16 = ((N + 15) / 16) * H * n_img;
        if (N <= 256 && items16 <= 640 && (long)n_img * N * 12 * D < (1l << 32))
            launch(attention_x2_short_kernel, dim3(8 * ((items16 + 7) / 8)), dim3(256), 4 * 16384, stream, (const hx2*)qkv, (hx2*)out, N, D, n_img);
        else if (N >= 2048 && (long)n_img * N * 12 * D < (1l << 32)) {
            // (measured on one box, generic | this kernel, back to back on random operands: 2 x 3137 x 12 275 | 252 us, 2 x 1370 x 16 70 | 74 us,
            //  2 x 485 x 6 11.5 | 18 us: it pays from a few thousand tokens on, where the generic kernel's 64-query workgroups stage
            //  K / V through registers twice as often)
            const int items = ((N + 127) / 128) * H * n_img, g_per_xcd = (items + 7) / 8;
            launch(attention_x2_long_kernel, dim3(8 * g_per_xcd), dim3(256), 3 * 2 * 32 * 256, stream, (const hx2*)qkv, (hx2*)out, N, D, n_img, g_per_xcd);
        } else if ((long)nt * H * n_img <= 256 && nt >= 2)
            launch((attention_x2_kernel<2>), grid, dim3(512), 2 * 2 * 64 * 256, stream, (const hx2*)qkv, (hx2*)out, N, D);
        else
            launch((attention_x2_kernel<1>), grid, dim3(256), 2 * 64 * 256, stream, (const hx2*)qkv, (hx2*)out, N, D);
    } else if (p == PREC_F32) {
        launch(attention_f32_kernel, grid, dim3(256), 0, stream, (const float*)qkv, (float*)out, N, D);
    } else if (p == PREC_F16) {
        rc = launch_attention_16<f16>((const f16*)qkv, (f16*)out, n_img, N, H, stream, ws, q_prescaled);
    } else {
        rc = launch_attention_16<bf16>((const bf16*)qkv, (bf16*)out, n_img, N, H, stream, ws, q_prescaled);
    }
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
