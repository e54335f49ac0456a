// Dense token correspondence: cosine-similarity Gram of the (already L2-normalised) descriptors
// of the desired and the current frame, fused with the row/column max+argmax, so the T x T
// similarity matrix is never written to HBM (it is 39 MB at T = 3136).
//
// Reference arithmetic being replaced (vitvs_v2.py):
//   chunk_cosine_sim            :49-56   S[i][j] = <d1_i/|d1_i|, d2_j/|d2_j|>   (python loop over T tokens)
//   sim_1, nn_1 = max(S, -1)    :80      best current-frame token for every desired-frame token
//   sim_2, nn_2 = max(S, -2)    :81      and the reverse; first index on ties
// The arithmetic is exact fp32 on the f32 MFMA (v_mfma_f32_16x16x4_f32): in the fp32 mode always, in the 16-bit modes up to 1023
// tokens.  From 1024 tokens on the 16-bit modes (whose descriptors carry the bf16 / fp16 forward's ~1e-2 error anyway) take the
// same Gram on the f16 matrix cores at 16x the rate from a two-term split x * 2^10 = hi + lo (fp16 each, 22 significant bits):
//   S * 2^20 = hi1.hi2 + hi1.lo2 + lo1.hi2      (the dropped lo1.lo2 term is < 2^-22 relative)
// as ONE contraction over 3 D: desired rows are stored [hi | hi | lo], current rows [hi | lo | hi] (split_desc_kernel).
// ViT-B/8 448² (3136 tokens): 156 us -> see profiles/r02_notes.md.
//
// Binned descriptors (use_feature_binning: the reference's shipped default, config.yaml:17; _log_bin,
// dinov2_extractor.py:265-311): the descriptor of token i is the concatenation of the 3 x 3 neighbourhood's tokens (replicate-
// clamped at the border), so the dot product of two binned descriptors is the sum over the nine offsets d of the PLAIN dot products
// <t_{n_d(i)}, t'_{n_d(j)}>: the 9 D-wide Gram is a 9-point "diagonal" stencil over the D-wide Gram of the raw tokens,
//   S[i][j] = ( sum_d G[n_d(i)][n_d(j)] ) / ( |b_i| |b'_j| ),   G[i][j] = <t_i, t'_j>,   |b_i|^2 = sum_d |t_{n_d(i)}|^2 .
// The velocity path therefore never builds the 9 D-wide descriptors: raw Gram (1 / 9 of the FLOPs and operand bytes) into a
// T x T workspace, then gram_stencil_argmax_kernel (nine L2-resident reads per similarity).  DINOv2 ViT-S/14 308² (484 tokens,
// 9 x 384 = 3456): 30 us + 5 us of descriptor building -> see profiles/r04_notes.md.  extract_descriptors(bin=True) still
// returns the concatenated descriptors (elementwise.hip desc_binned_kernel); vitvs_correspond_dev takes whatever rows it is given.
#include <algorithm>

#include "gemm_core.h"
#include "kernels.h"

namespace vitvs {

// Tile order.  A tile of the Gram reads one BM-row panel of the desired descriptors and one BN-row panel of the current ones, both
// over the whole contraction; panels are shared only through the XCD's private L2 (4 MB, not coherent with the other seven), and a
// workgroup with linear id i runs on XCD i % 8.  The plain (column, row) grid therefore put every panel on all eight XCDs:
// 313.8 MB fetched for 28.9 MB of split descriptors at 3136 tokens (10.9 x; profiles/r03_pmc_traffic_vitb8_448.json), 8.7 MB for
// 1.2 MB at 196 tokens.  Here the tiles are listed band by band (a band = `hb` row panels over all column panels, walked column
// by column, the band's rows fastest) and XCD x — workgroup ids x, x + 8, ... — takes the x-th eighth of that list, balanced to
// one tile: the tiles resident on an XCD at any time share a column panel `hb` ways and the band's `hb` row panels among all of
// them.  hb ~ sqrt(tiles / 8) makes an XCD's share square; with square shares the eight L2s cannot fetch less than sqrt(8) = 2.83 x
// the operands (each needs T / sqrt(8) rows of both frames).
struct GramMap { int i0, j0; bool valid; };
template <int BM, int BN>
__device__ __forceinline__ GramMap gram_tile(int T, int hb) {
    const int ty = (T + BM - 1) / BM, tx = (T + BN - 1) / BN, tiles = ty * tx;
    const int per = (tiles + 7) >> 3;
    const int local = (int)(blockIdx.x >> 3), t = (int)(blockIdx.x & 7) * per + local;
    if (local >= per || t >= tiles) return GramMap{0, 0, false};
    const int band = t / (hb * tx), rem = t - band * hb * tx;
    const int rows = min(hb, ty - band * hb);                  // the last band may be lower
    const int c = rem / rows, r = rem - c * rows;
    return GramMap{(band * hb + r) * BM, c * BN, true};
}
// host side: band height and workgroups per XCD
static int gram_band_rows(int T, int BM, int BN, int* per_xcd) {
    const int ty = (T + BM - 1) / BM, tx = (T + BN - 1) / BN;
    const int per = (ty * tx + 7) / 8;
    *per_xcd = per;
    int hb = 1;
    while ((hb + 1) * (hb + 1) <= per) ++hb;                    // floor(sqrt(per))
    if (hb * (hb + 1) <= per) ++hb;                             // round to nearest
    return std::min(hb, ty);
}

// E: float (Dp = descriptor length, scale 1) or f16 (Dp = 3 x descriptor length of the split rows, scale 2^-20)
template <typename E, int BM, int BN, int KG>
__global__ __launch_bounds__(256 * KG) void gram_argmax_kernel(const E* __restrict__ dn, int T, int Dp, int n_pairs,
                                                          int des_shared, unsigned long long* __restrict__ row_best,
                                                          unsigned long long* __restrict__ col_best, int hb) {
    using Tile = GemmTile<BM, BN, KG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    const int n_des = des_shared ? 1 : n_pairs;
    const E* d1 = dn + (size_t)(des_shared ? 0 : b) * T * Dp;   // desired frame tokens (rows i)
    const E* d2 = dn + (size_t)(n_des + b) * T * Dp;            // current frame tokens (cols j)
    const GramMap tile = gram_tile<BM, BN>(T, hb);
    if (!tile.valid) return;                                    // (the whole workgroup: before any barrier)
    const int i0 = tile.i0, j0 = tile.j0;
    f32x4 acc[Tile::NT][Tile::MT];
    gemm_mainloop<E, BM, BN, KG>(d1, d2, Dp, Dp, T, T, i0, j0, 0, Dp, smem, acc);
    if constexpr (sizeof(E) == 2) {                                 // undo the 2^10 x 2^10 of the split (exact)
#pragma unroll
        for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ni][mi][r] *= 0x1p-20f;
    }
    // with two k-groups each group holds full sums only for the column tiles it owns; mask the rest
    const int kg = (KG == 2) ? k_group() : 0;
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi)
            if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) acc[ni][mi] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = wave & 1, wn = wave >> 1;
    unsigned long long* rb = row_best + (size_t)b * T;
    unsigned long long* cb = col_best + (size_t)b * T;

    // row best (over j) for i = m: lane-local over (ni, reg), then across the 4 k-groups of lanes
#pragma unroll
    for (int mi = 0; mi < Tile::MT; ++mi) {
        const int i = i0 + wm * Tile::WM + mi * 16 + (lane & 15);
        unsigned long long key = 0ull;
#pragma unroll
        for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4) + r;
                const unsigned long long k = (j < T) ? pack_best(acc[ni][mi][r], (unsigned)j) : 0ull;
                key = (k > key) ? k : key;
            }
        unsigned long long o = shfl_xor_u64(key, 16);
        key = (o > key) ? o : key;
        o = shfl_xor_u64(key, 32);
        key = (o > key) ? o : key;
        if ((lane >> 4) == 0 && i < T) atomicMax(rb + i, key);
    }
    // column best (over i) for j = n: lane-local over mi, then across the 16 lanes of a k-group
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4) + r;
            unsigned long long key = 0ull;
#pragma unroll
            for (int mi = 0; mi < Tile::MT; ++mi) {
                const int i = i0 + wm * Tile::WM + mi * 16 + (lane & 15);
                const unsigned long long k = (i < T) ? pack_best(acc[ni][mi][r], (unsigned)i) : 0ull;
                key = (k > key) ? k : key;
            }
            key = row16_max_u64(key);
            if ((lane & 15) == 0 && j < T) atomicMax(cb + j, key);
        }
}

// S[b][i][j] = <row i of image a(b), row j of image c(b)> over Dp elements; rows of image k start at src + k * img_stride and are
// ld apart (normalised descriptors: img_stride = T * Dp, ld = Dp; raw tokens of the residual stream: src = x + D, img_stride =
// (T + 1) * D, ld = D — the cls row is skipped by the base and the stride).  Tiles in the band order of gram_tile.
template <int BM, int BN, int KG>
__global__ __launch_bounds__(256 * KG) void gram_dense_kernel(const float* __restrict__ src, long img_stride, int ld, int T, int Dp,
                                                              int n_pairs, int des_shared, float* __restrict__ S, int hb) {
    using Tile = GemmTile<BM, BN, KG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    const int n_des = des_shared ? 1 : n_pairs;
    const float* d1 = src + (size_t)(des_shared ? 0 : b) * img_stride;
    const float* d2 = src + (size_t)(n_des + b) * img_stride;
    const GramMap tile = gram_tile<BM, BN>(T, hb);
    if (!tile.valid) return;
    const int i0 = tile.i0, j0 = tile.j0;
    f32x4 acc[Tile::NT][Tile::MT];
    gemm_mainloop<float, BM, BN, KG>(d1, d2, ld, ld, T, T, i0, j0, 0, Dp, smem, acc);
    const int kg = (KG == 2) ? k_group() : 0;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = wave & 1, wn = wave >> 1;
    float* out = S + (size_t)b * T * T;
#pragma unroll
    for (int ni = 0; ni < Tile::NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < Tile::MT; ++mi) {
            if (KG == 2 && tile_owner<Tile::NT, Tile::MT>(ni, mi) != kg) continue;    // the other k-group holds this tile's sums
            const int i = i0 + wm * Tile::WM + mi * 16 + (lane & 15);
            const int j = j0 + wn * Tile::WN + ni * 16 + 4 * (lane >> 4);
            if (i >= T) continue;
            float* dst = out + (size_t)i * T + j;
            if (j + 3 < T && (T & 3) == 0) *reinterpret_cast<f32x4*>(dst) = acc[ni][mi];
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < T) dst[r] = acc[ni][mi][r];
        }
}

// Binned similarities from the raw Gram (header): one workgroup = a 32 x 32 tile of (desired token i, current token j).
//   G   [n_pairs][T][T]   raw dot products of the tokens (gram_dense_kernel on the residual stream)
//   sq  [frames][T]       |t|^2 of every token (token_sqnorm_kernel), frames in the call's order: desired first
// The nine terms are summed in the descriptor's order (dy, dx row-major: dinov2_extractor.py:302-307).
__global__ __launch_bounds__(256) void gram_stencil_argmax_kernel(const float* __restrict__ G, const float* __restrict__ sq, int T,
                                                                  int grid, int n_pairs, int des_shared,
                                                                  unsigned long long* __restrict__ row_best,
                                                                  unsigned long long* __restrict__ col_best) {
    __shared__ float tile[32][33];
    __shared__ float rn[64];                                       // 1 / |b| of the tile's 32 desired and 32 current tokens
    const int b = blockIdx.z, tid = threadIdx.x;
    const int n_des = des_shared ? 1 : n_pairs;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const float* Gb = G + (size_t)b * T * T;
    auto clampi = [&](int v) { return min(max(v, 0), grid - 1); };
    if (tid < 64) {
        const int tok = (tid < 32 ? i0 : j0 - 32) + tid;
        const float* s = sq + (size_t)(tid < 32 ? (des_shared ? 0 : b) : n_des + b) * T;
        float tot = 0.f;
        if (tok < T) {
            const int y = tok / grid, x = tok - y * grid;
#pragma unroll
            for (int o = 0; o < 9; ++o) tot += s[clampi(y + o / 3 - 1) * grid + clampi(x + o % 3 - 1)];
        }
        rn[tid] = __fdiv_rn(1.0f, fmaxf(sqrtf(tot), 1e-8f));
    }
    const int jj = tid & 31, ii0 = tid >> 5;
    const int j = j0 + jj, jy = j / grid, jx = j - jy * grid;
    int nj[9];
#pragma unroll
    for (int o = 0; o < 9; ++o) nj[o] = clampi(jy + o / 3 - 1) * grid + clampi(jx + o % 3 - 1);
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ii0 + 8 * r, iy = i / grid, ix = i - iy * grid;
        float a = 0.f;
        if (i < T && j < T) {
#pragma unroll
            for (int o = 0; o < 9; ++o) a += Gb[(size_t)(clampi(iy + o / 3 - 1) * grid + clampi(ix + o % 3 - 1)) * T + nj[o]];
        }
        acc[r] = a;
    }
    __syncthreads();                                               // rn is in place
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[ii0 + 8 * r][jj] = acc[r] * rn[ii0 + 8 * r] * rn[32 + jj];
    __syncthreads();
    if (tid < 32) {                                                // best current token for desired token i0 + tid
        const int i = i0 + tid;
        if (i < T) {
            unsigned long long key = 0ull;
            for (int c = 0; c < 32 && j0 + c < T; ++c) {
                const unsigned long long k = pack_best(tile[tid][c], (unsigned)(j0 + c));
                key = k > key ? k : key;
            }
            atomicMax(row_best + (size_t)b * T + i, key);
        }
    } else if (tid < 64) {                                         // best desired token for current token j0 + tid - 32
        const int c = tid - 32;
        if (j0 + c < T) {
            unsigned long long key = 0ull;
            for (int r = 0; r < 32 && i0 + r < T; ++r) {
                const unsigned long long k = pack_best(tile[r][c], (unsigned)(i0 + r));
                key = k > key ? k : key;
            }
            atomicMax(col_best + (size_t)b * T + j0 + c, key);
        }
    }
}

__global__ void decode_best_kernel(const unsigned long long* __restrict__ rb, const unsigned long long* __restrict__ cb,
                                   int T, int32_t* nn1, int32_t* nn2, float* sim1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    nn1[i] = (int32_t)best_index(rb[i]);
    sim1[i] = best_value(rb[i]);
    nn2[i] = (int32_t)best_index(cb[i]);
}

__global__ void encode_best_kernel(const int32_t* __restrict__ nn1, const int32_t* __restrict__ nn2,
                                   const float* __restrict__ sim1, int T, unsigned long long* rb,
                                   unsigned long long* cb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    rb[i] = pack_best(sim1[i], (unsigned)nn1[i]);
    cb[i] = pack_best(0.f, (unsigned)nn2[i]);
}

int launch_decode_best(const unsigned long long* row_best, const unsigned long long* col_best, int T, int32_t* nn1,
                       int32_t* nn2, float* sim1, hipStream_t stream) {
    launch(decode_best_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, row_best, col_best, T, nn1, nn2,
                       sim1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_encode_best(const int32_t* nn1, const int32_t* nn2, const float* sim1, int T, unsigned long long* row_best,
                       unsigned long long* col_best, hipStream_t stream) {
    launch(encode_best_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, nn1, nn2, sim1, T, row_best,
                       col_best);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// dn fp32 [frames][T][Dp] -> dh fp16 [frames][T][3 Dp]: frames below n_des are desired frames (rows of the Gram), the rest current
__global__ __launch_bounds__(256) void split_desc_kernel(const float* __restrict__ dn, f16* __restrict__ dh, int Dp, long chunks,
                                                         long des_chunks) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;           // one 8-element chunk of one row
    if (c >= chunks) return;
    const int per_row = Dp >> 3;
    const long row = c / per_row;
    const int col = (int)(c - row * per_row) << 3;
    const float4 a = *reinterpret_cast<const float4*>(dn + row * Dp + col);
    const float4 b = *reinterpret_cast<const float4*>(dn + row * Dp + col + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float s = x[i] * 1024.f;                             // |x| <= 1: no overflow; exact
        const f16 h = (f16)s;
        hi[i] = h;
        lo[i] = (f16)(s - (float)h);                               // exact difference, rounded once
    }
    f16* dst = dh + row * (3l * Dp) + col;
    const bool desired = c < des_chunks;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + Dp) = desired ? hi : lo;
    *reinterpret_cast<f16x8*>(dst + 2 * Dp) = desired ? lo : hi;
}

size_t gram_split_elems(int n_frames, int T, int Dp) { return (size_t)n_frames * T * 3 * Dp; }

int launch_split_desc(const float* dn, void* dh, int T, int Dp, int n_pairs, int des_shared, hipStream_t stream) {
    if (T <= 0 || n_pairs <= 0 || (Dp % 64) != 0 || !dh) return -2;
    const int n_des = des_shared ? 1 : n_pairs, n_frames = n_des + n_pairs;
    if ((long)n_frames * T * 3 * Dp * 2 >= (1l << 32)) return -2;  // 32-bit operand offsets in the main loop
    const long chunks = (long)n_frames * T * (Dp >> 3), des_chunks = (long)n_des * T * (Dp >> 3);
    launch(split_desc_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, dn, (f16*)dh, Dp, chunks, des_chunks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_gram_argmax_split(const void* dh, int T, int Dp, int n_pairs, int des_shared, unsigned long long* row_best,
                             unsigned long long* col_best, hipStream_t stream) {
    if (T <= 0 || n_pairs <= 0 || (Dp % 64) != 0 || !dh) return -2;
    const long t128 = (long)((T + 127) / 128) * ((T + 127) / 128) * n_pairs;
    if (t128 >= 256) {                                             // enough 128 x 128 tiles for every CU (3136 tokens: 625)
        using Tile = GemmTile<128, 128, 1>;
        static std::atomic<unsigned long long> raised{0};
        if (raise_lds_limit(reinterpret_cast<const void*>(&gram_argmax_kernel<f16, 128, 128, 1>), Tile::LDS_BYTES, raised)) return -1;
        int per = 0;
        const int hb = gram_band_rows(T, 128, 128, &per);
        launch((gram_argmax_kernel<f16, 128, 128, 1>), dim3(8 * per, 1, n_pairs), dim3(256), Tile::LDS_BYTES, stream, (const f16*)dh, T,
               3 * Dp, n_pairs, des_shared, row_best, col_best, hb);
    } else {                                                       // 1369 tokens: 121 tiles of 128 x 128, 484 of 64 x 64
        using Tile = GemmTile<64, 64, 1>;
        static std::atomic<unsigned long long> raised{0};
        if (raise_lds_limit(reinterpret_cast<const void*>(&gram_argmax_kernel<f16, 64, 64, 1>), Tile::LDS_BYTES, raised)) return -1;
        int per = 0;
        const int hb = gram_band_rows(T, 64, 64, &per);
        launch((gram_argmax_kernel<f16, 64, 64, 1>), dim3(8 * per, 1, n_pairs), dim3(256), Tile::LDS_BYTES, stream, (const f16*)dh, T,
               3 * Dp, n_pairs, des_shared, row_best, col_best, hb);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_gram_argmax(const float* dn, int T, int Dp, int n_pairs, int des_shared, unsigned long long* row_best,
                       unsigned long long* col_best, hipStream_t stream) {
    if (T <= 0 || n_pairs <= 0 || (Dp % 32) != 0) return -2;
    int per = 0;
    if (T <= 512 && (Dp / 32) % 2 == 0) {
        // few tokens: 32x32 tiles (49 workgroups at T = 196 instead of 16) with two k-groups
        using Tile = GemmTile<32, 32, 2>;
        const int hb = gram_band_rows(T, 32, 32, &per);
        launch((gram_argmax_kernel<float, 32, 32, 2>), dim3(8 * per, 1, n_pairs), dim3(Tile::THREADS), Tile::LDS_BYTES, stream, dn, T, Dp,
               n_pairs, des_shared, row_best, col_best, hb);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    constexpr int lds = GemmTile<64, 64, 1>::LDS_BYTES;
    const int hb = gram_band_rows(T, 64, 64, &per);
    launch((gram_argmax_kernel<float, 64, 64, 1>), dim3(8 * per, 1, n_pairs), dim3(256), lds, stream, dn, T, Dp, n_pairs, des_shared,
           row_best, col_best, hb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int launch_gram_dense_strided(const float* src, long img_stride, int ld, int T, int Dp, int n_pairs, int des_shared, float* S,
                                     hipStream_t stream) {
    if (T <= 0 || n_pairs <= 0 || (Dp % 32) != 0) return -2;
    // 32-bit byte offsets in the main loop: every row the launch touches lies within 4 GiB of its image's first row
    if ((long long)T * ld * 4 >= (1ll << 32)) return -2;
    int per = 0;
    if (T <= 512 && (Dp / 32) % 2 == 0) {                       // few tokens: 32 x 32 tiles with two k-groups, like the fused arg-max form
        using Tile = GemmTile<32, 32, 2>;
        const int hb = gram_band_rows(T, 32, 32, &per);
        launch((gram_dense_kernel<32, 32, 2>), dim3(8 * per, 1, n_pairs), dim3(Tile::THREADS), Tile::LDS_BYTES, stream, src, img_stride, ld,
               T, Dp, n_pairs, des_shared, S, hb);
    } else {
        constexpr int lds = GemmTile<64, 64, 1>::LDS_BYTES;
        const int hb = gram_band_rows(T, 64, 64, &per);
        launch((gram_dense_kernel<64, 64, 1>), dim3(8 * per, 1, n_pairs), dim3(256), lds, stream, src, img_stride, ld, T, Dp, n_pairs,
               des_shared, S, hb);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_gram_dense(const float* dn, int T, int Dp, int n_pairs, int des_shared, float* S, hipStream_t stream) {
    return launch_gram_dense_strided(dn, (long)T * Dp, Dp, T, Dp, n_pairs, des_shared, S, stream);
}

int launch_gram_raw_tokens(const float* x, int T, int D, int n_pairs, int des_shared, float* G, hipStream_t stream) {
    return launch_gram_dense_strided(x + D, (long)(T + 1) * D, D, T, D, n_pairs, des_shared, G, stream);
}

int launch_gram_stencil_argmax(const float* G, const float* sq, int T, int grid, int n_pairs, int des_shared,
                               unsigned long long* row_best, unsigned long long* col_best, hipStream_t stream) {
    if (T <= 0 || n_pairs <= 0 || grid * grid != T) return -2;
    launch(gram_stencil_argmax_kernel, dim3((T + 31) / 32, (T + 31) / 32, n_pairs), dim3(256), 0, stream, G, sq, T, grid, n_pairs,
           des_shared, row_best, col_best);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
