// Bandwidth-bound pieces of the forward: image -> patch rows, LayerNorm, descriptor preparation.
// Reference arithmetic being replaced:
//   ToTensor + Normalize            dinov2_extractor.py:177-191 (mean/std :49-50)
//   patch extraction (Conv2d input) dinov2_extractor.py:141, 259; cls + pos_embed from the 3rd-party prepare_tokens
//   nn.LayerNorm(eps=1e-6)          dino_patch/block.py:57,75 (norm1 / norm2)
//   descriptor = blocks[11] output minus cls   dinov2_extractor.py:326-334; 3x3 log-bin :289-308
//   cosine normalisation x / max(|x|, 1e-8)    vitvs_v2.py:55 (torch CosineSimilarity)
#include "common.h"
#include "kernels.h"

namespace vitvs {

// ------------------------------------------------------------------------------------ patchify
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(PatchifyArgs a, T* __restrict__ Ape, float* __restrict__ x) {
    const int Tn = a.grid * a.grid;
    const int n_img = a.n_des + a.n_cur;
    const int row = blockIdx.x;
    if (row >= n_img * Tn) {  // cls rows: x[img][0][:] = cls + pos[0]
        const int img = row - n_img * Tn;
        float* dst = x + (size_t)img * (Tn + 1) * a.D;
        for (int d = threadIdx.x; d < a.D; d += blockDim.x) dst[d] = a.cls[d] + a.pos[d];
        return;
    }
    const int img = row / Tn, t = row - img * Tn;
    const int ty = t / a.grid, tx = t - ty * a.grid;
    const uint8_t* src = (img < a.n_des) ? a.des + (size_t)img * a.S * a.S * 3
                                         : a.cur + (size_t)(img - a.n_des) * a.S * a.S * 3;
    const int pp = a.patch * a.patch;
    T* dst = Ape + (size_t)row * a.Kp;
    for (int k = threadIdx.x; k < a.Kp; k += blockDim.x) {
        float v = 0.f;
        if (k < 3 * pp) {
            const int c = k / pp, rem = k - c * pp;
            const int py = rem / a.patch, px = rem - py * a.patch;
            const int yy = ty * a.stride + py, xx = tx * a.stride + px;
            const float u = (float)src[((size_t)yy * a.S + xx) * 3 + c];
            v = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), a.mean[c]), a.std[c]);
        }
        dst[k] = from_float<T>(v);
    }
}

int launch_patchify(Precision p, const PatchifyArgs& a, void* Ape, float* x, hipStream_t stream) {
    const int n_img = a.n_des + a.n_cur;
    const int rows = n_img * a.grid * a.grid + n_img;
    if (rows <= 0 || a.Kp < 3 * a.patch * a.patch) return -2;
    if (p == PREC_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(rows), dim3(256), 0, stream, a, (float*)Ape, x);
    else
        hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(rows), dim3(256), 0, stream, a, (bf16*)Ape, x);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; each lane owns NV float2 (D = 128*NV).  Two-pass moments in registers.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int M,
                                                        float eps) {
    constexpr int D = 128 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float2* src = reinterpret_cast<const float2*>(x + (size_t)row * D);
    float2 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = src[i * 64 + lane];
        s += v[i].x + v[i].y;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean;
        q += a * a + b * b;
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float2* g2 = reinterpret_cast<const float2*>(gamma);
    const float2* b2 = reinterpret_cast<const float2*>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float2 g = g2[i * 64 + lane], b = b2[i * 64 + lane];
        const float y0 = (v[i].x - mean) * rstd * g.x + b.x;
        const float y1 = (v[i].y - mean) * rstd * g.y + b.y;
        T* dst = out + (size_t)row * D + 2 * (i * 64 + lane);
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float2*>(dst) = make_float2(y0, y1);
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 h = {(bf16)y0, (bf16)y1};
            *reinterpret_cast<bf16x2*>(dst) = h;
        }
    }
}

template <typename T>
static int launch_ln_t(const float* x, const float* g, const float* b, T* out, int M, int D, float eps,
                       hipStream_t stream) {
    dim3 grid(M), block(64);  // one wave per workgroup: 394 rows spread over all CUs
    switch (D) {
        case 384: hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, stream, x, g, b, out, M, eps); break;
        case 768: hipLaunchKernelGGL((layernorm_kernel<T, 6>), grid, block, 0, stream, x, g, b, out, M, eps); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<T, 8>), grid, block, 0, stream, x, g, b, out, M, eps); break;
        case 128: hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, stream, x, g, b, out, M, eps); break;
        case 256: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, stream, x, g, b, out, M, eps); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_layernorm(Precision p, const float* x, const float* gamma, const float* beta, void* out, int M, int D,
                     float eps, hipStream_t stream) {
    if (M <= 0) return -2;
    if (p == PREC_F32) return launch_ln_t<float>(x, gamma, beta, (float*)out, M, D, eps, stream);
    return launch_ln_t<bf16>(x, gamma, beta, (bf16*)out, M, D, eps, stream);
}

// ------------------------------------------------------------------------------------ residual + LayerNorm
// Finishes a split-K linear layer: x += ls * (sum_z part[z] + bias), slices summed in index order,
// then (optionally) the next LayerNorm of the freshly updated row — one pass, one wave per row.
// Reference: dino_patch/block.py:90-96,112-115 (x + ls(f(norm(x)))), followed by the next norm.
constexpr int SMAX = 8;  // most K slices splitk_slices() ever picks

template <typename T, int NV>
__global__ __launch_bounds__(256) void residual_ln_kernel(float* __restrict__ x, const float* __restrict__ part,
                                                          int splits, const float* __restrict__ bias,
                                                          const float* __restrict__ ls, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, int M,
                                                          float eps) {
    constexpr int D = 128 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    float2* xr = reinterpret_cast<float2*>(x + (size_t)row * D);
    const float2* b2 = reinterpret_cast<const float2*>(bias);
    const float2* l2 = reinterpret_cast<const float2*>(ls);
    // All loads of the row (x and every K slice) are issued before the first add, so the pass costs one
    // memory latency, not one per slice; the slices are still summed in index order (deterministic).
    float2 v[NV], pv[SMAX][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
#pragma unroll
    for (int z = 0; z < SMAX; ++z)
        if (z < splits) {
            const float2* pz = reinterpret_cast<const float2*>(part + ((size_t)z * M + row) * D);
#pragma unroll
            for (int i = 0; i < NV; ++i) pv[z][i] = pz[i * 64 + lane];
        }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        float2 acc = pv[0][i];
#pragma unroll
        for (int z = 1; z < SMAX; ++z)
            if (z < splits) {
                acc.x += pv[z][i].x;
                acc.y += pv[z][i].y;
            }
        const float2 b = b2[c];
        acc.x += b.x;
        acc.y += b.y;
        if (ls) {
            const float2 g = l2[c];
            acc.x *= g.x;
            acc.y *= g.y;
        }
        float2 r = v[i];
        r.x += acc.x;
        r.y += acc.y;
        xr[c] = r;
        v[i] = r;
        s += r.x + r.y;
    }
    if (!gamma) return;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean;
        q += a * a + b * b;
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float2* g2 = reinterpret_cast<const float2*>(gamma);
    const float2* be2 = reinterpret_cast<const float2*>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float2 g = g2[i * 64 + lane], b = be2[i * 64 + lane];
        const float y0 = (v[i].x - mean) * rstd * g.x + b.x;
        const float y1 = (v[i].y - mean) * rstd * g.y + b.y;
        T* dst = out + (size_t)row * D + 2 * (i * 64 + lane);
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float2*>(dst) = make_float2(y0, y1);
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 h = {(bf16)y0, (bf16)y1};
            *reinterpret_cast<bf16x2*>(dst) = h;
        }
    }
}

template <typename T>
static int launch_rln_t(float* x, const float* part, int splits, const float* bias, const float* ls, const float* g,
                        const float* b, T* out, int M, int D, float eps, hipStream_t stream) {
    dim3 grid(M), block(64);  // one wave per workgroup: 394 rows spread over all CUs
#define VITVS_RLN(NV) \
    hipLaunchKernelGGL((residual_ln_kernel<T, NV>), grid, block, 0, stream, x, part, splits, bias, ls, g, b, out, M, eps)
    switch (D) {
        case 128: VITVS_RLN(1); break;
        case 256: VITVS_RLN(2); break;
        case 384: VITVS_RLN(3); break;
        case 768: VITVS_RLN(6); break;
        case 1024: VITVS_RLN(8); break;
        default: return -2;
    }
#undef VITVS_RLN
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_residual_ln(Precision p, float* x, const float* part, int splits, const float* bias, const float* ls,
                       const float* gamma, const float* beta, void* out, int M, int D, float eps, hipStream_t stream) {
    if (M <= 0 || splits < 1 || splits > SMAX) return -2;
    if (p == PREC_F32) return launch_rln_t<float>(x, part, splits, bias, ls, gamma, beta, (float*)out, M, D, eps, stream);
    return launch_rln_t<bf16>(x, part, splits, bias, ls, gamma, beta, (bf16*)out, M, D, eps, stream);
}

// ------------------------------------------------------------------------------------ descriptors
// plain: one wave per patch token: dn = x / max(||x||, 1e-8)
__global__ __launch_bounds__(256) void desc_plain_kernel(const float* __restrict__ x, float* __restrict__ dn,
                                                         float* __restrict__ raw, int n_img, int T, int D,
                                                         unsigned long long* zero_a, unsigned long long* zero_b,
                                                         int zero_count) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    {   // clears the packed (similarity, index) keys the Gram kernel will atomicMax into
        const int gid = blockIdx.x * blockDim.x + threadIdx.x;
        if (gid < zero_count) { zero_a[gid] = 0ull; zero_b[gid] = 0ull; }
    }
    if (tok >= n_img * T) return;
    const int img = tok / T, t = tok - img * T;
    const float* src = x + ((size_t)img * (T + 1) + 1 + t) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = src[d];
        s += v * v;
    }
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-8f);
    float* dst = dn + (size_t)tok * D;
    for (int d = lane; d < D; d += 64) {
        const float v = src[d];
        dst[d] = __fdiv_rn(v, nrm);
        if (raw) raw[(size_t)tok * D + d] = v;
    }
}

__global__ __launch_bounds__(256) void token_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ sq,
                                                           int n_img, int T, int D, unsigned long long* zero_a,
                                                           unsigned long long* zero_b, int zero_count) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    {
        const int gid = blockIdx.x * blockDim.x + threadIdx.x;
        if (gid < zero_count) { zero_a[gid] = 0ull; zero_b[gid] = 0ull; }
    }
    if (tok >= n_img * T) return;
    const int img = tok / T, t = tok - img * T;
    const float* src = x + ((size_t)img * (T + 1) + 1 + t) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = src[d];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) sq[tok] = s;
}

// binned: block per token; the 9 neighbour tokens (row-major dy,dx, replicate-clamped) are
// concatenated and the 9D vector normalised.
__global__ __launch_bounds__(256) void desc_binned_kernel(const float* __restrict__ x, const float* __restrict__ sq,
                                                          float* __restrict__ dn, float* __restrict__ raw, int n_img,
                                                          int T, int grid, int D) {
    const int tok = blockIdx.x;
    const int img = tok / T, t = tok - img * T;
    const int ty = t / grid, tx = t - ty * grid;
    int nb[9];
    float tot = 0.f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const int yy = min(max(ty + o / 3 - 1, 0), grid - 1);
        const int xx = min(max(tx + o % 3 - 1, 0), grid - 1);
        nb[o] = yy * grid + xx;
        tot += sq[img * T + nb[o]];
    }
    const float nrm = fmaxf(sqrtf(tot), 1e-8f);
    float* dst = dn + (size_t)tok * 9 * D;
    for (int e = threadIdx.x; e < 9 * D; e += blockDim.x) {
        const int o = e / D, d = e - o * D;
        const float v = x[((size_t)img * (T + 1) + 1 + nb[o]) * D + d];
        dst[e] = __fdiv_rn(v, nrm);
        if (raw) raw[(size_t)tok * 9 * D + e] = v;
    }
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int rows, int Dp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = src + (size_t)row * Dp;
    float acc = 0.f;
    for (int d = lane; d < Dp; d += 64) acc += s[d] * s[d];
    const float nrm = fmaxf(sqrtf(wave_sum(acc)), 1e-8f);
    for (int d = lane; d < Dp; d += 64) dst[(size_t)row * Dp + d] = __fdiv_rn(s[d], nrm);
}

int launch_normalize_rows(const float* src, float* dst, int rows, int Dp, hipStream_t stream) {
    if (rows <= 0 || Dp <= 0) return -2;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, src, dst, rows, Dp);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_descriptors(const float* x, float* dn, float* raw, float* sqnorm_ws, int n_img, int T, int grid, int D,
                       int binned, unsigned long long* zero_a, unsigned long long* zero_b, int zero_count,
                       hipStream_t stream) {
    const int toks = n_img * T;
    if (toks <= 0 || grid * grid != T || zero_count > toks * 64) return -2;
    if (!binned) {
        hipLaunchKernelGGL(desc_plain_kernel, dim3((toks + 3) / 4), dim3(256), 0, stream, x, dn, raw, n_img, T, D,
                           zero_a, zero_b, zero_count);
    } else {
        hipLaunchKernelGGL(token_sqnorm_kernel, dim3((toks + 3) / 4), dim3(256), 0, stream, x, sqnorm_ws, n_img, T, D,
                           zero_a, zero_b, zero_count);
        hipLaunchKernelGGL(desc_binned_kernel, dim3(toks), dim3(256), 0, stream, x, sqnorm_ws, dn, raw, n_img, T, grid,
                           D);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
