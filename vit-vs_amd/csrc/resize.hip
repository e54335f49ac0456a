// Camera frame -> extractor input: the reference resizes both frames with PIL before the path
// (vitvs_v2.py:474-475: Image.resize((S, S)), default filter = BICUBIC).  Pillow is a third-party dependency that is not
// under /root/reference (unpinned there); this restates its published algorithm (Pillow src/libImaging/Resample.c,
// 8 bits per channel): separable convolution, support = 2 * max(scale, 1) (antialiased when shrinking),
// coefficients normalised per output sample in double precision and rounded to 22-bit fixed point, horizontal pass to
// an intermediate uint8 image (rounded and clipped), then the vertical pass.  Results are bit-identical to
// PIL.Image.resize (tests/test_resize.py pins the CPU restatement on PIL 12.2, tests/test_gpu_resize.py this kernel).
//
// One launch: workgroup (y, image) filters the <= ksize input rows that output row y needs horizontally into LDS
// (the intermediate image's rows, as uint8), then combines them vertically.  Each input row is filtered by ~4
// neighbouring workgroups again: 15 MFLOP of redundant integer work instead of a second launch and a round trip.
#include <math.h>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace vitvs {

constexpr int kPrecisionBits = kResizePrecisionBits;
__device__ __forceinline__ int clip8(int v) { return resize_clip8(v); }

__global__ __launch_bounds__(256) void resize_bicubic_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const int* __restrict__ xb, const int* __restrict__ xk,
                                                             const int* __restrict__ yb, const int* __restrict__ yk, int in_h,
                                                             int in_w, int out, int ksx, int ksy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [ksy][out * 3] intermediate rows
    const int y = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int row3 = out * 3;
    const int ymin = yb[2 * y], ycnt = yb[2 * y + 1];
    const uint8_t* im = src + (size_t)img * in_h * in_w * 3;
    for (int idx = tid; idx < ycnt * row3; idx += 256) {
        const int r = idx / row3, rem = idx - r * row3;
        const int xx = rem / 3, c = rem - xx * 3;
        const int xmin = xb[2 * xx], xcnt = xb[2 * xx + 1];
        const uint8_t* p = im + ((size_t)(ymin + r) * in_w + xmin) * 3 + c;
        const int* k = xk + xx * ksx;
        int ss = 1 << (kPrecisionBits - 1);
        for (int t = 0; t < xcnt; ++t) ss += (int)p[3 * t] * k[t];
        smem[idx] = (unsigned char)clip8(ss);
    }
    __syncthreads();
    const int* k = yk + y * ksy;
    uint8_t* o = dst + ((size_t)img * out + y) * row3;
    for (int idx = tid; idx < row3; idx += 256) {
        int ss = 1 << (kPrecisionBits - 1);
        for (int t = 0; t < ycnt; ++t) ss += (int)smem[t * row3 + idx] * k[t];
        o[idx] = (uint8_t)clip8(ss);
    }
}

// Resample.c: bicubic_filter (a = -0.5), precompute_coeffs, normalize_coeffs_8bpc — same expressions, same order, double.
static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int resize_coefficients(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& coeffs) {
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    coeffs.assign((size_t)out_size * ksize, 0);
    std::vector<double> pre(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            pre[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) pre[x] /= ww;
        for (int x = 0; x < xmax; ++x)
            coeffs[(size_t)xx * ksize + x] = pre[x] < 0 ? (int)(-0.5 + pre[x] * (1 << kPrecisionBits))
                                                        : (int)(0.5 + pre[x] * (1 << kPrecisionBits));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

int launch_resize_bicubic(const uint8_t* src, uint8_t* dst, int n, int in_h, int in_w, int out, const int* xb, const int* xk,
                          int ksx, const int* yb, const int* yk, int ksy, hipStream_t stream) {
    if (n <= 0 || in_h <= 0 || in_w <= 0 || out <= 0) return -2;
    const size_t lds = (size_t)ksy * out * 3;
    if (lds > 64 * 1024) return -3;
    launch(resize_bicubic_kernel, dim3(out, n), dim3(256), lds, stream, src, dst, xb, xk, yb, yk, in_h, in_w, out, ksx, ksy);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
