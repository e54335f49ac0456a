// How fast can a CU take in GEMM operands?  The frame-pair GEMMs are bound by per-CU operand intake
// (tools/op_chain: time = launch floor + ~0.7 us + (BM+BN)*K*2 bytes / ~80 GB/s per CU).  This bench issues
// exactly the GEMM's operand traffic (no MFMA) under several layouts / load instructions, 252 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o tools/intake_bench tools/intake_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum Mode { ROWS_DMA = 0, TILED_W_DMA = 1, TILED_BOTH_DMA = 2, ROWS_VGPR = 3, SAME_SLAB_DMA = 4, TILED_BOTH_VGPR = 5, ROWS_DMA_SWZ = 6 };

// grid (N/64, ceil(M/64)), WAVES waves.  Per k-tile (64 bf16 = 128 B per row) the workgroup copies 64 A rows and
// 64 W rows (16 KB).  ROWS: row-major operands with pitch K*2 bytes (what the GEMM does).  TILED: the 64x128-byte
// tile is one contiguous 8 KB block ([row tile][k tile][64][128 B]).
template <int MODE, int WAVES, int INFLIGHT>
__global__ __launch_bounds__(64 * WAVES) void intake_kernel(const unsigned char* __restrict__ A,
                                                            const unsigned char* __restrict__ W, int M, int N, int K,
                                                            unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K / 64;
    const int bx = (MODE == SAME_SLAB_DMA) ? 0 : blockIdx.x, by = (MODE == SAME_SLAB_DMA) ? 0 : blockIdx.y;
    constexpr int L = 16 / WAVES;            // copy instructions per wave per k-tile (16 groups of 8 rows)
    const unsigned char* src[L];
    size_t step[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int g8 = wave * L + j;         // 0..7: A rows, 8..15: W rows
        const int row = (g8 & 7) * 8 + (lane >> 3);
        const int c = (MODE == ROWS_DMA_SWZ) ? ((lane & 7) ^ ((row >> 1) & 7)) : (lane & 7);   // SWZ: the GEMM's source-side LDS swizzle
        const bool isA = g8 < 8;
        const bool tiled = (MODE == TILED_BOTH_DMA || MODE == TILED_BOTH_VGPR) || (MODE == TILED_W_DMA && !isA);
        const unsigned char* base = isA ? A : W;
        const int tile_row = isA ? by : bx;
        if (tiled) {
            src[j] = base + ((size_t)tile_row * nk) * 8192 + row * 128 + c * 16;
            step[j] = 8192;
        } else {
            const int r = tile_row * 64 + row;
            const int rc = isA ? min(r, M - 1) : min(r, N - 1);
            src[j] = base + (size_t)rc * K * 2 + c * 16;
            step[j] = 128;
        }
    }
    unsigned acc = 0;
    if constexpr (MODE == ROWS_VGPR || MODE == TILED_BOTH_VGPR) {
        for (int kt = 0; kt < nk; kt += INFLIGHT) {
            u32x4 v[INFLIGHT][L];
#pragma unroll
            for (int p = 0; p < INFLIGHT; ++p)
#pragma unroll
                for (int j = 0; j < L; ++j)
                    v[p][j] = *reinterpret_cast<const u32x4*>(src[j] + (size_t)(kt + p) * step[j]);
#pragma unroll
            for (int p = 0; p < INFLIGHT; ++p)
#pragma unroll
                for (int j = 0; j < L; ++j) acc ^= v[p][j][0] ^ v[p][j][3];
        }
    } else {
        // INFLIGHT k-tiles requested at once, then drained (ring of INFLIGHT stages)
        for (int kt = 0; kt < nk; kt += INFLIGHT) {
#pragma unroll
            for (int p = 0; p < INFLIGHT; ++p)
#pragma unroll
                for (int j = 0; j < L; ++j)
                    __builtin_amdgcn_global_load_lds((gbl_ptr)(src[j] + (size_t)(kt + p) * step[j]),
                                                     (lds_ptr)(smem + p * 16384 + (wave * L + j) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc = smem[tid * 4];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int WAVES, int INFLIGHT>
static void run(const char* name, hipStream_t st, const unsigned char* A, const unsigned char* W, int M, int N, int K,
                unsigned* sink, float floor_us) {
    auto kern = intake_kernel<MODE, WAVES, INFLIGHT>;
    const int lds = INFLIGHT * 16384;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    dim3 grid(N / 64, (M + 63) / 64);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 300;
    for (int i = 0; i < 20; ++i) kern<<<grid, 64 * WAVES, lds, st>>>(A, W, M, N, K, sink);
    CHECK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) kern<<<grid, 64 * WAVES, lds, st>>>(A, W, M, N, K, sink);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const float us = best * 1e3f / reps;
    const double per_wg = 128.0 * K * 2;
    printf("%-46s %6.2f us/launch   per-CU intake %6.1f GB/s (after %.1f us floor)\n", name, us, per_wg / ((us - floor_us) * 1e3), floor_us);
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int M = 394, Mp = 448;
    unsigned char *A, *W;
    unsigned* sink;
    CHECK(hipMalloc(&A, (size_t)Mp * 3072 * 2)); CHECK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(A, 1, (size_t)Mp * 3072 * 2)); CHECK(hipMemset(W, 1, (size_t)3072 * 3072 * 2));
    const float fl = 2.3f;
    for (int shape = 0; shape < 2; ++shape) {
        const int N = shape == 0 ? 2304 : 768, K = shape == 0 ? 768 : 3072;
        printf("--- %s: M %d N %d K %d, %d workgroups, %.0f KB per workgroup\n", shape == 0 ? "qkv-like" : "fc2-like (no split)", M, N,
               K, (N / 64) * 7, 128.0 * K * 2 / 1024);
        run<ROWS_DMA, 8, 2>("row-major, LDS-DMA, 8 waves, 2 tiles in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA, 8, 4>("row-major, LDS-DMA, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA_SWZ, 8, 2>("row-major, swizzled chunks, LDS-DMA, 8 waves, 2 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA_SWZ, 8, 4>("row-major, swizzled chunks, LDS-DMA, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA, 8, 6>("row-major, LDS-DMA, 8 waves, 6 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA, 4, 4>("row-major, LDS-DMA, 4 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_DMA, 16, 4>("row-major, LDS-DMA, 16 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<TILED_W_DMA, 8, 4>("W tiled, LDS-DMA, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<TILED_BOTH_DMA, 8, 4>("A+W tiled, LDS-DMA, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<TILED_BOTH_DMA, 8, 6>("A+W tiled, LDS-DMA, 8 waves, 6 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_VGPR, 8, 2>("row-major, dwordx4 -> VGPR, 8 waves, 2 in flight", st, A, W, M, N, K, sink, fl);
        run<ROWS_VGPR, 8, 4>("row-major, dwordx4 -> VGPR, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<TILED_BOTH_VGPR, 8, 4>("A+W tiled, dwordx4 -> VGPR, 8 waves, 4 in flight", st, A, W, M, N, K, sink, fl);
        run<SAME_SLAB_DMA, 8, 4>("every WG the same slab, LDS-DMA, 4 in flight", st, A, W, M, N, K, sink, fl);
    }
    return 0;
}
