// Issue cost of the vector instructions the attention softmax is made of (gfx950): cycles per wave instruction with 1, 2
// and 3 waves per SIMD, alone and mixed.  The question it answers: does v_exp_f32 (quarter rate) occupy the SIMD's vector
// issue for its 16 cycles, or can other vector work of the same or of another wave run beside it?
// Build: make -C tools valu_rate      Run: tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 256;

// 16 independent chains per lane so that no instruction waits for the one before it
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

enum Mode { FMA, EXP, PKFMA, MAX3, CVT, ADD, EXP_FMA3, EXP_PKFMA2, MFMA_ONLY, EXP_MFMA, FMA_MFMA, MFMA_IND, EXP_MFMA_IND };

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* cycles, float seed) {
    float v[16];
    f32x2 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = seed * (threadIdx.x + i); w[i] = f32x2{v[i], v[i] + 1.f}; }
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 1.f; }
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (short)(threadIdx.x + i); fb[i] = (short)(threadIdx.x * 3 + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < kIters; ++it) {
        if constexpr (MODE == FMA) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == EXP) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == PKFMA) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) & 15]));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == MAX3) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(seed), "v"(v[(i + 1) & 15]));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == CVT) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == ADD) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == EXP_FMA3) {                 // 16 exp and 48 fma, interleaved 1 : 3
#define OP(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %1, %1, %2, %1" \
                           : "+v"(v[i]), "+v"(w[i][0]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == EXP_PKFMA2) {               // 16 exp and 32 packed fma
#define OP(i) asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %1, %1, %2, %1\n v_pk_fma_f32 %1, %1, %2, %1" \
                           : "+v"(v[i]), "+v"(w[i]) : "v"(w[(i + 1) & 15]));
            REP16(OP)
#undef OP
        } else if constexpr (MODE == MFMA_ONLY) {                // 16 MFMA 32x32x16 (8 passes = 32 cycles each)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        } else if constexpr (MODE == EXP_MFMA) {                 // 16 MFMA with 2 exp behind each
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(v[i]), "+v"(w[i][0]));
            }
        } else if constexpr (MODE == FMA_MFMA) {                 // 16 MFMA with 4 fma behind each
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %0\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %0, %0, %2, %0\n v_fma_f32 %1, %1, %2, %1"
                             : "+v"(v[i]), "+v"(w[i][0]) : "v"(seed));
            }
        } else if constexpr (MODE == MFMA_IND) {                 // two independent accumulators, alternating
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc2, 0, 0, 0);
            }
        } else if constexpr (MODE == EXP_MFMA_IND) {             // the same with 2 exp behind each MFMA
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(v[2 * i]), "+v"(w[2 * i][0]));
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc2, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(v[2 * i + 1]), "+v"(w[2 * i + 1][0]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + w[i][0] + w[i][1] + acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Two roles in one workgroup of 8 waves (waves w and w + 4 share a SIMD): waves 0-3 issue only MFMAs, waves 4-7 only vector
// instructions.  which: 1 = MFMA waves only, 2 = vector waves only, 3 = both.  Do the two streams run beside each other?
template <int VOP>
__global__ __launch_bounds__(512) void role_kernel(float* out, unsigned long long* cycles, float seed, int which) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = wave < 4;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * (threadIdx.x + i);
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 1.f; }
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (short)(threadIdx.x + i); fb[i] = (short)(threadIdx.x * 3 + i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mfma_role) {
        if (which & 1)
            for (int it = 0; it < kIters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc2, 0, 0, 0);
                }
            }
    } else if (which & 2) {
        for (int it = 0; it < 4 * kIters; ++it) {
            if constexpr (VOP == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                REP16(OP)
#undef OP
            } else {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                REP16(OP)
#undef OP
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VOP>
static void measure_roles(const char* name, float* out, unsigned long long* cyc) {
    printf("%-44s", name);
    for (int which = 1; which <= 3; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(role_kernel<VOP>, dim3(256), dim3(512), 0, 0, out, cyc, 1e-3f, which);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long h[256 * 8];
        CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        m /= 256 * 4; v /= 256 * 4;
        printf("  %s: mfma %7.2f /instr  vector %6.2f /instr", which == 1 ? "MFMA waves alone" : which == 2 ? "vector waves alone" : "both",
               m / (16.0 * kIters), v / (64.0 * kIters));
    }
    printf("\n");
}

template <int MODE>
static void measure(const char* name, int instr_per_iter, float* out, unsigned long long* cyc) {
    printf("%-44s", name);
    for (int waves_per_simd = 1; waves_per_simd <= 3; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;                // one workgroup per CU: waves land round-robin on the 4 SIMDs
        if (threads > 1024) break;
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1e-3f);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1e-3f);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[256 * 12];
        const int n = 256 * threads / 64;
        CHECK(hipMemcpy(h, cyc, n * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int i = 0; i < n; ++i) sum += (double)h[i];
        // s_memtime counts at 100 MHz on this chip; report raw counts per instruction and let the fma line calibrate
        printf("  %d w/SIMD %8.3f", waves_per_simd, sum / n / kIters / instr_per_iter);
    }
    printf("   (counter ticks per wave instruction)\n");
}

int main() {
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc((void**)&out, 256 * 1024 * 4));
    CHECK(hipMalloc((void**)&cyc, 256 * 12 * 8));
    measure<FMA>("v_fma_f32", 16, out, cyc);
    measure<ADD>("v_add_f32", 16, out, cyc);
    measure<PKFMA>("v_pk_fma_f32 (2 per lane)", 16, out, cyc);
    measure<MAX3>("v_max3_f32", 16, out, cyc);
    measure<CVT>("v_cvt_pk_bf16_f32", 16, out, cyc);
    measure<EXP>("v_exp_f32", 16, out, cyc);
    measure<EXP_FMA3>("v_exp_f32 + 3 v_fma_f32 (per group of 4)", 16, out, cyc);
    measure<EXP_PKFMA2>("v_exp_f32 + 2 v_pk_fma_f32 (per group of 3)", 16, out, cyc);
    measure<MFMA_ONLY>("v_mfma_f32_32x32x16_bf16", 16, out, cyc);
    measure<EXP_MFMA>("mfma 32x32x16 + 2 v_exp_f32 (per group)", 16, out, cyc);
    measure<FMA_MFMA>("mfma 32x32x16 + 4 v_fma_f32 (per group)", 16, out, cyc);
    measure<MFMA_IND>("mfma 32x32x16, two accumulators", 16, out, cyc);
    measure<EXP_MFMA_IND>("mfma two accumulators + 2 v_exp_f32 each", 16, out, cyc);
    measure_roles<0>("roles: 16 mfma 32x32x16 | 64 v_fma_f32", out, cyc);
    measure_roles<1>("roles: 16 mfma 32x32x16 | 64 v_exp_f32", out, cyc);
    return 0;
}
