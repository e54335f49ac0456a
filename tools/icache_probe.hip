// Is kernel-entry time dominated by cold instruction fetch?  One wave per workgroup executes the same straight-line
// block of VALU code (about 4 KB) twice; the first pass is cold, the second hot.  Stamps: s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/icache_probe tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define OP4(a) asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP16(a) OP4(a) OP4(a) OP4(a) OP4(a)
#define OP64(a) OP16(a) OP16(a) OP16(a) OP16(a)
#define OP256(a) OP64(a) OP64(a) OP64(a) OP64(a)

__global__ void probe(float* out, unsigned long long* stamps, int passes) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float a = threadIdx.x, b = 1.0001f;
    unsigned long long t[4] = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        const unsigned long long s = __builtin_readcyclecounter();
        OP256(a) OP256(a)            // 512 dependent VALU ops, 8 bytes... (VOP2 = 4 B each: 2 KB of code)
        t[p & 3] = __builtin_readcyclecounter() - s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + blockIdx.x * 8;
        s[0] = t1 - t0; s[1] = t[0]; s[2] = t[1]; s[3] = t[2];
    }
}

int main() {
    const int wgs = 256;
    float* out; unsigned long long* st;
    CHECK(hipMalloc(&out, wgs * 64 * 4)); CHECK(hipMalloc(&st, wgs * 8 * 8));
    for (int i = 0; i < 10; ++i) probe<<<wgs, 64>>>(out, st, 3);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(wgs * 8);
    CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"two timer reads back to back", "pass 0 (first execution of the block in this launch)", "pass 1", "pass 2"};
    for (int k = 0; k < 4; ++k) {
        std::vector<unsigned long long> d;
        for (int g = 0; g < wgs; ++g) d.push_back(h[g * 8 + k]);
        std::sort(d.begin(), d.end());
        printf("%-58s median %6llu  min %6llu  max %6llu cycles\n", names[k], d[wgs / 2], d[0], d[wgs - 1]);
    }
    return 0;
}
