// In-kernel timeline of the GEMM launches of the frame-pair forward: median cycle counts of each phase, per k-group.
// Needs a probe build of the library:
//   make -C vit-vs_amd/csrc OUT=../variants/probe/libvitvs_hip.so BUILD=build_probe EXTRA=-DVITVS_PROBE
//   hipcc -O2 -o tools/gemm_probe tools/gemm_probe.cpp -Iinclude -Lvit-vs_amd/variants/probe -lvitvs_hip
//   LD_LIBRARY_PATH=vit-vs_amd/variants/probe HIP_FORCE_DEV_KERNARG=1 tools/gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "vitvs.h"
#include "vitvs_ops.h"
extern "C" int vitvs_debug_set_gemm_probe(void* p);
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void* dalloc(size_t bytes) { void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); return p; }

static void report(const char* name, unsigned long long* dbuf, int wgs) {
    std::vector<unsigned long long> h((size_t)wgs * 8 * 8);
    CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
    static const char* ph[6] = {"entry->requests issued", "->first tile landed", "->main loop done", "->k-group swap done", "->stores issued", "->stores acked"};
    printf("%s (%d workgroups)\n", name, wgs);
    for (int kg = 0; kg < 2; ++kg) {
        printf("  k-group %d:", kg);
        long long total = 0;
        for (int s = 0; s < 6; ++s) {
            std::vector<long long> d;
            for (int g = 0; g < wgs; ++g)
                for (int w = 4 * kg; w < 4 * kg + 4; ++w) { const unsigned long long* p = &h[((size_t)g * 8 + w) * 8]; if (p[0]) d.push_back((long long)(p[s + 1] - p[s])); }
            std::sort(d.begin(), d.end());
            const long long m = d.empty() ? 0 : d[d.size() / 2];
            total += m;
            printf("  %6lld", m);
        }
        printf("   = %lld cycles\n", total);
    }
    printf("  phases:"); for (int s = 0; s < 6; ++s) printf(" [%s]", ph[s]); printf("\n");
}

int main() {
    const int prec = VITVS_BF16, M = 394, D = 768, hidden = 3072;
    hipStream_t st; CHECK(hipStreamCreate(&st));
    void *xn = dalloc((size_t)M * D * 2), *qkv = dalloc((size_t)M * 3 * D * 2), *hid = dalloc((size_t)M * hidden * 2);
    void *w1 = dalloc((size_t)3 * D * D * 2), *w2 = dalloc((size_t)hidden * D * 2), *w3 = dalloc((size_t)D * hidden * 2), *w4 = dalloc((size_t)D * D * 2);
    float *bias = (float*)dalloc(hidden * 4), *part = (float*)dalloc((size_t)8 * M * D * 4);
    unsigned long long* dbuf = (unsigned long long*)dalloc((size_t)512 * 8 * 8 * 8);
    if (vitvs_debug_set_gemm_probe(dbuf)) { printf("probe symbol not set\n"); return 1; }
    const int s_proj = vitvs_op_splitk_slices(prec, M, D, D), s_fc2 = vitvs_op_splitk_slices(prec, M, D, hidden);
    for (int i = 0; i < 20; ++i) vitvs_op_linear(prec, xn, w1, bias, qkv, M, 3 * D, D, 0, st);
    CHECK(hipStreamSynchronize(st)); report("qkv 394x2304x768 (64x64 tiles)", dbuf, 36 * 7);
    CHECK(hipMemset(dbuf, 0, (size_t)512 * 8 * 8 * 8));
    for (int i = 0; i < 20; ++i) vitvs_op_linear(prec, xn, w2, bias, hid, M, hidden, D, 1, st);
    CHECK(hipStreamSynchronize(st)); report("fc1+GELU 394x3072x768 (64x96 tiles)", dbuf, 32 * 7);
    CHECK(hipMemset(dbuf, 0, (size_t)512 * 8 * 8 * 8));
    for (int i = 0; i < 20; ++i) vitvs_op_linear_partial(prec, hid, w3, part, M, D, hidden, s_fc2, st);
    CHECK(hipStreamSynchronize(st)); report("fc2 partial 394x768x3072, 3 slices", dbuf, 12 * 7 * s_fc2);
    CHECK(hipMemset(dbuf, 0, (size_t)512 * 8 * 8 * 8));
    for (int i = 0; i < 20; ++i) vitvs_op_linear_partial(prec, xn, w4, part, M, D, D, s_proj, st);
    CHECK(hipStreamSynchronize(st)); report("proj partial 394x768x768, 3 slices", dbuf, 12 * 7 * s_proj);
    return 0;
}
