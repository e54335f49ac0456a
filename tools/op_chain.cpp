// True per-launch cost of each operator of the ViT-B/16 224² frame-pair forward: N identical launches
// back to back on one stream, plain launches (no per-kernel events: an event-stamped launch costs ~1.7 us
// more than a plain one on this platform, see tools/launch_floor.hip), total time / N.
// Build: hipcc -O2 -o tools/op_chain tools/op_chain.cpp -Iinclude -Lvit-vs_amd -lvitvs_hip -Wl,-rpath,'$ORIGIN/../vit-vs_amd'
// Run  : HIP_FORCE_DEV_KERNARG=1 tools/op_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <time.h>
#include <vector>

#include "vitvs.h"
#include "vitvs_ops.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void* dalloc(size_t bytes, int fill) {
    void* p;
    CHECK(hipMalloc(&p, bytes));
    CHECK(hipMemset(p, fill, bytes));
    return p;
}

static void run(const char* name, hipStream_t st, int reps, const std::function<int()>& op) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) if (op()) { printf("%s: launch failed\n", name); return; }
    CHECK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) op();
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %6.2f us/launch\n", name, best * 1e3 / reps);
}

// `tools/op_chain queues`: the same operators as chains on k = 1 .. 3 high-priority streams (a hardware queue each, own
// activations, shared rotating weights), captured as graphs and replayed at the same time: which launches of the forward
// overlap with their like on another queue, and which only take turns?
struct Acts { void *xn, *qkv, *attn, *hid; float *x, *part; };
static int queues_mode(int prec) {
    const size_t es = prec == VITVS_F32 ? 4 : 2;
    const int n_img = getenv("OPC_IMAGES") ? atoi(getenv("OPC_IMAGES")) : 2;   // OPC_IMAGES=1: one frame per chain (197 rows)
    if (getenv("OPC_HINT")) vitvs_op_plan_in_flight(atoi(getenv("OPC_HINT")));   // plan as a handle with that "in_flight" option
    const int N = 197, M = n_img * N, D = 768, H = 12, hidden = 3072, reps = n_img > 4 ? 60 : 240;
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st[3];
    Acts a[3];
    for (int q = 0; q < 3; ++q) {
        CHECK(hipStreamCreateWithPriority(&st[q], hipStreamNonBlocking, hi));
        a[q] = Acts{dalloc((size_t)M * D * es, 0), dalloc((size_t)M * 3 * D * es, 0), dalloc((size_t)M * D * es, 0),
                    dalloc((size_t)M * hidden * es, 0), (float*)dalloc((size_t)M * D * 4, 0), (float*)dalloc((size_t)8 * M * D * 4, 0)};
    }
    void *wqkv[12], *wproj[12], *wfc1[12], *wfc2[12];
    for (int b = 0; b < 12; ++b) {
        wqkv[b] = dalloc((size_t)3 * D * D * es, 0); wproj[b] = dalloc((size_t)D * D * es, 0);
        wfc1[b] = dalloc((size_t)hidden * D * es, 0); wfc2[b] = dalloc((size_t)D * hidden * es, 0);
    }
    float* bias = (float*)dalloc((size_t)hidden * 4, 0);
    float* gamma = (float*)dalloc((size_t)D * 4, 0);
    float* beta = (float*)dalloc((size_t)D * 4, 0);
    const int s_proj = vitvs_op_splitk_slices(prec, M, D, D), s_fc2 = vitvs_op_splitk_slices(prec, M, D, hidden);
    struct Op { const char* name; int launches; std::function<int(int, int)> f; };   // f(queue, i)
    std::vector<Op> ops = {
        {"qkv   linear 394x2304x768", 1, [&](int q, int i) { return vitvs_op_linear(prec, a[q].xn, wqkv[i % 12], bias, a[q].qkv, M, 3 * D, D, 0, st[q]); }},
        {"attention 2 x 12 heads x 197", 1, [&](int q, int) { return vitvs_op_attention(prec, a[q].qkv, a[q].attn, n_img, N, H, st[q]); }},
        {"proj  partial 394x768x768", 1, [&](int q, int i) { return vitvs_op_linear_partial(prec, a[q].attn, wproj[i % 12], a[q].part, M, D, D, s_proj, st[q]); }},
        {"residual_ln (3 slices) + LayerNorm", 1, [&](int q, int) { return vitvs_op_residual_ln(prec, a[q].x, a[q].part, s_fc2, bias, nullptr, gamma, beta, a[q].xn, M, D, 1e-6f, st[q]); }},
        {"fc1   linear+GELU 394x3072x768", 1, [&](int q, int i) { return vitvs_op_linear(prec, a[q].xn, wfc1[i % 12], bias, a[q].hid, M, hidden, D, 1, st[q]); }},
        {"fc2   partial 394x768x3072", 1, [&](int q, int i) { return vitvs_op_linear_partial(prec, a[q].hid, wfc2[i % 12], a[q].part, M, D, hidden, s_fc2, st[q]); }},
        {"block (7 launches)", 7, [&](int q, int i) {
            int rc = vitvs_op_linear(prec, a[q].xn, wqkv[i % 12], bias, a[q].qkv, M, 3 * D, D, 0, st[q]);
            rc |= vitvs_op_attention(prec, a[q].qkv, a[q].attn, n_img, N, H, st[q]);
            rc |= vitvs_op_linear_partial(prec, a[q].attn, wproj[i % 12], a[q].part, M, D, D, s_proj, st[q]);
            rc |= vitvs_op_residual_ln(prec, a[q].x, a[q].part, s_proj, bias, nullptr, gamma, beta, a[q].xn, M, D, 1e-6f, st[q]);
            rc |= vitvs_op_linear(prec, a[q].xn, wfc1[i % 12], bias, a[q].hid, M, hidden, D, 1, st[q]);
            rc |= vitvs_op_linear_partial(prec, a[q].hid, wfc2[i % 12], a[q].part, M, D, hidden, s_fc2, st[q]);
            rc |= vitvs_op_residual_ln(prec, a[q].x, a[q].part, s_fc2, bias, nullptr, gamma, beta, a[q].xn, M, D, 1e-6f, st[q]);
            return rc; }},
    };
    printf("rows per launch %d, split-K slices proj %d fc2 %d\n", M, s_proj, s_fc2);
    printf("%-40s %28s %28s %28s\n", "us per launch: per queue / overall", "1 queue", "2 queues", "3 queues");
    for (auto& op : ops) {
        const int n = op.launches == 7 ? reps / 4 : reps;
        hipGraphExec_t ge[3];
        for (int q = 0; q < 3; ++q) {
            op.f(q, 0);                                  // LDS opt-in attributes are set outside the capture
            CHECK(hipStreamSynchronize(st[q]));
            hipGraph_t g;
            CHECK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n; ++i) if (op.f(q, i + 4 * q)) { printf("launch failed\n"); return 1; }
            CHECK(hipStreamEndCapture(st[q], &g));
            CHECK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
        }
        printf("%-40s", op.name);
        for (int k = 1; k <= 3; ++k) {
            double best = 1e30;
            for (int round = 0; round < 4; ++round) {
                CHECK(hipDeviceSynchronize());
                timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                for (int rep = 0; rep < 3; ++rep)
                    for (int q = 0; q < k; ++q) CHECK(hipGraphLaunch(ge[q], st[q]));
                CHECK(hipDeviceSynchronize());
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
                if (round && us < best) best = us;
            }
            const double per = best / (3.0 * n * op.launches);
            printf("        %8.2f / %8.2f", per, per / k);
        }
        printf("\n");
        for (int q = 0; q < 3; ++q) CHECK(hipGraphExecDestroy(ge[q]));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'q') return queues_mode(VITVS_BF16);
    const int prec = (argc > 1 && atoi(argv[1]) == 32) ? VITVS_F32 : VITVS_BF16;
    const size_t es = prec == VITVS_F32 ? 4 : 2;
    const int M = 394, D = 768, H = 12, N = 197, hidden = 3072;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    void* xn = dalloc((size_t)M * D * es, 0);
    void* qkv = dalloc((size_t)M * 3 * D * es, 0);
    void* attn = dalloc((size_t)M * D * es, 0);
    void* hid = dalloc((size_t)M * hidden * es, 0);
    float* x = (float*)dalloc((size_t)M * D * 4, 0);
    float* part = (float*)dalloc((size_t)8 * M * D * 4, 0);
    // 12 weight sets visited round-robin, like the 12 blocks of the real forward (171 MB in bf16)
    void *wqkv_[12], *wproj_[12], *wfc1_[12], *wfc2_[12];
    for (int b = 0; b < 12; ++b) {
        wqkv_[b] = dalloc((size_t)3 * D * D * es, 0);
        wproj_[b] = dalloc((size_t)D * D * es, 0);
        wfc1_[b] = dalloc((size_t)hidden * D * es, 0);
        wfc2_[b] = dalloc((size_t)D * hidden * es, 0);
    }
    const int nsets = (argc > 2) ? atoi(argv[2]) : 12;   // 1: the same weights every launch (cache-resident)
    int turn = 0;
#define wqkv wqkv_[(turn++) % nsets]
#define wproj wproj_[(turn++) % nsets]
#define wfc1 wfc1_[(turn++) % nsets]
#define wfc2 wfc2_[(turn++) % nsets]
    float* bias = (float*)dalloc((size_t)hidden * 4, 0);
    float* gamma = (float*)dalloc((size_t)D * 4, 0);
    float* beta = (float*)dalloc((size_t)D * 4, 0);
    const int reps = 400;
    const int s_proj = vitvs_op_splitk_slices(prec, M, D, D), s_fc2 = vitvs_op_splitk_slices(prec, M, D, hidden);
    printf("precision %s, split-K slices: proj %d, fc2 %d, weight sets %d\n", prec == VITVS_F32 ? "fp32" : "bf16", s_proj, s_fc2,
           nsets);
    run("qkv   linear 394x2304x768", st, reps, [&] { return vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st); });
    run("attention 2 x 12 heads x 197", st, reps, [&] { return vitvs_op_attention(prec, qkv, attn, 2, N, H, st); });
    run("proj  partial 394x768x768", st, reps, [&] { return vitvs_op_linear_partial(prec, attn, wproj, part, M, D, D, s_proj, st); });
    run("residual_ln (proj slices) + LayerNorm", st, reps, [&] { return vitvs_op_residual_ln(prec, x, part, s_proj, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st); });
    run("fc1   linear+GELU 394x3072x768", st, reps, [&] { return vitvs_op_linear(prec, xn, wfc1, bias, hid, M, hidden, D, 1, st); });
    run("fc2   partial 394x768x3072", st, reps, [&] { return vitvs_op_linear_partial(prec, hid, wfc2, part, M, D, hidden, s_fc2, st); });
    run("residual_ln (fc2 slices) + LayerNorm", st, reps, [&] { return vitvs_op_residual_ln(prec, x, part, s_fc2, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st); });
    run("layernorm 394x768", st, reps, [&] { return vitvs_op_layernorm(prec, x, gamma, beta, xn, M, D, 1e-6f, st); });
    // kernel-switch cost: alternate two different kernels with no data dependence between them
    run("pair: qkv, layernorm (independent)", st, reps / 2, [&] {
        int rc = vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st);
        rc |= vitvs_op_layernorm(prec, x, gamma, beta, attn, M, D, 1e-6f, st);
        return rc;
    });
    run("pair: layernorm -> qkv (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_layernorm(prec, x, gamma, beta, xn, M, D, 1e-6f, st);
        rc |= vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st);
        return rc;
    });
    run("pair: qkv -> attention (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st);
        rc |= vitvs_op_attention(prec, qkv, attn, 2, N, H, st);
        return rc;
    });
    void* qkv2 = dalloc((size_t)M * 3 * D * es, 0);
    run("pair: qkv, attention (independent buffers)", st, reps / 2, [&] {
        int rc = vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st);
        rc |= vitvs_op_attention(prec, qkv2, attn, 2, N, H, st);
        return rc;
    });
    run("pair: attention -> proj partial (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_attention(prec, qkv, attn, 2, N, H, st);
        rc |= vitvs_op_linear_partial(prec, attn, wproj, part, M, D, D, s_proj, st);
        return rc;
    });
    run("pair: fc1 -> fc2 partial (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_linear(prec, xn, wfc1, bias, hid, M, hidden, D, 1, st);
        rc |= vitvs_op_linear_partial(prec, hid, wfc2, part, M, D, hidden, s_fc2, st);
        return rc;
    });
    run("pair: residual_ln -> fc1 (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_residual_ln(prec, x, part, s_fc2, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st);
        rc |= vitvs_op_linear(prec, xn, wfc1, bias, hid, M, hidden, D, 1, st);
        return rc;
    });
    run("pair: fc2 partial -> residual_ln (dependent)", st, reps / 2, [&] {
        int rc = vitvs_op_linear_partial(prec, hid, wfc2, part, M, D, hidden, s_fc2, st);
        rc |= vitvs_op_residual_ln(prec, x, part, s_fc2, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st);
        return rc;
    });
    // one whole block as the forward issues it (7 launches)
    run("block (7 launches)", st, reps / 4, [&] {
        int rc = vitvs_op_linear(prec, xn, wqkv, bias, qkv, M, 3 * D, D, 0, st);
        rc |= vitvs_op_attention(prec, qkv, attn, 2, N, H, st);
        rc |= vitvs_op_linear_partial(prec, attn, wproj, part, M, D, D, s_proj, st);
        rc |= vitvs_op_residual_ln(prec, x, part, s_proj, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st);
        rc |= vitvs_op_linear(prec, xn, wfc1, bias, hid, M, hidden, D, 1, st);
        rc |= vitvs_op_linear_partial(prec, hid, wfc2, part, M, D, hidden, s_fc2, st);
        rc |= vitvs_op_residual_ln(prec, x, part, s_fc2, bias, nullptr, gamma, beta, xn, M, D, 1e-6f, st);
        return rc;
    });
    return 0;
}
