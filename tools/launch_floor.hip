// Micro-benchmark of the per-launch cost of short dependent kernels on one stream (gfx950):
// what does a launch cost as a function of grid size, workgroup size, LDS allocation and memory traffic?
// Build:  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o tools/launch_floor tools/launch_floor.hip
// Run  :  HIP_FORCE_DEV_KERNARG=1 tools/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1.f; }

__global__ void k_lds(float* p) {
    extern __shared__ float s[];
    if (p == nullptr && threadIdx.x == 12345) { s[threadIdx.x] = 1.f; p[0] = s[0]; }
}

// one dependent load -> store per thread (float4), like a LayerNorm row pass
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 v = in[i]; v.x += 1.f; out[i] = v; }
}

// writes `n` float4 without reading (dirty lines at the end of the kernel)
__global__ void k_fill(float4* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// 8 bytes per lane, 4 lanes cover a 32-byte segment of a row's 128-byte line, 4 passes complete the line:
// the pattern of an MFMA-layout epilogue that stores bf16x4 per lane (rows = lane & 15 of a 16-row group)
__global__ void k_fill_mfma_pattern(unsigned long long* __restrict__ out, int rows) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int row = wave * 16 + (lane & 15);
    if (row >= rows) return;
    unsigned long long* line = out + (size_t)row * 16;   // 128 B = 16 x 8 B
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) line[pass * 4 + (lane >> 4)] = 0x1234567812345678ull + pass;
}
// the same bytes as full 16-byte-per-lane rows: 8 lanes cover one 128-byte line
__global__ void k_fill_rows(float4* __restrict__ out, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * 8) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// dependent chain before the store: a load whose result feeds the stored value (store issued late)
__global__ void k_load_then_fill_rows(const float4* __restrict__ in, float4* __restrict__ out, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * 8) { float4 v = in[i & 1023]; out[i] = v; }
}

// holds the queue busy for `us` microseconds (bounded: s_memrealtime ticks at 100 MHz) so that the host can enqueue the
// whole chain behind it: what follows is the DEVICE-side cost per dependent launch, free of host enqueue time
__global__ void k_blocker(float* p, unsigned us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
    if (p == nullptr) p[0] = 1.f;
}

struct Timer {
    std::vector<hipEvent_t> a, b;
    explicit Timer(int n) : a(n), b(n) { for (int i = 0; i < n; ++i) { CHECK(hipEventCreate(&a[i])); CHECK(hipEventCreate(&b[i])); } }
};

template <class F>
static void run(const char* name, int reps, hipStream_t st, F launch) {
    Timer t(reps);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(nullptr, nullptr);
    CHECK(hipStreamSynchronize(st));
    // pass 1: plain launches, chain time
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch(nullptr, nullptr);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float chain_ms = 0.f;
    CHECK(hipEventElapsedTime(&chain_ms, e0, e1));
    // pass 2: per-kernel begin/end stamps
    for (int i = 0; i < reps; ++i) launch(t.a[i], t.b[i]);
    CHECK(hipStreamSynchronize(st));
    double sum = 0.0, mn = 1e9;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, t.a[i], t.b[i]));
        sum += ms; if (ms < mn) mn = ms;
    }
    printf("%-44s chain %6.2f us/launch   kernel avg %6.2f us  min %6.2f us\n", name, chain_ms * 1e3 / reps, sum * 1e3 / reps, mn * 1e3);
}

// the same chain enqueued behind a blocker kernel (device-side floor) and replayed as a hipGraph
template <class F>
static void run_queued(const char* name, int reps, hipStream_t st, float* buf, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int round = 0; round < 3; ++round) {
        hipLaunchKernelGGL(k_blocker, dim3(1), dim3(64), 0, st, buf, 4000u);   // 4 ms: the host enqueues `reps` launches meanwhile
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch(nullptr, nullptr);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // graph replay of the same chain
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(nullptr, nullptr);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    float gbest = 1e9f;
    for (int round = 0; round < 3; ++round) {
        CHECK(hipEventRecord(e0, st));
        CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < gbest) gbest = ms;
    }
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    printf("%-44s queued behind a blocker %6.2f us/launch   graph replay %6.2f us/launch\n", name, best * 1e3 / reps, gbest * 1e3 / reps);
}

// every wave sleeps ~`us` microseconds without touching memory, LDS or the vector pipes: a stand-in for a launch whose
// body is latency (what a one-pair GEMM launch is), so that k chains can only be limited by dispatch, not by resources
__global__ void k_sleep(float* p, unsigned ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (p == nullptr) p[0] = 1.f;
}

// k chains of `reps` dependent launches, one chain per high-priority stream (a queue each), replayed as graphs at the same
// time: the AGGREGATE dispatch rate of the command processor (does the 1.76 us launch-to-launch floor of one queue overlap
// across queues?)
static void run_queues(const char* name, int reps, int k, dim3 grid, dim3 block, unsigned sleep_ticks, float* buf,
                       int copy_n4 = 0, int lds = 0) {
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (const char* e = getenv("LF_PRIO")) hi = atoi(e);      // LF_PRIO=0: default-class streams; -1: what torch calls high priority
    static bool said = false;
    if (!said) { printf("stream priority range: least %d, greatest %d; using %d\n", lo, hi, hi); said = true; }
    std::vector<hipStream_t> st(k);
    std::vector<hipGraphExec_t> ge(k);
    for (int q = 0; q < k; ++q) {
        CHECK(hipStreamCreateWithPriority(&st[q], hipStreamNonBlocking, hi));
        hipGraph_t g;
        CHECK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; ++i) {
            if (copy_n4) {                    // a load -> store row pass per launch, ping-pong between two buffers of this queue
                float4* a = reinterpret_cast<float4*>(buf) + (size_t)(2 * q) * copy_n4;
                float4* b = a + copy_n4;
                hipLaunchKernelGGL(k_copy, dim3((copy_n4 + 255) / 256), dim3(256), 0, st[q], (i & 1) ? b : a, (i & 1) ? a : b, copy_n4);
            } else if (lds) hipLaunchKernelGGL(k_lds, grid, block, lds, st[q], buf);
            else if (sleep_ticks) hipLaunchKernelGGL(k_sleep, grid, block, 0, st[q], buf, sleep_ticks);
            else hipLaunchKernelGGL(k_empty, grid, block, 0, st[q], buf);
        }
        CHECK(hipStreamEndCapture(st[q], &g));
        CHECK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge[q], st[q]));
    }
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int round = 0; round < 3; ++round) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int rep = 0; rep < 4; ++rep)
            for (int q = 0; q < k; ++q) CHECK(hipGraphLaunch(ge[q], st[q]));
        CHECK(hipDeviceSynchronize());
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
        if (us < best) best = us;
    }
    printf("%-36s %d queue(s): %6.2f us per launch per queue, %6.2f us per launch overall\n", name, k, best / (4.0 * reps),
           best / (4.0 * reps * k));
    for (int q = 0; q < k; ++q) { CHECK(hipGraphExecDestroy(ge[q])); CHECK(hipStreamDestroy(st[q])); }
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'q') {          // launch_floor queues
        float* buf; CHECK(hipMalloc(&buf, 4096));
        const int kmax = getenv("LF_KMAX") ? atoi(getenv("LF_KMAX")) : 4;
        if (getenv("LF_SHORT")) {                 // only the two informative rows, up to LF_KMAX queues
            for (int k = 1; k <= kmax; ++k) run_queues("empty 252 x 256", 400, k, dim3(252), dim3(256), 0, buf);
            for (int k = 1; k <= kmax; ++k) run_queues("sleep 3 us 252 x 256", 400, k, dim3(252), dim3(256), 300, buf);
            return 0;
        }
        for (int k = 1; k <= 4; ++k) run_queues("empty 256 x 64", 400, k, dim3(256), dim3(64), 0, buf);
        for (int k = 1; k <= 4; ++k) run_queues("empty 252 x 256", 400, k, dim3(252), dim3(256), 0, buf);
        for (int k = 1; k <= 4; ++k) run_queues("sleep 3 us 252 x 256", 400, k, dim3(252), dim3(256), 300, buf);
        for (int k = 1; k <= 4; ++k) run_queues("sleep 3 us 1 x 64", 400, k, dim3(1), dim3(64), 300, buf);
        for (int k = 1; k <= 4; ++k) run_queues("sleep 10 us 252 x 256", 200, k, dim3(252), dim3(256), 1000, buf);
        const int n4 = 394 * 768 / 4;
        float* big; CHECK(hipMalloc(&big, (size_t)8 * n4 * 16 * 4)); CHECK(hipMemset(big, 0, (size_t)8 * n4 * 16 * 4));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int k = 1; k <= 4; ++k) run_queues("copy 1.2 MB 296 x 256 (ping-pong)", 400, k, dim3(1), dim3(1), 0, big, n4);
        for (int k = 1; k <= 4; ++k) run_queues("copy 4.8 MB (ping-pong)", 400, k, dim3(1), dim3(1), 0, big, 4 * n4);
        for (int k = 1; k <= 4; ++k) run_queues("empty, 112 KB LDS 252 x 512", 400, k, dim3(252), dim3(512), 0, buf, 0, 112 * 1024);
        for (int k = 1; k <= 4; ++k) run_queues("empty, 64 KB LDS 252 x 256", 400, k, dim3(252), dim3(256), 0, buf, 0, 64 * 1024);
        return 0;
    }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int n4 = 394 * 768 / 4;   // one fp32 activation matrix of the headline workload
    float4 *bufa, *bufb, *big;
    CHECK(hipMalloc(&bufa, (size_t)n4 * 16)); CHECK(hipMalloc(&bufb, (size_t)n4 * 16));
    CHECK(hipMalloc(&big, (size_t)16 << 20));
    CHECK(hipMemset(bufa, 0, (size_t)n4 * 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int reps = 400;
#define L(kern, grid, block, lds, ...) [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(kern, grid, block, lds, st, a, b, 0, __VA_ARGS__); }
    run_queued("empty 256 x 64", reps, st, (float*)bufa, L(k_empty, dim3(256), dim3(64), 0, (float*)bufa));
    run_queued("empty 252 x 512", reps, st, (float*)bufa, L(k_empty, dim3(252), dim3(512), 0, (float*)bufa));
    run_queued("copy 1.2 MB 296 x 256", reps, st, (float*)bufa, L(k_copy, dim3((n4 + 255) / 256), dim3(256), 0, bufa, bufb, n4));
    run("empty 1 x 64", reps, st, L(k_empty, dim3(1), dim3(64), 0, (float*)bufa));
    run("empty 256 x 64", reps, st, L(k_empty, dim3(256), dim3(64), 0, (float*)bufa));
    run("empty 394 x 64", reps, st, L(k_empty, dim3(394), dim3(64), 0, (float*)bufa));
    run("empty 252 x 256", reps, st, L(k_empty, dim3(252), dim3(256), 0, (float*)bufa));
    run("empty 252 x 512", reps, st, L(k_empty, dim3(252), dim3(512), 0, (float*)bufa));
    run("empty 252 x 1024", reps, st, L(k_empty, dim3(252), dim3(1024), 0, (float*)bufa));
    run("empty 1024 x 256", reps, st, L(k_empty, dim3(1024), dim3(256), 0, (float*)bufa));
    run("lds 48 KB 252 x 512", reps, st, L(k_lds, dim3(252), dim3(512), 48 * 1024, (float*)bufa));
    run("lds 96 KB 252 x 512", reps, st, L(k_lds, dim3(252), dim3(512), 96 * 1024, (float*)bufa));
    run("lds 160 KB 252 x 512", reps, st, L(k_lds, dim3(252), dim3(512), 160 * 1024, (float*)bufa));
    run("copy 1.2 MB (394x768 f32) 296 x 256", reps, st, L(k_copy, dim3((n4 + 255) / 256), dim3(256), 0, bufa, bufb, n4));
    run("copy 1.2 MB 1182 x 64", reps, st, L(k_copy, dim3((n4 + 63) / 64), dim3(64), 0, bufa, bufb, n4));
    run("fill 1.2 MB 296 x 256", reps, st, L(k_fill, dim3((n4 + 255) / 256), dim3(256), 0, bufb, n4));
    run("fill 4.8 MB 1182 x 256", reps, st, L(k_fill, dim3((4 * n4 + 255) / 256), dim3(256), 0, big, 4 * n4));
    run("fill 16 MB 4096 x 256", reps, st, L(k_fill, dim3(4096), dim3(256), 0, big, 1 << 20));
    {
        const int rows = 394 * 18;   // 394 x 2304 bf16 = 18 lines of 128 B per token row: the qkv output (1.8 MB)
        run("store 1.8 MB, MFMA pattern (8 B/lane, 32-B runs)", reps, st, L(k_fill_mfma_pattern, dim3((rows / 16 * 64 + 255) / 256 + 1), dim3(256), 0, (unsigned long long*)big, rows));
        run("store 1.8 MB, full rows (16 B/lane)", reps, st, L(k_fill_rows, dim3((rows * 8 + 255) / 256), dim3(256), 0, big, rows));
        run("load then store 1.8 MB, full rows", reps, st, L(k_load_then_fill_rows, dim3((rows * 8 + 255) / 256), dim3(256), 0, bufa, big, rows));
    }
    // ping-pong: every launch reads what the previous one wrote (cross-XCD visibility on the critical path)
    {
        int flip = 0;
        auto pp = [&](hipEvent_t a, hipEvent_t b) {
            const float4* in = flip ? bufb : bufa; float4* out = flip ? bufa : bufb; flip ^= 1;
            hipExtLaunchKernelGGL(k_copy, dim3((n4 + 255) / 256), dim3(256), 0, st, a, b, 0, in, out, n4);
        };
        run("copy ping-pong 1.2 MB 296 x 256", reps, st, pp);
    }
    return 0;
}
