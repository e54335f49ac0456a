// Many-row operators on RANDOM data (zero-filled operands read 15-20 % high on this chip: the clock it holds depends on
// the data): each linear layer of the many-row configurations on every tile family, and the long-sequence attention.
// Build: make -C tools      Run: HIP_FORCE_DEV_KERNARG=1 tools/big_ops [fast]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <array>
#include <functional>
#include <vector>

#include "vitvs.h"
#include "vitvs_ops.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static void* rand_bf16(size_t n, float scale, unsigned seed) {
    std::vector<unsigned short> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;      // uniform [-1, 1)
        h[i] = bf16_of(u * scale);
    }
    void* p;
    CHECK(hipMalloc(&p, n * 2));
    CHECK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
    return p;
}

static double run(hipStream_t st, int reps, const std::function<int()>& op) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) if (op()) return -1.0;
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
    return best * 1e3 / reps;   // us
}

// "fixed" (probe build of the library: LD_LIBRARY_PATH=vit-vs_amd/variants/probe): what the fixed cost of a many-row launch is made
// of.  Per shape on the 256 x 256 tile: the plain-chain time per launch with and without the epilogue's global stores (the output
// burst), and in-kernel 100 MHz stamps of every workgroup (entry, first operands landed, end of the k-loop, exit) from single
// launches with cold and with warm operands.
static int fixed_cost_table(hipStream_t st) {
    typedef int (*set_probe_t)(int, void*);
    set_probe_t set_probe = (set_probe_t)dlsym(RTLD_DEFAULT, "vitvs_debug_set_big_probe");
    if (!set_probe) { printf("fixed needs a probe build of the library\n"); return 1; }
    struct Shape { const char* name; int M, N, K, variant; };
    const Shape shapes[] = {{"B/8 qkv 6274x2304 K=128", 6274, 2304, 128, 256}, {"B/8 qkv 6274x2304 K=768", 6274, 2304, 768, 256},
                            {"B/8 qkv 6274x2304 K=3072", 6274, 2304, 3072, 256}, {"8 pairs fc1 3152x3072 K=768 (256x192)", 3152, 3072, 768, 192},
                            {"ViT-L fc1 2740x4096 K=1024 (192x256)", 2740, 4096, 1024, 1256}};
    unsigned long long* stamps;
    CHECK(hipMalloc((void**)&stamps, 256 * 4 * 8));
    printf("%-40s | chain us/launch: stores, no stores | single launch, in-kernel (us after the first workgroup's entry; median over workgroups): "
           "entry, operands landed, k-loop done, exit | the same without stores\n", "shape (256-row tiles, bf16, random)");
    for (const Shape& s : shapes) {
        void* A = rand_bf16((size_t)s.M * s.K, 1.0f, 1);
        void* W[4];
        for (int i = 0; i < 4; ++i) W[i] = rand_bf16((size_t)s.N * s.K, 0.05f, 2 + i);
        void* out;
        CHECK(hipMalloc(&out, (size_t)2 * s.M * s.N + 256));
        float* bias;
        CHECK(hipMalloc((void**)&bias, s.N * 4));
        CHECK(hipMemset(bias, 0, s.N * 4));
        double chain[2];
        std::array<double, 4> med[2];
        for (int nostore = 0; nostore < 2; ++nostore) {
            if (set_probe(nostore, nullptr)) return 1;
            int turn = 0;
            chain[nostore] = run(st, 60, [&] { return vitvs_op_linear_variant(VITVS_BF16, s.variant, A, W[(turn++) % 4], bias, out, s.M, s.N, s.K, 0, 0, st); });
            if (set_probe(nostore, stamps)) return 1;
            std::vector<std::array<double, 4>> rows;
            std::vector<std::vector<double>> cols(4);
            for (int rep = 0; rep < 9; ++rep) {
                CHECK(hipMemset(stamps, 0, 256 * 4 * 8));
                if (vitvs_op_linear_variant(VITVS_BF16, s.variant, A, W[rep % 4], bias, out, s.M, s.N, s.K, 0, 0, st)) return 1;
                CHECK(hipStreamSynchronize(st));
                std::vector<unsigned long long> h(256 * 4);
                CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < 256; ++w) if (h[4 * w]) t0 = std::min(t0, h[4 * w]);
                if (rep < 2) continue;                             // the first launches warm the code object and the caches
                for (int w = 0; w < 256; ++w)
                    if (h[4 * w])
                        for (int c = 0; c < 4; ++c) cols[c].push_back((double)(h[4 * w + c] - t0) * 0.01);
            }
            for (int c = 0; c < 4; ++c) { std::sort(cols[c].begin(), cols[c].end()); med[nostore][c] = cols[c].empty() ? 0.0 : cols[c][cols[c].size() / 2]; }
        }
        printf("%-40s | %6.1f %6.1f | %5.2f %5.2f %6.2f %6.2f | %5.2f %5.2f %6.2f %6.2f\n", s.name, chain[0], chain[1], med[0][0], med[0][1], med[0][2],
               med[0][3], med[1][0], med[1][1], med[1][2], med[1][3]);
        fflush(stdout);
        CHECK(hipFree(A)); CHECK(hipFree(out)); CHECK(hipFree(bias));
        for (int i = 0; i < 4; ++i) CHECK(hipFree(W[i]));
    }
    set_probe(0, nullptr);
    return 0;
}

int main(int argc, char** argv) {
    const bool fast = argc > 1 && !strcmp(argv[1], "fast");
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    if (argc > 1 && !strcmp(argv[1], "fixed")) return fixed_cost_table(st);
    struct Shape { const char* name; int M, N, K, gelu, slices; };
    const Shape shapes[] = {
        {"ViT-B/8 448  qkv ", 6274, 2304, 768, 0, 0},  {"ViT-B/8 448  fc1 ", 6274, 3072, 768, 1, 0},
        {"ViT-B/8 448  fc2 ", 6274, 768, 3072, 0, 1},  {"ViT-B/8 448  proj", 6274, 768, 768, 0, 1},
        {"8 x ViT-B/16 qkv ", 3152, 2304, 768, 0, 0},  {"8 x ViT-B/16 fc1 ", 3152, 3072, 768, 1, 0},
        {"8 x ViT-B/16 fc2 ", 3152, 768, 3072, 0, 1},  {"8 x ViT-B/16 proj", 3152, 768, 768, 0, 1},
        {"ViT-L/14 518 qkv ", 2740, 3072, 1024, 0, 0}, {"ViT-L/14 518 fc1 ", 2740, 4096, 1024, 1, 0},
        {"ViT-L/14 518 fc2 ", 2740, 1024, 4096, 0, 1}, {"ViT-L/14 518 proj", 2740, 1024, 1024, 0, 1},
        {"4 x ViT-B/16 fc1 ", 1576, 3072, 768, 1, 0},  {"2 x ViT-B/16 fc1 ", 788, 3072, 768, 1, 0},
    };
    const bool sweep = argc > 1 && !strcmp(argv[1], "sweep");
    // K sweep at fixed M x N: separates the per-k-tile time from the fixed cost of a tile (prologue, epilogue, launch)
    const Shape sweep_shapes[] = {
        {"6274x2304 K=128  ", 6274, 2304, 128, 0, 0},  {"6274x2304 K=256  ", 6274, 2304, 256, 0, 0},
        {"6274x2304 K=512  ", 6274, 2304, 512, 0, 0},  {"6274x2304 K=768  ", 6274, 2304, 768, 0, 0},
        {"6274x2304 K=1536 ", 6274, 2304, 1536, 0, 0}, {"6274x2304 K=3072 ", 6274, 2304, 3072, 0, 0},
        {"1500x2304 K=128  ", 1500, 2304, 128, 0, 0},  {"1500x2304 K=768  ", 1500, 2304, 768, 0, 0},
        {"1500x2304 K=3072 ", 1500, 2304, 3072, 0, 0},
    };
    // "mid": 2..6 frame pairs (788..2364 rows) through the four layer shapes of ViT-B/16: where does each tile family win?
    // "midl": the same for ViT-L/14 at 224² (257 tokens per frame)
    const bool midl = argc > 1 && !strcmp(argv[1], "midl");
    const bool mid_mode = midl || (argc > 1 && !strcmp(argv[1], "mid"));
    std::vector<Shape> mid_shapes;
    static char mid_names[64][24];
    if (mid_mode) {
        int q = 0;
        for (int rows : midl ? std::vector<int>{514, 1028, 1542, 2056} : std::vector<int>{788, 985, 1182, 1576, 1970, 2364})
            for (int layer = 0; layer < 4; ++layer) {
                const int D = midl ? 1024 : 768;
                const int N = layer == 0 ? 3 * D : layer == 1 ? 4 * D : D, K = layer == 2 ? 4 * D : D;
                snprintf(mid_names[q], sizeof(mid_names[q]), "%4d rows %s", rows, layer == 0 ? "qkv " : layer == 1 ? "fc1 " : layer == 2 ? "fc2 " : "proj");
                mid_shapes.push_back(Shape{mid_names[q], rows, N, K, layer == 1, layer >= 2 ? -1 : 0});
                ++q;
            }
    }
    // "slices": the narrow layers at 3152 / 2740 rows with 1..4 K slices (the forward adds ~1.8 us of residual_ln per extra slice)
    const bool slices_mode = argc > 1 && !strcmp(argv[1], "slices");
    const Shape slice_shapes[] = {
        {"3152 proj 1 slice ", 3152, 768, 768, 0, 1},  {"3152 proj 2 slices", 3152, 768, 768, 0, 2},  {"3152 proj 3 slices", 3152, 768, 768, 0, 3},
        {"3152 fc2  2 slices", 3152, 768, 3072, 0, 2}, {"3152 fc2  3 slices", 3152, 768, 3072, 0, 3}, {"3152 fc2  4 slices", 3152, 768, 3072, 0, 4},
        {"2740 proj 1 slice ", 2740, 1024, 1024, 0, 1}, {"2740 proj 2 slices", 2740, 1024, 1024, 0, 2}, {"2740 proj 4 slices", 2740, 1024, 1024, 0, 4},
        {"2740 fc2  2 slices", 2740, 1024, 4096, 0, 2}, {"2740 fc2  4 slices", 2740, 1024, 4096, 0, 4},
        {"1576 proj 1 slice ", 1576, 768, 768, 0, 1},  {"1576 proj 2 slices", 1576, 768, 768, 0, 2},  {"1576 proj 3 slices", 1576, 768, 768, 0, 3},
        {"1576 fc2  3 slices", 1576, 768, 3072, 0, 3}, {"1576 fc2  4 slices", 1576, 768, 3072, 0, 4}, {"1576 fc2  6 slices", 1576, 768, 3072, 0, 6},
    };
    // weight sets a layer rotates through.  4 (the default, as in the profiles of rounds 2-3) keeps them in the 256 MB
    // infinity cache; VITVS_WEIGHT_MB=512 rotates through that many MB per layer shape, so that every launch fetches its
    // weights from HBM as the forward does (12 layers x 14 MB + activations per update do not stay cached)
    const long weight_mb = getenv("VITVS_WEIGHT_MB") ? atol(getenv("VITVS_WEIGHT_MB")) : 0;
    const int max_sets = 512;
    printf("%-20s %6s %6s %6s | %-28s\n", "layer (bf16, random)", "M", "N", "K", "us / TFLOP/s per tile family: auto, 64-row tiles, 256x256, 256x128, 128x128, 256x192, 192x128, 192x256");
    // "occ": attention only, on a probe build of the library, with the workgroups per CU limited through their LDS size
    const bool occ = argc > 1 && !strcmp(argv[1], "occ");
    const bool attn_only = occ || (argc > 1 && (!strcmp(argv[1], "attn") || !strcmp(argv[1], "attnmid")));
    std::vector<Shape> todo;
    if (attn_only) todo.clear();
    else if (sweep) todo.assign(std::begin(sweep_shapes), std::end(sweep_shapes));
    else if (slices_mode) todo.assign(std::begin(slice_shapes), std::end(slice_shapes));
    else if (mid_mode) todo = mid_shapes;
    else todo.assign(std::begin(shapes), std::end(shapes));
    for (const Shape& s : todo) {
        void* A = rand_bf16((size_t)s.M * s.K, 1.0f, 1);
        const int nsets = weight_mb > 0 ? (int)std::min<long>(max_sets, std::max<long>(4, weight_mb * 1000000 / ((long)s.N * s.K * 2))) : 4;
        void* Wt[max_sets];
        for (int i = 0; i < nsets; ++i) Wt[i] = rand_bf16((size_t)s.N * s.K, 0.05f, 2 + i);
        void* out;
        const int sl = s.slices < 0 ? vitvs_op_splitk_slices(VITVS_BF16, s.M, s.N, s.K) : s.slices;   // -1: the library's choice
        CHECK(hipMalloc(&out, (size_t)(sl ? sl * 4 : 2) * s.M * s.N + 256));
        float* bias;
        CHECK(hipMalloc((void**)&bias, s.N * 4));
        CHECK(hipMemset(bias, 0, s.N * 4));
        const double flop = 2.0 * s.M * s.N * s.K;
        printf("%-20s %6d %6d %6d |", s.name, s.M, s.N, s.K);
        if (s.slices < 0) printf(" %d slice(s) |", sl);
        for (int variant : {0, 1, 256, 128, 2, 192, 1192, 1256}) {
            int turn = 0;
            const double us = run(st, fast ? 20 : 60, [&] {
                return vitvs_op_linear_variant(VITVS_BF16, variant, A, Wt[(turn++) % nsets], bias, out, s.M, s.N, s.K, s.gelu, sl, st);
            });
            if (us < 0) printf("     n/a      ");
            else printf(" %7.1f %6.0f", us, flop / us * 1e-6);
        }
        printf("\n");
        fflush(stdout);
        CHECK(hipFree(A)); CHECK(hipFree(out)); CHECK(hipFree(bias));
        for (int i = 0; i < nsets; ++i) CHECK(hipFree(Wt[i]));
    }
    if (sweep || slices_mode || mid_mode) return 0;
    // long-sequence attention: (images, tokens, heads)
    // "attn": the long sequences of the 448 / 518 inputs and batches of short ones; "attnmid": 3..16 images of 197 / 485 tokens
    const bool attn_mid = argc > 1 && !strcmp(argv[1], "attnmid");
    std::vector<std::array<int, 3>> att = {{2, 3137, 12}, {2, 1370, 16}, {16, 197, 12}, {2, 785, 12}, {8, 785, 12}, {4, 3137, 12},
                                           {2, 577, 12}, {8, 197, 12}, {4, 197, 12}, {16, 257, 12}, {16, 485, 6}};
    if (attn_mid) {
        att.clear();
        for (int n : {3, 4, 5, 6, 7, 8, 10, 12, 16}) att.push_back({n, 197, 12});
        for (int n : {2, 3, 4, 5, 6, 8, 10, 12}) att.push_back({n, 485, 6});
        for (int n : {2, 4, 6, 8}) att.push_back({n, 257, 16});
    }
    typedef int (*set_lds_t)(int);
    set_lds_t set_lds = (set_lds_t)dlsym(RTLD_DEFAULT, "vitvs_debug_set_attn_lds");
    if (occ && !set_lds) { printf("occ needs a probe build of the library\n"); return 1; }
    for (int lds_kb : {48, 64, 100}) {
    if (!occ && lds_kb != 48) break;
    if (occ) { set_lds(lds_kb * 1024); printf("== %d KB of LDS per workgroup: %d workgroup(s) of 4 waves per CU\n", lds_kb, 160 / lds_kb); }
    for (auto& a : att) {
        const int n_img = a[0], N = a[1], H = a[2], D = H * 64;
        if (occ && N < 1024) continue;
        void* qkv = rand_bf16((size_t)n_img * N * 3 * D, 1.0f, 9);
        void* out;
        CHECK(hipMalloc(&out, (size_t)n_img * N * D * 2));
        const double us = run(st, fast ? 10 : 30, [&] { return vitvs_op_attention(VITVS_BF16, qkv, out, n_img, N, H, st); });
        printf("attention %d x %d tokens x %d heads: %8.1f us  %6.0f TFLOP/s\n", n_img, N, H, us, 4.0 * n_img * N * (double)N * D / us * 1e-6);
        // probe builds of the library (-DVITVS_PROBE) export a hook: per-wave cycle sums of the long kernel's tile loop
        typedef int (*set_probe_t)(void*);
        set_probe_t set_probe = (set_probe_t)dlsym(RTLD_DEFAULT, "vitvs_debug_set_attn_probe");
        if (set_probe && N >= 512) {
            const int wgs = 4 * 8 * ((((N + 127) / 128) * H * n_img + 7) / 8) + 1024;      // every plan's grid fits
            const int REC = 14;
            unsigned long long* buf;
            CHECK(hipMalloc((void**)&buf, (size_t)wgs * 4 * REC * 8));
            CHECK(hipMemset(buf, 0, (size_t)wgs * 4 * REC * 8));
            for (int i = 0; i < 3; ++i) vitvs_op_attention(VITVS_BF16, qkv, out, n_img, N, H, st);
            CHECK(hipStreamSynchronize(st));
            set_probe(buf);
            vitvs_op_attention(VITVS_BF16, qkv, out, n_img, N, H, st);
            CHECK(hipStreamSynchronize(st));
            set_probe(nullptr);
            std::vector<unsigned long long> hbuf((size_t)wgs * 4 * REC);
            CHECK(hipMemcpy(hbuf.data(), buf, hbuf.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> col[REC];
            unsigned long long t_first = ~0ull;
            for (int w = 0; w < wgs * 4; ++w) if (hbuf[(size_t)w * REC + 7]) t_first = std::min(t_first, hbuf[(size_t)w * REC + 8]);
            for (int w = 0; w < wgs * 4; ++w) if (hbuf[(size_t)w * REC + 7]) {
                for (int k = 0; k < REC; ++k) col[k].push_back((double)hbuf[(size_t)w * REC + k]);
                col[8].back() -= (double)t_first; col[9].back() -= (double)t_first;
            }
            auto pct = [](std::vector<double> v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
            auto med = [&](const std::vector<double>& v) { return pct(v, 0.5); };
            auto mean = [](const std::vector<double>& v) { double a = 0; for (double x : v) a += x; return v.empty() ? 0.0 : a / v.size(); };
            const double nt = mean(col[7]);
            printf("  probe (mean over %zu waves, cycles per tile): wait-dma %.0f  barrier %.0f  copy-issue+K-reads+scores %.0f  softmax %.0f  V-wait+PV %.0f | tiles/wave %.1f (slow path %.2f)  segments %.2f  exchange+finish %.0f cycles/wave | lifetime median %.0f cycles in %.1f us -> %.2f GHz\n",
                   col[0].size(), mean(col[0]) / nt, mean(col[1]) / nt, mean(col[2]) / nt, mean(col[3]) / nt, mean(col[4]) / nt, nt, mean(col[12]), mean(col[11]), mean(col[10]), med(col[5]), med(col[6]) / 100.0,
                   med(col[5]) / (med(col[6]) * 10.0));
            printf("  timeline (us after the first wave's start): starts p0 %.1f p50 %.1f p90 %.1f p100 %.1f | ends p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n",
                   pct(col[8], 0) / 100, pct(col[8], .5) / 100, pct(col[8], .9) / 100, pct(col[8], 1) / 100, pct(col[9], 0) / 100, pct(col[9], .1) / 100,
                   pct(col[9], .5) / 100, pct(col[9], .9) / 100, pct(col[9], 1) / 100);
            {   // per XCD (workgroup id % 8): median end time, median lifetime cycles, median clock; and the spread of the waves' cycle counts
                printf("  per XCD: ");
                for (int x = 0; x < 8; ++x) {
                    std::vector<double> e, c, f;
                    for (int w = 0; w < wgs * 4; ++w) if (hbuf[(size_t)w * REC + 7] && ((w / 4) & 7) == x) {
                        e.push_back((double)(hbuf[(size_t)w * REC + 9] - t_first) / 100.0);
                        c.push_back((double)hbuf[(size_t)w * REC + 5]);
                        f.push_back((double)hbuf[(size_t)w * REC + 5] / ((double)hbuf[(size_t)w * REC + 6] * 10.0));
                    }
                    printf("[%d] end %.0f..%.0f us, %.0fk cyc, %.2f GHz  ", x, pct(e, 0), pct(e, 1), med(c) / 1e3, med(f));
                }
                printf("\n  lifetime kcycles by (workgroup id >> 8): ");
                for (int cl = 0; cl < 6; ++cl) {
                    std::vector<double> c;
                    for (int w = 0; w < wgs * 4; ++w) if (hbuf[(size_t)w * REC + 7] && ((w / 4) >> 8) == cl) c.push_back((double)hbuf[(size_t)w * REC + 5] / 1e3);
                    if (!c.empty()) printf("[%d] %.0f..%.0f..%.0f  ", cl, pct(c, 0), med(c), pct(c, 1));
                }
                printf("| by ((workgroup id >> 3) %% 3): ");
                for (int cl = 0; cl < 3; ++cl) {
                    std::vector<double> c;
                    for (int w = 0; w < wgs * 4; ++w) if (hbuf[(size_t)w * REC + 7] && (((w / 4) >> 3) % 3) == cl) c.push_back((double)hbuf[(size_t)w * REC + 5] / 1e3);
                    if (!c.empty()) printf("[%d] %.0f..%.0f..%.0f  ", cl, pct(c, 0), med(c), pct(c, 1));
                }
                printf("\n  lifetime cycles p0 %.0fk p10 %.0fk p50 %.0fk p90 %.0fk p100 %.0fk\n", pct(col[5], 0) / 1e3, pct(col[5], .1) / 1e3, pct(col[5], .5) / 1e3, pct(col[5], .9) / 1e3, pct(col[5], 1) / 1e3);
            }
            CHECK(hipFree(buf));
        }
        CHECK(hipFree(qkv)); CHECK(hipFree(out));
    }
    }
    return 0;
}
