// From nearest-neighbour tables to the camera twist: mutual-NN filter, feature selection,
// pixel features, depth lookup, interaction matrix L_e and v_c = -lambda * pinv(L_e) e.
// One workgroup per frame pair; the SVD-based pseudo-inverse runs in fp64 on one wavefront.
//
// Reference arithmetic being replaced (vitvs_v2.py):
//   same-image shortcut  mean(sim_1) > 0.99                   :84-101
//   cyclic filter        nn_2[nn_1[i]], distance, normalise, >= 1 mask   :105-131
//   patch centres        p*scale + scale/2 (fp32)             :511-513
//   calculate_uv         flip, scale, round-half-even, zero padding, <4 quirk   :525-553
//   transform_to_real_world  (u-cx)/fx, (v-cy)/fy             :634-648
//   get_depth            mm -> m, 0 / out-of-bounds -> 100    :566-586
//   interaction matrix   :650-659 ;  e, pinv (rcond 1e-15), -lambda   :613-622
//
// Cyclic filter, restated: with dist_i = -|rc(nn_2[nn_1[i]]) - rc(i) + 1e-6|, the reference keeps
// tokens whose min/max-normalised distance is >= 1.  The global maximum of S is both a row and a
// column maximum, so at least one token is a mutual NN (dist = -1.41e-6, the largest possible);
// every non-mutual token has dist <= -0.999999, hence the range is >= 0.99, the +1e-8 in the
// normaliser is below half an fp32 ulp and exactly the mutual NNs normalise to 1.0.  If EVERY token
// is mutual the range is 0, every normalised value is 0 and the reference returns None.
// So: candidates = { i : nn_2[nn_1[i]] == i } when their count is < T, nothing when it equals T.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace vitvs {

constexpr int kLdsRows = 128;  // L rows kept in LDS; larger systems use the global workspace

// LDS words (4 bytes) in front of the fp64 area: nn1, nn2, flag, simL [T each], zraw [T, only when the depth
// prefetch is on: T <= 256], sel [max_rows], 16 ints, 4 floats
__host__ __device__ constexpr size_t servo_words(int T, int max_rows) {
    return (size_t)4 * T + (T <= 256 ? T : 0) + max_rows + 16 + 4;
}
__host__ __device__ constexpr size_t servo_f64_offset(int T, int max_rows) {
    return (servo_words(T, max_rows) * 4 + 15) & ~(size_t)15;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load AND store (vmcnt(0)); this kernel is one serial chain of short phases, and its detail
// stores and prefetched loads must stay in flight across the phase boundaries.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pixel of a token's patch centre in camera resolution: vitvs_v2.py:511-513 (fp32 centre) and :544-549
// (scale in fp64, round half to even).
__device__ __forceinline__ void token_pixel(const ServoArgs& a, int tok, long& u, long& v) {
    const int g = a.grid;
    const float r = __fadd_rn(__fmul_rn((float)(tok / g), a.scale_f), a.half_f);
    const float c = __fadd_rn(__fmul_rn((float)(tok % g), a.scale_f), a.half_f);
    u = (long)rint((double)c * a.scale_x);
    v = (long)rint((double)r * a.scale_y);
}

// The leading flat arguments repeat the fields of `a` the first memory requests depend on: they are
// preloaded into SGPRs by the command processor (kernarg preload), the struct is fetched by the wave.
__global__ __launch_bounds__(256) void servo_kernel(const unsigned long long* __restrict__ row_best,
                                                    const unsigned long long* __restrict__ col_best,
                                                    const double* __restrict__ Kin, const int32_t* __restrict__ selection,
                                                    const uint16_t* __restrict__ depth_all, int T, int mode, int sel_stride,
                                                    ServoArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* nn1 = reinterpret_cast<int*>(smem);
    int* nn2 = nn1 + T;
    int* flag = nn2 + T;            // candidate flags
    int* zraw = flag + T;           // [T <= 256 only] prefetched depth (mm) of each token's match; 0 = hole / outside
    float* simL = reinterpret_cast<float*>(zraw + (T <= 256 ? T : 0));   // sim_1, kept on chip for the feature stage
    int* sel = reinterpret_cast<int*>(simL + T);        // [max_rows]
    int* iscr = sel + a.max_rows;                       // [16]
    float* fscr = reinterpret_cast<float*>(iscr + 16);  // [4]
    double* Llds = reinterpret_cast<double*>(smem + servo_f64_offset(T, a.max_rows));
    double* Gs = Llds + 7 * kLdsRows;   // [0..27) G (21, upper triangle row-major) + g (6); [40 ..) 8 x 27 slice sums

    const unsigned long long* rb = row_best + (size_t)b * T;
    const unsigned long long* cb = col_best + (size_t)b * T;

    // 0. everything whose address is known up front is requested now and consumed phases later:
    // camera intrinsics, the first entry of this thread's slice of the visiting order / explicit ids
    const double fx = Kin[b * 4 + 0], fy = Kin[b * 4 + 1], cx = Kin[b * 4 + 2], cy = Kin[b * 4 + 3];
    const int32_t* order = (mode == SEL_PRIORITY) ? selection + (size_t)b * sel_stride : nullptr;
    const int per = (T + 255) / 256;
    const int p0 = tid * per, p1 = min(p0 + per, T);
    int first_id = p0;
    if (order && p0 < T) first_id = order[p0];
    int n_explicit = 0, explicit_id = 0;
    if (mode == SEL_EXPLICIT) {
        n_explicit = min(a.n_selected[b], a.num_pairs);
        if (tid < n_explicit) explicit_id = selection[(size_t)b * sel_stride + tid];
    }

    // 1. decode the packed (similarity, index) keys.  With one token per thread the depth of the
    // token's match is requested here as well (its address depends on nn_1 only), two phases early.
    const bool prefetch_depth = depth_all != nullptr && T <= 256;
    int dpre = 0;
    float ssum = 0.f;
    unsigned long long kr0 = 0, kc0 = 0;
    if (tid < T) { kr0 = rb[tid]; kc0 = cb[tid]; }
    // first use of a field of `a` (fetched by the wave itself): after the requests above are in flight
    const uint16_t* depth = depth_all ? depth_all + (size_t)b * a.depth_h * a.depth_w : nullptr;
    for (int i = tid; i < T; i += 256) {
        const unsigned long long kr = (i == tid) ? kr0 : rb[i], kc = (i == tid) ? kc0 : cb[i];
        const int n1 = (int)best_index(kr), n2 = (int)best_index(kc);
        const float s1 = best_value(kr);
        if (prefetch_depth) {
            long u, v;
            token_pixel(a, min(max(n1, 0), T - 1), u, v);
            if (u >= 0 && u < a.depth_w && v >= 0 && v < a.depth_h) dpre = depth[(size_t)v * a.depth_w + u];
        }
        nn1[i] = n1;
        nn2[i] = n2;
        simL[i] = s1;
        a.nn1[(size_t)b * T + i] = n1;
        a.nn2[(size_t)b * T + i] = n2;
        a.sim1[(size_t)b * T + i] = s1;
        ssum += s1;
    }
    ssum = wave_sum(ssum);
    if (lane == 0) fscr[wave] = ssum;
    lds_barrier();
    const float mean_sim = (fscr[0] + fscr[1] + fscr[2] + fscr[3]) / (float)T;
    const bool same_image = mean_sim > 0.99f;

    // 2. mutual nearest neighbours
    int cnt = 0;
    for (int i = tid; i < T; i += 256) {
        const int n1 = nn1[i];
        const int m = (n1 >= 0 && n1 < T && nn2[n1] == i) ? 1 : 0;
        flag[i] = same_image ? 1 : m;
        cnt += m;
    }
    cnt = wave_sum(cnt);
    if (lane == 0) iscr[wave] = cnt;
    lds_barrier();
    const int n_mutual = iscr[0] + iscr[1] + iscr[2] + iscr[3];
    const bool none = !same_image && (n_mutual == T || n_mutual == 0);
    const int n_cand = same_image ? T : n_mutual;

    // 3. selection -> sel[0 .. n_matched)
    int n_matched = 0;
    const int want = (a.mode == SEL_DENSE) ? min(n_cand, a.max_rows) : a.num_pairs;
    if (a.mode == SEL_EXPLICIT) {
        n_matched = n_explicit;
        for (int k = tid; k < n_matched; k += 256) {
            const int i = (k == tid) ? explicit_id : selection[(size_t)b * sel_stride + k];
            sel[k] = min(max(i, 0), T - 1);
        }
    } else if (!none) {
        // visit tokens in the given order (identity for DENSE); keep the first `want` candidates:
        // per-thread count over a contiguous slice of the order, wave scan, wave totals through LDS
        int local = 0;
        for (int p = p0; p < p1; ++p) {
            int i = (p == p0) ? first_id : (order ? order[p] : p);
            i = min(max(i, 0), T - 1);
            local += flag[i];
        }
        int incl = local;
        if (per == 1) {   // one position per thread: the wave scan is a ballot and a bit count (no LDS-crossbar shuffles)
            const unsigned long long mask = __ballot(local != 0);
            incl = local + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        } else {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, WAVE);
                if (lane >= o) incl += up;
            }
        }
        if (lane == 63) iscr[4 + wave] = incl;
        lds_barrier();
        int rank = incl - local;
        for (int w = 0; w < wave; ++w) rank += iscr[4 + w];
        for (int p = p0; p < p1; ++p) {
            int i = (p == p0) ? first_id : (order ? order[p] : p);
            i = min(max(i, 0), T - 1);
            if (flag[i]) {
                if (rank < want) sel[rank] = i;
                ++rank;
            }
        }
        n_matched = min(n_cand, want);
    }
    if (prefetch_depth && tid < T) zraw[tid] = dpre;
    lds_barrier();

    // 4. pixel features, depth, interaction matrix
    const int n_rows = (a.mode == SEL_DENSE) ? n_matched : a.num_pairs;   // feature pairs entering L
    const bool too_few = (a.mode != SEL_DENSE) && (n_matched != a.num_pairs) && (n_matched < 4);
    const int R = 2 * n_rows;
    const bool use_lds = R <= kLdsRows;
    const int rcap = use_lds ? kLdsRows : 2 * a.max_rows;
    const int gcap = 2 * a.max_rows;
    double* Lg = a.L_ws + (size_t)b * 7 * gcap;            // global copy of L and e (output for the parity tests)
    double* Lc = use_lds ? Llds : Lg;                      // working copy, 6 columns + e, column-major
    int32_t* uv_out = a.s_uv + (size_t)b * a.max_rows * 4;
    double* feat_out = a.feat + (size_t)b * a.max_rows * 4;
    int32_t* sel_out = a.sel_out + (size_t)b * a.max_rows;
    for (int k = tid; k < n_rows; k += 256) {
        long us = 0, vs = 0, u = 0, v = 0;
        int tok = -1;
        double simk = 0.0;
        const bool live = !none && !too_few && k < n_matched;
        if (live) {
            tok = sel[k];
            token_pixel(a, tok, us, vs);
            token_pixel(a, same_image ? tok : nn1[tok], u, v);
            simk = same_image ? 1.0 : (double)simL[tok];
        }
        const double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
        const double xs = ((double)us - cx) / fx, ys = ((double)vs - cy) / fy;
        unsigned d = 0;
        if (live && prefetch_depth && !same_image) d = (unsigned)zraw[tok];
        else if (depth && u >= 0 && u < a.depth_w && v >= 0 && v < a.depth_h) d = depth[(size_t)v * a.depth_w + u];
        const double z = d != 0 ? (double)d / 1000.0 : 100.0;
        const int r0 = 2 * k, r1i = 2 * k + 1;
        const double l0[7] = {-1.0 / z, 0.0, x / z, x * y, -(1.0 + x * x), y, x - xs};
        const double l1[7] = {0.0, -1.0 / z, y / z, 1.0 + y * y, -(x * y), -x, y - ys};
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            Lc[c * rcap + r0] = l0[c];
            Lc[c * rcap + r1i] = l1[c];
            if (use_lds) {
                Lg[(size_t)c * gcap + r0] = l0[c];
                Lg[(size_t)c * gcap + r1i] = l1[c];
            }
        }
        uv_out[k * 4 + 0] = (int32_t)us; uv_out[k * 4 + 1] = (int32_t)vs;
        uv_out[k * 4 + 2] = (int32_t)u;  uv_out[k * 4 + 3] = (int32_t)v;
        feat_out[k * 4 + 0] = z; feat_out[k * 4 + 1] = x; feat_out[k * 4 + 2] = y; feat_out[k * 4 + 3] = simk;
        sel_out[k] = tok;
    }
    if (use_lds) lds_barrier();
    else __syncthreads();   // L lives in global memory: full fence

    // 5. v_c = -lambda * pinv(L) e, fp64.
    // Fast path (L_e of full column rank and well conditioned, the normal servo case): pinv(L) e is the
    // least-squares solution, obtained from the 6x6 normal equations G = L^T L, g = L^T e by an LDL^T
    // factorisation.  If a pivot falls below 1e-8 of its diagonal (cond(L) > ~1e4, or rank deficiency,
    // e.g. all-identical zero-padded rows) the general path below runs instead: one-sided Jacobi SVD
    // with numpy.linalg.pinv's rcond = 1e-15 cut-off.
    int status = ST_OK;
    if (!depth) status = ST_NO_DEPTH;
    else if (none) status = ST_NO_CORRESPONDENCE;
    else if (too_few) status = ST_TOO_FEW;
    const bool try_fast = status == ST_OK && R > 0;   // L in LDS or (dense selections) in the global workspace
    if (try_fast) {
        // 27 quantities x 8 row slices on 216 threads (fixed slice order -> deterministic)
        const int qid = tid & 31, slice = tid >> 5;
        if (qid < 27) {
            int ca, cb;
            if (qid < 21) {
                int q = qid;
                ca = 0;
                while (q >= 6 - ca) { q -= 6 - ca; ++ca; }
                cb = ca + q;
            } else {
                ca = qid - 21;
                cb = 6;
            }
            // 4 independent chains keep 8 loads in flight (dense selections read L from the global workspace);
            // fixed combination order -> still deterministic
            double acc4[4] = {0.0, 0.0, 0.0, 0.0};
            int r = slice;
            for (; r + 24 < R; r += 32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc4[u] += Lc[ca * rcap + r + 8 * u] * Lc[cb * rcap + r + 8 * u];
            }
            for (; r < R; r += 8) acc4[0] += Lc[ca * rcap + r] * Lc[cb * rcap + r];
            Gs[40 + slice * 27 + qid] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        }
        lds_barrier();
    }
    if (wave != 0) return;
    double vout[6] = {0, 0, 0, 0, 0, 0};
    int sweeps = 0;
    bool solved = false;
    if (try_fast) {
        // wave 0 only from here: 27 lanes add the 8 slices, every lane then factors the same 6x6 system in
        // registers (fully unrolled: no private-memory arrays, no cross-lane traffic)
        if (lane < 27) {
            double acc = 0.0;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) acc += Gs[40 + sl * 27 + lane];
            Gs[lane] = acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: LDS writes above are visible below
        double Gm[6][6], rhs[6];
        {
            int q = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) { Gm[j][i] = Gs[q]; ++q; }   // lower triangle
#pragma unroll
            for (int i = 0; i < 6; ++i) rhs[i] = Gs[21 + i];
        }
        // G = L D L^T (unit lower-triangular L, no square roots, one reciprocal per pivot)
        bool good = true;
        double Lf[6][6], dinv[6], dpiv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double d = Gm[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) d -= Lf[j][k] * Lf[j][k] * dpiv[k];
            good = good && (d > 1e-8 * Gm[j][j]) && (Gm[j][j] > 0.0);
            dpiv[j] = d;
            dinv[j] = 1.0 / d;
#pragma unroll
            for (int i = j + 1; i < 6; ++i) {
                double t = Gm[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) t -= Lf[i][k] * Lf[j][k] * dpiv[k];
                Lf[i][j] = t * dinv[j];
            }
        }
        if (good) {
            double y[6], xsol[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {          // L y = rhs
                double t = rhs[i];
#pragma unroll
                for (int k = 0; k < i; ++k) t -= Lf[i][k] * y[k];
                y[i] = t;
            }
#pragma unroll
            for (int i = 5; i >= 0; --i) {         // L^T x = D^-1 y
                double t = y[i] * dinv[i];
#pragma unroll
                for (int k = i + 1; k < 6; ++k) t -= Lf[k][i] * xsol[k];
                xsol[i] = t;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) vout[i] = -a.lambda * xsol[i];
            solved = true;
            sweeps = -1;
        }
    }
    if (status == ST_OK && R > 0 && !solved) {
        double V[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
        const double tol = 4e-15;
        for (sweeps = 0; sweeps < 40; ++sweeps) {
            int rotated = 0;
#pragma unroll
            for (int p = 0; p < 5; ++p)
#pragma unroll
                for (int q = p + 1; q < 6; ++q) {
                    double al = 0.0, be = 0.0, ga = 0.0;
                    for (int r = lane; r < R; r += 64) {
                        const double ap = Lc[p * rcap + r], aq = Lc[q * rcap + r];
                        al += ap * ap; be += aq * aq; ga += ap * aq;
                    }
                    al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                    if (fabs(ga) > tol * sqrt(al * be) && al > 0.0 && be > 0.0) {
                        ++rotated;
                        const double zeta = (be - al) / (2.0 * ga);
                        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                        for (int r = lane; r < R; r += 64) {
                            const double ap = Lc[p * rcap + r], aq = Lc[q * rcap + r];
                            Lc[p * rcap + r] = c * ap - s * aq;
                            Lc[q * rcap + r] = s * ap + c * aq;
                        }
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const double vp = V[i][p], vq = V[i][q];
                            V[i][p] = c * vp - s * vq;
                            V[i][q] = s * vp + c * vq;
                        }
                    }
                }
            if (rotated == 0) break;
        }
        double sig2[6], w[6], smax2 = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s2 = 0.0, dot = 0.0;
            for (int r = lane; r < R; r += 64) {
                const double aj = Lc[j * rcap + r];
                s2 += aj * aj;
                dot += aj * Lc[6 * rcap + r];
            }
            sig2[j] = wave_sum(s2);
            w[j] = wave_sum(dot);
            smax2 = fmax(smax2, sig2[j]);
        }
        const double cutoff = 1e-15 * sqrt(smax2);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (sqrt(sig2[j]) > cutoff) {
                const double coef = w[j] / sig2[j];
#pragma unroll
                for (int i = 0; i < 6; ++i) vout[i] += V[i][j] * coef;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) vout[i] = -a.lambda * vout[i];
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.v_c[(size_t)b * 6 + i] = vout[i];
        a.status[b] = status;
        int32_t* info = a.info + (size_t)b * 8;
        info[0] = n_mutual; info[1] = n_rows; info[2] = same_image ? 1 : 0; info[3] = n_matched;
        info[4] = sweeps; info[5] = R; info[6] = 0; info[7] = 0;
    }
}

int launch_servo(const ServoArgs& a, hipStream_t stream) {
    if (a.n_pairs <= 0 || a.T <= 0 || a.grid * a.grid != a.T || a.max_rows < a.num_pairs || a.num_pairs <= 0) return -2;
    if (a.mode == SEL_DENSE && a.max_rows < a.T) return -2;
    const size_t lds = servo_f64_offset(a.T, a.max_rows) + (size_t)7 * kLdsRows * 8 + (40 + 8 * 27) * 8;
    if (lds > 160 * 1024) return -3;
    static std::atomic<unsigned long long> raised{0};   // > 64 KiB of dynamic LDS (dense selection over thousands of tokens): per-device opt-in
    if (lds > 64 * 1024 && raise_lds_limit(reinterpret_cast<const void*>(&servo_kernel), 160 * 1024, raised)) return -3;
    launch(servo_kernel, dim3(a.n_pairs), dim3(256), lds, stream, a.row_best, a.col_best, a.K, a.selection, a.depth, a.T,
           a.mode, a.sel_stride, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
