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

__device__ __forceinline__ int block_sum_int(int v, int* scratch) {
    // scratch: 5 ints; all 256 threads call
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void servo_kernel(ServoArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T, g = a.grid;
    int* nn1 = reinterpret_cast<int*>(smem);
    int* nn2 = nn1 + T;
    int* flag = nn2 + T;            // candidate flags, later: per-position "taken" rank
    int* sel = flag + T;            // [max_rows]
    int* scratch = sel + a.max_rows;  // [8 + 256]
    float* fscr = reinterpret_cast<float*>(scratch + 8 + 256);  // [4]
    float* simL = fscr + 4;                                     // [T] sim_1, kept on chip for the feature stage
    double* Llds = reinterpret_cast<double*>(smem + (((size_t)(4 * T + a.max_rows + 8 + 256 + 4) * 4 + 15) & ~(size_t)15));
    // camera intrinsics: issued now, needed only in the feature stage (one memory round trip hidden)
    const double fx = a.K[b * 4 + 0], fy = a.K[b * 4 + 1], cx = a.K[b * 4 + 2], cy = a.K[b * 4 + 3];

    const unsigned long long* rb = a.row_best + (size_t)b * T;
    const unsigned long long* cb = a.col_best + (size_t)b * T;

    // 1. decode the packed (similarity, index) keys
    float ssum = 0.f;
    for (int i = tid; i < T; i += 256) {
        const unsigned long long kr = rb[i], kc = cb[i];
        const int n1 = (int)best_index(kr), n2 = (int)best_index(kc);
        const float s1 = best_value(kr);
        nn1[i] = n1;
        nn2[i] = n2;
        a.nn1[(size_t)b * T + i] = n1;
        a.nn2[(size_t)b * T + i] = n2;
        a.sim1[(size_t)b * T + i] = s1;
        simL[i] = s1;
        ssum += s1;
    }
    ssum = wave_sum(ssum);
    if (lane == 0) fscr[wave] = ssum;
    __syncthreads();
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
    const int n_mutual = block_sum_int(cnt, scratch);
    const bool none = !same_image && (n_mutual == T || n_mutual == 0);
    const int n_cand = same_image ? T : n_mutual;

    // 3. selection -> sel[0 .. n_matched)
    int n_matched = 0;
    const int want = (a.mode == SEL_DENSE) ? min(n_cand, a.max_rows) : a.num_pairs;
    if (a.mode == SEL_EXPLICIT) {
        n_matched = min(a.n_selected[b], a.num_pairs);
        for (int k = tid; k < n_matched; k += 256) {
            int i = a.selection[(size_t)b * a.sel_stride + k];
            sel[k] = min(max(i, 0), T - 1);
        }
    } else if (!none) {
        // visit tokens in the given order (identity for DENSE); keep the first `want` candidates
        const int32_t* order = (a.mode == SEL_PRIORITY) ? a.selection + (size_t)b * a.sel_stride : nullptr;
        const int per = (T + 255) / 256;
        const int p0 = tid * per, p1 = min(p0 + per, T);
        int local = 0;
        for (int p = p0; p < p1; ++p) {
            int i = order ? order[p] : p;
            i = min(max(i, 0), T - 1);
            local += flag[i];
        }
        // exclusive scan of the 256 per-thread counts
        int* cnts = scratch + 8;
        __syncthreads();
        cnts[tid] = local;
        __syncthreads();
        if (wave == 0) {
            int v[4], run = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = cnts[lane * 4 + q]; run += v[q]; }
            int incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, WAVE);
                if (lane >= o) incl += up;
            }
            int excl = incl - run;
#pragma unroll
            for (int q = 0; q < 4; ++q) { cnts[lane * 4 + q] = excl; excl += v[q]; }
        }
        __syncthreads();
        int rank = cnts[tid];
        for (int p = p0; p < p1; ++p) {
            int i = order ? order[p] : p;
            i = min(max(i, 0), T - 1);
            if (flag[i]) {
                if (rank < want) sel[rank] = i;
                ++rank;
            }
        }
        n_matched = min(n_cand, want);
    }
    __syncthreads();

    // 4. pixel features, depth, interaction matrix
    const int n_rows = (a.mode == SEL_DENSE) ? n_matched : a.num_pairs;   // feature pairs entering L
    const bool too_few = (a.mode != SEL_DENSE) && (n_matched != a.num_pairs) && (n_matched < 4);
    const int R = 2 * n_rows;
    const bool use_lds = R <= kLdsRows;
    const int rcap = use_lds ? kLdsRows : 2 * a.max_rows;
    double* Lc = use_lds ? Llds : a.L_ws + (size_t)b * 7 * 2 * a.max_rows;   // 6 columns + e, column-major
    const uint16_t* depth = a.depth ? a.depth + (size_t)b * a.depth_h * a.depth_w : nullptr;
    int32_t* uv_out = a.s_uv + (size_t)b * a.max_rows * 4;
    double* feat_out = a.feat + (size_t)b * a.max_rows * 4;
    int32_t* sel_out = a.sel_out + (size_t)b * a.max_rows;
    for (int k = tid; k < n_rows; k += 256) {
        long us = 0, vs = 0, u = 0, v = 0;
        int tok = -1;
        double simk = 0.0;
        if (!none && !too_few && k < n_matched) {
            tok = sel[k];
            const int j = same_image ? tok : nn1[tok];
            const float r1 = __fadd_rn(__fmul_rn((float)(tok / g), a.scale_f), a.half_f);
            const float c1 = __fadd_rn(__fmul_rn((float)(tok % g), a.scale_f), a.half_f);
            const float r2 = __fadd_rn(__fmul_rn((float)(j / g), a.scale_f), a.half_f);
            const float c2 = __fadd_rn(__fmul_rn((float)(j % g), a.scale_f), a.half_f);
            us = (long)rint((double)c1 * a.scale_x);
            vs = (long)rint((double)r1 * a.scale_y);
            u = (long)rint((double)c2 * a.scale_x);
            v = (long)rint((double)r2 * a.scale_y);
            simk = same_image ? 1.0 : (double)simL[tok];
        }
        const double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
        const double xs = ((double)us - cx) / fx, ys = ((double)vs - cy) / fy;
        double z = 100.0;
        if (depth && u >= 0 && u < a.depth_w && v >= 0 && v < a.depth_h) {
            const unsigned d = depth[(size_t)v * a.depth_w + u];
            z = d != 0 ? (double)d / 1000.0 : 100.0;
        }
        const int r0 = 2 * k, r1i = 2 * k + 1;
        Lc[0 * rcap + r0] = -1.0 / z;  Lc[1 * rcap + r0] = 0.0;        Lc[2 * rcap + r0] = x / z;
        Lc[3 * rcap + r0] = x * y;     Lc[4 * rcap + r0] = -(1.0 + x * x);  Lc[5 * rcap + r0] = y;
        Lc[0 * rcap + r1i] = 0.0;      Lc[1 * rcap + r1i] = -1.0 / z;  Lc[2 * rcap + r1i] = y / z;
        Lc[3 * rcap + r1i] = 1.0 + y * y;  Lc[4 * rcap + r1i] = -(x * y);  Lc[5 * rcap + r1i] = -x;
        Lc[6 * rcap + r0] = x - xs;
        Lc[6 * rcap + r1i] = y - ys;
        uv_out[k * 4 + 0] = (int32_t)us; uv_out[k * 4 + 1] = (int32_t)vs;
        uv_out[k * 4 + 2] = (int32_t)u;  uv_out[k * 4 + 3] = (int32_t)v;
        feat_out[k * 4 + 0] = z; feat_out[k * 4 + 1] = x; feat_out[k * 4 + 2] = y; feat_out[k * 4 + 3] = simk;
        sel_out[k] = tok;
    }
    __syncthreads();
    if (use_lds) {  // keep a copy of the untouched L and e for the parity tests
        double* Lg = a.L_ws + (size_t)b * 7 * 2 * a.max_rows;
        for (int idx = tid; idx < 7 * R; idx += 256) {
            const int c = idx / R, r = idx - c * R;
            Lg[(size_t)c * 2 * a.max_rows + r] = Lc[c * rcap + r];
        }
    }

    // 5. v_c = -lambda * pinv(L) e, fp64.
    // Fast path (L_e of full column rank and well conditioned, the normal servo case): pinv(L) e is the
    // least-squares solution, obtained from the 6x6 normal equations by Cholesky; 27 threads form
    // G = L^T L and g = L^T e, one thread factors (LDL^T) and solves.  If a pivot falls below 1e-8 of its
    // diagonal (cond(L) > ~1e4, or rank deficiency, e.g. all-identical zero-padded rows) the general
    // path below runs instead: one-sided Jacobi SVD with numpy.linalg.pinv's rcond = 1e-15 cut-off.
    int status = ST_OK;
    if (!depth) status = ST_NO_DEPTH;
    else if (none) status = ST_NO_CORRESPONDENCE;
    else if (too_few) status = ST_TOO_FEW;
    double* Gs = Llds + 7 * kLdsRows;   // [27] G (21, upper triangle row-major) + g (6); [27..33] solution, [34] flag
    bool solved = false;
    if (status == ST_OK && R > 0 && use_lds) {
        // 27 quantities x 8 row slices on 216 threads (fixed slice order -> deterministic), then 27 sums of 8
        {
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
                double acc = 0.0;
                for (int r = slice; r < R; r += 8) acc += Lc[ca * rcap + r] * Lc[cb * rcap + r];
                Gs[40 + slice * 27 + qid] = acc;
            }
        }
        __syncthreads();
        if (tid < 27) {
            double acc = 0.0;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) acc += Gs[40 + sl * 27 + tid];
            Gs[tid] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            double Gm[6][6], rhs[6];
            int q = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) { Gm[i][j] = Gs[q]; Gm[j][i] = Gs[q]; ++q; }
            for (int i = 0; i < 6; ++i) rhs[i] = Gs[21 + i];
            // G = L D L^T (unit lower-triangular L, no square roots, one reciprocal per pivot)
            bool good = true;
            double Lf[6][6], dinv[6], dpiv[6];
            for (int j = 0; j < 6 && good; ++j) {
                double d = Gm[j][j];
                for (int k = 0; k < j; ++k) d -= Lf[j][k] * Lf[j][k] * dpiv[k];
                if (!(d > 1e-8 * Gm[j][j]) || !(Gm[j][j] > 0.0)) { good = false; break; }
                dpiv[j] = d;
                dinv[j] = 1.0 / d;
                for (int i = j + 1; i < 6; ++i) {
                    double t = Gm[i][j];
                    for (int k = 0; k < j; ++k) t -= Lf[i][k] * Lf[j][k] * dpiv[k];
                    Lf[i][j] = t * dinv[j];
                }
            }
            if (good) {
                double y[6], xsol[6];
                for (int i = 0; i < 6; ++i) {          // L y = rhs
                    double t = rhs[i];
                    for (int k = 0; k < i; ++k) t -= Lf[i][k] * y[k];
                    y[i] = t;
                }
                for (int i = 5; i >= 0; --i) {         // L^T x = D^-1 y
                    double t = y[i] * dinv[i];
                    for (int k = i + 1; k < 6; ++k) t -= Lf[k][i] * xsol[k];
                    xsol[i] = t;
                }
                for (int i = 0; i < 6; ++i) Gs[27 + i] = -a.lambda * xsol[i];
            }
            Gs[34] = good ? 1.0 : 0.0;
        }
        __syncthreads();
        solved = Gs[34] != 0.0;
    }
    if (wave != 0) return;
    double vout[6] = {0, 0, 0, 0, 0, 0};
    int sweeps = 0;
    if (solved) {
#pragma unroll
        for (int i = 0; i < 6; ++i) vout[i] = Gs[27 + i];
        sweeps = -1;
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
    size_t ints = (size_t)4 * a.T + a.max_rows + 8 + 256 + 4;
    size_t lds = ((ints * 4 + 15) & ~(size_t)15) + (size_t)7 * kLdsRows * 8 + (40 + 8 * 27) * 8;
    if (lds > 64 * 1024) return -3;
    launch(servo_kernel, dim3(a.n_pairs), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace vitvs
