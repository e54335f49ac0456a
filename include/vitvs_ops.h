/*
 * vitvs_ops.h — single-operator entry points of libvitvs_hip.so, exported so that the
 * kernel-level parity tests (tests/test_gpu_ops.py) can drive each HIP kernel through the C ABI.
 * Not part of the drop-in surface; the reference issues these as stock PyTorch ops:
 *   linear / linear_residual   nn.Linear (+ GELU, + LayerScale and residual add)  dino_patch/attention.py:72,79; block.py:78-96
 *   layernorm                  nn.LayerNorm(eps=1e-6)                              dino_patch/block.py:57,75
 *   attention                  scaled_dot_product_attention over [B,H,N,64]        dino_patch/attention.py:73-78
 *   linear_partial+residual_ln the same nn.Linear + residual (+ next LayerNorm), as K slices  dino_patch/block.py:90-115
 * All pointers are device pointers; `precision` is enum vitvs_precision; `stream` a hipStream_t.
 */
#ifndef VITVS_OPS_H
#define VITVS_OPS_H
#include <stdint.h>
#ifndef VITVS_API
#define VITVS_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* out[M][N] = act(A[M][K] . W[N][K]^T + bias[N]); A, W, out in `precision`; K % 64 == 0, N % 64 == 0 */
VITVS_API int vitvs_op_linear(int32_t precision, const void* A, const void* W, const float* bias, void* out, int32_t M,
                    int32_t N, int32_t K, int32_t gelu, void* stream);
/* The same operator on a chosen tile family (the library picks one by itself in vitvs_op_linear): variant 0 = the
 * library's choice, 1 = the 64/128-row tiles of gemm.hip as that file picks them, 2 = its 128 x 128 tiles whatever the shape
 * (16-bit precisions, N % 128 == 0), 256 / 192 / 128 = the 256-row tiles of gemm_big.hip with that column width, 1192 = its
 * 192 x 128 tile (16-bit precisions, N % width == 0, K >= 128).  For the parity tests of both families on one shape and for
 * tools/big_ops.  slices > 0 selects the split-K form: out = fp32 part[slices][M][N], bias / gelu ignored. */
VITVS_API int vitvs_op_linear_variant(int32_t precision, int32_t variant, const void* A, const void* W, const float* bias,
                            void* out, int32_t M, int32_t N, int32_t K, int32_t gelu, int32_t slices, void* stream);
/* x[M][N] (fp32) += ls[N] * (A . W^T + bias); ls may be NULL */
VITVS_API int vitvs_op_linear_residual(int32_t precision, const void* A, const void* W, const float* bias, const float* ls,
                             float* x, int32_t M, int32_t N, int32_t K, void* stream);
/* out[M][D] = LayerNorm(x[M][D]) * gamma + beta; x fp32, out in `precision`; D in {128,256,384,768,1024} */
VITVS_API int vitvs_op_layernorm(int32_t precision, const float* x, const float* gamma, const float* beta, void* out, int32_t M,
                       int32_t D, float eps, void* stream);
/* out[n_img*N][H*64] = softmax(q k^T / 8) v per (image, head); qkv [n_img*N][3*H*64].  From 512 tokens on the keys of a query
 * block may be split over several workgroups that merge through a workspace this hook owns, one per (device, stream): calls
 * on different streams do not share state (handles own their workspace).
 * vitvs_op_attention takes the raw q a plain qkv projection produces and applies 1/8 (and the log2(e) of its exp2) inside the
 * kernels; in the 16-bit precisions the long-sequence kernel does that by one more 16-bit rounding of q.
 * vitvs_op_attention_q with q_prescaled != 0 is the form the handle's forward uses in the 16-bit precisions: the q third of qkv
 * already carries 0.125 * log2(e) (the handle folds it into the q rows of attn.qkv.weight / bias in fp32, before their one
 * rounding to 16 bits), and the kernels apply nothing.  q_prescaled is ignored for VITVS_F32. */
VITVS_API int vitvs_op_attention(int32_t precision, const void* qkv, void* out, int32_t n_img, int32_t N, int32_t H,
                       void* stream);
VITVS_API int vitvs_op_attention_q(int32_t precision, const void* qkv, void* out, int32_t n_img, int32_t N, int32_t H,
                         int32_t q_prescaled, void* stream);

/* Split-K pair used for the narrow layers (proj, fc2, patch embedding):
 *   slices = vitvs_op_splitk_slices(precision, M, N, K)           (>= 1; the plan the forward uses)
 *   part[z][M][N] (fp32) = A[:, z-th K slice] . W[:, z-th K slice]^T   for z < slices
 *   x[M][D] (fp32) += ls[D] * (sum_z part[z] + bias)  (slices summed in index order); then, if gamma != NULL,
 *   out[M][D] = LayerNorm(x) * gamma + beta in `precision` (out may be NULL when gamma is NULL). */
VITVS_API int vitvs_op_splitk_slices(int32_t precision, int32_t M, int32_t N, int32_t K);
/* Measurement hook (tools/l2_warm_probe.py): the private L2 of every XCD reads all `bytes` of p (share_xcds != 0: XCD x only the
 * x-th eighth), so that a following launch finds the operand in L2 rather than in the Infinity Cache.  No result. */
VITVS_API int vitvs_op_touch(const void* p, int64_t bytes, int32_t share_xcds, void* stream);
/* VITVS_F16X2 operands of the hooks: A / qkv rows and W rows hold 2 C fp16 per C logical columns, [hi of 32 columns | lo of the
 * same 32] per 64 fp16 (csrc/common.h), outputs likewise; W may carry a power of two 2^e (e = 0 .. 31, what the handle's weight
 * upload does so that the lo halves are normal fp16 numbers): this sets e for the calling thread's later vitvs_op_linear* calls
 * (the sums leave multiplied by 2^-e).  e outside 0 .. 31 only reads.  Returns the previous value. */
VITVS_API int vitvs_op_weight_exponent(int32_t e);
/* The plan hint a handle carries as its "in_flight" option (vitvs.h), for the pointer-only hooks of this header: the calling
 * thread's later vitvs_op_* calls plan as if n updates were in flight (n >= 1; n < 1 only reads).  Returns the previous value. */
VITVS_API int vitvs_op_plan_in_flight(int32_t n);
/* the tile the library launches for a linear layer: tile[0..2] = rows, columns, k-groups (k-groups 0: the 256-row kernels of
 * gemm_big.hip); slices = 0: vitvs_op_linear, > 0: vitvs_op_linear_partial with that many K slices.  No device work. */
VITVS_API int vitvs_op_linear_tile(int32_t precision, int32_t M, int32_t N, int32_t K, int32_t slices, int32_t* tile);
VITVS_API int vitvs_op_linear_partial(int32_t precision, const void* A, const void* W, float* part, int32_t M, int32_t N,
                            int32_t K, int32_t slices, void* stream);
VITVS_API int vitvs_op_residual_ln(int32_t precision, float* x, const float* part, int32_t slices, const float* bias,
                         const float* ls, const float* gamma, const float* beta, void* out, int32_t M, int32_t D,
                         float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
