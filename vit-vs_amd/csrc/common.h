// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the ViT-VS hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitvs {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // 16-byte register image (native vector: stays in VGPRs)

constexpr int WAVE = 64;



template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int PER_CHUNK = 4; };
template <> struct Elem<bf16> { static constexpr int PER_CHUNK = 8; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, o, WAVE);
    hi = __shfl_xor(hi, o, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

// LDS operand tile: rows of 128 bytes (8 chunks of 16 B).  Chunk c of row r lives at
// chunk slot c ^ ((r >> 1) & 7): the 16 rows x 4 k-groups one ds_read_b128 touches then
// hit 16 distinct 16-B slots of the 256-B bank row in each of its lane groups.
__device__ __forceinline__ int tile128_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// LDS operand tile with 256-byte rows (16 chunks): slot c ^ (r & 15).
__device__ __forceinline__ int tile256_off(int row, int chunk) {
    return row * 256 + ((chunk ^ (row & 15)) << 4);
}

template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_float<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16 v) { return (float)v; }

// total order on floats as unsigned keys (larger float -> larger key)
__device__ __forceinline__ unsigned ordered_key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}
// (similarity, index) packed so that integer max == (largest similarity, then smallest index):
// torch.max's first-max-on-ties rule (reference: vitvs_v2.py:80-81).
__device__ __forceinline__ unsigned long long pack_best(float sim, unsigned idx) {
    return ((unsigned long long)ordered_key(sim) << 32) | (unsigned long long)(0xffffffffu - idx);
}
__device__ __forceinline__ unsigned best_index(unsigned long long k) { return 0xffffffffu - (unsigned)k; }
__device__ __forceinline__ float best_value(unsigned long long k) { return ordered_value((unsigned)(k >> 32)); }

}  // namespace vitvs
