// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the ViT-VS hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitvs {

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // 16-byte register image (native vector: stays in VGPRs)

constexpr int WAVE = 64;



// Split-f16 ("f16x2") storage: a value v is kept as TWO fp16 numbers, hi = fp16(v) and lo = fp16(v - hi) (22 significant
// bits), and a contraction runs as hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation — the dropped
// lo.lo term is below 2^-22 of the product, i.e. fp32-class results at three 16-bit MFMAs per k-step instead of the
// sixteen-times-slower fp32 matrix pipe (reference arithmetic: fp32 throughout, dinov2_extractor.py:245-263).
// Layout of a row of C logical columns (C % 32 == 0): 2 C fp16, in groups of 64 = [hi of 32 columns | lo of the same 32
// columns]; a 128-byte k-tile of the GEMM ring is then 32 k with both halves, and MFMA step 0 / 1 of the tile loop
// (gemm_core.h) reads the hi / lo fragments.  hx2 is the storage unit (one fp16); sizeof == 2 like the 16-bit types.
struct hx2 { unsigned short bits; };
template <typename T> inline constexpr bool kSplit = false;
template <> inline constexpr bool kSplit<hx2> = true;
// f16 index of logical column c inside an f16x2 row (hi half; the lo half is 32 further on)
__device__ __host__ __forceinline__ int x2_index(int c) { return ((c >> 5) << 6) | (c & 31); }

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int PER_CHUNK = 4; };
template <> struct Elem<bf16> { static constexpr int PER_CHUNK = 8; };
template <> struct Elem<f16> { static constexpr int PER_CHUNK = 8; };
template <> struct Elem<hx2> { static constexpr int PER_CHUNK = 8; };
// the two 16-bit operand types share every kernel: vectors of 4 / 8 elements and the f32-accumulating 16x16x32 MFMA
template <typename H> struct Vec16;
template <> struct Vec16<bf16> { typedef bf16x4 x4; typedef bf16x8 x8; };
template <> struct Vec16<f16> { typedef f16x4 x4; typedef f16x8 x8; };
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Stores of data the NEXT launch consumes.  WT = true gives them the sc1 bit (agent scope: written through to memory
// as they are issued instead of staying dirty in this XCD's L2 until the end-of-kernel write-back).  In the one-wave
// (latency-bound) launches that shortens the producer's tail and the consumer's first reads: +4 % updates/s at one
// frame pair; in the many-row (bandwidth-bound) launches it costs 2-10 %, so those keep plain stores.
// The 16-byte form has no builtin: inline asm, followed by the two wait states gfx940+ requires between a VMEM store
// of more than 8 bytes and a VALU write of its data registers (the compiler cannot see into the asm to add them).
template <bool WT>
__device__ __forceinline__ void store_out8(void* p, unsigned long long v) {
    if constexpr (WT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<unsigned long long*>(p) = v;
}
template <bool WT>
__device__ __forceinline__ void store_out16(void* p, u32x4 v) {
    if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(p) = v;
}
template <bool WT>
__device__ __forceinline__ void store_out(float* p, f32x4 v) { store_out16<WT>(p, __builtin_bit_cast(u32x4, v)); }
template <bool WT>
__device__ __forceinline__ void store_out(float* p, float4 v) {
    store_out16<WT>(p, u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
}
template <bool WT>
__device__ __forceinline__ void store_out(bf16* p, bf16x4 v) { store_out8<WT>(p, __builtin_bit_cast(unsigned long long, v)); }
template <bool WT>
__device__ __forceinline__ void store_out(f16* p, f16x4 v) { store_out8<WT>(p, __builtin_bit_cast(unsigned long long, v)); }

// f16x2: the hi / lo halves of four values.  hi saturates at the fp16 range instead of becoming infinite (lo then carries
// what is left, so magnitudes up to 2 x 65504 stay finite); below 2^-3 the lo half is an fp16 subnormal — exact to 2^-25
// absolute, which the matrix cores honour (tools/denorm_probe.py).
struct Split4 { f16x4 hi, lo; };
__device__ __forceinline__ Split4 split4(f32x4 v) {
    Split4 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16 h = (f16)__builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f);
        s.hi[i] = h;
        s.lo[i] = (f16)(v[i] - (float)h);
    }
    return s;
}
// four consecutive logical columns c .. c + 3 (c % 4 == 0) of an f16x2 row
template <bool WT>
__device__ __forceinline__ void store_x2(hx2* row, int c, f32x4 v) {
    const Split4 s = split4(v);
    f16* p = reinterpret_cast<f16*>(row) + x2_index(c);
    store_out<WT>(p, s.hi);
    store_out<WT>(p + 32, s.lo);
}
__device__ __forceinline__ void store_x2_one(hx2* row, int c, float v) {
    f16* p = reinterpret_cast<f16*>(row) + x2_index(c);
    const f16 h = (f16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    p[0] = h;
    p[32] = (f16)(v - (float)h);
}
__device__ __forceinline__ float load_x2(const hx2* row, int c) {
    const f16* p = reinterpret_cast<const f16*>(row) + x2_index(c);
    return (float)p[0] + (float)p[32];
}

// Cross-lane reductions on the VALU (DPP row operations, v_readlane and gfx950's v_permlane{16,32}_swap):
// the HIP __shfl_* intrinsics go through the LDS crossbar (ds_bpermute, ~100+ cycles of latency per step; a
// 6-step butterfly is ~0.3 us), which is most of the run time of a one-wave LayerNorm row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned v) {   // lanes outside ROW_MASK (or with no source) get `old`
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __uint_as_float(dpp_mov<CTRL, ROW_MASK>(__float_as_uint(old), __float_as_uint(v)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double old, double v) {
    const unsigned long long o = __builtin_bit_cast(unsigned long long, old), x = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = dpp_mov<CTRL, ROW_MASK>((unsigned)o, (unsigned)x);
    const unsigned hi = dpp_mov<CTRL, ROW_MASK>((unsigned)(o >> 32), (unsigned)(x >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;               // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;     // lane i <-> 7-i within 8, i <-> 15-i within 16
constexpr int DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;        // last lane of a row -> the next row / rows 2,3

// Sum over the 64 lanes, returned to every lane (wave-uniform).  Fixed order: pairs, quads, 8, 16, rows.
template <typename F>
__device__ __forceinline__ F wave_sum_dpp(F v) {
    v += dpp_mov<DPP_XOR1, 0xF>(F(0), v);
    v += dpp_mov<DPP_XOR2, 0xF>(F(0), v);
    v += dpp_mov<DPP_HALF_MIRROR, 0xF>(F(0), v);
    v += dpp_mov<DPP_MIRROR, 0xF>(F(0), v);
    v += dpp_mov<DPP_BCAST15, 0xA>(F(0), v);
    v += dpp_mov<DPP_BCAST31, 0xC>(F(0), v);
    return v;   // lane 63 holds the total
}
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_sum_dpp(v);
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ int wave_sum(int v) {
    v += (int)dpp_mov<DPP_XOR1, 0xF>(0u, (unsigned)v);
    v += (int)dpp_mov<DPP_XOR2, 0xF>(0u, (unsigned)v);
    v += (int)dpp_mov<DPP_HALF_MIRROR, 0xF>(0u, (unsigned)v);
    v += (int)dpp_mov<DPP_MIRROR, 0xF>(0u, (unsigned)v);
    v += (int)dpp_mov<DPP_BCAST15, 0xA>(0u, (unsigned)v);
    v += (int)dpp_mov<DPP_BCAST31, 0xC>(0u, (unsigned)v);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = wave_sum_dpp(v);
    const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_huge_valf();
    v = fmaxf(v, dpp_mov<DPP_XOR1, 0xF>(ninf, v));
    v = fmaxf(v, dpp_mov<DPP_XOR2, 0xF>(ninf, v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR, 0xF>(ninf, v));
    v = fmaxf(v, dpp_mov<DPP_MIRROR, 0xF>(ninf, v));
    v = fmaxf(v, dpp_mov<DPP_BCAST15, 0xA>(ninf, v));
    v = fmaxf(v, dpp_mov<DPP_BCAST31, 0xC>(ninf, v));
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
// The value held by lane ^ 16 / lane ^ 32 (v_permlane16_swap / v_permlane32_swap of two copies of v).
__device__ __forceinline__ unsigned lane_xor16(unsigned v) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((threadIdx.x >> 4) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned lane_xor32(unsigned v) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return ((threadIdx.x >> 5) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ float lane_xor16(float v) { return __uint_as_float(lane_xor16(__float_as_uint(v))); }
__device__ __forceinline__ float lane_xor32(float v) { return __uint_as_float(lane_xor32(__float_as_uint(v))); }
// Reductions over the 4 rows of 16 lanes (same lane & 15), result in all 4: sum / max of {v, r[0], r[1]} needs no select
__device__ __forceinline__ float rows_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    if (o == 16) { lo = lane_xor16(lo); hi = lane_xor16(hi); }
    else if (o == 32) { lo = lane_xor32(lo); hi = lane_xor32(hi); }
    else { lo = __shfl_xor(lo, o, WAVE); hi = __shfl_xor(hi, o, WAVE); }
    return ((unsigned long long)hi << 32) | lo;
}

// max of a 64-bit key over the 16 lanes of a DPP row (lane & 15), result in all 16
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long key) {
    const unsigned lo = dpp_mov<CTRL, 0xF>(0u, (unsigned)key), hi = dpp_mov<CTRL, 0xF>(0u, (unsigned)(key >> 32));
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return (o > key) ? o : key;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long key) {
    key = dpp_max_u64<DPP_XOR1>(key);
    key = dpp_max_u64<DPP_XOR2>(key);
    key = dpp_max_u64<DPP_HALF_MIRROR>(key);
    return dpp_max_u64<DPP_MIRROR>(key);
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
template <> __device__ __forceinline__ f16 from_float<f16>(float v) { return (f16)v; }

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_float(f16 v) { return (float)v; }

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
