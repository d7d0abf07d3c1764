// Shared device helpers for the gfx950 kernels (wave64, bf16 bit tricks, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcad {

typedef uint16_t bf16_t;   // raw bfloat16 bits

typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

// round-to-nearest-even fp32 -> bf16: the compiler lowers the __bf16 conversion to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float round_to_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// storage-type traits: T is `float` or `bf16_t`
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float round(float v) { return v; }
    static __device__ __forceinline__ f32x2_t round2(f32x2_t v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ float round(float v) { return round_to_bf16(v); }
    static __device__ __forceinline__ f32x2_t round2(f32x2_t v) {         // one v_cvt_pk_bf16_f32 for the pair
        const uint32_t pk = pack_bf16x2(v[0], v[1]);
        return f32x2_t{bf16lo_to_f32(pk), bf16hi_to_f32(pk)};
    }
};

// 8 consecutive elements <-> 8 floats (16-byte accesses for bf16, 2x16 bytes for fp32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(a[i]); v[2 * i + 1] = bf16hi_to_f32(a[i]); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32

// "Blocked" activation layout used for the tensors that an MFMA GEMM streams as its A operand (xc, y):
// [row / 8][k-piece][row % 8][128 bytes], k-piece = (byte offset in the row) / 128.  One 1 KiB LDS-DMA instruction of
// the GEMM (8 rows x 128 B) then reads ONE contiguous 1 KiB block instead of 8 pieces from 8 DRAM rows.
// byte offset of (row r, byte cb of the row) for rows of `pieces` x 128 bytes:
__host__ __device__ __forceinline__ int64_t blocked_off(int64_t r, int64_t cb, int64_t pieces) {
    return (((r >> 3) * pieces + (cb >> 7)) << 10) + ((r & 7) << 7) + (cb & 127);
}

// "Fragment" layout of the fp32 residual stream in the norm-folded layer form (api.hip "norm_fold"): the order in which the 4-wave
// GEMM's lanes hold a 256 x 256 output tile in their accumulators, so that the read-modify-write in out_proj's epilogue is
// whole 1 KiB runs per instruction (in plain rows a lane's 16 bytes sit in a different row than its neighbour's: 64 separate
// 16-byte requests per store instruction, which made the epilogue L2-request-bound: +23 us per tile measured).
//   [tile_m][tile_n][wave = 2 wm + wn][i = 0..7][jg = 0..1][k = 0..3][lane = 16 lg + li][r = 0..3]   (floats)
//   row = 256 tile_m + 128 wm + 16 i + li,   col = 256 tile_n + 128 wn + 64 jg + 16 lg + 4 k + r
// float offset of the 4-float quad that holds (row, col), col % 4 == 0, for a [rows, D] tensor (rows % 256 == 0, D % 256 == 0):
__host__ __device__ __forceinline__ int64_t res_frag_off(int64_t row, int col, int D) {
    const int64_t tile = (row >> 8) * (D >> 8) + (col >> 8);
    const int wave = (int)((row >> 7) & 1) * 2 + ((col >> 7) & 1);
    const int i = (int)(row >> 4) & 7, li = (int)row & 15;
    const int jg = (col >> 6) & 1, lg = (col >> 4) & 3, k = (col >> 2) & 3;
    return (((tile * 4 + wave) * 8 + i) * 2 + jg) * 1024 + k * 256 + (lg * 16 + li) * 4;
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// silu(v) = v / (1 + exp(-v))
__device__ __forceinline__ float silu(float v) { return v * fast_rcp(1.0f + fast_exp2(-v * kLog2e)); }

// softplus with torch's threshold (20): log1p(e), e = exp(x).  For e < 2^-6 the series e - e^2/2 + e^3/3
// (truncation e^3/4 < 1e-6 relative) keeps full relative accuracy for small time-steps (softplus(-7) ~ 1e-3) where
// log(1 + e) would lose it to the rounding of 1 + e; above, log2(1 + e) is accurate to < 4e-6 relative.
// 2 transcendentals (v_exp_f32, v_log_f32) + 6 plain ops.
__device__ __forceinline__ float softplus(float x) {
    const float e = fast_exp2(x * kLog2e);
    const float big = fast_log2(1.0f + e) * kLn2;
    float p = 1.0f / 3.0f;
    p = p * e - 0.5f;
    p = p * e + 1.0f;
    const float small = p * e;
    const float l = e < 0.015625f ? small : big;
    return x > 20.0f ? x : l;
}

// Two values at once on the packed fp32 pipe, branch- and select-free:
//   softplus(x) = max(x, 0) + log1p(e),  e = exp(-|x|) in (0, 1]
//   log1p(e)    = log2(w) ln 2 + r (1 - e),  w = fl(1 + e),  r = e - (w - 1)  (the rounding error of 1 + e, exact)
// r / w is the first-order correction of log(w + r); 1 / w ~ 1 - e where it matters (e << 1: the result is ~e and log2(w)
// alone would lose it to the rounding of 1 + e), and for e -> 1 the term is < 2^-25 absolute against a result > 0.4.
// For x > 20 the sum is x + 2e-9 = x exactly (torch's threshold branch needs no select) and nothing overflows.
// Per pair: 7 packed ops + 2 v_max + 2 v_exp_f32 + 2 v_log_f32 (the select form above: 20 plain ops + 4 transcendentals).
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t softplus2(f2_t x) {
    const f2_t t = x * f2_t{kLog2e, kLog2e};
    const f2_t e = {fast_exp2(-__builtin_fabsf(t[0])), fast_exp2(-__builtin_fabsf(t[1]))};
    const f2_t w = e + f2_t{1.0f, 1.0f};
    const f2_t r = e - (w - f2_t{1.0f, 1.0f});
    const f2_t lg = {fast_log2(w[0]), fast_log2(w[1])};
    const f2_t c = r - r * e;
    const f2_t m = {__builtin_fmaxf(x[0], 0.0f), __builtin_fmaxf(x[1], 0.0f)};
    return lg * f2_t{kLn2, kLn2} + (c + m);
}

}  // namespace pcad
