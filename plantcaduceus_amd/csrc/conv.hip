// Depth-wise conv1d (width 4) + bias + SiLU, causal AND anti-causal in one pass, token-major.
//
// Replaces causal_conv1d.causal_conv1d_fn(x, weight, bias, activation="silu") (causal-conv1d 1.4.0;
// SURVEY.md §2b K2).  The reference calls it 4x per layer on channels-first (B,E,L) tensors and feeds the
// reverse Mamba a flipped COPY of the sequence; here the reverse direction is the anti-causal conv
//   y_rev[t] = silu(b + sum_k w[k] * x[t + 3 - k])
// on the same rows, so one read of x produces both directions (HBM-bound: 1 read + 2 writes of S*L*E).
//
// A thread owns 8 channels (16-byte vectors; a wave covers 1 KiB of a row) and slides a 7-row window
// x[t-3 .. t+3] along SEG consecutive timesteps: the 2 x (4 taps + bias) x 8 channel weights are loaded once
// per thread instead of once per output, every x row is read once (+6 halo rows per segment), and the rows
// of the next 4-step group are in flight while the current group is computed.
#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

constexpr int CONV_SEG = 64;   // timesteps per thread
constexpr int CONV_UN = 4;     // timesteps per unrolled group

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { typedef u32x4 type; };
template <> struct Raw8<float> { struct type { f32x4 a, b; }; };

template <typename T> __device__ __forceinline__ typename Raw8<T>::type raw_load8(const T* p);
template <> __device__ __forceinline__ u32x4 raw_load8<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
template <> __device__ __forceinline__ Raw8<float>::type raw_load8<float>(const float* p) {
    Raw8<float>::type r;
    r.a = *reinterpret_cast<const f32x4*>(p);
    r.b = *reinterpret_cast<const f32x4*>(p + 4);
    return r;
}
__device__ __forceinline__ void raw_unpack(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(r[i]); v[2 * i + 1] = bf16hi_to_f32(r[i]); }
}
__device__ __forceinline__ void raw_unpack(const Raw8<float>::type& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
}
template <typename R> __device__ __forceinline__ R raw_zero();
template <> __device__ __forceinline__ u32x4 raw_zero<u32x4>() { return u32x4{0u, 0u, 0u, 0u}; }
template <> __device__ __forceinline__ Raw8<float>::type raw_zero<Raw8<float>::type>() {
    Raw8<float>::type r;
    r.a = f32x4{0.f, 0.f, 0.f, 0.f};
    r.b = r.a;
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void conv_bidir_kernel(const T* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ wf, const float* __restrict__ bfw,
                                                         const float* __restrict__ wr, const float* __restrict__ brw,
                                                         T* __restrict__ yf, T* __restrict__ yr, int S, int L, int E,
                                                         int out_blocked, int in_blocked) {
    typedef typename Raw8<T>::type raw_t;
    const int nchunk = E >> 3;
    const int nseg = (L + CONV_SEG - 1) / CONV_SEG;
    const int64_t total = (int64_t)S * nseg * nchunk;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % nchunk) * 8;
    const int64_t rb = i / nchunk;
    const int s = (int)(rb / nseg);
    const int t0 = (int)(rb - (int64_t)s * nseg) * CONV_SEG;
    const int t1 = min(L, t0 + CONV_SEG);
    const T* xs = x + (int64_t)s * L * ldx + c;
    const int64_t pieces = ((int64_t)E * sizeof(T)) >> 7;

    float wfv[8][4], wrv[8][4], bfv[8], brv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wf + (int64_t)(c + e) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(wr + (int64_t)(c + e) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { wfv[e][k] = a[k]; wrv[e][k] = b[k]; }
    }
    load8<float>(bfw + c, bfv);
    load8<float>(brw + c, brv);

    auto row = [&](int t) -> raw_t {
        if (t < 0 || t >= L) return raw_zero<raw_t>();
        if (in_blocked) return raw_load8<T>(x + blocked_off((int64_t)s * L + t, (int64_t)c * sizeof(T), pieces) / (int64_t)sizeof(T));
        return raw_load8<T>(xs + (int64_t)t * ldx);
    };

    // ext[j] = x[t - 3 + j], j = 0..10 for the group starting at t; ext[0..6] carried, ext[7..10] prefetched
    float ext[7 + CONV_UN][8];
#pragma unroll
    for (int j = 0; j < 7; ++j) raw_unpack(row(t0 - 3 + j), ext[j]);
    raw_t nxt[CONV_UN];
#pragma unroll
    for (int j = 0; j < CONV_UN; ++j) nxt[j] = row(t0 + 4 + j);

    for (int t = t0; t < t1; t += CONV_UN) {
#pragma unroll
        for (int j = 0; j < CONV_UN; ++j) raw_unpack(nxt[j], ext[7 + j]);
#pragma unroll
        for (int j = 0; j < CONV_UN; ++j) nxt[j] = row(t + CONV_UN + 4 + j);     // rows of the NEXT group, in flight
#pragma unroll
        for (int k = 0; k < CONV_UN; ++k) {
            if (t + k < t1) {
                float of[8], orv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float af = bfv[e], ar = brv[e];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        af += wfv[e][q] * ext[k + q][e];          // x[t+k - 3 + q]
                        ar += wrv[e][q] * ext[k + 6 - q][e];      // x[t+k + 3 - q]
                    }
                    of[e] = silu(af);
                    orv[e] = silu(ar);
                }
                const int64_t orow = (int64_t)s * L + t + k;
                const int64_t o = out_blocked ? blocked_off(orow, (int64_t)c * sizeof(T), ((int64_t)E * sizeof(T)) >> 7) / (int64_t)sizeof(T)
                                              : orow * E + c;
                if (yf != nullptr) store8<T>(yf + o, of);
                if (yr != nullptr) store8<T>(yr + o, orv);
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) ext[j][e] = ext[j + CONV_UN][e];
    }
}

hipError_t launch_conv_bidir(const void* x, int64_t ldx, const float* wf, const float* bf, const float* wr,
                             const float* br, void* yf, void* yr, int S, int L, int E, int dt, bool out_blocked,
                             hipStream_t s, bool in_blocked) {
    if (S <= 0 || L <= 0) return hipSuccess;
    if (E % 8) return hipErrorInvalidValue;
    if (out_blocked && (E * (dt == BF16 ? 2 : 4)) % 128) return hipErrorInvalidValue;
    const int64_t total = (int64_t)S * ((L + CONV_SEG - 1) / CONV_SEG) * (E >> 3);
    const int64_t nb = (total + 255) / 256;
    if (nb > 0x7fffffff) return hipErrorInvalidValue;
    if (dt == BF16)
        hipLaunchKernelGGL(conv_bidir_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x, ldx, wf,
                           bf, wr, br, (bf16_t*)yf, (bf16_t*)yr, S, L, E, (int)out_blocked, (int)in_blocked);
    else
        hipLaunchKernelGGL(conv_bidir_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)x, ldx, wf, bf,
                           wr, br, (float*)yf, (float*)yr, S, L, E, (int)out_blocked, (int)in_blocked);
    return hipGetLastError();
}

}  // namespace pcad
