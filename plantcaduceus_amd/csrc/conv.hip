// Depth-wise conv1d (width 4) + bias + SiLU, causal AND anti-causal in one pass, token-major.
//
// Replaces causal_conv1d.causal_conv1d_fn(x, weight, bias, activation="silu") (causal-conv1d 1.4.0;
// SURVEY.md §2b K2).  The reference calls it 4x per layer on channels-first (B,E,L) tensors and feeds the
// reverse Mamba a flipped COPY of the sequence; here the reverse direction is the anti-causal conv
//   y_rev[t] = silu(b + sum_k w[k] * x[t + 3 - k])
// on the same rows, so one read of x produces both directions (HBM-bound: 1 read + 2 writes of S*L*E).
// A thread owns 8 channels (16-byte vectors) x 4 consecutive positions: 10 row loads -> 8 output rows.
#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

template <typename T>
__global__ __launch_bounds__(256) void conv_bidir_kernel(const T* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ wf, const float* __restrict__ bfw,
                                                         const float* __restrict__ wr, const float* __restrict__ brw,
                                                         T* __restrict__ yf, T* __restrict__ yr, int S, int L, int E) {
    constexpr int TT = 4;
    const int nchunk = E >> 3;
    const int ntb = (L + TT - 1) / TT;
    const int64_t total = (int64_t)S * ntb * nchunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % nchunk);
        const int64_t rb = i / nchunk;
        const int s = (int)(rb / ntb);
        const int t0 = (int)(rb - (int64_t)s * ntb) * TT;
        const int c = c8 * 8;
        float xin[TT + 6][8];
#pragma unroll
        for (int j = 0; j < TT + 6; ++j) {
            const int t = t0 - 3 + j;
            if (t >= 0 && t < L) {
                load8<T>(x + ((int64_t)s * L + t) * ldx + c, xin[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xin[j][e] = 0.f;
            }
        }
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
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + tt;
            if (t >= L) break;
            float of[8], orv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float af = bfv[e], ar = brv[e];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    af += wfv[e][k] * xin[tt + k][e];        // x[t - 3 + k]
                    ar += wrv[e][k] * xin[tt + 6 - k][e];    // x[t + 3 - k]
                }
                of[e] = silu(af);
                orv[e] = silu(ar);
            }
            const int64_t o = ((int64_t)s * L + t) * E + c;
            if (yf != nullptr) store8<T>(yf + o, of);
            if (yr != nullptr) store8<T>(yr + o, orv);
        }
    }
}

hipError_t launch_conv_bidir(const void* x, int64_t ldx, const float* wf, const float* bf, const float* wr,
                             const float* br, void* yf, void* yr, int S, int L, int E, int dt, hipStream_t s) {
    if (S <= 0 || L <= 0) return hipSuccess;
    if (E % 8) return hipErrorInvalidValue;
    const int64_t total = (int64_t)S * ((L + 3) / 4) * (E >> 3);
    int64_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (dt == BF16)
        hipLaunchKernelGGL(conv_bidir_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x, ldx, wf,
                           bf, wr, br, (bf16_t*)yf, (bf16_t*)yr, S, L, E);
    else
        hipLaunchKernelGGL(conv_bidir_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)x, ldx, wf, bf,
                           wr, br, (float*)yf, (float*)yr, S, L, E);
    return hipGetLastError();
}

}  // namespace pcad
