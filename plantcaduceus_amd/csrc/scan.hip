// Selective-scan recurrence (Mamba-v1 S6), token-major, one direction per launch.
//
// Replaces selective_scan_cuda.fwd behind mamba_ssm's selective_scan_fn / mamba_inner_fn
// (mamba-ssm 2.2.2):
//     delta = softplus(delta + delta_bias)
//     h_t   = exp(delta_t * A) (.) h_{t-1} + delta_t * B_t * u_t          (A in R^{E x 16}, B_t, C_t in R^16)
//     y_t   = <h_t, C_t> + D * u_t ;   out_t = y_t * silu(z_t)
// The reference maps one CUDA block to a (batch, channel) pair and scans along time with a block scan.
// With 2B*E independent (strand, channel) recurrences per layer there is no need to parallelise time on
// MI355X: a wave owns 64 consecutive channels of one strand, lane = channel, the 16 states live in VGPRs
// and time is walked sequentially with zero cross-lane traffic.  u/delta/z rows are 128/256-byte
// coalesced reads of the token-major tensors, software-prefetched one 8-step chunk ahead; B_t / C_t are
// wave-uniform and come in through the scalar unit (s_load -> SGPR operands of the VALU ops).
// The reverse direction walks t = L-1..0 on the same rows (no flipped copy); `accumulate` adds the
// forward direction's output (BiMambaWrapper strategy "add", tied out_proj folded by linearity).
// The kernel is VALU/transcendental bound: 16 v_exp_f32 + ~64 fp32 ops per (t, channel).
#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

constexpr int NSTATE = 16;
constexpr int CH = 8;   // timesteps per prefetch chunk

// 16 wave-uniform B_t / C_t values.  The address is uniform, so these become s_load_dwordx{8,16} and the
// bf16 unpack runs on the scalar ALU; the values are then SGPR operands of the VALU recurrence.
template <typename T> __device__ __forceinline__ void load_state16(const T* __restrict__ p, float (&o)[NSTATE]);
template <> __device__ __forceinline__ void load_state16<float>(const float* __restrict__ p, float (&o)[NSTATE]) {
#pragma unroll
    for (int n = 0; n < NSTATE; ++n) o[n] = p[n];
}
template <> __device__ __forceinline__ void load_state16<bf16_t>(const bf16_t* __restrict__ p, float (&o)[NSTATE]) {
    const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int k = 0; k < NSTATE / 2; ++k) {
        const uint32_t w = q[k];
        o[2 * k] = __uint_as_float(w << 16);
        o[2 * k + 1] = __uint_as_float(w & 0xffff0000u);
    }
}

template <typename T, bool REV, bool ACC, bool HASZ>
__global__ __launch_bounds__(64) void scan_kernel(const T* __restrict__ u, const T* __restrict__ delta,
                                                  const T* __restrict__ z, int64_t ldz,
                                                  const T* __restrict__ Bm, const T* __restrict__ Cm,
                                                  int64_t ldbc, const float* __restrict__ A2, float a_scale,
                                                  const float* __restrict__ Dskip, const float* __restrict__ dbias,
                                                  const T* yin, T* y, int L, int E) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int s = blockIdx.y;
    const int64_t row0 = (int64_t)s * L;

    float a2[NSTATE], h[NSTATE];
#pragma unroll
    for (int n = 0; n < NSTATE; n += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A2 + (int64_t)c * NSTATE + n);
        a2[n] = v[0] * a_scale; a2[n + 1] = v[1] * a_scale; a2[n + 2] = v[2] * a_scale; a2[n + 3] = v[3] * a_scale;
    }
#pragma unroll
    for (int n = 0; n < NSTATE; ++n) h[n] = 0.f;
    const float dsk = Dskip[c];
    const float db = dbias[c];

    const int nchunks = (L + CH - 1) / CH;
    float ub[CH], dbuf[CH], zb[CH], yb[CH];
    auto tof = [&](int step) { return REV ? (L - 1 - step) : step; };
    auto load_chunk = [&](int ci, float (&uu)[CH], float (&dd)[CH], float (&zz)[CH], float (&yy)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            int step = ci * CH + i;
            if (step > L - 1) step = L - 1;
            const int64_t r = row0 + tof(step);
            uu[i] = Elem<T>::load(u + r * E + c);
            dd[i] = Elem<T>::load(delta + r * E + c);
            if constexpr (HASZ) zz[i] = Elem<T>::load(z + r * ldz + c);
            if constexpr (ACC) yy[i] = Elem<T>::load(yin + r * E + c);
        }
    };
    load_chunk(0, ub, dbuf, zb, yb);
    for (int ci = 0; ci < nchunks; ++ci) {
        float un[CH], dn[CH], zn[CH], yn[CH];
        if (ci + 1 < nchunks) load_chunk(ci + 1, un, dn, zn, yn);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int step = ci * CH + i;
            if (step < L) {
                const int64_t r = row0 + tof(step);
                float bv[NSTATE], cv[NSTATE];
                load_state16<T>(Bm + r * ldbc, bv);
                load_state16<T>(Cm + r * ldbc, cv);
                const float dv = softplus(dbuf[i] + db);
                const float uv = ub[i];
                const float du = dv * uv;
                float yv = dsk * uv;
#pragma unroll
                for (int n = 0; n < NSTATE; ++n) {
                    const float a = fast_exp2(dv * a2[n]);
                    h[n] = a * h[n] + du * bv[n];
                    yv += h[n] * cv[n];
                }
                if constexpr (HASZ) yv *= silu(zb[i]);
                yv = Elem<T>::round(yv);
                if constexpr (ACC) yv += yb[i];
                Elem<T>::store(y + r * E + c, yv);
            }
        }
        if (ci + 1 < nchunks) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { ub[i] = un[i]; dbuf[i] = dn[i]; zb[i] = zn[i]; yb[i] = yn[i]; }
        }
    }
}

template <typename T>
static hipError_t launch_scan_t(const void* u, const void* delta, const void* z, int64_t ldz, const void* Bm,
                                const void* Cm, int64_t ldbc, const float* A2, float a_scale, const float* Dskip,
                                const float* dbias, void* y, int S, int L, int E, bool reverse, bool accumulate,
                                hipStream_t s) {
    dim3 grid((unsigned)(E / 64), (unsigned)S), block(64);
#define PCAD_SCAN(REV, ACC, HZ)                                                                                   \
    hipLaunchKernelGGL((scan_kernel<T, REV, ACC, HZ>), grid, block, 0, s, (const T*)u, (const T*)delta, (const T*)z, \
                       ldz, (const T*)Bm, (const T*)Cm, ldbc, A2, a_scale, Dskip, dbias, (const T*)y, (T*)y, L, E)
    const bool hz = z != nullptr;
    if (!reverse && !accumulate) { if (hz) PCAD_SCAN(false, false, true); else PCAD_SCAN(false, false, false); }
    else if (!reverse && accumulate) { if (hz) PCAD_SCAN(false, true, true); else PCAD_SCAN(false, true, false); }
    else if (reverse && !accumulate) { if (hz) PCAD_SCAN(true, false, true); else PCAD_SCAN(true, false, false); }
    else { if (hz) PCAD_SCAN(true, true, true); else PCAD_SCAN(true, true, false); }
#undef PCAD_SCAN
    return hipGetLastError();
}

hipError_t launch_scan(const void* u, const void* delta, const void* z, int64_t ldz, const void* Bm, const void* Cm,
                       int64_t ldbc, const float* A2, float a_scale, const float* Dskip, const float* dbias, void* y, int S,
                       int L, int E, bool reverse, bool accumulate, int dt, hipStream_t s) {
    if (S <= 0 || L <= 0) return hipSuccess;
    if (E % 64) return hipErrorInvalidValue;
    if (dt == BF16)
        return launch_scan_t<bf16_t>(u, delta, z, ldz, Bm, Cm, ldbc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s);
    return launch_scan_t<float>(u, delta, z, ldz, Bm, Cm, ldbc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s);
}

}  // namespace pcad
