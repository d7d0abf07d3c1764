// Micro-benchmark (developer tool, not part of the product): can an MFMA-bound kernel and a VALU/transcendental-bound kernel
// share the CUs of an MI355X and both make progress, or do they only time-slice?  Upper bound for running the selective scan
// (VALU) of one chunk beside the in_proj/out_proj GEMMs (MFMA) of another.
//
//   kernel M: persistent, 256 blocks x 4 waves (one wave per SIMD), register-resident v_mfma_f32_16x16x32_bf16 on random
//             operands, `duty` MFMAs out of every 16 issue slots replaced by nothing (s_nop) to emulate a GEMM whose matrix
//             pipe is busy ~60 % of the time.  ~100 VGPRs, no LDS: leaves room for kernel V on the same SIMDs.
//   kernel V: the scan's inner loop shape, per step 16 v_exp_f32 + 16 v_pk_mul_f32 + 16 v_pk_fma_f32 on registers, 64-thread
//             blocks, launch_bounds(64, 4) (<=128 VGPRs), W waves per SIMD.
// Timed alone and together on two streams (HIP events on each stream + host wall clock around both).
//   hipcc --offload-arch=gfx950 -O3 -o coreside_microbench tools/coreside_microbench.hip && ./coreside_microbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int GAP>   // GAP: s_nop states between MFMAs (0 = back to back)
__global__ __launch_bounds__(256, 1) void mfma_kernel(const u32x4* __restrict__ src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane * 3 + 64 * i + 7) & 1023]; }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i & 3]),
                                                           __builtin_bit_cast(bf16x8_t, b[i >> 2]), acc[i], 0, 0, 0);
            if constexpr (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float s_exp(float a) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(a)); return r; }

// same recurrence with SCALAR fp32 instructions only (v_mul_f32 / v_fma_f32 / v_exp_f32 through inline asm, so the
// compiler cannot re-pack them): MODE 1 full step, MODE 2 the 16 exps only, MODE 3 the 64 plain ops only
template <int LDS_FLOATS, int MODE, int PRIO>
__global__ __launch_bounds__(64, 4) void valu_scalar_kernel(const float* __restrict__ src, float* out, int steps) {
    __shared__ float pad[LDS_FLOATS];
    const int gid = blockIdx.x * 64 + threadIdx.x;
    pad[threadIdx.x] = src[gid & 1023];
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    float a2[16], h[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) { a2[n] = -0.01f * (n + 1) - 1e-4f * src[gid & 1023]; h[n] = 0.f; }
    float y = 0.f;
    float dv = 0.3f + 1e-3f * src[(gid + 5) & 1023];
    for (int s = 0; s < steps; ++s) {
        const float uv = 0.5f + 1e-4f * (float)(s & 7);
        const float du = dv * uv;
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int n = 0; n < 16; n += 2) {
            const float bb = 0.25f + 0.01f * n, cc = 0.75f - 0.01f * n;
            float x0, x1;
            if (MODE == 3) { x0 = a2[n]; x1 = a2[n + 1]; }
            else { x0 = s_exp(MODE == 2 ? a2[n] + dv : s_mul(dv, a2[n])); x1 = s_exp(MODE == 2 ? a2[n + 1] + dv : s_mul(dv, a2[n + 1])); }
            if (MODE == 2) { y0 += x0; y1 += x1; continue; }
            if (MODE == 3) { x0 = s_mul(dv, x0); x1 = s_mul(dv, x1); }
            h[n] = s_fma(x0, h[n], s_mul(du, bb));
            h[n + 1] = s_fma(x1, h[n + 1], s_mul(du, cc));
            y0 = s_fma(h[n], cc, y0);
            y1 = s_fma(h[n + 1], bb, y1);
        }
        y += y0 + y1;
        dv = 0.3f + 1e-6f * y;
    }
    out[gid] = y + pad[(threadIdx.x + 1) & 63];
}

template <int LDS_FLOATS>   // static LDS per 1-wave block sets the occupancy: 2560 floats (10 KiB) -> 16 blocks/CU = 4 waves/SIMD
__global__ __launch_bounds__(64, 4) void valu_kernel(const float* __restrict__ src, float* out, int steps) {
    __shared__ float pad[LDS_FLOATS];
    const int gid = blockIdx.x * 64 + threadIdx.x;
    pad[threadIdx.x] = src[gid & 1023];
    f2 a2[8], h[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        a2[p] = f2{-0.01f * (2 * p + 1) - 1e-4f * src[gid & 1023], -0.01f * (2 * p + 2)};
        h[p] = f2{0.f, 0.f};
    }
    float y = 0.f;
    float dv = 0.3f + 1e-3f * src[(gid + 5) & 1023];
    for (int s = 0; s < steps; ++s) {
        const float uv = 0.5f + 1e-4f * (float)(s & 7);
        const f2 dv2 = {dv, dv}, du2 = {dv * uv, dv * uv};
        f2 y0 = {0.f, 0.f}, y1 = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 8; p += 2) {
            const f2 e0 = dv2 * a2[p], e1 = dv2 * a2[p + 1];
            const f2 x0 = {__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])};
            const f2 x1 = {__builtin_amdgcn_exp2f(e1[0]), __builtin_amdgcn_exp2f(e1[1])};
            const f2 bb = {0.25f + 0.01f * p, 0.5f - 0.01f * p}, cc = {0.75f - 0.01f * p, 0.1f + 0.02f * p};
            h[p] = x0 * h[p] + du2 * bb;
            h[p + 1] = x1 * h[p + 1] + du2 * cc;
            y0 = h[p] * cc + y0;
            y1 = h[p + 1] * bb + y1;
        }
        y += y0[0] + y0[1] + y1[0] + y1[1];
        dv = 0.3f + 1e-6f * y;      // loop-carried so nothing is hoisted
    }
    out[gid] = y + pad[(threadIdx.x + 1) & 63];
}

// ONE kernel, wave-specialised: waves 0-3 of each 16-wave block (one per SIMD) run the MFMA loop, waves 4-15 (three per SIMD)
// the scalar VALU recurrence.  A single dispatch, so rocprofv3 --pmc sees both pipes in the same pass:
//   tools/coreside_microbench fused        (three dispatches per GAP: M only, V only, both)
template <int GAP>
__global__ __launch_bounds__(1024) void fused_kernel(const u32x4* __restrict__ src, float* out, int iters_m, int steps_v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 1024 + threadIdx.x;
    if (wave < 4) {
        if (iters_m == 0) return;
        u32x4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane * 3 + 64 * i + 7) & 1023]; }
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i & 3]),
                                                               __builtin_bit_cast(bf16x8_t, b[i >> 2]), acc[i], 0, 0, 0);
                if constexpr (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[gid] = s;
    } else {
        if (steps_v == 0) return;
        const float* fs = (const float*)src;
        float a2[16], h[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) { a2[n] = -0.01f * (n + 1) - 1e-4f * fs[gid & 1023]; h[n] = 0.f; }
        float y = 0.f;
        float dv = 0.3f + 1e-3f * fs[(gid + 5) & 1023];
        for (int s = 0; s < steps_v; ++s) {
            const float uv = 0.5f + 1e-4f * (float)(s & 7);
            const float du = dv * uv;
            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int n = 0; n < 16; n += 2) {
                const float bb = 0.25f + 0.01f * n, cc = 0.75f - 0.01f * n;
                const float x0 = s_exp(s_mul(dv, a2[n])), x1 = s_exp(s_mul(dv, a2[n + 1]));
                h[n] = s_fma(x0, h[n], s_mul(du, bb));
                h[n + 1] = s_fma(x1, h[n + 1], s_mul(du, cc));
                y0 = s_fma(h[n], cc, y0);
                y1 = s_fma(h[n + 1], bb, y1);
            }
            y += y0 + y1;
            dv = 0.3f + 1e-6f * y;
        }
        out[gid] = y;
    }
}

template <int GAP>
void fused_experiment(const char* name, const u32x4* src, float* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&](int im, int sv) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fused_kernel<GAP>, dim3(256), dim3(1024), 0, 0, src, out, im, sv);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    const int sv = 16384;                 // 3 V waves per SIMD x 16384 steps
    go(1000, 100);
    const float tv = go(0, sv);
    int im = 20000;
    float tm = go(im, 0);
    im = (int)(im * tv / tm);
    tm = go(im, 0);
    const float tb = go(im, sv);
    printf("fused %-20s  M only %.2f ms (%.0f TF)   V only %.2f ms   both in one dispatch %.2f ms   serial sum %.2f -> speedup %.2fx\n", name, tm,
           256.0 * 4 * (double)im * 16 * 2 * 16 * 16 * 32 / (tm * 1e-3) / 1e12, tv, tb, tm + tv, (tm + tv) / tb);
}

struct Timing { float ms_m, ms_v; double wall_ms; };

template <int GAP, int LF, int MODE = 0, int PRIO = 0>
Timing run(bool do_m, bool do_v, int iters_m, int v_blocks, int v_steps, const u32x4* src, float* out_m, float* out_v,
           hipStream_t sm, hipStream_t sv) {
    hipEvent_t m0, m1, v0, v1;
    CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    if (do_m) { CK(hipEventRecord(m0, sm)); hipLaunchKernelGGL(mfma_kernel<GAP>, dim3(256), dim3(256), 0, sm, src, out_m, iters_m); CK(hipEventRecord(m1, sm)); }
    if (do_v) { CK(hipEventRecord(v0, sv));
        if (MODE == 0) hipLaunchKernelGGL(valu_kernel<LF>, dim3(v_blocks), dim3(64), 0, sv, (const float*)src, out_v, v_steps);
        else hipLaunchKernelGGL((valu_scalar_kernel<LF, MODE, PRIO>), dim3(v_blocks), dim3(64), 0, sv, (const float*)src, out_v, v_steps); CK(hipEventRecord(v1, sv)); }
    CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    Timing t{0.f, 0.f, std::chrono::duration<double, std::milli>(t1 - t0).count()};
    if (do_m) CK(hipEventElapsedTime(&t.ms_m, m0, m1));
    if (do_v) CK(hipEventElapsedTime(&t.ms_v, v0, v1));
    return t;
}

template <int GAP, int LF, int MODE = 0, int PRIO = 0>
void experiment(const char* name, const u32x4* src, float* out_m, float* out_v, hipStream_t sm, hipStream_t sv) {
    const int v_steps = 512;
    // V alone at 4 waves/SIMD: 16 rounds of 4096 blocks (like a 1024-strand scan launch at l32: 65536 blocks)
    const int v_blocks = 65536;
    run<GAP, LF, MODE, PRIO>(false, true, 0, v_blocks, v_steps, src, out_m, out_v, sm, sv);                    // warm
    Timing tv = run<GAP, LF, MODE, PRIO>(false, true, 0, v_blocks, v_steps, src, out_m, out_v, sm, sv);
    // M alone, iterations sized to about the same duration
    int iters_m = 20000;
    run<GAP, LF, MODE, PRIO>(true, false, iters_m, 0, 0, src, out_m, out_v, sm, sv);
    Timing tm = run<GAP, LF, MODE, PRIO>(true, false, iters_m, 0, 0, src, out_m, out_v, sm, sv);
    iters_m = (int)(iters_m * tv.ms_v / tm.ms_m);
    tm = run<GAP, LF, MODE, PRIO>(true, false, iters_m, 0, 0, src, out_m, out_v, sm, sv);
    const double mfma_rate = 256.0 * 4 * (double)iters_m * 16 * 2 * 16 * 16 * 32 / (tm.ms_m * 1e-3) / 1e12;
    Timing tb = run<GAP, LF, MODE, PRIO>(true, true, iters_m, v_blocks, v_steps, src, out_m, out_v, sm, sv);
    tb = run<GAP, LF, MODE, PRIO>(true, true, iters_m, v_blocks, v_steps, src, out_m, out_v, sm, sv);
    printf("%-22s  M alone %.2f ms (%.0f TF)   V alone %.2f ms   together: M %.2f ms, V %.2f ms, wall %.2f ms   serial sum %.2f ms  -> speedup %.2fx\n",
           name, tm.ms_m, mfma_rate, tv.ms_v, tb.ms_m, tb.ms_v, tb.wall_ms, tm.ms_m + tv.ms_v,
           (tm.ms_m + tv.ms_v) / (tb.ms_m > tb.ms_v ? tb.ms_m : tb.ms_v));
}

int main(int argc, char** argv) {
    std::vector<unsigned> h(4096);
    srand(1);
    for (auto& v : h) {   // two random bf16 values in [-1, 1) per word
        auto bf = [&]() { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); return u >> 16; };
        v = bf() | (bf() << 16);
    }
    u32x4* src; float *out_m, *out_v;
    CK(hipMalloc(&src, 16384)); CK(hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice));
    CK(hipMalloc(&out_m, 256 * 256 * 4)); CK(hipMalloc(&out_v, 65536 * 64 * 4));
    hipStream_t sm, sv;
    CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    if (argc > 1 && !strcmp(argv[1], "fused")) {
        float* out;
        CK(hipMalloc(&out, 256 * 1024 * 4));
        fused_experiment<0>("MFMA back-to-back", src, out);
        fused_experiment<2>("MFMA + 2 nop states", src, out);
        fused_experiment<4>("MFMA + 4 nop states", src, out);
        fused_experiment<8>("MFMA + 8 nop states", src, out);
        return 0;
    }
    printf("== V packed (v_pk_mul/v_pk_fma_f32 + v_exp_f32), 4 waves/SIMD\n");
    experiment<0, 2560>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<2, 2560>("MFMA + 2 nop states", src, out_m, out_v, sm, sv);
    experiment<4, 2560>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    experiment<8, 2560>("MFMA + 8 nop states", src, out_m, out_v, sm, sv);
    printf("== V scalar (v_mul/v_fma_f32 + v_exp_f32), 4 waves/SIMD\n");
    experiment<0, 2560, 1>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<2, 2560, 1>("MFMA + 2 nop states", src, out_m, out_v, sm, sv);
    experiment<4, 2560, 1>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    experiment<8, 2560, 1>("MFMA + 8 nop states", src, out_m, out_v, sm, sv);
    printf("== V scalar, s_setprio 3, 4 waves/SIMD\n");
    experiment<0, 2560, 1, 3>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<4, 2560, 1, 3>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    printf("== V = 16 v_exp_f32 only, 4 waves/SIMD\n");
    experiment<0, 2560, 2>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<4, 2560, 2>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    printf("== V = 64 scalar v_mul/v_fma only, 4 waves/SIMD\n");
    experiment<0, 2560, 3>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<4, 2560, 3>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    printf("== V scalar, 2 waves/SIMD\n");
    experiment<0, 5120, 1>("MFMA back-to-back", src, out_m, out_v, sm, sv);
    experiment<4, 5120, 1>("MFMA + 4 nop states", src, out_m, out_v, sm, sv);
    return 0;
}
