// Developer tool (run on the GPU box): accuracy and issue cost of the two softplus formulations in csrc/common.hpp
// (select form `softplus`, packed select-free form `softplus2`) against a double-precision host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plantcaduceus_amd/csrc -o /tmp/softplus_test tools/softplus_test.hip && /tmp/softplus_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "common.hpp"
using namespace pcad;

__global__ void eval(const float* x, float* a, float* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    a[2 * i] = softplus(x[2 * i]);
    a[2 * i + 1] = softplus(x[2 * i + 1]);
    const f2_t r = softplus2(f2_t{x[2 * i], x[2 * i + 1]});
    b[2 * i] = r[0];
    b[2 * i + 1] = r[1];
}

template <int MODE> __global__ __launch_bounds__(256) void cost(float* out, int iters, float seed) {
    f2_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f2_t{seed * (i + 1) - 3.f + threadIdx.x * 1e-3f, seed * i - 2.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) v[i] = f2_t{softplus(v[i][0]) - 1.f, softplus(v[i][1]) - 1.f};
            else v[i] = softplus2(v[i]) - f2_t{1.f, 1.f};
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v[0][0] + v[1][1] + v[2][0] + v[3][1];
}

int main() {
    std::vector<float> xs;
    for (double x = -40.0; x <= 40.0; x += 1e-3) xs.push_back((float)x);
    for (int k = 0; k < 200000; ++k) xs.push_back(-20.f + 45.f * (float)((k * 2654435761u) % 1000003) / 1000003.f);
    for (float x : {-87.f, -88.5f, -100.f, -1000.f, 20.f, 20.0001f, 19.9999f, 50.f, 88.f, 89.f, 100.f, 1000.f, 1e6f, 0.f, -0.f, 1e-8f, -1e-8f})
        xs.push_back(x);
    if (xs.size() & 1) xs.push_back(0.f);
    const int n = (int)xs.size();
    float *dx, *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    eval<<<(n / 2 + 255) / 256, 256>>>(dx, da, db, n);
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0; float xa = 0, xb = 0; int nonfinite = 0, thr_mismatch = 0;
    for (int i = 0; i < n; ++i) {
        const double x = xs[i];
        const double ref = x > 20.0 ? x : std::log1p(std::exp(x));
        if (!std::isfinite(b[i]) || !std::isfinite(a[i])) { ++nonfinite; continue; }
        if (x > 20.0 && b[i] != xs[i]) ++thr_mismatch;
        if (ref < 1e-37) continue;                       // below the fp32 normal range: both flush
        const double ra = std::fabs(a[i] - ref) / ref, rb = std::fabs(b[i] - ref) / ref;
        if (ra > ea) { ea = ra; xa = xs[i]; }
        if (rb > eb) { eb = rb; xb = xs[i]; }
    }
    printf("points %d  nonfinite %d  x>20 results != x (packed form): %d\n", n, nonfinite, thr_mismatch);
    printf("max relative error vs double: select form %.3e (at x = %g)   packed form %.3e (at x = %g)\n", ea, xa, eb, xb);
    float* out; hipMalloc(&out, 4 * 256 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 20000;
        if (mode == 0) cost<0><<<1024, 256>>>(out, 100, 1.f); else cost<1><<<1024, 256>>>(out, 100, 1.f);
        hipEventRecord(e0);
        if (mode == 0) cost<0><<<1024, 256>>>(out, iters, 1.f); else cost<1><<<1024, 256>>>(out, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 4 waves per SIMD, 8 values per lane per iteration
        printf("%s: %.3f ms -> %.1f SIMD-cycles@2.4GHz per value per wave\n", mode ? "packed form" : "select form", ms,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8) / 4);
    }
    return 0;
}
