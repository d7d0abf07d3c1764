// Developer tool (GPU box): rate of the GEMM epilogue's store patterns, nothing else running.  One 256-thread block per CU writes
// 256x256 bf16 output tiles of a [rows, N] tensor (plain rows or the engine's blocked layout) with 32 global_store_dwordx4 per wave
// and tile, in three lane->address mappings:
//   0  the epilogue's: lane (li = row of 16, lg = 16-column group) stores 16 B at columns lg*16 + {0, 8}: every instruction
//      writes four separate 16-byte runs in each of 16 rows (half of each 128-byte line, interleaved)
//   1  after one v_permlane32_swap per register pair: each instruction writes one contiguous 64-byte half line in each of 16 rows
//   2  full lines: each instruction writes 8 rows x 128 contiguous bytes
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/spm tools/store_pattern_microbench.hip && /tmp/spm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t blocked_off(int64_t r, int64_t cb, int64_t pieces) {
    return (((r >> 3) * pieces + (cb >> 7)) << 10) + ((r & 7) << 7) + (cb & 127);
}

template <int PAT, bool BLOCKED>
__global__ __launch_bounds__(256) void k(char* out, int64_t rows, int N, int tiles_n, int ntiles, uint32_t seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t rowb = (int64_t)N * 2, pieces = rowb >> 7;
    u32x4 v = {seed + lane, seed * 3u + lane, seed ^ lane, seed + 7u};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t m0 = (int64_t)(t / tiles_n) * 256;
        const int n0 = (t % tiles_n) * 256;
#pragma unroll
        for (int jg = 0; jg < 2; ++jg)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int64_t row; int64_t cb;                         // row, byte offset in the row
                    const int64_t cbase = (int64_t)(n0 + wn * 128 + jg * 64) * 2;      // 128-byte group of this (wave, jg)
                    if (PAT == 0) { row = m0 + wm * 128 + i * 16 + (lane & 15); cb = cbase + (lane >> 4) * 32 + h * 16; }
                    else if (PAT == 1) { row = m0 + wm * 128 + i * 16 + (lane & 15); cb = cbase + h * 64 + (lane >> 4) * 16; }
                    else { row = m0 + wm * 128 + i * 16 + h * 8 + (lane >> 3); cb = cbase + (lane & 7) * 16; }
                    char* dst = out + (BLOCKED ? blocked_off(row, cb, pieces) : row * rowb + cb);
                    *reinterpret_cast<u32x4*>(dst) = v;
                    v[0] += 1u;
                }
    }
}

template <int PAT, bool BLOCKED> void run(char* buf, int64_t rows, int N) {
    const int tiles_n = N / 256, ntiles = (int)(rows / 256) * tiles_n;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PAT, BLOCKED><<<256, 256>>>(buf, rows, N, tiles_n, ntiles, 1u);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) k<PAT, BLOCKED><<<256, 256>>>(buf, rows, N, tiles_n, ntiles, 2u + r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double bytes = (double)rows * N * 2;
    printf("pattern %d %-7s N=%d: %.3f ms for %.2f GB -> %.2f TB/s = %.1f B/cycle/CU at 2.4 GHz; %.2f us per 256x256 tile per CU\n", PAT,
           BLOCKED ? "blocked" : "plain", N, ms, bytes / 1e9, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e3 / (rows / 256 * (N / 256) / 256.0));
}

int main() {
    const int64_t rows = 262144;
    char* buf; hipMalloc(&buf, rows * 4096 * 2 + (1 << 20));
    run<0, true>(buf, rows, 2048); run<1, true>(buf, rows, 2048); run<2, true>(buf, rows, 2048);      // in_proj's x / z outputs (blocked, E = 2048)
    run<0, false>(buf, rows, 1024); run<1, false>(buf, rows, 1024); run<2, false>(buf, rows, 1024);   // out_proj's h (plain rows, D = 1024)
    return 0;
}
