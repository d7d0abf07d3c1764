// NT GEMM on the gfx950 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T   (F.linear without bias).
//
// Replaces the cuBLAS GEMMs behind in_proj / x_proj / dt_proj / out_proj of mamba_ssm.Mamba
// (shapes from reference notebooks/examples.ipynb:73-80).  Both operands are
// K-contiguous (token-major activations x nn.Linear weights), which is the natural MFMA layout.
//
// Design (wave64, CDNA4):
//   * 128x128 block tile, 4 waves as 2(M) x 2(N), each wave 64x64 = 4x4 MFMA 16x16 tiles.
//   * K tile = 128 BYTES per row for every dtype (64 bf16 / 32 fp32) so the LDS image, the staging code and
//     the ds_read_b128 fragment reads are byte-identical; only the MFMA differs:
//       bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte fragment pair),
//       fp32: 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the same fragments (exact fp32, k permuted
//             identically on both operands).
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per K tile.
//     The LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle (16-byte chunk index XOR a
//     row key) is applied to the per-lane SOURCE address and again on the fragment read.
//   * operands swapped (W is the MFMA "A" operand) with W rows permuted inside the wave tile so that each
//     lane ends up holding 16 CONSECUTIVE output columns of one output row: 32/64-byte vector stores.
//   * blockIdx -> tile map is XCD-aware: each XCD walks a contiguous range of tiles, m-fastest inside
//     groups of 8 m-panels, so the A panels and the current W rows stay in that XCD's 4 MiB L2.
#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, ROWB = 128;          // ROWB: bytes of K per tile row
constexpr int TILE_BYTES = BM * ROWB;                  // 16 KiB per operand tile
constexpr int GEMM_LDS = 2 * 2 * TILE_BYTES;           // 64 KiB: 2 buffers x (A, W)
constexpr int GROUP_M = 8;

__device__ __forceinline__ int key_a(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int key_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const u32x4& w, const u32x4& a, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w),
                                                       __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(const u32x4& w, const u32x4& a, f32x4 c) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[s]), __uint_as_float(a[s]), c, 0, 0, 0);
        return c;
    }
};

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, typename OutT, bool ROUND, bool VEC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const T* __restrict__ A, int64_t lda,
                                                         const T* __restrict__ W, int64_t ldw,
                                                         OutT* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                         int tiles_m, int tiles_n, float* __restrict__ C2,
                                                         int64_t ldc2, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile map -------------------------------------------------------------------
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gsz = GROUP_M * tiles_n;
    const int g = logical / gsz;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int in_g = logical - g * gsz;
    const int tm = first_m + in_g % gm;
    const int tn = in_g / gm;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    // ---- staging addresses: wave stages rows [wave*32 + i*8, +8) of both tiles --------------------
    const char* pa[4];
    const char* pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int ca = (lane & 7) ^ key_a(row);
        const int cw = (lane & 7) ^ key_w(row);
        int64_t ga = m0 + row; if (ga > M - 1) ga = M - 1;
        int gw = n0 + row; if (gw > N - 1) gw = N - 1;
        pa[i] = reinterpret_cast<const char*>(A + ga * lda) + ca * 16;
        pw[i] = reinterpret_cast<const char*>(W + (int64_t)gw * ldw) + cw * 16;
    }
    const int nkt = (K * (int)sizeof(T)) / ROWB;

    auto stage = [&](int kt, int buf) {
        char* as = smem + buf * (2 * TILE_BYTES) + (wave * 32) * ROWB;
        char* ws = as + TILE_BYTES;
        const int64_t ko = (int64_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(pa[i] + ko, as + i * 8 * ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(pw[i] + ko, ws + i * 8 * ROWB);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    // fragment row offsets (bytes) and swizzle keys
    int a_off[4], a_key[4], w_off[4], w_key[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + li;
        a_off[i] = ra * ROWB; a_key[i] = key_a(ra);
        const int rw = wn * 64 + (li >> 2) * 16 + i * 4 + (li & 3);
        w_off[i] = rw * ROWB; w_key[i] = key_w(rw);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
        const char* as = smem + buf * (2 * TILE_BYTES);
        const char* ws = as + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + lg;
            u32x4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const u32x4*>(as + a_off[i] + ((chunk ^ a_key[i]) << 4));
                wf[i] = *reinterpret_cast<const u32x4*>(ws + w_off[i] + ((chunk ^ w_key[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(wf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: lane holds, per mi, 16 consecutive columns n = nb .. nb+15 of row m ------------
    const int nb = n0 + wn * 64 + lg * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + li;
        if (m >= M) continue;
        float o[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = acc[i][j][rr];
                if constexpr (ROUND) v = round_to_bf16(v);
                o[j * 4 + rr] = v;
            }
        if constexpr (SPLIT) {
            if (nb >= nsplit) {   // fp32 side output (x_proj: B_t | C_t rows for the scan's scalar loads)
                if (nb + 16 <= N) {
                    float* d2 = C2 + m * ldc2 + (nb - nsplit);
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        f32x4 v = {Elem<T>::round(o[e]), Elem<T>::round(o[e + 1]), Elem<T>::round(o[e + 2]),
                                   Elem<T>::round(o[e + 3])};
                        *reinterpret_cast<f32x4*>(d2 + e) = v;
                    }
                }
                continue;
            }
        }
        OutT* dst = C + m * ldc + nb;
        const int Nmain = SPLIT ? nsplit : N;
        if (VEC && nb + 16 <= Nmain) {
            float lo[8], hi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = o[e]; hi[e] = o[8 + e]; }
            store8<OutT>(dst, lo);
            store8<OutT>(dst + 8, hi);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (nb + e < Nmain) Elem<OutT>::store(dst + e, o[e]);
        }
    }
}

template <typename T, typename OutT, bool ROUND>
static hipError_t launch_gemm_t(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                int64_t M, int N, int K, hipStream_t s) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    const bool vec = ((ldc * (int64_t)sizeof(OutT)) % 16 == 0) && (((uintptr_t)C) % 16 == 0);
    dim3 grid((unsigned)(tiles_m * tiles_n)), block(256);
    static bool attr_done_v = false, attr_done_s = false;
    if (vec) {
        auto kfn = gemm_nt_kernel<T, OutT, ROUND, true, false>;
        if (!attr_done_v) {
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            attr_done_v = true;
        }
        hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K,
                           tiles_m, tiles_n, (float*)nullptr, (int64_t)0, 0);
    } else {
        auto kfn = gemm_nt_kernel<T, OutT, ROUND, false, false>;
        if (!attr_done_s) {
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            attr_done_s = true;
        }
        hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K,
                           tiles_m, tiles_n, (float*)nullptr, (int64_t)0, 0);
    }
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_gemm_split_t(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                      float* C2, int64_t ldc2, int nsplit, int64_t M, int N, int K, hipStream_t s) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    dim3 grid((unsigned)(tiles_m * tiles_n)), block(256);
    auto kfn = gemm_nt_kernel<T, T, false, true, true>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (T*)C, ldc, M, N, K, tiles_m,
                       tiles_n, C2, ldc2, nsplit);
    return hipGetLastError();
}

hipError_t launch_gemm_nt_split(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, float* C2,
                                int64_t ldc2, int nsplit, int64_t M, int N, int K, int dt, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB || nsplit % 16 || (N - nsplit) % 16) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16) return hipErrorInvalidValue;
    if ((ldc * esz) % 16 || ((uintptr_t)C) % 16 || (ldc2 * 4) % 16 || ((uintptr_t)C2) % 16) return hipErrorInvalidValue;
    if (dt == BF16) return launch_gemm_split_t<bf16_t>(A, lda, W, ldw, C, ldc, C2, ldc2, nsplit, M, N, K, s);
    return launch_gemm_split_t<float>(A, lda, W, ldw, C, ldc, C2, ldc2, nsplit, M, N, K, s);
}

hipError_t launch_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                          int N, int K, int dt, int out_dt, bool round_bf16, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16)
        return hipErrorInvalidValue;
    if (dt == BF16 && out_dt == BF16)
        return launch_gemm_t<bf16_t, bf16_t, false>(A, lda, W, ldw, C, ldc, M, N, K, s);
    if (dt == BF16 && out_dt == F32)
        return round_bf16 ? launch_gemm_t<bf16_t, float, true>(A, lda, W, ldw, C, ldc, M, N, K, s)
                          : launch_gemm_t<bf16_t, float, false>(A, lda, W, ldw, C, ldc, M, N, K, s);
    if (dt == F32 && out_dt == F32)
        return launch_gemm_t<float, float, false>(A, lda, W, ldw, C, ldc, M, N, K, s);
    return hipErrorInvalidValue;
}

}  // namespace pcad
