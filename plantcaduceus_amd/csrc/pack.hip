// Weight packing (run once at bind time): dtype conversion, zero padding, pre-exponentiated A.
#include <mutex>
#include <set>
#include <utility>

#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void pack2d_kernel(const SrcT* __restrict__ src, int64_t src_ld,
                                                     DstT* __restrict__ dst, int64_t dst_ld, int rows, int cols,
                                                     int dst_rows, int dst_cols) {
    const int64_t total = (int64_t)dst_rows * dst_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / dst_cols), c = (int)(i - (int64_t)r * dst_cols);
        float v = 0.f;
        if (r < rows && c < cols) v = Elem<SrcT>::load(src + (int64_t)r * src_ld + c);
        Elem<DstT>::store(dst + (int64_t)r * dst_ld + c, v);
    }
}

hipError_t launch_pack2d(const void* src, int src_dt, int64_t src_ld, void* dst, int dst_dt, int64_t dst_ld, int rows,
                         int cols, int dst_rows, int dst_cols, hipStream_t s) {
    const int64_t total = (int64_t)dst_rows * dst_cols;
    if (total <= 0) return hipSuccess;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    dim3 grid((unsigned)nb), block(256);
#define PCAD_PACK(ST, DT)                                                                                      \
    hipLaunchKernelGGL((pack2d_kernel<ST, DT>), grid, block, 0, s, (const ST*)src, src_ld, (DT*)dst, dst_ld, rows, \
                       cols, dst_rows, dst_cols)
    if (src_dt == F32 && dst_dt == F32) PCAD_PACK(float, float);
    else if (src_dt == F32 && dst_dt == BF16) PCAD_PACK(float, bf16_t);
    else if (src_dt == BF16 && dst_dt == F32) PCAD_PACK(bf16_t, float);
    else if (src_dt == BF16 && dst_dt == BF16) PCAD_PACK(bf16_t, bf16_t);
    else return hipErrorInvalidValue;
#undef PCAD_PACK
    return hipGetLastError();
}

// A = -exp(A_log.float())  (mamba_ssm.Mamba.forward), then pre-scaled by log2(e) so that the scan's
// exp(delta * A) is a single v_exp_f32 (2^x) — the same folding selective_scan_fwd_kernel.cuh does.
template <typename SrcT>
__global__ __launch_bounds__(256) void pack_A_kernel(const SrcT* __restrict__ A_log, float* __restrict__ A2, int64_t n,
                                                     float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A2[i] = -expf(Elem<SrcT>::load(A_log + i)) * scale;
}

template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void pack_scale_cols_kernel(const SrcT* __restrict__ src, int64_t src_ld, const float* __restrict__ scale,
                                                              DstT* __restrict__ dst, int64_t dst_ld, int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        Elem<DstT>::store(dst + (int64_t)r * dst_ld + c, Elem<SrcT>::load(src + (int64_t)r * src_ld + c) * scale[c]);
    }
}

hipError_t launch_pack_scale_cols(const void* src, int src_dt, int64_t src_ld, const float* scale, void* dst, int dst_dt,
                                  int64_t dst_ld, int rows, int cols, hipStream_t s) {
    const int64_t total = (int64_t)rows * cols;
    if (total <= 0) return hipSuccess;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    dim3 grid((unsigned)nb), block(256);
#define PCAD_PSC(ST, DT) hipLaunchKernelGGL((pack_scale_cols_kernel<ST, DT>), grid, block, 0, s, (const ST*)src, src_ld, scale, (DT*)dst, dst_ld, rows, cols)
    if (src_dt == F32 && dst_dt == F32) PCAD_PSC(float, float);
    else if (src_dt == F32 && dst_dt == BF16) PCAD_PSC(float, bf16_t);
    else if (src_dt == BF16 && dst_dt == F32) PCAD_PSC(bf16_t, float);
    else PCAD_PSC(bf16_t, bf16_t);
#undef PCAD_PSC
    return hipGetLastError();
}

// ---- split-bf16 operands of the fp32 model's big GEMMs (api.hip "f32_gemm_split") ------------------------------------------------
// An fp32 value v is carried as two bf16 values hi = bf16(v), lo = bf16(v - hi) (v = hi + lo up to 2^-17 |v|), and
//     a . w  ~  a_hi w_hi + a_lo w_hi + a_hi w_lo          (the dropped a_lo w_lo term is 2^-16 relative)
// runs as ONE bf16 GEMM of 3 K / 64 K-tiles on operands stored once as A' = [a_hi | a_lo], W' = [w_hi | w_lo]: the GEMM's K-tile
// cursor wraps around (A: hi, lo, hi; W: hi, hi, lo - gemm.hip), fp32 accumulation in the MFMA, fp32 result - three
// v_mfma_f32_16x16x32_bf16 products (3/16 of the fp32-MFMA cost per flop).  (Round 5 stored the concatenations [hi | lo | hi] and
// [hi | hi | lo]: a third more bytes written by every producer and 1.5x the operand footprint.)
// Weights (bind time): src [rows, cols] fp32 or bf16 -> dst [rows, 2 cols] bf16 = [hi | lo].
template <typename SrcT>
__global__ __launch_bounds__(256) void pack_split_w_kernel(const SrcT* __restrict__ src, int64_t src_ld, bf16_t* __restrict__ dst,
                                                           int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        const float v = Elem<SrcT>::load(src + (int64_t)r * src_ld + c);
        const bf16_t hi = f32_to_bf16(v);
        const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
        bf16_t* d = dst + (int64_t)r * 2 * cols + c;
        d[0] = hi; d[cols] = lo;
    }
}

hipError_t launch_pack_split_w(const void* src, int src_dt, int64_t src_ld, void* dst, int rows, int cols, hipStream_t s) {
    const int64_t total = (int64_t)rows * cols;
    if (total <= 0) return hipSuccess;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (src_dt == F32) hipLaunchKernelGGL(pack_split_w_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)src, src_ld, (bf16_t*)dst, rows, cols);
    else hipLaunchKernelGGL(pack_split_w_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)src, src_ld, (bf16_t*)dst, rows, cols);
    return hipGetLastError();
}

// Activations (per layer): src fp32 [rows, K] plain rows or blocked (32 floats per 128-byte piece) -> dst bf16 [rows, 2 K] =
// [hi | lo] in plain rows or blocked (64 bf16 per piece, rows of 2 K * 2 / 128 pieces).  One thread = 8 consecutive values:
// 32 bytes in, 2 x 16 bytes out.  K % 64 == 0 (whole pieces of the destination per half).
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int64_t src_ld, bf16_t* __restrict__ dst, int64_t rows, int K,
                                                          int src_blocked, int dst_blocked) {
    const int nch = K >> 3;
    const int64_t total = rows * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nch;
        const int c = (int)(i - r * nch) * 8;
        const float* sp = src_blocked ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + blocked_off(r, (int64_t)c * 4, (int64_t)K * 4 >> 7))
                                      : src + r * src_ld + c;
        float v[8], lo[8];
        load8<float>(sp, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) lo[k] = v[k] - round_to_bf16(v[k]);
        auto at = [&](int col) -> bf16_t* {
            return dst_blocked ? reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(dst) + blocked_off(r, (int64_t)col * 2, (int64_t)2 * K * 2 >> 7))
                               : dst + r * 2 * K + col;
        };
        store8<bf16_t>(at(c), v);
        store8<bf16_t>(at(K + c), lo);
    }
}

hipError_t launch_split_rows(const float* src, int64_t src_ld, void* dst, int64_t rows, int K, bool src_blocked, bool dst_blocked, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (K <= 0 || K % 64 || (!src_blocked && (src_ld < K || src_ld % 4)) || ((uintptr_t)src) % 16 || ((uintptr_t)dst) % 16) return hipErrorInvalidValue;
    const int64_t total = rows * (K >> 3);
    int64_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)nb), dim3(256), 0, s, src, src_ld, (bf16_t*)dst, rows, K, (int)src_blocked, (int)dst_blocked);
    return hipGetLastError();
}

// ---- layer 0 of the norm-folded form: in_proj's operand is one of the V embedding rows, so its output is one of V rows ------------
//   tab[tok][n] = round( (sum_k emb[tok][k] * Wf[n][k]) * rstd(emb[tok]) ),   rstd = rsqrt(mean_k emb[tok][k]^2 + eps)
// (bind time; fp32 sums in index order; emb and Wf already in the model dtype)
template <typename T>
__global__ __launch_bounds__(256) void embed_inproj_table_kernel(const T* __restrict__ emb, const T* __restrict__ Wf, T* __restrict__ tab,
                                                                 int D, int N2, float eps) {
    const int n = blockIdx.x * 256 + threadIdx.x, tok = blockIdx.y;
    if (n >= N2) return;
    const T* er = emb + (int64_t)tok * D;
    const T* wr = Wf + (int64_t)n * D;
    float acc = 0.f, ss = 0.f;
    for (int k = 0; k < D; ++k) {
        const float ev = Elem<T>::load(er + k);
        acc = __builtin_fmaf(ev, Elem<T>::load(wr + k), acc);
        ss = __builtin_fmaf(ev, ev, ss);
    }
    Elem<T>::store(tab + (int64_t)tok * N2 + n, acc * rsqrtf(ss / (float)D + eps));
}

hipError_t launch_embed_inproj_table(const void* emb, const void* Wf, void* tab, int V, int D, int N2, float eps, int dt, hipStream_t s) {
    dim3 grid((unsigned)((N2 + 255) / 256), (unsigned)V), block(256);
    if (dt == BF16) hipLaunchKernelGGL(embed_inproj_table_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)emb, (const bf16_t*)Wf, (bf16_t*)tab, D, N2, eps);
    else hipLaunchKernelGGL(embed_inproj_table_kernel<float>, grid, block, 0, s, (const float*)emb, (const float*)Wf, (float*)tab, D, N2, eps);
    return hipGetLastError();
}

// x / z of layer 0 (both blocked [rows8, E]) gathered from the table: row (strand s, position t) copies tab[token(s, t)][0, E) to x
// and [E, 2E) to z; the rc strand's token is the complement of the reversed ids (RCPSEmbedding by index arithmetic, as in norm.hip).
// One wave per row, 16-byte accesses; E * elem a multiple of 128 bytes.
template <typename T>
__global__ __launch_bounds__(256) void embed_xz_gather_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ comp8,
                                                              const T* __restrict__ tab, T* __restrict__ x, T* __restrict__ z, int B, int L, int E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)2 * B * L) return;
    const int s = (int)(row / L), t = (int)(row - (int64_t)s * L);
    int tok;
    if (s < B) tok = ids[(int64_t)s * L + t];
    else tok = comp8[ids[(int64_t)(s - B) * L + (L - 1 - t)] & 7];
    const char* src = reinterpret_cast<const char*>(tab + (int64_t)(tok & 7) * 2 * E);
    const int64_t rowb = (int64_t)E * sizeof(T), pieces = rowb >> 7;
    char* xb = reinterpret_cast<char*>(x);
    char* zb = reinterpret_cast<char*>(z);
    for (int64_t cb = (int64_t)lane * 16; cb < rowb; cb += 64 * 16) {
        const int64_t o = blocked_off(row, cb, pieces);
        *reinterpret_cast<u32x4*>(xb + o) = *reinterpret_cast<const u32x4*>(src + cb);
        *reinterpret_cast<u32x4*>(zb + o) = *reinterpret_cast<const u32x4*>(src + rowb + cb);
    }
}

hipError_t launch_embed_xz_gather(const int32_t* ids, const int32_t* comp8, const void* tab, void* x, void* z, int B, int L, int E, int dt,
                                  hipStream_t s) {
    const int64_t rows = (int64_t)2 * B * L;
    if (rows <= 0) return hipSuccess;
    if ((E * (dt == BF16 ? 2 : 4)) % 128) return hipErrorInvalidValue;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dt == BF16) hipLaunchKernelGGL(embed_xz_gather_kernel<bf16_t>, grid, block, 0, s, ids, comp8, (const bf16_t*)tab, (bf16_t*)x, (bf16_t*)z, B, L, E);
    else hipLaunchKernelGGL(embed_xz_gather_kernel<float>, grid, block, 0, s, ids, comp8, (const float*)tab, (float*)x, (float*)z, B, L, E);
    return hipGetLastError();
}

hipError_t launch_pack_A(const void* A_log, int src_dt, float* A2, int64_t n, float scale, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (src_dt == F32) hipLaunchKernelGGL(pack_A_kernel<float>, grid, block, 0, s, (const float*)A_log, A2, n, scale);
    else hipLaunchKernelGGL(pack_A_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)A_log, A2, n, scale);
    return hipGetLastError();
}

// Rows of the 2B-strand activation tensor that a positions-only forward consumes after the last mixer: strand b row p_q,
// strand B + b row L - 1 - p_q  ->  compact out[(strand * P + q), E] (plain rows).  One wave per row, 16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ out, int B, int L, int E,
                                                          Positions pos, int blocked) {
    const int P = pos.n;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= 2 * B * P) return;
    const int strand = wave / P, q = wave - strand * P;
    int p = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i == q) p = pos.p[i];
    const int64_t row = (int64_t)strand * L + (strand < B ? p : L - 1 - p);
    const int64_t rowb = (int64_t)E * sizeof(T);
    const char* sb = reinterpret_cast<const char*>(src);
    char* db = reinterpret_cast<char*>(out) + (int64_t)wave * rowb;
    for (int64_t cb = (int64_t)lane * 16; cb < rowb; cb += 64 * 16) {
        const int64_t so = blocked ? blocked_off(row, cb, rowb >> 7) : row * rowb + cb;
        *reinterpret_cast<u32x4*>(db + cb) = *reinterpret_cast<const u32x4*>(sb + so);
    }
}

hipError_t launch_gather_rows(const void* src, void* out, int B, int L, int E, Positions pos, int dt, bool blocked,
                              hipStream_t s) {
    const int64_t rows = (int64_t)2 * B * pos.n;
    if (rows <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if ((E * esz) % 16 || (blocked && (E * esz) % 128)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dt == BF16) hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (bf16_t*)out, B, L, E, pos, (int)blocked);
    else hipLaunchKernelGGL(gather_rows_kernel<float>, grid, block, 0, s, (const float*)src, (float*)out, B, L, E, pos, (int)blocked);
    return hipGetLastError();
}

// ---- per-device launch state (kernels.hpp) -----------------------------------------------------------------------------
hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;        // (device ordinal, kernel)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

int device_cu_count() {
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

}  // namespace pcad
