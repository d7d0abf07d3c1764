// Fused depth-wise conv1d + SiLU (causal AND anti-causal) + x_proj of both directions.
//
// Replaces, per layer, causal_conv1d_fn x4 and the x_proj GEMM x4 of the reference — in this
// engine previously three launches: conv_bidir (read x, write xc_f / xc_r) and two x_proj GEMMs (each re-reading its xc).
// Here x is read ONCE: a block owns 128 consecutive timesteps of one strand and walks the channels in 128-byte K-tiles;
// per K-tile
//   1. the raw x tile (128 rows + 3 halo rows on each side) arrives by LDS-DMA from the blocked x tensor (1 KiB blocks),
//      two K-tiles ahead, together with that K-tile's Wx slabs (both directions) and conv taps;
//   2. conv pass (VALU): thread = (direction, 16-byte channel chunk, 4 consecutive rows): 7 raw rows and 5 tap vectors
//      from LDS, fp32 taps + bias + SiLU, result rounded to the model dtype and written into the LDS tiles cf / cr in
//      the MFMA A-fragment image (XOR swizzle);
//   3. MFMA pass: wave = (direction, 32 rows): x_dbl += c{f,r} . Wx^T, accumulators stay in registers across K-tiles;
//      the same waves copy cf / cr to the blocked xc tensors the scan reads, 1 KiB contiguous per store instruction.
// Epilogue: dt_low [rows, 64] (model dtype) and B_t | C_t [rows, 32] (fp32) per direction, as the split-epilogue GEMM wrote.
// HBM traffic per row: read E*s (x) + write 2*E*s (xc) instead of read 3*E*s + write 2*E*s, and one launch instead of three.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int CX_ROWS = 128;                 // output rows (timesteps) per block
constexpr int CX_RAW = CX_ROWS + 8;          // raw rows staged per K-tile (3 + 128 + 3, padded to 17 groups of 8)
constexpr int CX_ROWB = 128;                 // bytes of K per K-tile row
constexpr int CX_RAW_BYTES = CX_RAW * CX_ROWB;           // 17408
constexpr int CX_NRAW = 3;                   // raw ring depth
constexpr int CX_TILE_BYTES = CX_ROWS * CX_ROWB;         // 16384: cf, cr
constexpr int CX_WROWS = 96;
constexpr int CX_W_BYTES = CX_WROWS * CX_ROWB;           // 12288 per direction
constexpr int CX_CW_BYTES = 3072;            // conv taps of one K-tile: [dir][5][KC] fp32, padded
constexpr int CX_OFF_CF = CX_NRAW * CX_RAW_BYTES;
constexpr int CX_OFF_CR = CX_OFF_CF + CX_TILE_BYTES;
constexpr int CX_OFF_W = CX_OFF_CR + CX_TILE_BYTES;      // [2 stages][2 dirs]
constexpr int CX_OFF_CW = CX_OFF_W + 2 * 2 * CX_W_BYTES; // [2 stages]
constexpr int CX_LDS = CX_OFF_CW + 2 * CX_CW_BYTES;      // 140288
constexpr int CX_THREADS = 512;

__device__ __forceinline__ int cx_key(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ void cx_glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T> struct CxMma;
template <> struct CxMma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const u32x4& w, const u32x4& a, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c,
                                                       0, 0, 0);
    }
};
template <> struct CxMma<float> {
    static __device__ __forceinline__ f32x4 run(const u32x4& w, const u32x4& a, f32x4 c) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[s]), __uint_as_float(a[s]), c, 0, 0, 0);
        return c;
    }
};

// 16 bytes <-> CPC floats
template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int CPC = 8;
    static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(r[i]); v[2 * i + 1] = bf16hi_to_f32(r[i]); }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        return r;
    }
};
template <> struct Chunk<float> {
    static constexpr int CPC = 4;
    static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[4]) {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    }
};

struct ConvxDir {
    const void* Wx;      // [96, E] model dtype (rows [R, 64) zero)
    void* xc;            // [rows8, E] blocked
    void* dtl;           // [rows, 64]
    float* bc;           // [rows, 32]
};

// convw: per K-tile CX_CW_BYTES of fp32 [dir][tap 0..3, bias][KC]  (packed at bind time by launch_pack_convw)
template <typename T>
__global__ __launch_bounds__(CX_THREADS, 2) void convx_kernel(const T* __restrict__ x, const float* __restrict__ convw,
                                                              ConvxDir d0, ConvxDir d1, int S, int L, int E) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPC = Chunk<T>::CPC;
    constexpr int KC = CX_ROWB / (int)sizeof(T);            // channels per K-tile: 64 (bf16) / 32 (fp32)
    constexpr int CW_PIECES = (2 * 5 * KC * 4 + 1023) / 1024;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // 0..7
    const int tiles_per_strand = (L + CX_ROWS - 1) / CX_ROWS;
    const int strand = blockIdx.x / tiles_per_strand;
    const int t0 = (blockIdx.x - strand * tiles_per_strand) * CX_ROWS;
    const int64_t row0 = (int64_t)strand * L;                     // whole-tensor row of t = 0
    const int nkt = E / KC;
    const int64_t pieces = nkt;                                   // 128-byte pieces per row of x / xc

    // ---- staging (every wave issues exactly 7 LDS-DMAs per K-tile: 3 raw + 3 Wx + 1 taps) -------------------------
    // raw: 17 groups of 8 rows; wave w stages groups w, w + 8 and (all waves, redundantly) group 16
    const char* xb = reinterpret_cast<const char*>(x);
    int64_t raw_src[3];
    int raw_dst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int grp = i < 2 ? wave + 8 * i : 16;
        const int q = grp * 8 + (lane >> 3);                      // raw row index: t = t0 - 3 + q
        const int t = min(max(t0 - 3 + q, 0), L - 1);             // clamped address; masked in the conv
        raw_src[i] = blocked_off(row0 + t, 0, pieces) + ((lane & 7) << 4);
        raw_dst[i] = grp * 8 * CX_ROWB;
    }
    // Wx slabs: per direction 96 rows = 12 groups of 8; 24 groups over 8 waves = 3 each
    const char* w_src[3];
    int w_dst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int grp = wave * 3 + i;                              // 0..23
        const int dir = grp / 12, g = grp - dir * 12;
        const int r = g * 8 + (lane >> 3);
        const T* Wd = (const T*)(dir == 0 ? d0.Wx : d1.Wx);
        w_src[i] = reinterpret_cast<const char*>(Wd + (int64_t)r * E) + (((lane & 7) ^ cx_key(r)) << 4);
        w_dst[i] = dir * CX_W_BYTES + g * 8 * CX_ROWB;
    }
    const char* cw_src = reinterpret_cast<const char*>(convw) + (wave % CW_PIECES) * 1024 + lane * 16;
    const int cw_dst = (wave % CW_PIECES) * 1024;

    auto stage_raw = [&](int kt) {
        char* base = smem + (kt % CX_NRAW) * CX_RAW_BYTES;
#pragma unroll
        for (int i = 0; i < 3; ++i) cx_glds16(xb + raw_src[i] + (int64_t)kt * 1024, base + raw_dst[i]);
    };
    auto stage_w = [&](int kt) {
        char* wb = smem + CX_OFF_W + (kt & 1) * 2 * CX_W_BYTES;
#pragma unroll
        for (int i = 0; i < 3; ++i) cx_glds16(w_src[i] + (int64_t)kt * CX_ROWB, wb + w_dst[i]);
        cx_glds16(cw_src + (int64_t)kt * CX_CW_BYTES, smem + CX_OFF_CW + (kt & 1) * CX_CW_BYTES + cw_dst);
    };

    // ---- conv-pass mapping: direction, 16-byte chunk, 4 consecutive rows ---------------------------------------
    const int cdir = __builtin_amdgcn_readfirstlane(tid >> 8);     // waves 0-3: causal, 4-7: anti-causal (wave-uniform)
    const int c8 = tid & 7;
    const int g4 = ((tid >> 3) & 31) * 4;      // first output row (tile-relative)
    const int qbase = g4 + (cdir ? 3 : 0);     // first raw row of the 7-row window: fwd q = r .. r+3, rev q = r+3 .. r+6
    // ---- MFMA-pass mapping: direction, 32 rows ------------------------------------------------------------------
    const int mdir = wave >> 2, mq = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    int a_off[2], a_key[2], wf_off[6], wf_key[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = mq * 32 + i * 16 + li; a_off[i] = r * CX_ROWB; a_key[i] = cx_key(r); }
#pragma unroll
    for (int j = 0; j < 6; ++j) { const int r = j * 16 + li; wf_off[j] = r * CX_ROWB; wf_key[j] = cx_key(r); }
    f32x4 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- xc store-pass mapping: wave w copies rows 16w .. 16w+15 of cf and of cr (8 rows x 128 B per instruction) ---
    T* xcf = (T*)d0.xc;
    T* xcr = (T*)d1.xc;

    // prologue: K-tiles 0 and 1 of raw, K-tile 0 of W / taps
    stage_w(0);
    stage_raw(0);
    if (nkt > 1) stage_raw(1);
    for (int kt = 0; kt < nkt; ++kt) {
        // issue order per iteration: W(kt+1), taps(kt+1), raw(kt+2); needed now: raw(kt), W(kt), taps(kt).  DMAs retire in
        // order, so "at most 3 outstanding" (the raw DMAs of kt+1) means everything needed has landed.
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // (1) this K-tile's inputs are in LDS; cf / cr are free (MFMA + copy of kt-1 done)
        asm volatile("" ::: "memory");
        if (kt + 1 < nkt) stage_w(kt + 1);
        if (kt + 2 < nkt) stage_raw(kt + 2);

        // ---------------- conv pass ----------------
        auto conv_pass = [&](auto rev_tag) {
            constexpr bool REVC = decltype(rev_tag)::value;       // static window indices (no dynamic register indexing)
            const char* raw = smem + (kt % CX_NRAW) * CX_RAW_BYTES;
            const float* cw = reinterpret_cast<const float*>(smem + CX_OFF_CW + (kt & 1) * CX_CW_BYTES) + (REVC ? 5 * KC : 0) + c8 * CPC;
            float wt[4][CPC], bias[CPC];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < CPC; e += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(cw + k * KC + e);
                    wt[k][e] = v[0]; wt[k][e + 1] = v[1]; wt[k][e + 2] = v[2]; wt[k][e + 3] = v[3];
                }
#pragma unroll
            for (int e = 0; e < CPC; e += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(cw + 4 * KC + e);
                bias[e] = v[0]; bias[e + 1] = v[1]; bias[e + 2] = v[2]; bias[e + 3] = v[3];
            }
            float win[7][CPC];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int q = qbase + j;
                const int t = t0 - 3 + q;
                u32x4 r = *reinterpret_cast<const u32x4*>(raw + q * CX_ROWB + c8 * 16);
                const unsigned keep = 0u - (unsigned)((unsigned)t < (unsigned)L);   // zero padding at the sequence ends,
                r &= u32x4{keep, keep, keep, keep};                                  // branch-free (loads stay batched)
                Chunk<T>::unpack(r, win[j]);
            }
            char* ct = smem + (REVC ? CX_OFF_CR : CX_OFF_CF);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o[CPC];
#pragma unroll
                for (int e = 0; e < CPC; ++e) {
                    float a = bias[e];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a += wt[k][e] * win[REVC ? (j + 3 - k) : (j + k)][e];   // x[t+3-k] / x[t-3+k]
                    o[e] = silu(a);
                }
                const int r = g4 + j;
                *reinterpret_cast<u32x4*>(ct + r * CX_ROWB + ((c8 ^ cx_key(r)) << 4)) = Chunk<T>::pack(o);
            }
        };
        if (cdir) conv_pass(std::true_type{});
        else conv_pass(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // (2) cf / cr complete
        asm volatile("" ::: "memory");

        // ---------------- MFMA pass + copy of cf / cr to the blocked xc tensors ----------------
        {
            const char* at = smem + (mdir ? CX_OFF_CR : CX_OFF_CF);
            const char* wb = smem + CX_OFF_W + (kt & 1) * 2 * CX_W_BYTES + mdir * CX_W_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 4 + lg;
                u32x4 af[2], wfr[6];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const u32x4*>(at + a_off[i] + ((chunk ^ a_key[i]) << 4));
#pragma unroll
                for (int j = 0; j < 6; ++j) wfr[j] = *reinterpret_cast<const u32x4*>(wb + wf_off[j] + ((chunk ^ wf_key[j]) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = CxMma<T>::run(wfr[j], af[i], acc[i][j]);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = wave * 16 + i * 8 + (lane >> 3);
                    const int t = t0 + r;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (d ? CX_OFF_CR : CX_OFF_CF) + r * CX_ROWB +
                                                                   (((lane & 7) ^ cx_key(r)) << 4));
                    if (t < L) {
                        char* dst = reinterpret_cast<char*>(d ? xcr : xcf) + blocked_off(row0 + t, 0, pieces) + (int64_t)kt * 1024 +
                                    ((lane & 7) << 4);
                        *reinterpret_cast<u32x4*>(dst) = v;
                    }
                }
            }
        }
    }

    // ---- epilogue: lane (li = row, lg): fragment j -> columns j*16 + lg*4 .. +3 --------------------------------
    const ConvxDir dd = mdir ? d1 : d0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = t0 + mq * 32 + i * 16 + li;
        if (t >= L) continue;
        const int64_t row = row0 + t;
        T* dl = (T*)dd.dtl + row * 64;
        float* bcr = dd.bc + row * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (sizeof(T) == 2) {
                u32x2 v = {pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
                *reinterpret_cast<u32x2*>(dl + j * 16 + lg * 4) = v;
            } else {
                *reinterpret_cast<f32x4*>(dl + j * 16 + lg * 4) = acc[i][j];
            }
        }
#pragma unroll
        for (int j = 4; j < 6; ++j) {
            f32x4 v = {Elem<T>::round(acc[i][j][0]), Elem<T>::round(acc[i][j][1]), Elem<T>::round(acc[i][j][2]),
                       Elem<T>::round(acc[i][j][3])};
            *reinterpret_cast<f32x4*>(bcr + (j - 4) * 16 + lg * 4) = v;
        }
    }
}

// conv taps of both directions -> per K-tile [dir][tap 0..3, bias][KC] fp32 (CX_CW_BYTES per K-tile, zero padded)
__global__ __launch_bounds__(256) void pack_convw_kernel(const float* __restrict__ wf, const float* __restrict__ bfw,
                                                         const float* __restrict__ wr, const float* __restrict__ brw,
                                                         float* __restrict__ out, int E, int KC) {
    const int per = CX_CW_BYTES / 4;
    const int64_t total = (int64_t)(E / KC) * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kt = (int)(i / per), o = (int)(i - (int64_t)kt * per);
        float v = 0.f;
        if (o < 2 * 5 * KC) {
            const int dir = o / (5 * KC), k = (o - dir * 5 * KC) / KC, c = kt * KC + (o % KC);
            const float* w = dir ? wr : wf;
            const float* b = dir ? brw : bfw;
            v = k < 4 ? w[(int64_t)c * 4 + k] : b[c];
        }
        out[i] = v;
    }
}

size_t convx_packed_bytes(int E, int dt) {
    const int KC = CX_ROWB / (dt == BF16 ? 2 : 4);
    return (size_t)(E / KC) * CX_CW_BYTES;
}

hipError_t launch_pack_convw(const float* wf, const float* bf, const float* wr, const float* br, float* out, int E, int dt,
                             hipStream_t s) {
    const int KC = CX_ROWB / (dt == BF16 ? 2 : 4);
    if (E % KC) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_convw_kernel, dim3(64), dim3(256), 0, s, wf, bf, wr, br, out, E, KC);
    return hipGetLastError();
}

hipError_t launch_convx(const void* x, const float* convw, const void* Wx0, void* xc0, void* dtl0, float* bc0,
                        const void* Wx1, void* xc1, void* dtl1, float* bc1, int S, int L, int E, int dt, hipStream_t s) {
    if (S <= 0 || L <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if ((E * esz) % CX_ROWB) return hipErrorInvalidValue;
    ConvxDir d0{Wx0, xc0, dtl0, bc0}, d1{Wx1, xc1, dtl1, bc1};
    const int tiles = S * ((L + CX_ROWS - 1) / CX_ROWS);
    static bool attr_b = false, attr_f = false;
    if (dt == BF16) {
        auto k = convx_kernel<bf16_t>;
        if (!attr_b) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS); attr_b = true; }
        hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(CX_THREADS), CX_LDS, s, (const bf16_t*)x, convw, d0, d1, S, L, E);
    } else {
        auto k = convx_kernel<float>;
        if (!attr_f) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS); attr_f = true; }
        hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(CX_THREADS), CX_LDS, s, (const float*)x, convw, d0, d1, S, L, E);
    }
    return hipGetLastError();
}

}  // namespace pcad
