// Fused depth-wise conv1d + SiLU (causal AND anti-causal) + x_proj of both directions.
//
// Replaces, per layer, causal_conv1d_fn x4 and the x_proj GEMM x4 of the reference — in this
// engine previously three launches: conv_bidir (read x, write xc_f / xc_r) and two x_proj GEMMs (each re-reading its xc).
// Here x is read ONCE: a block (8 waves; waves 0-3 own the causal direction, 4-7 the anti-causal one) owns 128
// consecutive timesteps of one strand and walks the channels in 128-byte K-tiles as a software pipeline with ONE barrier
// per K-tile.  In iteration `it`
//   * the LDS-DMAs (buffer_load ... lds: constant 32-bit lane offsets + scalar offsets) of Wx(it), taps(it+1) and the raw
//     x tile (it+1) (128 rows + 3 halo rows on each side, from the blocked x tensor) are issued, each into the slot whose
//     last reader finished one iteration earlier;
//   * conv of K-tile it (VALU): thread = (16-byte channel chunk, 4 consecutive rows), convolved in two 8-byte halves
//     (7 raw rows and 5 tap vectors from LDS each; channel pairs through the packed fp32 pipe; fp32 taps + bias + SiLU),
//     result rounded to the model dtype and written into conv stage it & 1 in the MFMA A-fragment image (XOR swizzle);
//   * MFMA of K-tile it-1 (matrix pipe): wave = 32 rows, x_dbl += c . Wx^T from conv stage (it-1) & 1, accumulators stay
//     in registers across K-tiles; the same waves copy that conv stage to the blocked xc tensors the scan reads (1 KiB
//     contiguous per store instruction).  The causal waves convolve first and multiply / copy second, the anti-causal waves the
//     other way round (the two waves of a SIMD are one of each).
// Epilogue: dt_low [rows, 64] (model dtype) and B_t | C_t [rows, 32] (fp32) per direction, as the split-epilogue GEMM wrote.
// HBM traffic per row: read E*s (x) + write 2*E*s (xc) instead of read 3*E*s + write 2*E*s, and one launch instead of three.
// Measured: instruction-issue bound (VALU busy 44 %, ~500 wave-instructions per K-tile of which the conv
// arithmetic is about half); removing the conv math, the MFMAs or the xc stores from the kernel saves 7 %, 6 % and 16 %.
// Register pressure is a hard constraint: a spilled value is reloaded by a scratch load that shares vmcnt with the
// DMAs in flight and drains them (an early version with 19 spilled dwords ran 1.8x slower).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int CX_ROWS = 128;                 // output rows (timesteps) per block
constexpr int CX_RAW = CX_ROWS + 8;          // raw rows staged per K-tile (3 + 128 + 3, padded to 17 groups of 8)
constexpr int CX_ROWB = 128;                 // bytes of K per K-tile row
constexpr int CX_RAW_BYTES = CX_RAW * CX_ROWB;           // 17408
constexpr int CX_NRAW = 2;                   // raw ring depth
constexpr int CX_TILE_BYTES = CX_ROWS * CX_ROWB;         // 16384: cf, cr
constexpr int CX_CW_BYTES = 3072;            // conv taps of one K-tile: [dir][5][KC] fp32, padded
constexpr int CX_OFF_C = CX_NRAW * CX_RAW_BYTES;         // conv outputs: [2 stages][cf, cr]
constexpr int CX_OFF_W = CX_OFF_C + 2 * 2 * CX_TILE_BYTES;   // Wx slabs: [W stages][2 dirs]
constexpr int CX_THREADS = 512;
// NJ = x_proj output columns / 16 = (Rp + 32) / 16: 6 (dt_rank <= 64, every PlantCaduceus size and PlantCAD2 Small / Medium) or
// 8 (dt_rank 65..96: PlantCAD2 Large, d_model 1536).  The Wx slab of one direction and K-tile is 16 NJ rows of 128 bytes.
// NJ == 6: two W stages (Wx(it) lands while the MFMAs of K-tile it-1 read the other stage), 152 KiB of LDS.
// NJ == 8: two such stages would need 168 KiB, so there is ONE: the MFMAs of K-tile it-1 run first in iteration it, a second
//          barrier marks the end of every wave's fragment reads, and only then is Wx(it) fetched into the same slab (it has the
//          conv pass of the iteration to land).  136 KiB.
template <int NJ> struct CxGeom {
    static constexpr int WROWS = 16 * NJ;
    static constexpr int W_BYTES = WROWS * CX_ROWB;                       // per direction
    static constexpr int WSTAGES = NJ <= 6 ? 2 : 1;
    static constexpr int OFF_CW = CX_OFF_W + WSTAGES * 2 * W_BYTES;       // taps: [2 stages]
    static constexpr int LDS = OFF_CW + 2 * CX_CW_BYTES;                  // NJ 6: 155648, NJ 8: 139264
    static constexpr int WPW = (2 * WROWS / 8) / 8;                       // Wx row groups (of 8 rows) staged per wave: 3 / 4
};
static_assert(CxGeom<6>::LDS <= 160 * 1024 && CxGeom<8>::LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int cx_key(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ void cx_glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: descriptor base + per-lane 32-bit offset + wave-uniform 32-bit offset
__device__ __forceinline__ void cx_blds16(const void* base, uint32_t voff, uint32_t soff, char* lds_wave_base,
                                          uint32_t num_records = 0xfffffffcu) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)num_records, 0x00020000),
                                             (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
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
    static __device__ __forceinline__ void unpack_half(const u32x2& r, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { v[2 * i] = bf16lo_to_f32(r[i]); v[2 * i + 1] = bf16hi_to_f32(r[i]); }
    }
    static __device__ __forceinline__ u32x2 pack_half(const float (&v)[4]) { return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; }
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
    static __device__ __forceinline__ void unpack_half(const u32x2& r, float (&v)[2]) { v[0] = __uint_as_float(r[0]); v[1] = __uint_as_float(r[1]); }
    static __device__ __forceinline__ u32x2 pack_half(const float (&v)[2]) { return u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}; }
    static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[4]) {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    }
};

struct ConvxDir {
    const void* Wx;      // [Rp + 32, E] model dtype (rows [R, Rp) zero)
    void* xc;            // [rows8, E] blocked
    void* dtl;           // [rows, Rp]
    float* bc;           // [rows, 32]
    int dtl_split;       // T == float only: dtl is a bf16 tensor [rows, 2 Rp] = [hi | lo] (the scan's split-bf16 dt_proj operand)
};

// convw: per K-tile CX_CW_BYTES of fp32 [dir][tap 0..3, bias][KC]  (packed at bind time by launch_pack_convw)
// ZFILL (L % 8 == 0: a strand is a contiguous byte range of the blocked tensor): the raw tile is fetched through a
// descriptor that covers exactly the strand, so the halo rows outside [0, L) are out of range and arrive as zeros (a
// negative strand-relative offset wraps to a huge unsigned one) - the conv pass then needs no per-element masking.
// XS (T == float only; api.hip "f32_gemm_split"): x_proj on the bf16 matrix pipes as three bf16 products per fp32 product.  Wx arrives
// as bf16 [XP, 2E] whose K-tile kt (32 channels) is the 128-byte piece [hi (32) | lo (32)] (launch_pack_convx_wsplit: same row pitch, same
// LDS slab and staging as the fp32 weight); the conv stage stays fp32 (it is also what the xc copy-out reads) and the A fragment - 8
// consecutive channels per lane - is split into hi / lo when it is loaded: 36 v_mfma_f32_16x16x32_bf16 per K-tile and wave instead of
// 96 v_mfma_f32_16x16x4_f32 at a quarter of the rate (the fp32 kernel is MFMA-bound: 3 072 of its cycles per K-tile and wave).
// K-split (small launches, launch_convx): gridDim.y = KS blocks share a row tile, block ks convolves / copies out / multiplies the K-tiles
// [ks * nkt / KS, (ks + 1) * nkt / KS) only and writes its raw fp32 x_proj accumulators to `part` ([KS][2 dirs][rows][16 NJ]);
// convx_reduce_kernel adds the KS partials in index order (deterministic) and writes dt_low / B|C in the formats of the epilogue below.
template <typename T, bool ZFILL, int NJ = 6, bool XS = false>
__global__ __launch_bounds__(CX_THREADS, 2) void convx_kernel(const T* __restrict__ x, const float* __restrict__ convw,
                                                              ConvxDir d0, ConvxDir d1, int S, int L, int E, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = CxGeom<NJ>;
    constexpr int CX_W_BYTES = G::W_BYTES, CX_OFF_CW = G::OFF_CW;
    constexpr bool W1 = G::WSTAGES == 1;                    // single Wx slab, second barrier per K-tile (see CxGeom)
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
    const int64_t pieces = E / KC;                                // 128-byte pieces per row of x / xc
    const int nkt = (E / KC) / (int)gridDim.y;                    // K-tiles THIS block walks (K-split: a gridDim.y-th of them)
    const int kt0 = (int)blockIdx.y * nkt;                        // ... starting at this one

    // ---- staging (every wave issues exactly 7 LDS-DMAs per K-tile: 3 raw + 3 Wx + 1 taps) -------------------------
    // raw: 17 groups of 8 rows; wave w stages groups w, w + 8 and (all waves, redundantly) group 16
    uint32_t raw_src[3];                                          // 32-bit offsets (launcher checks the tensor sizes)
    int raw_dst[3];
    const char* raw_base = reinterpret_cast<const char*>(x);
    uint32_t raw_records = 0xfffffffcu;
    if (ZFILL) {                                                  // strand-relative offsets, descriptor = the strand
        raw_base += (row0 >> 3) * pieces * 1024;
        raw_records = (uint32_t)((int64_t)(L >> 3) * pieces * 1024);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int grp = i < 2 ? wave + 8 * i : 16;
        const int q = grp * 8 + (lane >> 3);                      // raw row index: t = t0 - 3 + q
        if (ZFILL) {
            const int t = t0 - 3 + q;                             // t < 0 wraps out of range, t >= L is out of range
            raw_src[i] = (uint32_t)((int64_t)(t >> 3) * pieces * 1024 + ((t & 7) << 7) + ((lane & 7) << 4));
        } else {
            const int t = min(max(t0 - 3 + q, 0), L - 1);         // clamped address; masked in the conv
            raw_src[i] = (uint32_t)(blocked_off(row0 + t, 0, pieces) + ((lane & 7) << 4));
        }
        raw_dst[i] = grp * 8 * CX_ROWB;
    }
    // Wx slabs: per direction 16 NJ rows = 2 NJ groups of 8; 4 NJ groups over 8 waves = NJ / 2 each (3 or 4)
    constexpr int WPW = G::WPW;
    uint32_t w_src[WPW];
    int w_dst[WPW], w_dir[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int grp = wave * WPW + i;                            // 0 .. 4 NJ - 1
        const int dir = grp / (2 * NJ), g = grp - dir * (2 * NJ);
        const int r = g * 8 + (lane >> 3);
        w_dir[i] = dir;
        w_src[i] = (uint32_t)((int64_t)r * E * (int64_t)sizeof(T) + (((lane & 7) ^ cx_key(r)) << 4));
        w_dst[i] = dir * CX_W_BYTES + g * 8 * CX_ROWB;
    }
    const uint32_t cw_src = (uint32_t)((wave % CW_PIECES) * 1024 + lane * 16);
    const int cw_dst = (wave % CW_PIECES) * 1024;

    auto stage_raw = [&](int kt, int par) __attribute__((always_inline)) {          // par = kt & 1, a compile-time constant at every call site
        char* base = smem + par * CX_RAW_BYTES;
#pragma unroll
        for (int i = 0; i < 3; ++i) cx_blds16(raw_base, raw_src[i], (uint32_t)kt * 1024u, base + raw_dst[i], raw_records);
    };
    auto stage_w = [&](int kt, int par) __attribute__((always_inline)) {
        char* wb = smem + CX_OFF_W + (W1 ? 0 : par) * 2 * CX_W_BYTES;
#pragma unroll
        for (int i = 0; i < WPW; ++i) cx_blds16(w_dir[i] ? d1.Wx : d0.Wx, w_src[i], (uint32_t)kt * (uint32_t)CX_ROWB, wb + w_dst[i]);
    };
    auto stage_taps = [&](int kt, int par) __attribute__((always_inline)) {
        cx_blds16(convw, cw_src, (uint32_t)kt * (uint32_t)CX_CW_BYTES, smem + CX_OFF_CW + par * CX_CW_BYTES + cw_dst);
    };

    // ---- a wave works on ONE direction in both passes: waves 0-3 causal, 4-7 anti-causal (wave-uniform) ---------
    const int cdir = __builtin_amdgcn_readfirstlane(tid >> 8);
    // conv mapping: 16-byte chunk, 4 consecutive rows
    const int c8 = tid & 7;
    const int g4 = ((tid >> 3) & 31) * 4;      // first output row (tile-relative)
    const int qbase = g4 + (cdir ? 3 : 0);     // first raw row of the 7-row window: fwd q = r .. r+3, rev q = r+3 .. r+6
    // MFMA mapping: 32 rows
    const int mq = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    // fragment byte offset inside a tile for k-step kk: rows of one fragment are 16 apart and cx_key(r + 16) == cx_key(r),
    // so ONE lane offset per kk serves every A and W fragment (+ a compile-time row-block offset)
    int frag_lo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) frag_lo[kk] = li * CX_ROWB + (((kk * 4 + lg) ^ cx_key(li)) << 4);
    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // xc store mapping: wave w copies rows 16w .. 16w+15 of cf and of cr (8 rows x 128 B per instruction)
    T* xcf = (T*)d0.xc;
    T* xcr = (T*)d1.xc;
    const bool full_tile = t0 + CX_ROWS <= L;          // every xc store instruction is issued

    // Software pipeline, ONE barrier per K-tile: iteration `it` runs the conv of K-tile it (VALU, writes conv stage it&1)
    // and the MFMAs + xc copy-out of K-tile it-1 (matrix pipe + LDS, reads conv stage (it-1)&1) in the same barrier
    // interval, so the two pipes overlap instead of alternating.  DMAs issued at the start of iteration it: Wx(it),
    // taps(it+1), raw(it+1), each into the slot whose last reader finished in iteration it-1.
    // XS: A fragment of the bf16 MFMA = channels 8 lg .. 8 lg + 7 of row li = fp32 chunks 2 lg and 2 lg + 1 of the (swizzled) conv stage row
    int frag_xs[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) frag_xs[c] = li * CX_ROWB + (((2 * lg + c) ^ cx_key(li)) << 4);
    auto mfma_half = [&](int mpar, int kk) __attribute__((always_inline)) {          // mpar: parity of the K-tile being multiplied
        const char* at = smem + CX_OFF_C + mpar * 2 * CX_TILE_BYTES + cdir * CX_TILE_BYTES;
        const char* wb = smem + CX_OFF_W + (W1 ? 0 : mpar) * 2 * CX_W_BYTES + cdir * CX_W_BYTES;
        if constexpr (XS) {
            if (kk != 0) return;                              // the whole K-tile (K = 32) in the kk == 0 call
            u32x4 ahi[2], alo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* rowp = at + (mq * 32 + i * 16) * CX_ROWB;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(rowp + frag_xs[0]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(rowp + frag_xs[1]);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t h = pack_bf16x2(v[2 * q], v[2 * q + 1]);
                    ahi[i][q] = h;
                    alo[i][q] = pack_bf16x2(v[2 * q] - bf16lo_to_f32(h), v[2 * q + 1] - bf16hi_to_f32(h));
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const u32x4 whi = *reinterpret_cast<const u32x4*>(wb + j * 16 * CX_ROWB + frag_lo[0]);      // chunk lg of [hi | lo]
                const u32x4 wlo = *reinterpret_cast<const u32x4*>(wb + j * 16 * CX_ROWB + frag_lo[1]);      // chunk 4 + lg
                // three products per accumulator, the two row blocks alternating (no back-to-back dependent MFMAs)
                acc[0][j] = CxMma<bf16_t>::run(whi, ahi[0], acc[0][j]);
                acc[1][j] = CxMma<bf16_t>::run(whi, ahi[1], acc[1][j]);
                acc[0][j] = CxMma<bf16_t>::run(whi, alo[0], acc[0][j]);
                acc[1][j] = CxMma<bf16_t>::run(whi, alo[1], acc[1][j]);
                acc[0][j] = CxMma<bf16_t>::run(wlo, ahi[0], acc[0][j]);
                acc[1][j] = CxMma<bf16_t>::run(wlo, ahi[1], acc[1][j]);
            }
            return;
        }
        u32x4 af[2], wfr[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const u32x4*>(at + (mq * 32 + i * 16) * CX_ROWB + frag_lo[kk]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wfr[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * CX_ROWB + frag_lo[kk]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = CxMma<T>::run(wfr[j], af[i], acc[i][j]);
    };
    // xc copy-out: lane offset of row r0 = 16 * wave + lane / 8 (the i = 1 row is 8 rows = one 1 KiB row-block further)
    const int cr0 = wave * 16 + (lane >> 3);
    const uint32_t xc_lo = (uint32_t)(blocked_off(row0 + t0 + cr0, 0, pieces) + ((lane & 7) << 4));
    const uint32_t xc_blk = (uint32_t)(pieces << 10);
    int c_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) c_lds[i] = (cr0 + i * 8) * CX_ROWB + (((lane & 7) ^ cx_key(cr0 + i * 8)) << 4);
    auto copy_out = [&](int mt, int mpar) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + CX_OFF_C + mpar * 2 * CX_TILE_BYTES + d * CX_TILE_BYTES +
                                                               c_lds[i]);
                if (t0 + cr0 + i * 8 < L) {
                    char* dst = reinterpret_cast<char*>(d ? xcr : xcf) + (xc_lo + (uint32_t)i * xc_blk + (uint32_t)mt * 1024u);
                    *reinterpret_cast<u32x4*>(dst) = v;
                }
            }
        }
    };
    auto conv_pass = [&](int kt, int par, auto rev_tag, bool with_mfma) __attribute__((always_inline)) {
        constexpr bool REVC = decltype(rev_tag)::value;       // static window indices (no dynamic register indexing)
        const char* raw = smem + par * CX_RAW_BYTES;
        const float* cw = reinterpret_cast<const float*>(smem + CX_OFF_CW + par * CX_CW_BYTES) + (REVC ? 5 * KC : 0) + c8 * CPC;
        // phase-shifted directions (round 4, -1.5 % in 3 of 3 interleaved pairs, r04u): the anti-causal waves run the MFMAs + copy-out of
        // K-tile it-1 first and convolve K-tile it second, the causal waves the other way round, so that on every SIMD (one wave of each
        // direction) one wave's VALU phase faces the other's LDS / MFMA / store phase.  (Before: both interleaved MFMA half 0, conv
        // half 0, MFMA half 1, conv half 1, copy-out in the same order.)
        if (with_mfma && REVC) { mfma_half(1 - par, 0); mfma_half(1 - par, 1); copy_out(kt - 1, 1 - par); }
        char* ct = smem + CX_OFF_C + par * 2 * CX_TILE_BYTES + (REVC ? CX_TILE_BYTES : 0);
        const f32x2_t nl2e = {-kLog2e, -kLog2e}, one = {1.0f, 1.0f};
        // The 16-byte chunk is convolved in two halves of HC channels (8-byte LDS accesses: same LDS cycles, half the live
        // registers: a spill here costs far more than its load - a scratch reload shares vmcnt with the LDS-DMAs in flight
        // and drains them).  Channel pairs go through the packed fp32 pipe: taps, bias and window as (e, e+1) pairs.
        constexpr int HC = CPC / 2;
        // ZFILL (the engine's case): the 7-row window of BOTH halves in one 16-byte read per row (7 ds_read_b128 instead of 14
        // ds_read_b64: half the LDS cycles of the window reads — a 2-way instead of a 4-way bank conflict between the rows a lane
        // group touches — and one LDS round trip per K-tile instead of two), kept packed: 28 registers.  The masked (ragged
        // length) form keeps the two 8-byte reads: with the masks it no longer fits the register file.
#ifdef PCAD_CX_B64
        constexpr bool B128 = false;
#else
        constexpr bool B128 = ZFILL && NJ <= 6;           // NJ 8: 16 more accumulator registers - the packed window no longer fits
#endif
        u32x4 rwin[7];
        if constexpr (B128) {
#pragma unroll
            for (int j = 0; j < 7; ++j) rwin[j] = *reinterpret_cast<const u32x4*>(raw + (qbase + j) * CX_ROWB + c8 * 16);
        }
        auto conv_half = [&](int h) __attribute__((always_inline)) {
            f32x2_t wt[4][HC / 2], bias[HC / 2];
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int e = 0; e < HC; e += 2) {
                    const f32x2_t v = *reinterpret_cast<const f32x2_t*>(cw + k * KC + h * HC + e);
                    if (k < 4) wt[k][e / 2] = v; else bias[e / 2] = v;
                }
            f32x2_t win[7][HC / 2];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int q = qbase + j;
                const int t = t0 - 3 + q;
                u32x2 r;
                if constexpr (B128) {
                    r = u32x2{rwin[j][2 * h], rwin[j][2 * h + 1]};
                } else {
                    r = *reinterpret_cast<const u32x2*>(raw + q * CX_ROWB + c8 * 16 + h * 8);
                    if (!ZFILL) {
                        const unsigned keep = 0u - (unsigned)((unsigned)t < (unsigned)L);   // zero padding at the sequence ends,
                        r &= u32x2{keep, keep};                                              // branch-free (loads stay batched)
                    }
                }
                float w1[HC];
                Chunk<T>::unpack_half(r, w1);
#pragma unroll
                for (int e = 0; e < HC / 2; ++e) win[j][e] = f32x2_t{w1[2 * e], w1[2 * e + 1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o[HC];
#pragma unroll
                for (int e = 0; e < HC / 2; ++e) {
                    f32x2_t a = bias[e];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a = wt[k][e] * win[REVC ? (j + 3 - k) : (j + k)][e] + a;   // x[t+3-k] / x[t-3+k]
                    const f32x2_t x2 = a * nl2e;                                  // silu(a) = a / (1 + 2^(-a log2 e))
                    const f32x2_t den = f32x2_t{fast_exp2(x2[0]), fast_exp2(x2[1])} + one;
                    const f32x2_t sv = a * f32x2_t{fast_rcp(den[0]), fast_rcp(den[1])};
                    o[2 * e] = sv[0]; o[2 * e + 1] = sv[1];
                }
                const int r = g4 + j;
                *reinterpret_cast<u32x2*>(ct + r * CX_ROWB + ((c8 ^ cx_key(r)) << 4) + h * 8) = Chunk<T>::pack_half(o);
            }
        };
        conv_half(0);
        __builtin_amdgcn_sched_barrier(0);
        conv_half(1);
        if (with_mfma && !REVC) { __builtin_amdgcn_sched_barrier(0); mfma_half(1 - par, 0); mfma_half(1 - par, 1); copy_out(kt - 1, 1 - par); }
    };

    // prologue: taps and raw tile of the block's first K-tile
    stage_taps(kt0, 0);
    stage_raw(kt0, 0);
    // the loop is unrolled by two so that every LDS stage offset is a compile-time constant (addresses = a lane register
    // + an immediate instead of per-access SALU/VALU address arithmetic; the kernel is instruction-issue bound)
    auto iteration = [&](int it, auto par_tag) __attribute__((always_inline)) {
        constexpr int P = decltype(par_tag)::value;               // it & 1
        // everything issued in the previous iteration (Wx(it-1), taps(it), raw(it)) must have landed; younger than those are
        // only that iteration's 4 xc stores (vmcnt retires in issue order, loads and stores alike)
        if (full_tile && it >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (W1) {
            // single Wx slab: taps / raw tile of K-tile it+1 first, then ALL MFMAs of K-tile it-1 (they read Wx(it-1)), a second
            // barrier (every wave's fragment reads are complete), and only then the fetch of Wx(it) into the same slab
            if (it + 1 < nkt) { stage_taps(kt0 + it + 1, 1 - P); stage_raw(kt0 + it + 1, 1 - P); }
            if (it > 0) { mfma_half(1 - P, 0); mfma_half(1 - P, 1); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it < nkt) {
                stage_w(kt0 + it, P);
                if (cdir) conv_pass(kt0 + it, P, std::true_type{}, false);
                else conv_pass(kt0 + it, P, std::false_type{}, false);
            }
            if (it > 0) copy_out(kt0 + it - 1, 1 - P);
            return;
        }
        if (it < nkt) stage_w(kt0 + it, P);
        if (it + 1 < nkt) { stage_taps(kt0 + it + 1, 1 - P); stage_raw(kt0 + it + 1, 1 - P); }
        if (it < nkt) {
            if (cdir) conv_pass(kt0 + it, P, std::true_type{}, it > 0);
            else conv_pass(kt0 + it, P, std::false_type{}, it > 0);
        } else {
            mfma_half(1 - P, 0);
            mfma_half(1 - P, 1);
            copy_out(kt0 + it - 1, 1 - P);
        }
    };
    for (int it = 0; it <= nkt; it += 2) {
        iteration(it, std::integral_constant<int, 0>{});
        if (it + 1 <= nkt) iteration(it + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: lane (li = row, lg): fragment j -> columns j*16 + lg*4 .. +3 --------------------------------
    const ConvxDir dd = cdir ? d1 : d0;
    if (part != nullptr) {          // K-split: raw partial sums, reduced by convx_reduce_kernel
        const int64_t rows_all = (int64_t)S * L;
        float* pb = part + ((int64_t)blockIdx.y * 2 + cdir) * rows_all * (16 * NJ);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = t0 + mq * 32 + i * 16 + li;
            if (t >= L) continue;
            float* pr = pb + (row0 + t) * (16 * NJ);
#pragma unroll
            for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(pr + j * 16 + lg * 4) = acc[i][j];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = t0 + mq * 32 + i * 16 + li;
        if (t >= L) continue;
        const int64_t row = row0 + t;
        T* dl = (T*)dd.dtl + row * (16 * (NJ - 2));
        float* bcr = dd.bc + row * 32;
#pragma unroll
        for (int j = 0; j < NJ - 2; ++j) {
            if constexpr (sizeof(T) == 2) {
                u32x2 v = {pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
                *reinterpret_cast<u32x2*>(dl + j * 16 + lg * 4) = v;
            } else if (dd.dtl_split) {
                constexpr int RPW = 16 * (NJ - 2);
                bf16_t* ds = (bf16_t*)dd.dtl + row * (2 * RPW) + j * 16 + lg * 4;
                const u32x2 hi = {pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
                const u32x2 lo = {pack_bf16x2(acc[i][j][0] - bf16lo_to_f32(hi[0]), acc[i][j][1] - bf16hi_to_f32(hi[0])),
                                  pack_bf16x2(acc[i][j][2] - bf16lo_to_f32(hi[1]), acc[i][j][3] - bf16hi_to_f32(hi[1]))};
                *reinterpret_cast<u32x2*>(ds) = hi;
                *reinterpret_cast<u32x2*>(ds + RPW) = lo;
            } else {
                *reinterpret_cast<f32x4*>(dl + j * 16 + lg * 4) = acc[i][j];
            }
        }
#pragma unroll
        for (int j = NJ - 2; j < NJ; ++j) {
            f32x4 v = {Elem<T>::round(acc[i][j][0]), Elem<T>::round(acc[i][j][1]), Elem<T>::round(acc[i][j][2]),
                       Elem<T>::round(acc[i][j][3])};
            *reinterpret_cast<f32x4*>(bcr + (j - (NJ - 2)) * 16 + lg * 4) = v;
        }
    }
}

// K-split reduction: x_dbl[dir][row][col] = sum_ks part[ks][dir][row][col] (index order), written as the epilogue above writes it:
// columns [0, 16 (NJ - 2)) -> dt_low (model dtype, or the bf16 [hi | lo] form), the last 32 -> B_t | C_t fp32 rounded to the model dtype
template <typename T>
__global__ __launch_bounds__(256) void convx_reduce_kernel(const float* __restrict__ part, int KS, int64_t rows, int XPW, ConvxDir d0, ConvxDir d1) {
    const int q4 = XPW / 4;
    const int64_t total = 2 * rows * q4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % q4) * 4;
        const int64_t rr = i / q4;
        const int dir = (int)(rr / rows);
        const int64_t row = rr - (int64_t)dir * rows;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) v += *reinterpret_cast<const f32x4*>(part + (((int64_t)ks * 2 + dir) * rows + row) * XPW + c4);
        const ConvxDir dd = dir ? d1 : d0;
        const int RPW = XPW - 32;
        if (c4 < RPW) {
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<u32x2*>((T*)dd.dtl + row * RPW + c4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            } else if (dd.dtl_split) {
                bf16_t* ds = (bf16_t*)dd.dtl + row * (2 * RPW) + c4;
                const u32x2 hi = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                const u32x2 lo = {pack_bf16x2(v[0] - bf16lo_to_f32(hi[0]), v[1] - bf16hi_to_f32(hi[0])),
                                  pack_bf16x2(v[2] - bf16lo_to_f32(hi[1]), v[3] - bf16hi_to_f32(hi[1]))};
                *reinterpret_cast<u32x2*>(ds) = hi;
                *reinterpret_cast<u32x2*>(ds + RPW) = lo;
            } else {
                *reinterpret_cast<f32x4*>((T*)dd.dtl + row * RPW + c4) = v;
            }
        } else {
            *reinterpret_cast<f32x4*>(dd.bc + row * 32 + (c4 - RPW)) = f32x4{Elem<T>::round(v[0]), Elem<T>::round(v[1]), Elem<T>::round(v[2]), Elem<T>::round(v[3])};
        }
    }
}

// K-split policy (a function of the launch shape only): split when the row tiles alone leave at least half of the chip's 256 CUs
// idle - the 512-bp windows of the reference's interactive use in batches of up to 8 at l32 - into the largest divisor KS of the
// K-tile count with tiles * KS <= 256 and at least 2 K-tiles per block.  One window at l32: 8 blocks x 32 K-tiles -> 128 blocks x 2.
int convx_ksplit(int S, int L, int E, int dt) {
    const int KC = CX_ROWB / (dt == BF16 ? 2 : 4);
    const int nkt = E / KC;
    const int64_t tiles = (int64_t)S * ((L + CX_ROWS - 1) / CX_ROWS);
    static const int max_tiles = [] { const char* v = dev_env("PCAD_CX_SPLIT_TILES"); return v ? atoi(v) : 64; }();     // PCAD_DEV=1 A/B knob
    if (tiles <= 0 || tiles > max_tiles) return 1;
    int best = 1;
    for (int ks = 2; ks <= nkt / 2; ++ks)
        if (nkt % ks == 0 && tiles * ks <= 256) best = ks;
    return best;
}

size_t convx_split_bytes(int S, int L, int E, int dt, int Rp, int policy_S) {
    const int ks = convx_ksplit(policy_S > 0 ? policy_S : S, L, E, dt);
    return ks > 1 ? (size_t)ks * 2 * (size_t)S * L * (Rp + 32) * sizeof(float) : 0;
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

// x_proj weight for the XS form: src fp32 [rows, E] (ld) -> dst bf16 [rows, 2E]: K-tile kt (channels 32 kt .. 32 kt + 31) becomes the
// 64 bf16 [hi (32) | lo (32)], hi = bf16(w), lo = bf16(w - hi)
__global__ __launch_bounds__(256) void pack_convx_wsplit_kernel(const float* __restrict__ src, int64_t ld, bf16_t* __restrict__ dst, int rows, int E) {
    const int64_t total = (int64_t)rows * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / E), c = (int)(i - (int64_t)r * E);
        const float v = src[(int64_t)r * ld + c];
        const bf16_t hi = f32_to_bf16(v);
        bf16_t* d = dst + (int64_t)r * 2 * E + (c >> 5) * 64 + (c & 31);
        d[0] = hi;
        d[32] = f32_to_bf16(v - bf16_to_f32(hi));
    }
}

hipError_t launch_pack_convx_wsplit(const float* src, int64_t ld, void* dst, int rows, int E, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (E % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_convx_wsplit_kernel, dim3(256), dim3(256), 0, s, src, ld, (bf16_t*)dst, rows, E);
    return hipGetLastError();
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
                        const void* Wx1, void* xc1, void* dtl1, float* bc1, int S, int L, int E, int dt, hipStream_t s, int Rp, bool dtl_split,
                        bool w_split, float* part_ws, int policy_S) {
    if (S <= 0 || L <= 0) return hipSuccess;
    if ((dtl_split || w_split) && dt != F32) return hipErrorInvalidValue;
    const int esz = dt == BF16 ? 2 : 4;
    if ((E * esz) % CX_ROWB || (Rp != 64 && Rp != 96)) return hipErrorInvalidValue;
    if (((int64_t)S * L + 16) * E * esz >= ((int64_t)1 << 32)) return hipErrorInvalidValue;    // unsigned 32-bit in-tensor offsets
    ConvxDir d0{Wx0, xc0, dtl0, bc0, dtl_split ? 1 : 0}, d1{Wx1, xc1, dtl1, bc1, dtl_split ? 1 : 0};
    const int tiles = S * ((L + CX_ROWS - 1) / CX_ROWS);
    const bool zfill = L % 8 == 0;
    const int ks = part_ws ? convx_ksplit(policy_S > 0 ? policy_S : S, L, E, dt) : 1;
    float* part = ks > 1 ? part_ws : nullptr;
#define PCAD_CONVX(T, Z, NJ_)                                                                                         \
    do {                                                                                                                \
        auto k = convx_kernel<T, Z, NJ_>;                                                                               \
        if constexpr (std::is_same<T, float>::value) { if (w_split) k = convx_kernel<T, Z, NJ_, true>; }                \
        if (hipError_t ae = ensure_dynamic_lds((const void*)k, CxGeom<NJ_>::LDS)) return ae;                            \
        hipLaunchKernelGGL(k, dim3((unsigned)tiles, (unsigned)ks), dim3(CX_THREADS), CxGeom<NJ_>::LDS, s, (const T*)x, convw, d0, d1, S, L, E, part); \
    } while (0)
#define PCAD_CONVX_NJ(NJ_)                                                                                            \
    do {                                                                                                                \
        if (dt == BF16) { if (zfill) PCAD_CONVX(bf16_t, true, NJ_); else PCAD_CONVX(bf16_t, false, NJ_); }              \
        else { if (zfill) PCAD_CONVX(float, true, NJ_); else PCAD_CONVX(float, false, NJ_); }                           \
    } while (0)
    if (Rp == 64) PCAD_CONVX_NJ(6); else PCAD_CONVX_NJ(8);
#undef PCAD_CONVX_NJ
#undef PCAD_CONVX
    if (part) {
        const int64_t rows = (int64_t)S * L;
        const unsigned nb = (unsigned)((2 * rows * ((Rp + 32) / 4) + 255) / 256);
        if (dt == BF16) hipLaunchKernelGGL(convx_reduce_kernel<bf16_t>, dim3(nb > 4096 ? 4096 : nb), dim3(256), 0, s, part, ks, rows, Rp + 32, d0, d1);
        else hipLaunchKernelGGL(convx_reduce_kernel<float>, dim3(nb > 4096 ? 4096 : nb), dim3(256), 0, s, part, ks, rows, Rp + 32, d0, d1);
    }
    return hipGetLastError();
}

}  // namespace pcad
