// Selective-scan recurrence (Mamba-v1 S6) with the dt_proj contraction fused in, token-major, one
// direction per launch.
//
// Replaces selective_scan_cuda.fwd behind mamba_ssm's selective_scan_fn / mamba_inner_fn and the
// `delta = dt_proj.weight @ x_dbl[:, :R]` GEMM in front of it (mamba-ssm 2.2.2):
//     delta = softplus(round(dt_low . Wdt^T) + delta_bias)
//     h_t   = exp(delta_t * A) (.) h_{t-1} + delta_t * B_t * u_t          (A in R^{E x 16}, B_t, C_t in R^16)
//     y_t   = <h_t, C_t> + D * u_t ;   out_t = y_t * silu(z_t)
// The reference maps one CUDA block to a (batch, channel) pair and scans along time with a block scan.
// With 2B*E independent (strand, channel) recurrences per layer there is no need to parallelise time on
// MI355X: a wave owns 64 consecutive channels of one strand, lane = channel, the 16 states live in VGPRs
// as 8 packed pairs (v_pk_mul/v_pk_fma_f32) and time is walked sequentially with zero cross-lane traffic.
//
//   * delta never exists in HBM: per 32-timestep block the wave computes its [32 t x 64 ch] tile on the
//     matrix cores (2 x v_mfma_f32_32x32x16_bf16 per 16 of K; fp32 model: v_mfma_f32_32x32x2_f32), 16
//     v_permlane32_swap move every lane's own channel into place, softplus is applied once and the 32 values
//     are parked in a wave-private LDS slab (used as dynamically indexed spill space, lane-linear, no barrier).
//   * B_t / C_t (shared by all channels) are fp32 rows written by the x_proj GEMM epilogue; the address is
//     wave-uniform so they arrive through the scalar unit (s_load_dwordx16) as SGPR operands of the VALU ops.
//   * u / z (/ y of the other direction) are 128-byte coalesced row reads, prefetched one 4-step chunk ahead.
//   * The reverse direction walks t = L-1..0 on the same rows (no flipped copy); `accumulate` (1: after the gate, each
//     direction rounded as the reference does; 2: before the gate, one SiLU(z) for both directions) adds the forward
//     direction's output (BiMambaWrapper strategy "add", tied out_proj folded by linearity).
//   * Block order is XCD-affine (round 6): the E/64 channel-block waves of a strand all run on ONE XCD (block b -> XCD b % 8 is the
//     observed dispatch rule; speed only), so a strand's B_t | C_t rows and dt_low rows are fetched from HBM into one L2 instead of
//     eight, and the rows of the NEXT 32-step block are pulled into that L2 one block ahead (two dword loads per lane whose values
//     are never used): the per-step scalar loads then hit L2.  At full occupancy (4 waves per SIMD) the other waves hid those misses;
//     with few long strands (PlantCAD2's 8 192-bp windows: 2 waves per SIMD) they were 45 % of a wave's cycles
//     (profiles/r06a_kpmc_scan_pc2m_8192.txt: SQ_WAIT_ANY).
// VALU/transcendental bound (measured on gfx950: v_exp_f32 8.5, v_fma_f32 3.7, v_pk_fma_f32 5.2 cycles per
// wave-instruction at 4 waves/SIMD, not overlapping): ~19 cycles per (t, channel, state).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int NSTATE = 16;
constexpr int TB = 32;    // timesteps per delta tile
#ifndef PCAD_SCAN_CH
#define PCAD_SCAN_CH 4
#endif
#ifndef PCAD_SCAN_OCC
#define PCAD_SCAN_OCC 4
#endif
#ifndef PCAD_SCAN_OCC96
#define PCAD_SCAN_OCC96 PCAD_SCAN_OCC
#endif
constexpr int CH = PCAD_SCAN_CH;     // timesteps per prefetch chunk (4 or 8: a chunk must not cross an 8-row block of the blocked layout)

__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }

// Row accesses through a wave-uniform buffer descriptor: SGPR row offset + constant per-lane byte offset, so a load
// or store costs no VALU address arithmetic (buffer_load_ushort v, v_lane, s[rsrc], s_row offen).
typedef __attribute__((address_space(8))) void* rsrc_t;
__device__ __forceinline__ auto make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// every row of u / z / y is read or written exactly once per launch and the tensors (2 GiB each at the benchmark chunk) are far
// larger than the caches: the accesses carry the streaming hint (slc / "nt"), which keeps them from displacing the dt_proj, B|C and
// weight lines other waves re-use (+0.5 % end to end in two interleaved same-box pairs, and the GEMMs after the scan run
// faster; loads-only or stores-only: no effect)
constexpr int SCAN_AUX = 2;
template <typename T> struct BufIO;
template <> struct BufIO<bf16_t> {
    template <typename R> static __device__ __forceinline__ bf16_t load(R r, int voff, uint32_t soff) {
        return (bf16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, (int)soff, SCAN_AUX);
    }
    template <typename R> static __device__ __forceinline__ void store(bf16_t v, R r, int voff, uint32_t soff) {
        __builtin_amdgcn_raw_buffer_store_b16(v, r, voff, (int)soff, SCAN_AUX);
    }
};
template <> struct BufIO<float> {
    template <typename R> static __device__ __forceinline__ float load(R r, int voff, uint32_t soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (int)soff, SCAN_AUX));
    }
    template <typename R> static __device__ __forceinline__ void store(float v, R r, int voff, uint32_t soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, (int)soff, SCAN_AUX);
    }
};

// ---- delta tile: acc[half][r] <- sum_k dtl[t0 + tt][k] * Wdt[c][k],  tt = (r&3) + 8*(r>>2) + 4*half ------
// After the MFMAs lane l holds column (l & 31) of channel tile q (channels 32q..32q+31) for the rows
// 4*(l>>5) + {0..3, 8..11, 16..19, 24..27}; swapping the upper half of tile 0 with the lower half of tile 1
// (v_permlane32_swap) leaves every lane with its own channel: acc0 = rows {0-3,8-11,..}, acc1 = rows {4-7,..}.
template <typename T> struct DeltaTile;

template <> struct DeltaTile<bf16_t> {
    static __device__ __forceinline__ void run(const bf16_t* __restrict__ dtl, int64_t lddt, int64_t row_base, int t0,
                                               int L, const bf16_t* __restrict__ Wdt, int c0, int Rp, int lane,
                                               f32x16& acc0, f32x16& acc1) {
        const int tr = max(0, min(t0 + (lane & 31), L - 1));
        const bf16_t* arow = dtl + (row_base + tr) * lddt + (lane >> 5) * 8;
        const bf16_t* b0 = Wdt + (int64_t)(c0 + (lane & 31)) * Rp + (lane >> 5) * 8;
        const bf16_t* b1 = b0 + (int64_t)32 * Rp;
#pragma unroll 4
        for (int k = 0; k < Rp; k += 16) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(arow + k);
            const u32x4 w0 = *reinterpret_cast<const u32x4*>(b0 + k);
            const u32x4 w1 = *reinterpret_cast<const u32x4*>(b1 + k);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                           __builtin_bit_cast(bf16x8_t, w0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                           __builtin_bit_cast(bf16x8_t, w1), acc1, 0, 0, 0);
        }
    }
};

// Rp == 64 (every PlantCaduceus size: dt_rank <= 64) or 96 (PlantCAD2 Large): the dt_low operand of the NEXT block (Rp / 16 x 16
// bytes per lane) is loaded one block ahead and held in registers through the walk, so its HBM/L2 latency is never exposed.
template <int RP> struct DeltaPre { u32x4 a[RP / 16]; };
template <int RP>
__device__ __forceinline__ DeltaPre<RP> delta_prefetch(const bf16_t* __restrict__ dtl, int64_t lddt, int64_t row_base,
                                                       int t0, int L, int lane) {
    const int tr = max(0, min(t0 + (lane & 31), L - 1));
    const bf16_t* arow = dtl + (row_base + tr) * lddt + (lane >> 5) * 8;
    DeltaPre<RP> p;
#pragma unroll
    for (int k = 0; k < RP / 16; ++k) p.a[k] = *reinterpret_cast<const u32x4*>(arow + k * 16);
    return p;
}
template <int RP>
__device__ __forceinline__ void delta_run_pre(const DeltaPre<RP>& p, const bf16_t* __restrict__ Wdt, int c0, int lane,
                                              f32x16& acc0, f32x16& acc1) {
    const bf16_t* b0 = Wdt + (int64_t)(c0 + (lane & 31)) * RP + (lane >> 5) * 8;
    const bf16_t* b1 = b0 + (int64_t)32 * RP;
#pragma unroll
    for (int k = 0; k < RP / 16; ++k) {
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(b0 + k * 16);
        const u32x4 w1 = *reinterpret_cast<const u32x4*>(b1 + k * 16);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, p.a[k]),
                                                       __builtin_bit_cast(bf16x8_t, w0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, p.a[k]),
                                                       __builtin_bit_cast(bf16x8_t, w1), acc1, 0, 0, 0);
    }
}

template <> struct DeltaTile<float> {
    // exact fp32 on v_mfma_f32_32x32x2_f32; lane half kh = l>>5 owns k in [kh*Rp/2, (kh+1)*Rp/2) (same
    // permutation of k on both operands, so the contraction is unchanged).
    static __device__ __forceinline__ void run(const float* __restrict__ dtl, int64_t lddt, int64_t row_base, int t0,
                                               int L, const float* __restrict__ Wdt, int c0, int Rp, int lane,
                                               f32x16& acc0, f32x16& acc1) {
        const int tr = max(0, min(t0 + (lane & 31), L - 1));
        const int kh = (lane >> 5) * (Rp >> 1);
        const float* arow = dtl + (row_base + tr) * lddt + kh;
        const float* b0 = Wdt + (int64_t)(c0 + (lane & 31)) * Rp + kh;
        const float* b1 = b0 + (int64_t)32 * Rp;
        for (int k = 0; k < (Rp >> 1); k += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + k);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(b0 + k);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(b1 + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w0[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w1[j], acc1, 0, 0, 0);
            }
        }
    }
};

// Split-bf16 dt_proj of the fp32 model (api.hip "f32_gemm_split"): dt_low arrives as bf16 [rows, lddt >= 2 R'] = [hi | lo] (written by the
// fused conv + x_proj epilogue) and Wdt as bf16 [E, 2 R'] = [hi | lo] (bind time), hi = bf16(v), lo = bf16(v - hi).  The K walk is the
// wrap-around cursor of the split GEMMs (gemm.hip): (a_hi, w_hi), (a_lo, w_hi), (a_hi, w_lo) - three bf16 products per fp32 product
// on v_mfma_f32_32x32x16_bf16, 3 R' / 16 x 2 MFMAs per 32-step tile instead of R' / 2 x 2 fp32 ones (4x slower each).
struct DeltaTileSplit {
    // R' = 64 (every PlantCaduceus size): every operand chunk is loaded ONCE - a_hi, a_lo, w_hi up front (16 x 16 bytes per lane), then
    // w_lo - and the three products run from registers in the same part-major order as the generic walk below (identical results).
    // The generic walk re-requests a_hi and w_hi for their second product; with 16 waves per CU streaming rows through the 32 KiB L1
    // those re-reads miss it: 1 135 L1 -> L2 read requests per 32-step tile and wave against 128 for the walk's own rows
    // (profiles/r06_kpmcm_scan_f32_split.txt), which also tripled the latency of the per-step scalar B | C loads.
    static __device__ __forceinline__ void run64(const bf16_t* __restrict__ arow, const bf16_t* __restrict__ b0, const bf16_t* __restrict__ b1,
                                                 f32x16& acc0, f32x16& acc1) {
        constexpr int NK = 4, RP = 64;
        u32x4 ah[NK], al[NK], w0[NK], w1[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            ah[k] = *reinterpret_cast<const u32x4*>(arow + k * 16);
            al[k] = *reinterpret_cast<const u32x4*>(arow + RP + k * 16);
            w0[k] = *reinterpret_cast<const u32x4*>(b0 + k * 16);
            w1[k] = *reinterpret_cast<const u32x4*>(b1 + k * 16);
        }
        auto mm = [&](const u32x4& a, const u32x4& x0, const u32x4& x1) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, x0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, x1), acc1, 0, 0, 0);
        };
#pragma unroll
        for (int k = 0; k < NK; ++k) mm(ah[k], w0[k], w1[k]);          // a_hi w_hi
#pragma unroll
        for (int k = 0; k < NK; ++k) mm(al[k], w0[k], w1[k]);          // a_lo w_hi
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            w0[k] = *reinterpret_cast<const u32x4*>(b0 + RP + k * 16);
            w1[k] = *reinterpret_cast<const u32x4*>(b1 + RP + k * 16);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) mm(ah[k], w0[k], w1[k]);          // a_hi w_lo
    }
    static __device__ __forceinline__ void run(const bf16_t* __restrict__ dtl, int64_t lddt, int64_t row_base, int t0,
                                               int L, const bf16_t* __restrict__ Wdt, int c0, int Rp, int lane,
                                               f32x16& acc0, f32x16& acc1) {
        const int tr = max(0, min(t0 + (lane & 31), L - 1));
        const bf16_t* arow = dtl + (row_base + tr) * lddt + (lane >> 5) * 8;
        const bf16_t* b0 = Wdt + (int64_t)(c0 + (lane & 31)) * 2 * Rp + (lane >> 5) * 8;
        const bf16_t* b1 = b0 + (int64_t)32 * 2 * Rp;
#ifndef PCAD_SCAN_TILE_GENERIC      // timing-only A/B: the generic three-pass walk for R' = 64 too
        if (Rp == 64) { run64(arow, b0, b1, acc0, acc1); return; }
#endif
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const bf16_t* ap = arow + (part == 1 ? Rp : 0);
            const int wo = part == 2 ? Rp : 0;
#pragma unroll 4
            for (int k = 0; k < Rp; k += 16) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(ap + k);
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(b0 + wo + k);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(b1 + wo + k);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                               __builtin_bit_cast(bf16x8_t, w0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                               __builtin_bit_cast(bf16x8_t, w1), acc1, 0, 0, 0);
            }
        }
    }
};

// FUSED: delta comes from dt_low . Wdt^T (above);  !FUSED: delta is read from memory like u (operator entry).
// PRE (bf16, FUSED; 0: off, else = Rp, 64 or 96): dt_low operand prefetched one block ahead.
// BLK8: u / y in the blocked layout AND L % 8 == 0 (the engine's case): one scalar block offset per 4-step chunk, the
// per-step +-128 bytes ride in the buffer instruction's immediate offset.
// SEG (long sequences with few strands, launch_scan below): the walk of a strand is cut into G segments of `seg_blocks` blocks that
// run as separate workgroups.  SEG = 1, pass A: from a ZERO state, no output - stores the segment's end state and its sum of delta
// (the product of its decays is exp2(A2 * sum delta));  SEG = 2, pass B: the normal walk of the segment from the true initial state
// that scan_carry_kernel derived from pass A.  SEG = 0: the whole strand in one workgroup (G = 1).
// SPLITY (fp32 engine with "f32_gemm_split", BLK8 only): the output is NOT written as fp32 rows but as out_proj's split-bf16
// operand - ysplit, bf16 [rows8, 2E] blocked = [hi | lo] with hi = bf16(y), lo = bf16(y - hi) (pack.hip launch_split_rows'
// format) - which saves the separate conversion pass over y (read 4E + write 4E bytes per row).
// SEG = 3 / 4 (launch_scan_pair below; "pair" walks of launches with few waves): both directions of a layer run in ONE launch, each
// walking HALF of its strand per launch.  SEG = 3: the first half of the walk from a zero state, WITH output, end state stored;
// SEG = 4: the second half from that state (ACC / HASZ as in the plain walk: it adds the other direction's first-half output).
// Same arithmetic as two plain launches - forward rows [0, L/2) then [L/2, L), reverse rows [L/2, L) then [0, L/2) - at twice the
// waves per launch and no extra pass: what the first launch of a direction leaves in y is exactly what the second launch of the
// OTHER direction accumulates onto.
template <typename T, bool REV, int ACC, bool HASZ, bool FUSED, int PRE, bool BLK8, int SEG = 0, bool ZB = false, bool SPLITY = false>
__device__ __forceinline__ void scan_body(float (*dvs)[64], const int block_id, const int num_blocks,
                                          const T* __restrict__ u, const T* __restrict__ z, int64_t ldz,
                                          const T* __restrict__ dsrc, int64_t ldd,
                                          const T* __restrict__ Wdt, int Rp,
                                          const float* __restrict__ bc,
                                          const float* __restrict__ A2, float a_scale,
                                          const float* __restrict__ Dskip, const float* __restrict__ dbias,
                                          const T* yin, T* y, int L, int E, int uyb, int zblk, int G, int seg_blocks,
                                          float* __restrict__ seg_state, int Lw, bf16_t* __restrict__ ysplit, int dts) {
    // dvs - delta slab [TB + 2][64] in LDS: rows 1..TB hold the block's TB steps; rows 0 and TB + 1 are never-consumed landing rows
    // for the one-step-ahead read at the block's ends, so that read needs no wrap (its address is a per-chunk base + a compile-time offset)
    const int lane = threadIdx.x;
    // 1-D grid of (strands x segments) x (E / 64) blocks in XCD-affine order: block b runs on XCD b % 8 (observed; speed only), so
    // the E / 64 channel blocks of strand-segment 8 q + x are the consecutive blocks j = b / 8 of XCD x = b % 8; a tail of fewer
    // than 8 strand-segments keeps the natural order
    const int ncb = E >> 6;
    const int nss = num_blocks / ncb;                 // strands x segments
    int ss, cb;
    {
        const int b = block_id, full = (nss & ~7) * ncb;
        if (b < full) { const int j = b >> 3, q = j / ncb; ss = q * 8 + (b & 7); cb = j - q * ncb; }
        else { const int r = b - full, q = r / ncb; ss = (nss & ~7) + q; cb = r - q * ncb; }
    }
    const int c0 = cb * 64;
    const int c = c0 + lane;
    constexpr bool PAIR = SEG == 3 || SEG == 4;       // one block per strand: the segment is the launch's (first / second half)
    const int strand = PAIR ? ss : (SEG ? ss / G : ss);
    const int seg = PAIR ? (SEG == 4 ? 1 : 0) : (SEG ? ss - strand * G : 0);
    const int nstrands = PAIR ? nss : (SEG ? nss / G : nss);
    const int64_t row0 = (int64_t)strand * L;

    f2 a2p[NSTATE / 2], hp[NSTATE / 2];
#pragma unroll
    for (int p = 0; p < NSTATE / 2; p += 2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A2 + (int64_t)c * NSTATE + 2 * p);
        a2p[p] = f2{v[0] * a_scale, v[1] * a_scale};
        a2p[p + 1] = f2{v[2] * a_scale, v[3] * a_scale};
    }
    // seg_state: [strand][segment][E][16] states, then [strand][segment][E] delta sums (pass A output); pass B reads its initial
    // state from the same [strand][segment][E][16] block (rewritten in place by scan_carry_kernel)
    float* __restrict__ seg_h = seg_state + (((int64_t)strand * G + seg) * E + c) * NSTATE;
    if constexpr (SEG == 2 || SEG == 4) {
#pragma unroll
        for (int p = 0; p < NSTATE / 2; p += 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(seg_h + 2 * p);
            hp[p] = f2{v[0], v[1]};
            hp[p + 1] = f2{v[2], v[3]};
        }
    } else {
#pragma unroll
        for (int p = 0; p < NSTATE / 2; ++p) hp[p] = f2{0.f, 0.f};
    }
    float dsum = 0.f;
    const float dsk = Dskip[c];
    const float db = dbias[c];

    // Lw <= L: number of walk steps to run (the last layer of a forward that is evaluated at a few positions only needs the
    // walk up to the furthest of them: rows past it are left as they were and are never read).  Addresses still clamp at L.
    const int nblk = (Lw + TB - 1) / TB;
    // walk order: step s = 0..L-1 visits t = REV ? L-1-s : s.  Block b covers walk steps [b*TB, (b+1)*TB), i.e.
    // memory rows [tb0, tb0+TB) with tb0 = REV ? L-(b+1)*TB : b*TB (tb0 < 0 / rows >= L are clamped loads whose
    // delta is never consumed); blocks are aligned in WALK space so the CH-step prefetch chunks never straddle.
    // Steps past the end of the sequence (last chunk when L % CH != 0) run on clamped rows and only their store
    // is suppressed, which keeps the chunk body branch-free.
    // Wave-uniform per-strand bases + 32-bit in-strand offsets (L * ld < 2^31): SGPR base, lane offset in a VGPR.
    const uint32_t esz = (uint32_t)sizeof(T);
    const uint32_t rowE = (uint32_t)E * esz, rowZ = (uint32_t)ldz * esz, rowD = (uint32_t)ldd * esz;
    // u and y: plain rows (descriptor based at the strand, offset t * rowE) or the blocked layout the GEMMs stream
    // (descriptor based at the tensor, offset = blocked_off of the whole-tensor row; the wave's 64 channels are one or two
    // 128-byte pieces, so the per-lane part is a constant)
    const uint32_t pieces = rowE >> 7;
    const uint32_t tot_rows = ((uint32_t)nstrands * (uint32_t)L + 7u) & ~7u;
    const bool blk = BLK8 || uyb;       // BLK8 instantiations: known at compile time
    const auto u_r = blk ? make_rsrc(u, tot_rows * rowE) : make_rsrc(u + row0 * E + c0, (uint32_t)L * rowE);
    const auto y_r = blk ? make_rsrc(y, tot_rows * rowE) : make_rsrc(y + row0 * E + c0, (uint32_t)L * rowE);   // also the ACC input
    const bool zb_ = ZB || (HASZ && blk && zblk);      // z: separate tensor in the same blocked layout as u -> same offsets (ZB: known at compile time)
    const auto z_r = zb_ ? make_rsrc(z, tot_rows * rowE) : make_rsrc(HASZ ? z + row0 * ldz + c0 : u, (uint32_t)L * rowZ);
    const auto d_r = make_rsrc(FUSED ? u : dsrc + row0 * ldd + c0, (uint32_t)L * rowD);
    const float* __restrict__ bc_s = bc + row0 * (2 * NSTATE);
    const int voff = lane * (int)esz;
    const int voff_uy = blk ? (int)((((uint32_t)voff >> 7) << 10) + ((uint32_t)voff & 127u)) : voff;
    const uint32_t c0b = (uint32_t)c0 * esz;
    auto off_uy = [&](uint32_t t) -> uint32_t {
        if (blk) {
            const uint32_t r = (uint32_t)row0 + t;
            return (((r >> 3) * pieces + (c0b >> 7)) << 10) + ((r & 7u) << 7);
        }
        return t * rowE;
    };
    // scalar offset / per-lane offset of walk step s0 + i (s0 a multiple of CH).  BLK8: a CH-step chunk (CH = 4, aligned
    // in walk space, rows s*L + t with L % 8 == 0) never crosses an 8-row block: scalar = offset of the chunk's lowest row,
    // and the step's +128*k goes into the per-lane offset as a compile-time constant (folded into the immediate).
    auto uy_soff = [&](int s0, int i) -> uint32_t {
        if constexpr (BLK8) {
            const int slo = min(REV ? s0 + CH - 1 : s0, L - 1);
            return off_uy((uint32_t)(REV ? (L - 1 - slo) : slo));
        } else {
            const int sc = min(s0 + i, L - 1);
            return off_uy((uint32_t)(REV ? (L - 1 - sc) : sc));
        }
    };
    auto uy_voff = [&](int i) -> int { return BLK8 ? voff_uy + 128 * (REV ? CH - 1 - i : i) : voff_uy; };
    // SPLITY: rows of 2E bf16 in the blocked layout; the wave's 64 channels are exactly one 128-byte piece of each half
    const uint32_t ys_pieces = (2u * (uint32_t)E * 2u) >> 7, ys_half = ((uint32_t)E >> 6) << 10;        // pieces per row; bytes from the hi to the lo piece
    const auto ys_r = make_rsrc(SPLITY ? (const void*)ysplit : (const void*)u, SPLITY ? tot_rows * 2u * (uint32_t)E * 2u : 4u);
    auto ys_soff = [&](int s0) -> uint32_t {            // scalar offset of the chunk's lowest row (BLK8: a chunk never crosses an 8-row block)
        const int slo = min(REV ? s0 + CH - 1 : s0, L - 1);
        const uint32_t r = (uint32_t)row0 + (uint32_t)(REV ? (L - 1 - slo) : slo);
        return (((r >> 3) * ys_pieces + ((uint32_t)c0 >> 6)) << 10) + ((r & 7u) << 7);
    };
    auto ys_voff = [&](int i) -> int { return lane * 2 + 128 * (REV ? CH - 1 - i : i); };

    T ub[CH], zb[CH], yb[CH], dr[CH];      // RAW prefetched values: converted at use, so no early vmcnt wait
    auto tclamp = [&](int s) { const int sc = min(s, L - 1); return REV ? (L - 1 - sc) : sc; };
    auto load_chunk = [&](int s0, T (&uu)[CH], T (&zz)[CH], T (&yy)[CH], T (&dd)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const uint32_t t = (uint32_t)tclamp(s0 + i);
#ifdef PCAD_SCAN_HOT      // timing-only ablation: every row load comes from the first 64 KiB of its tensor (cache hits)
            const uint32_t so = uy_soff(s0, i) & 0xffffu;
#else
            const uint32_t so = uy_soff(s0, i);
#endif
            uu[i] = BufIO<T>::load(u_r, uy_voff(i), so);
            if constexpr (HASZ && ZB) zz[i] = BufIO<T>::load(z_r, uy_voff(i), so);
            else if constexpr (HASZ) zz[i] = zb_ ? BufIO<T>::load(z_r, uy_voff(i), so) : BufIO<T>::load(z_r, voff, t * rowZ);
            if constexpr (ACC != 0) yy[i] = BufIO<T>::load(y_r, uy_voff(i), so);
            if constexpr (!FUSED) dd[i] = BufIO<T>::load(d_r, voff, t * rowD);
        }
    };
    const int b_begin = SEG ? seg * seg_blocks : 0;                        // first block (walk space) of this workgroup
    const int s_first = b_begin * TB;
    load_chunk(s_first, ub, zb, yb, dr);

    // B_t | C_t of the step being computed (SGPRs), software-pipelined one step ahead of the VALU work
    f2 bcc[NSTATE];
    auto load_bc = [&](int s, f2 (&dst)[NSTATE]) {
        const f2* __restrict__ r = reinterpret_cast<const f2*>(bc_s + (uint32_t)tclamp(s) * (uint32_t)(2 * NSTATE));
#pragma unroll
        for (int p = 0; p < NSTATE; ++p) dst[p] = r[p];
    };
    load_bc(s_first, bcc);

    // one recurrence step on raw inputs (uraw, zraw, yraw, draw) at walk step s
    float dv_cur = 0.f;    // FUSED: delta of the step about to run, read from LDS one step ahead
    // dvp: this lane's column of the delta slab at the chunk's base row (walk step s0 of the block, k0 = s0 - s_begin):
    //   forward &dvs[k0][lane], reverse &dvs[TB - CH - k0][lane]; the NEXT step's delta of chunk step i is a compile-time
    //   number of rows from there (slab row = block row + 1; at the block's ends the landing rows are read and never consumed)
    // uv / zv / yv: this step's u, z and (ACC) other-direction y as fp32.  Returns the value to store (SEG == 1: nothing).
    auto step = [&](int s, int i, const float* dvp, float uv, float zv, float yprev, T draw) -> float {
        f2 bcn[NSTATE];
        load_bc(s + 1, bcn);
        float dv, dv_next = 0.f;
        if constexpr (FUSED) {
            dv = dv_cur;
            dv_next = dvp[(REV ? CH - 1 - i : i + 2) * 64];
        } else {
            dv = softplus(Elem<T>::to_f32(draw) + db);
        }
        const float du = dv * uv;
        const f2 dv2 = {dv, dv}, du2 = {du, du};
        // D skip (and, gated-once reverse pass, the other direction's output) seed the accumulation: one fma, no separate add
        f2 yacc = {ACC == 2 ? __builtin_fmaf(dsk, uv, yprev) : dsk * uv, 0.f};
        if constexpr (SEG == 1) dsum += dv;
#pragma unroll
        for (int p = 0; p < NSTATE / 2; ++p) {
            const f2 e0 = dv2 * a2p[p];
            const f2 a0 = {exp2_hw(e0[0]), exp2_hw(e0[1])};
            hp[p] = a0 * hp[p] + du2 * bcc[p];
            if constexpr (SEG != 1) yacc = hp[p] * bcc[NSTATE / 2 + p] + yacc;
        }
        float yv = 0.f;
        if constexpr (SEG != 1) {
            yv = yacc[0] + yacc[1];
            if constexpr (HASZ) yv *= silu(zv);
            if constexpr (ACC == 1) yv = Elem<T>::round(yv) + yprev;    // each direction is rounded, then summed
        }
#pragma unroll
        for (int p = 0; p < NSTATE; ++p) bcc[p] = bcn[p];
        dv_cur = dv_next;
        return yv;
    };

    DeltaPre<PRE ? PRE : 16> pre;
    if constexpr (PRE != 0) pre = delta_prefetch<PRE>((const bf16_t*)dsrc, ldd, row0, REV ? (L - (b_begin + 1) * TB) : b_begin * TB, L, lane);

    // L2 prefetch of the NEXT block's B_t | C_t rows (32 x 128 bytes, contiguous): one dword per 32-byte sector, values never used.
    // They are "consumed" by an empty asm at the top of the next block (a full 32-step walk later: long landed), which is what keeps
    // the registers reserved until the loads have returned.
    float pfb0 = 0.f, pfb1 = 0.f;
    const int b_end = SEG ? min(nblk, b_begin + seg_blocks) : nblk;
    for (int b = b_begin; b < b_end; ++b) {
        const int tb0 = REV ? (L - (b + 1) * TB) : b * TB;
#ifndef PCAD_SCAN_NOPF      // timing-only ablation: without the L2 prefetch
        // (B | C rows only: also prefetching the fp32 split model's dt_low rows - six more gather loads and registers per block at the
        // 128-VGPR cap - cost its scans 10 %, 3.30 -> 3.63 ms per launch; with the B | C rows alone they gain 1 %: profiles/r06_f32_split_ab.txt)
        if constexpr (FUSED) {
            asm volatile("" ::"v"(pfb0), "v"(pfb1));
            const int tbn = max(0, min(REV ? (L - (b + 2) * TB) : (b + 1) * TB, L - 1));          // first row of the next block in memory order (clamped)
            {
                const int lim = L * (2 * NSTATE * 4) - 4;                                          // last dword of the strand's rows
                const char* pb = reinterpret_cast<const char*>(bc_s);
                pfb0 = *reinterpret_cast<const float*>(pb + min(tbn * (2 * NSTATE * 4) + lane * 32, lim));
                pfb1 = *reinterpret_cast<const float*>(pb + min(tbn * (2 * NSTATE * 4) + (64 + lane) * 32, lim));
            }
        }
#endif
        if constexpr (FUSED) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            if constexpr (PRE != 0) {
                delta_run_pre<PRE>(pre, (const bf16_t*)Wdt, c0, lane, acc0, acc1);
                // next block's operand: in flight during this block's 32-step walk (rows clamp at the sequence ends)
                pre = delta_prefetch<PRE>((const bf16_t*)dsrc, ldd, row0, REV ? (L - (b + 2) * TB) : (b + 1) * TB, L, lane);
            } else if constexpr (std::is_same<T, float>::value) {
                // dts (fp32 engine with "f32_gemm_split"): dt_low bf16 [rows, 2 R'] = [hi | lo], Wdt bf16 [E, 2 R'] = [hi | lo], Rp = R':
                // 24 bf16 MFMAs instead of 64 (4x slower) fp32 ones per 32-step tile at R' = 64 (wave-uniform branch, once per tile)
                if (dts) DeltaTileSplit::run(reinterpret_cast<const bf16_t*>(dsrc), ldd, row0, tb0, L, reinterpret_cast<const bf16_t*>(Wdt), c0, Rp, lane, acc0, acc1);
                else DeltaTile<T>::run(dsrc, ldd, row0, tb0, L, Wdt, c0, Rp, lane, acc0, acc1);
            } else {
                DeltaTile<T>::run(dsrc, ldd, row0, tb0, L, Wdt, c0, Rp, lane, acc0, acc1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // lanes 32..63 of acc0[r] <-> lanes 0..31 of acc1[r]
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[r]), __float_as_uint(acc1[r]),
                                                                 false, false);
                const int tt = (r & 3) + 8 * (r >> 2);
                // dt_proj output rounded to the model dtype (as F.linear returns it), + bias, softplus: both values on the packed pipe
                const f2 dl = softplus2(Elem<T>::round2(f2{__uint_as_float(sw[0]), __uint_as_float(sw[1])}) + f2{db, db});
                dvs[tt + 1][lane] = dl[0];
                dvs[tt + 5][lane] = dl[1];
            }
        }
        const int s_begin = b * TB;                                      // first walk step of the block
        const int s_end = min(Lw, s_begin + TB);                         // one past the last
        if constexpr (FUSED) dv_cur = dvs[REV ? TB : 1][lane];          // block row of the first walk step: 0 forward, TB - 1 reverse
        auto dv_base = [&](int s0) -> const float* { return &dvs[REV ? TB - CH - (s0 - s_begin) : (s0 - s_begin)][lane]; };
        auto run_step = [&](int s, int i, const float* dvp, uint32_t oy, int vy, T uraw, T zraw, T yraw, T draw, uint32_t oys) {
            const float yv0 = step(s, i, dvp, Elem<T>::to_f32(uraw), HASZ ? Elem<T>::to_f32(zraw) : 0.f,
                                   ACC != 0 ? Elem<T>::to_f32(yraw) : 0.f, draw);
            if constexpr (SPLITY && SEG != 1) {
                // y as the fp32 VALUE the plain walk would store (no contraction of the gate's multiply into the subtraction below):
                // hi / lo are then exactly what pack.hip's split_rows_kernel makes of the stored y, and the last-layer shortcut
                // (gathered fp32 rows -> split_rows) is bit-identical to the full layer
                float yq = yv0;
                asm volatile("" : "+v"(yq));
                const float yv = yq;
                const uint32_t pk = pack_bf16x2(yv, 0.f);                                  // hi = bf16(y)
                const bf16_t hi = (bf16_t)(pk & 0xffffu);
                const bf16_t lo = f32_to_bf16(yv - bf16lo_to_f32(pk));                       // lo = bf16(y - hi)
#ifdef PCAD_SCAN_HOTSTORE  // timing-only ablation: every row store goes to the first 64 KiB of its tensor (results are garbage)
                BufIO<bf16_t>::store(hi, ys_r, ys_voff(i), oys & 0xffffu);
                BufIO<bf16_t>::store(lo, ys_r, ys_voff(i), (oys & 0xffffu) + ys_half);
            } else if constexpr (SEG != 1) BufIO<T>::store(Elem<T>::from_f32(yv0), y_r, vy, oy & 0xffffu);
#else
                BufIO<bf16_t>::store(hi, ys_r, ys_voff(i), oys);
                BufIO<bf16_t>::store(lo, ys_r, ys_voff(i), oys + ys_half);
            } else if constexpr (SEG != 1) BufIO<T>::store(Elem<T>::from_f32(yv0), y_r, vy, oy);
#endif
        };
        auto run_chunk = [&](int s0, T (&uu)[CH], T (&zz)[CH], T (&yy)[CH], T (&dd)[CH]) {
            const float* dvp = dv_base(s0);
            const uint32_t oys = SPLITY ? ys_soff(s0) : 0u;
#pragma unroll
            for (int i = 0; i < CH; ++i) run_step(s0 + i, i, dvp, uy_soff(s0, i), uy_voff(i), uu[i], zz[i], yy[i], dd[i], oys);
        };
        int s0 = s_begin;
        // two chunks per iteration on alternating register sets: no copies between the sets, and each set is waited for at
        // its first use (a full chunk of work after its loads were issued) instead of at the end of the previous chunk
        T un[CH], zn[CH], yn[CH], dn[CH];
        for (; s0 + 2 * CH <= s_end; s0 += 2 * CH) {
            load_chunk(s0 + CH, un, zn, yn, dn);
            run_chunk(s0, ub, zb, yb, dr);
            load_chunk(s0 + 2 * CH, ub, zb, yb, dr);
            run_chunk(s0 + CH, un, zn, yn, dn);
        }
        if (s0 + CH <= s_end) {                                          // odd full chunk (only at the end of a sequence)
            load_chunk(s0 + CH, un, zn, yn, dn);
            run_chunk(s0, ub, zb, yb, dr);
#pragma unroll
            for (int i = 0; i < CH; ++i) { ub[i] = un[i]; zb[i] = zn[i]; yb[i] = yn[i]; dr[i] = dn[i]; }
            s0 += CH;
        }
        if (s0 < s_end) {                                                // tail (< CH steps, end of the sequence)
            const float* dvp = dv_base(s0);
#pragma unroll
            for (int i = 0; i < CH - 1; ++i)
                if (s0 + i < s_end) run_step(s0 + i, i, dvp, uy_soff(s0, i), uy_voff(i), ub[i], zb[i], yb[i], dr[i], SPLITY ? ys_soff(s0) : 0u);
        }
    }
    if constexpr (SEG == 1) {
#pragma unroll
        for (int p = 0; p < NSTATE / 2; p += 2)
            *reinterpret_cast<f32x4*>(seg_h + 2 * p) = f32x4{hp[p][0], hp[p][1], hp[p + 1][0], hp[p + 1][1]};
        seg_state[(int64_t)nstrands * G * E * NSTATE + ((int64_t)strand * G + seg) * E + c] = dsum;
    }
    if constexpr (SEG == 3) {             // end state of the first half = initial state of the second (slot of segment 1)
        float* __restrict__ nxt = seg_h + (int64_t)E * NSTATE;
#pragma unroll
        for (int p = 0; p < NSTATE / 2; p += 2)
            *reinterpret_cast<f32x4*>(nxt + 2 * p) = f32x4{hp[p][0], hp[p][1], hp[p + 1][0], hp[p + 1][1]};
    }
}

template <typename T, bool REV, int ACC, bool HASZ, bool FUSED, int PRE, bool BLK8, int SEG = 0, bool ZB = false, bool SPLITY = false>
__global__ __launch_bounds__(64, PRE == 96 ? PCAD_SCAN_OCC96 : PCAD_SCAN_OCC) void scan_kernel(const T* __restrict__ u, const T* __restrict__ z, int64_t ldz,
                                                  const T* __restrict__ dsrc, int64_t ldd,
                                                  const T* __restrict__ Wdt, int Rp,
                                                  const float* __restrict__ bc,
                                                  const float* __restrict__ A2, float a_scale,
                                                  const float* __restrict__ Dskip, const float* __restrict__ dbias,
                                                  const T* yin, T* y, int L, int E, int uyb, int zblk, int G, int seg_blocks,
                                                  float* __restrict__ seg_state, int Lw, bf16_t* __restrict__ ysplit, int dts) {
    __shared__ float dvs[TB + 2][64];
    scan_body<T, REV, ACC, HASZ, FUSED, PRE, BLK8, SEG, ZB, SPLITY>(dvs, (int)blockIdx.x, (int)gridDim.x, u, z, ldz, dsrc, ldd, Wdt, Rp, bc, A2, a_scale,
                                                                    Dskip, dbias, yin, y, L, E, uyb, zblk, G, seg_blocks, seg_state, Lw, ysplit, dts);
}

// Both directions of one layer in ONE launch (engine layouts only: fused dt_proj, blocked u / y / z, L % 64 == 0): blocks [0, n) run
// the forward body on direction 0's operands, blocks [n, 2n) the reverse body on direction 1's.  PHASE 3 / 4 = scan_body's SEG.
template <typename T>
struct ScanDirArgs {
    const T* u;            // conv output of the direction (xc)
    const T* dsrc;         // dt_low
    const T* Wdt;
    const float* bc;
    const float* A2;
    const float* Dskip;
    const float* dbias;
    float* seg_state;      // [S][2][E][16] fp32: slot (strand, 1) carries the state between the two launches
};

// (Every operand is its own __restrict__ kernel argument: B_t | C_t are fetched through the scalar unit only when the compiler can
// prove that the kernel's stores do not clobber them - with the pointers inside a by-value struct it could not, the B | C rows
// arrived through eight vector loads per step and the kernel ran 2.3x slower.)
template <typename T, int ACC, bool HASZ, int PRE, int PHASE, bool SPLITY>
__global__ __launch_bounds__(64, PRE == 96 ? PCAD_SCAN_OCC96 : PCAD_SCAN_OCC) void scan_pair_kernel(
    const T* __restrict__ u0, const T* __restrict__ dsrc0, const T* __restrict__ Wdt0, const float* __restrict__ bc0, const float* __restrict__ A2_0,
    const float* __restrict__ Dskip0, const float* __restrict__ dbias0, float* __restrict__ seg0,
    const T* __restrict__ u1, const T* __restrict__ dsrc1, const T* __restrict__ Wdt1, const float* __restrict__ bc1, const float* __restrict__ A2_1,
    const float* __restrict__ Dskip1, const float* __restrict__ dbias1, float* __restrict__ seg1,
    const T* __restrict__ z, int64_t ldd, int Rp, float a_scale, T* y, int L, int E, int seg_blocks, bf16_t* __restrict__ ysplit, int dts) {
    __shared__ float dvs[TB + 2][64];
    const int half = (int)gridDim.x >> 1;
    if ((int)blockIdx.x < half)
        scan_body<T, false, ACC, HASZ, true, PRE, true, PHASE, HASZ, SPLITY>(dvs, (int)blockIdx.x, half, u0, z, (int64_t)E, dsrc0, ldd, Wdt0, Rp, bc0, A2_0, a_scale,
                                                                             Dskip0, dbias0, y, y, L, E, 1, 1, 2, seg_blocks, seg0, L, ysplit, dts);
    else
        scan_body<T, true, ACC, HASZ, true, PRE, true, PHASE, HASZ, SPLITY>(dvs, (int)blockIdx.x - half, half, u1, z, (int64_t)E, dsrc1, ldd, Wdt1, Rp, bc1, A2_1, a_scale,
                                                                            Dskip1, dbias1, y, y, L, E, 1, 1, 2, seg_blocks, seg1, L, ysplit, dts);
}

// Carry between the segments of a strand, in walk order: the state a segment starts from is
//   h0[g] = exp2(A2 * a_scale * dsum[g-1]) (.) h0[g-1] + h_end[g-1],   h0[0] = 0
// (the product of a segment's per-step decays exp2(delta_t * A2) is exp2(A2 * sum_t delta_t)).  In place: h_end[g] -> h0[g].
__global__ __launch_bounds__(256) void scan_carry_kernel(float* __restrict__ seg_state, const float* __restrict__ A2, float a_scale,
                                                         int S, int G, int E) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (strand, channel)
    if (i >= (int64_t)S * E) return;
    const int strand = (int)(i / E), c = (int)(i - (int64_t)strand * E);
    const float* dsum = seg_state + (int64_t)S * G * E * NSTATE;
    float a[NSTATE], h[NSTATE];
#pragma unroll
    for (int n = 0; n < NSTATE; ++n) { a[n] = A2[(int64_t)c * NSTATE + n] * a_scale; h[n] = 0.f; }
    for (int g = 0; g < G; ++g) {
        float* hs = seg_state + (((int64_t)strand * G + g) * E + c) * NSTATE;
        const float ds = dsum[((int64_t)strand * G + g) * E + c];
#pragma unroll
        for (int n = 0; n < NSTATE; ++n) {
            const float hend = hs[n];
            hs[n] = h[n];                                             // initial state of segment g
            h[n] = exp2_hw(a[n] * ds) * h[n] + hend;
        }
    }
}

template <typename T, bool FUSED, int PRE = 0, bool BLK8 = false, bool ZB = false>
static hipError_t launch_scan_t(const void* u, const void* z, int64_t ldz, const void* dsrc, int64_t ldd,
                                const void* Wdt, int Rp, const float* bc, const float* A2, float a_scale,
                                const float* Dskip, const float* dbias, void* y, int S, int L, int E, bool reverse,
                                int accumulate, hipStream_t s, bool uyb, bool zblk = false, float* seg_ws = nullptr, int walk_len = 0,
                                void* ysplit = nullptr, bool dt_split = false, int policy_S = 0) {
    const int dts = dt_split ? 1 : 0;
    const int Sp = policy_S > 0 ? policy_S : S;          // strands the segment policy is evaluated for (kernels.hpp scan_segment_bytes)
    dim3 grid((unsigned)((int64_t)(E / 64) * S)), block(64);          // 1-D: the kernel maps blocks to (strand, channel block) XCD-affinely
    const bool hz = z != nullptr;
    if (ysplit != nullptr) {
        // out_proj's split-bf16 operand written by the walk itself: the fp32 engine's reverse (gating) launch only
        if constexpr (std::is_same<T, float>::value && FUSED && BLK8 && ZB) {
            const bool seg = seg_ws && scan_segments(Sp, L, E, nullptr) > 1;
            if (!reverse || !hz || (accumulate != 1 && accumulate != 2) || seg || (walk_len > 0 && walk_len < L)) return hipErrorInvalidValue;
#define PCAD_SCAN_SPLITY(ACCM)                                                                                                       \
            hipLaunchKernelGGL((scan_kernel<T, true, ACCM, true, FUSED, PRE, BLK8, 0, true, true>), grid, block, 0, s, (const T*)u, (const T*)z, ldz, \
                               (const T*)dsrc, ldd, (const T*)Wdt, Rp, bc, A2, a_scale, Dskip, dbias, (const T*)y, (T*)y, L, E, (int)uyb, (int)zblk, \
                               1, 0, (float*)nullptr, L, (bf16_t*)ysplit, dts)
            if (accumulate == 2) PCAD_SCAN_SPLITY(2); else PCAD_SCAN_SPLITY(1);
#undef PCAD_SCAN_SPLITY
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
#define PCAD_SCAN_ARGS(ZP) (const T*)u, (const T*)(ZP), ldz, (const T*)dsrc, ldd, (const T*)Wdt, Rp, bc, A2, a_scale, Dskip, dbias, \
                           (const T*)y, (T*)y, L, E, (int)uyb, (int)zblk
#define PCAD_WALK (walk_len > 0 && walk_len < L ? walk_len : L)
    // ---- long strands, few of them: G segments per strand as separate workgroups (pass A, carry, pass B) --------------------
    if constexpr (FUSED) {
        int sb = 0;
        const int G = seg_ws ? scan_segments(Sp, L, E, &sb) : 1;
        const bool combo = (!reverse && accumulate == 0) || (reverse && accumulate == 2 && hz) || (reverse && accumulate == 1 && hz) ||
                           (reverse && accumulate == 0 && hz);
        if (G > 1 && combo) {
            dim3 gseg((unsigned)((int64_t)(E / 64) * S * G));
#define PCAD_SEG(REV, ACC, HZ, SEGM, ZP)                                                                                  \
            hipLaunchKernelGGL((scan_kernel<T, REV, ACC, HZ, FUSED, PRE, BLK8, SEGM, ZB && HZ>), gseg, block, 0, s, PCAD_SCAN_ARGS(ZP), G, sb, seg_ws, L, (bf16_t*)nullptr, dts)
            if (reverse) PCAD_SEG(true, 0, false, 1, nullptr); else PCAD_SEG(false, 0, false, 1, nullptr);
            hipLaunchKernelGGL(scan_carry_kernel, dim3((unsigned)(((int64_t)S * E + 255) / 256)), dim3(256), 0, s, seg_ws, A2, a_scale, S, G, E);
            if (!reverse) { if (hz) PCAD_SEG(false, 0, true, 2, z); else PCAD_SEG(false, 0, false, 2, nullptr); }
            else if (accumulate == 2) PCAD_SEG(true, 2, true, 2, z);
            else if (accumulate == 1) PCAD_SEG(true, 1, true, 2, z);
            else PCAD_SEG(true, 0, true, 2, z);
#undef PCAD_SEG
            return hipGetLastError();
        }
    }
#define PCAD_SCAN(REV, ACC, HZ)                                                                                    \
    hipLaunchKernelGGL((scan_kernel<T, REV, ACC, HZ, FUSED, PRE, BLK8, 0, ZB && HZ>), grid, block, 0, s, PCAD_SCAN_ARGS(z), 1, 0, (float*)nullptr, PCAD_WALK, (bf16_t*)nullptr, dts)
    if (accumulate == 2) {                    // (y_prev + y) * silu(z): the bi-directional sum gated once
        if (!hz) return hipErrorInvalidValue;
        if (reverse) PCAD_SCAN(true, 2, true); else PCAD_SCAN(false, 2, true);
    }
    else if (!reverse && !accumulate) { if (hz) PCAD_SCAN(false, 0, true); else PCAD_SCAN(false, 0, false); }
    else if (!reverse && accumulate) { if (hz) PCAD_SCAN(false, 1, true); else PCAD_SCAN(false, 1, false); }
    else if (reverse && !accumulate) { if (hz) PCAD_SCAN(true, 0, true); else PCAD_SCAN(true, 0, false); }
    else { if (hz) PCAD_SCAN(true, 1, true); else PCAD_SCAN(true, 1, false); }
#undef PCAD_SCAN
#undef PCAD_WALK
#undef PCAD_SCAN_ARGS
    return hipGetLastError();
}

template <typename T, int PRE>
static hipError_t launch_scan_pair_t(const ScanDirection& f, const ScanDirection& r, const void* z, int64_t lddt, int Rp, void* y, int S, int L, int E,
                                     bool gate_each, hipStream_t s, float* ws, void* ysplit, bool dt_split, int phases) {
    const int dts = dt_split ? 1 : 0;
    const int sb = (L / TB) / 2;                                          // blocks (of TB steps) per half: L % (2 TB) == 0
    const size_t per_dir = (size_t)S * 2 * E * NSTATE;                   // floats
    ScanDirArgs<T> d0{(const T*)f.u, (const T*)f.dt_low, (const T*)f.Wdt, f.bc, f.A2, f.Dskip, f.dbias, ws};
    ScanDirArgs<T> d1{(const T*)r.u, (const T*)r.dt_low, (const T*)r.Wdt, r.bc, r.A2, r.Dskip, r.dbias, ws + per_dir};
    dim3 grid((unsigned)((int64_t)2 * S * (E / 64))), block(64);
#define PCAD_PAIR(ACCM, HZ, PH, SPY)                                                                                            \
    hipLaunchKernelGGL((scan_pair_kernel<T, ACCM, HZ, PRE, PH, SPY>), grid, block, 0, s, d0.u, d0.dsrc, d0.Wdt, d0.bc, d0.A2, d0.Dskip, d0.dbias, \
                       d0.seg_state, d1.u, d1.dsrc, d1.Wdt, d1.bc, d1.A2, d1.Dskip, d1.dbias, d1.seg_state, (const T*)z, lddt, Rp, 1.0f, (T*)y, L, E, sb, \
                       (bf16_t*)ysplit, dts)
    // first halves: forward rows [0, L/2), reverse rows [L/2, L) - ungated (or, "gate_each", each gated and rounded)
    if (phases & 1) { if (gate_each) PCAD_PAIR(0, true, 3, false); else PCAD_PAIR(0, false, 3, false); }
    if (!(phases & 2)) return hipGetLastError();
    // second halves on top of the other direction's first-half output, gated
    if constexpr (std::is_same<T, float>::value) {
        if (ysplit) { if (gate_each) PCAD_PAIR(1, true, 4, true); else PCAD_PAIR(2, true, 4, true); return hipGetLastError(); }
    } else if (ysplit) {
        return hipErrorInvalidValue;
    }
    if (gate_each) PCAD_PAIR(1, true, 4, false); else PCAD_PAIR(2, true, 4, false);
#undef PCAD_PAIR
    return hipGetLastError();
}

hipError_t launch_scan_pair(const ScanDirection& fwd, const ScanDirection& rev, const void* z, int64_t lddt, int Rp, void* y, int S, int L, int E,
                            bool gate_each, int dt, hipStream_t s, float* ws, void* ysplit, bool dt_split, int phases) {
    if (S <= 0 || L <= 0) return hipSuccess;
    if (!ws || !z || E % 64 || L % (2 * TB) || Rp <= 0 || Rp % 32) return hipErrorInvalidValue;
    const int esz = dt == BF16 ? 2 : 4;
    if ((E * esz) % 128 || ((int64_t)S * L + 7) / 8 * 8 * E * esz >= ((int64_t)1 << 32)) return hipErrorInvalidValue;      // blocked layouts, 32-bit offsets
    if ((int64_t)L * E * 4 >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    if (dt_split && !(dt == F32 && Rp <= 96 && lddt % 8 == 0 && lddt >= 2 * Rp)) return hipErrorInvalidValue;
    if (ysplit && !(dt == F32 && ((int64_t)S * L + 7) / 8 * 8 * 2 * E * 2 < ((int64_t)1 << 32))) return hipErrorInvalidValue;
    if (dt == BF16) {
        if (lddt % 8) return hipErrorInvalidValue;
        if (Rp == 64) return launch_scan_pair_t<bf16_t, 64>(fwd, rev, z, lddt, Rp, y, S, L, E, gate_each, s, ws, nullptr, false, phases);
        if (Rp == 96) return launch_scan_pair_t<bf16_t, 96>(fwd, rev, z, lddt, Rp, y, S, L, E, gate_each, s, ws, nullptr, false, phases);
        return hipErrorInvalidValue;
    }
    return launch_scan_pair_t<float, 0>(fwd, rev, z, lddt, Rp, y, S, L, E, gate_each, s, ws, ysplit, dt_split, phases);
}

hipError_t launch_scan(const void* u, const void* z, int64_t ldz, const void* delta, const void* dt_low, int64_t lddt,
                       const void* Wdt, int Rp, const float* bc, const float* A2, float a_scale, const float* Dskip,
                       const float* dbias, void* y, int S, int L, int E, bool reverse, int accumulate, int dt,
                       hipStream_t s, bool uyb, bool zblk, float* seg_ws, int walk_len, void* ysplit, bool dt_split, int policy_S) {
    if (zblk && !uyb) return hipErrorInvalidValue;
    if (dt_split && !(dt == F32 && delta == nullptr && Rp % 32 == 0 && Rp <= 96 && lddt % 8 == 0 && lddt >= 2 * Rp)) return hipErrorInvalidValue;   // bf16 [hi | lo] operands
    if (ysplit && !(dt == F32 && delta == nullptr && uyb && zblk && L % 8 == 0 && ((int64_t)S * L + 7) / 8 * 8 * 2 * E * 2 < ((int64_t)1 << 32)))
        return hipErrorInvalidValue;        // the split output exists in the fp32 engine's compile-time-layout instantiation only
    if (S <= 0 || L <= 0) return hipSuccess;
    if (E % 64) return hipErrorInvalidValue;
    if ((int64_t)L * (ldz > E ? ldz : E) * 4 >= ((int64_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit in-strand offsets
    if (uyb && (((int64_t)S * L + 7) / 8 * 8 * E * (dt == BF16 ? 2 : 4) >= ((int64_t)1 << 32) || (E * (dt == BF16 ? 2 : 4)) % 128))
        return hipErrorInvalidValue;     // blocked layout: 32-bit whole-tensor offsets
    const bool fused = delta == nullptr;
    if (fused && (!dt_low || !Wdt || Rp <= 0 || Rp % 32)) return hipErrorInvalidValue;
    if (dt == BF16) {
        if (fused && Rp == 64 && lddt % 8 == 0 && uyb && L % 8 == 0 && zblk)      // the engine's case: every layout known at compile time
            return launch_scan_t<bf16_t, true, 64, true, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
        static const bool nopre96 = dev_env("PCAD_SCAN_NOPRE96") != nullptr;       // PCAD_DEV=1 A/B: the non-prefetching walk instead
        if (!nopre96 && fused && Rp == 96 && lddt % 8 == 0 && uyb && L % 8 == 0 && zblk)      // the engine's case at dt_rank 65..96 (PlantCAD2 Large)
            return launch_scan_t<bf16_t, true, 96, true, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
        if (fused && Rp == 64 && lddt % 8 == 0 && uyb && L % 8 == 0)
            return launch_scan_t<bf16_t, true, 64, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
        if (fused && Rp == 64 && lddt % 8 == 0)
            return launch_scan_t<bf16_t, true, 64>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
        if (fused) return launch_scan_t<bf16_t, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
        return launch_scan_t<bf16_t, false>(u, z, ldz, delta, E, nullptr, 0, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
    }
    // the fp32 engine's case (blocked u / y / z, L % 8 == 0): layouts known at compile time like the bf16 instantiation above (one
    // scalar block offset per 4-step chunk, per-step offsets in the immediates); dt_proj stays on v_mfma_f32_32x32x2_f32, unprefetched
    static const bool f32_generic = dev_env("PCAD_SCAN_F32_GENERIC") != nullptr;       // PCAD_DEV=1 A/B: the run-time-layout instantiation
    if (!f32_generic && fused && uyb && L % 8 == 0 && zblk)
        return launch_scan_t<float, true, 0, true, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, ysplit, dt_split, policy_S);
    if (ysplit) return hipErrorInvalidValue;
    if (fused) return launch_scan_t<float, true>(u, z, ldz, dt_low, lddt, Wdt, Rp, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, dt_split, policy_S);
    return launch_scan_t<float, false>(u, z, ldz, delta, E, nullptr, 0, bc, A2, a_scale, Dskip, dbias, y, S, L, E, reverse, accumulate, s, uyb, zblk, seg_ws, walk_len, nullptr, false, policy_S);
}

}  // namespace pcad
