// NT GEMM on the gfx950 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T   (F.linear without bias).
//
// Replaces the cuBLAS GEMMs behind in_proj / x_proj / dt_proj / out_proj of mamba_ssm.Mamba
// (shapes from reference notebooks/examples.ipynb:73-80).  Both operands are
// K-contiguous (token-major activations x nn.Linear weights), which is the natural MFMA layout.
//
// Design (wave64, CDNA4):
//   * 256x128 block tile, 8 waves as 4(M) x 2(N), each wave 64x64 = 4x4 MFMA 16x16 tiles; ONE persistent block per
//     CU walking its share of the output tiles.
//   * K tile = 128 BYTES per row for every dtype (64 bf16 / 32 fp32) so the LDS image, the staging code and
//     the ds_read_b128 fragment reads are byte-identical; only the MFMA differs:
//       bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte fragment pair),
//       fp32: 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the same fragments (exact fp32, k permuted
//             identically on both operands).
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring (3 x 48 KiB): the loads of
//     K-tile g+3 are issued in the middle of K-tile g and waited for with a COUNTED s_waitcnt vmcnt(6), so 2.5
//     K-tiles (~2500 cycles of MFMA work per SIMD) cover the HBM/L2 latency; one raw s_barrier per K-tile.
//   * MFMA operand fragments are double-buffered in registers (ds_read_b128 of the next k-half under the 16 MFMAs
//     of the current one), so LDS latency is never exposed.  (Measured: with a 2-stage ring and loads one K-tile ahead the loop was latency-bound
//     at ~950 TF whatever the tile prologue/epilogue did.)  The ring runs continuously across the block's output
//     tiles, so a tile's first K-tiles load under the previous tile's last MFMAs and its stores.
//   * The LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle (16-byte chunk index XOR a
//     row key) is applied to the per-lane SOURCE address and again on the fragment read (0 conflicts measured).
//   * operands swapped (W is the MFMA "A" operand) with W rows permuted inside the wave tile so that each
//     lane ends up holding 16 CONSECUTIVE output columns of one output row: 32/64-byte vector stores.
//   * tile -> (m, n) map is XCD-aware: each XCD walks a contiguous range of tiles, m-fastest inside
//     groups of 8 m-panels, so the A panels and the current W rows stay in that XCD's 4 MiB L2.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace pcad {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 128, ROWB = 128;          // ROWB: bytes of K per tile row
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB; // 32 KiB + 16 KiB per stage
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int NSTAGE = 3;
constexpr int GEMM_LDS = NSTAGE * STAGE_BYTES;         // 144 KiB
constexpr int GEMM_THREADS = 512;
// Tile walk of the persistent kernels: every XCD walks a contiguous range of the logical tile order, its 32 CUs running 32
// consecutive tiles at a time.  Bits 0-7: GROUP_M - the logical order is m-fastest inside groups of GROUP_M m-panels (a "round" of
// 32 tiles = GROUP_M A panels x 32 / GROUP_M W panels); bit 8: n-fastest instead (a round = the fewest A panels, every W panel).
// A COMPILE-TIME constant: as a run-time kernel argument it cost the 4-wave kernel 12 % (863 / 1013 vs 988 / 1135 TF, same box).
// Swept in round 4 at the benchmark's launch size (tools/gemm_walk.sh, one build per walk: profiles/r04_gemm_walk.txt):
//   in_proj  ms / L2->fabric read GB:  2: 3.86 / 10.0   4: 3.72 / 6.8   8: 3.72 / 6.5   16: 3.78 / 9.7   32: 4.07 / 17.7   n-fastest: 3.82 / 10.1
//   out_proj                         :  2: 1.70 / 3.2    4: 1.69 / 3.2   8: 1.70 / 3.2   16: 1.73 / 4.8   32: 1.95 / 8.9    n-fastest: 1.70 / 3.2
// 8 is the minimum of both for in_proj (a round's 8 A panels + 4 W panels = 6 MB per 32 tiles is the least a 32-tile round can
// fetch: (gm + gn) x 0.5 MB with gm x gn = 32; the 4 MiB L2 keeps nothing across rounds, the re-reads are served by the
// Infinity Cache) and out_proj has only 4 n-tiles, which every walk <= 8 covers in one round.
#ifndef PCAD_WALK_CONST
#define PCAD_WALK_CONST 8
#endif
constexpr int walk = PCAD_WALK_CONST;

// Wrap-around K cursor of the split-bf16 GEMMs (K = 3 Ko, operands stored [hi | lo], nk = Ko / 64 K-tiles per part): the memory
// K-tile of logical K-tile kt - three passes over K.  (Measured and removed, profiles/r06_f32_split_ab.txt: the three products of one
// Ko-tile on consecutive K-tiles, kt = 3 k + part, so that the second use of a tile would hit the L2 - in_proj 7.3 -> 7.9 ms,
// out_proj 3.35 -> 3.77 ms per 350 208-row launch: re-requesting a tile one or two K-tiles after its first request is slower than
// streaming it again 16-32 K-tiles later.)
__device__ __forceinline__ int ksplit_a(int kt, int nk) { return kt >= 2 * nk ? kt - 2 * nk : kt; }      // A: hi, lo, hi
__device__ __forceinline__ int ksplit_w(int kt, int nk) { return kt >= nk ? kt - nk : kt; }              // W: hi, hi, lo
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
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(const T* __restrict__ A, int64_t lda,
                                                                  const T* __restrict__ W, int64_t ldw,
                                                                  OutT* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                  int K, int tiles_m, int tiles_n,
                                                                  float* __restrict__ C2, int64_t ldc2, int nsplit,
                                                                  int a_blocked, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // 0..7
    const int nblk = tiles_m * tiles_n;
    const int nkt = (K * (int)sizeof(T)) / ROWB;
    // ksplit != 0 (split-bf16 product of two fp32 operands, see "wrap-around K cursor" below): K = 3 Ko, both operands are stored
    // [hi | lo] (2 Ko columns) and K-tile kt reads A at kt mod-wrapped after 2 ksplit tiles, W after ksplit tiles

    // ---- XCD-aware tile map: block b runs on XCD b & 7 (observed dispatch order; speed only) and walks the
    // contiguous logical range of that XCD, m-fastest inside groups of GROUP_M m-panels -------------------------
    auto tile_coords = [&](int tile, int64_t& m0, int& n0) {
        const int xcd = tile & 7, idx = tile >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (walk & 256) {                  // n-fastest: the tiles an XCD runs at one time share as few A panels as possible
            const int tm = logical / tiles_n;
            m0 = (int64_t)tm * BM;
            n0 = (logical - tm * tiles_n) * BN;
            return;
        }
        const int group_m = walk & 255;
        const int gsz = group_m * tiles_n;
        const int g = logical / gsz;
        const int first_m = g * group_m;
        const int gm = min(group_m, tiles_m - first_m);
        const int in_g = logical - g * gsz;
        m0 = (int64_t)(first_m + in_g % gm) * BM;
        n0 = (in_g / gm) * BN;
    };

    // ---- staging cursor (runs two K-tiles ahead of the MFMAs): wave stages A rows [wave*32, +32) and W rows
    // [wave*16, +16) of the cursor's tile: 4 + 2 LDS-DMA instructions of 1 KiB ------------------------------------
    const char* pa[4];
    const char* pw[2];
    auto set_ptrs = [&](int64_t m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            int64_t ga = m0 + row; if (ga > M - 1) ga = M - 1;
            const int64_t abyte = a_blocked ? blocked_off(ga, 0, (lda * (int64_t)sizeof(T)) >> 7) : ga * lda * (int64_t)sizeof(T);
            pa[i] = reinterpret_cast<const char*>(A) + abyte + (((lane & 7) ^ key_a(row)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + (lane >> 3);
            int gw = n0 + row; if (gw > N - 1) gw = N - 1;
            pw[i] = reinterpret_cast<const char*>(W + (int64_t)gw * ldw) + (((lane & 7) ^ key_w(row)) << 4);
        }
    };
    auto stage = [&](int kt, int st) {
        char* as = smem + st * STAGE_BYTES + (wave * 32) * ROWB;
        char* ws = smem + st * STAGE_BYTES + A_BYTES + (wave * 16) * ROWB;
        const int kta = ksplit ? ksplit_a(kt, ksplit) : kt;
        const int ktw = ksplit ? ksplit_w(kt, ksplit) : kt;
        const int64_t ko = (int64_t)ktw * ROWB;
        const int64_t koa = a_blocked ? (int64_t)kta * 1024 : (int64_t)kta * ROWB;     // blocked A: consecutive k-pieces are 1 KiB apart
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(pa[i] + koa, as + i * 8 * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(pw[i] + ko, ws + i * 8 * ROWB);
    };
    int stile = blockIdx.x, skt = 0;        // what the next stage() call loads
    bool s_valid = stile < nblk;
    auto stage_next = [&](int st) -> bool {     // returns whether 6 DMAs were issued
        const bool issued = s_valid;
        if (s_valid) {
            stage(skt, st);
            if (++skt == nkt) {
                skt = 0;
                stile += (int)gridDim.x;
                s_valid = stile < nblk;
                if (s_valid) {
                    int64_t sm0; int sn0;
                    tile_coords(stile, sm0, sn0);
                    set_ptrs(sm0, sn0);
                }
            }
        }
        return issued;
    };
    // wait until the DMAs of the K-tile about to be read have landed: if `younger` DMAs (6 per wave, issued after
    // them; loads retire in order) are in flight they may stay in flight, otherwise drain everything
    auto wait_landed = [&](bool younger) {
        if (younger) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        w_off[i] = A_BYTES + rw * ROWB; w_key[i] = key_w(rw);
    }

    f32x4 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- epilogue: lane holds, per mi, 16 consecutive columns n = nb .. nb+15 of row m ------------
    auto epilogue = [&](int64_t m0, int n0) {
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
    };

    // ---- persistent loop: one continuous K-tile ring across this block's output tiles --------------------------
    // Fragment reads are software-pipelined in registers: the k-half-1 fragments of K-tile g are read before its
    // k-half-0 MFMAs, and the k-half-0 fragments of K-tile g+1 before the k-half-1 MFMAs of g, so ds_read latency
    // is always under 16 MFMAs.  That puts the "g+1 has landed" barrier in the MIDDLE of iteration g.
    auto load_frags = [&](int st, int kk, u32x4 (&af)[4], u32x4 (&wf)[4]) {
        const char* base = smem + st * STAGE_BYTES;
        const int chunk = kk * 4 + lg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = *reinterpret_cast<const u32x4*>(base + a_off[i] + ((chunk ^ a_key[i]) << 4));
            wf[i] = *reinterpret_cast<const u32x4*>(base + w_off[i] + ((chunk ^ w_key[i]) << 4));
        }
    };
    auto mma16 = [&](const u32x4 (&af)[4], const u32x4 (&wf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(wf[j], af[i], acc[i][j]);
    };

    int tile = blockIdx.x;
    if (tile >= nblk) return;
    int64_t m0; int n0;
    tile_coords(tile, m0, n0);
    set_ptrs(m0, n0);
    stage_next(0);                                   // K-tile g = 0
    wait_landed(stage_next(1));                      // g = 1 (possibly of the next output tile, or nothing) stays in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 af0[4], wf0[4], af1[4], wf1[4];
    load_frags(0, 0, af0, wf0);
    bool younger = stage_next(2);                    // g = 2
    zero_acc();
    int st = 0, kt = 0;
    bool pending = false;
    int64_t pm0 = 0; int pn0 = 0;
    while (true) {
        const int stn = st == 2 ? 0 : st + 1;
        load_frags(st, 1, af1, wf1);
        if (pending) {                    // previous tile's result (its MFMAs were issued before the last barrier)
            epilogue(pm0, pn0);
            zero_acc();
            pending = false;
        }
        mma16(af0, wf0);
        // K-tile g+1 must have landed before anyone reads it; the 6 DMAs of g+2 may stay in flight.  (The epilogue's
        // stores also count in vmcnt on CDNA4; they are younger than g+1's DMAs, and loads retire in order among
        // themselves, so "<= 6 outstanding" still implies g+1 has landed.)  lgkmcnt(0): this wave's reads of stage
        // `st` are complete, so after the barrier the stage may be overwritten by the DMAs of g+3.
        wait_landed(younger);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags(stn, 0, af0, wf0);     // K-tile g+1, k-half 0 (garbage after the last K-tile: never multiplied)
        younger = stage_next(st);         // K-tile g+3 -> the stage whose last reads completed above
        mma16(af1, wf1);
        st = stn;
        if (kt + 1 < nkt) {
            ++kt;
        } else {
            pending = true; pm0 = m0; pn0 = n0;
            tile += (int)gridDim.x;
            if (tile >= nblk) break;
            tile_coords(tile, m0, n0);
            kt = 0;
        }
    }
    epilogue(pm0, pn0);
}

// =====================================================================================================================
// 256x256-tile variant for the square-ish projections (in_proj, out_proj): 8 waves as 2(M) x 4(N), wave tile 128x64 =
// 8x4 MFMA tiles (128 accumulator registers).  Per flop it moves 33 % fewer L2->LDS bytes and 25 % fewer LDS->register
// bytes than the 256x128 / 64x64 kernel above, which the ablation showed to be what that kernel is bound by.  Each K-tile
// is four phases of 16 MFMAs (m-half x k-half); the ds_read_b128 of the next phase's fragments are issued before the
// current phase's MFMAs (four 16-register fragment sets rotate).  (Its first form, a 2-stage ring with all eight DMAs of
// K-tile g+2 behind one barrier, is gone: the ring kernels below replaced it — DESIGN.md §3.)
constexpr int BM2 = 256, BN2 = 256;
constexpr int A2_BYTES = BM2 * ROWB, W2_BYTES = BN2 * ROWB;     // 32 KiB each

// =====================================================================================================================
// 256x256 tile, A in a 3-stage ring and W in a 2-stage ring (3 x 32 + 2 x 32 KiB = all 160 KiB of the CU's LDS).
// 8 waves, 128x64 wave tiles, four phases per K-tile as described above; WHEN the LDS-DMAs are issued:
//   * A streams from HBM (each element once), W is re-read from L2 by every m-panel, so A gets the deeper ring:
//     A(g+3) is issued in four single pieces, one per phase (phase 4 of iteration g ... phase 3 of g+1) and has at least
//     one whole K-tile to land; W(g+2) is issued in phases 3 and 4 of iteration g (two pieces each) after a second
//     barrier that marks the end of all reads of W(g) (its last fragment read is in phase 2).
//   * every phase therefore carries 1, 1, 3, 3 DMA instructions per wave instead of 0, 0, 0, 8, and the two waves of a
//     SIMD (w, w + 4) issue theirs on opposite sides of the phase's 16 MFMAs, so one wave's MFMAs cover the other's
//     DMA issue time (measured: an LDS-DMA costs its wave 60-185 issue cycles).
//   * one counted wait per K-tile: at the barrier that releases K-tile g+1 the younger DMAs still in flight are
//     A(g+2) (4) and W(g+2)'s first half (2): s_waitcnt vmcnt(6).
constexpr int GEMM3_A_STAGES = 3, GEMM3_W_STAGES = 2;
constexpr int GEMM3_OFF_W = GEMM3_A_STAGES * A2_BYTES;
constexpr int GEMM3_LDS = GEMM3_OFF_W + GEMM3_W_STAGES * W2_BYTES;     // 160 KiB

template <typename T, typename OutT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm256r_kernel(const T* __restrict__ A, int64_t lda,
                                                                   const T* __restrict__ W, int64_t ldw,
                                                                   OutT* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                   int K, int tiles_m, int tiles_n, int a_blocked,
                                                                   OutT* __restrict__ C2, int nsplit, int out_blocked, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // 0..7
    const int gstride = (int)gridDim.x;
    const int nblk = tiles_m * tiles_n;
    const int nkt = (K * (int)sizeof(T)) / ROWB;
    if ((int)blockIdx.x >= nblk) return;
    const int my_tiles = (nblk - (int)blockIdx.x + gstride - 1) / gstride;
    const int G = my_tiles * nkt;                                 // K-tiles this block walks

    auto tile_coords = [&](int tile, int64_t& m0, int& n0) {
        const int xcd = tile & 7, idx = tile >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (walk & 256) {                  // n-fastest: the tiles an XCD runs at one time share as few A panels as possible
            const int tm = logical / tiles_n;
            m0 = (int64_t)tm * BM2;
            n0 = (logical - tm * tiles_n) * BN2;
            return;
        }
        const int group_m = walk & 255;
        const int gsz = group_m * tiles_n;
        const int g = logical / gsz;
        const int first_m = g * group_m;
        const int gm = min(group_m, tiles_m - first_m);
        const int in_g = logical - g * gsz;
        m0 = (int64_t)(first_m + in_g % gm) * BM2;
        n0 = (in_g / gm) * BN2;
    };

    // two staging cursors (A runs one K-tile further ahead than W); wave stages rows [wave*32, +32) of either tile
    const char* pa[4];
    const char* pw[4];
    auto set_pa = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            int64_t ga = m0 + row; if (ga > M - 1) ga = M - 1;
            const int64_t abyte = a_blocked ? blocked_off(ga, 0, (lda * (int64_t)sizeof(T)) >> 7) : ga * lda * (int64_t)sizeof(T);
            pa[i] = reinterpret_cast<const char*>(A) + abyte + (((lane & 7) ^ key_a(row)) << 4);
        }
    };
    auto set_pw = [&](int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            int gw = n0 + row; if (gw > N - 1) gw = N - 1;
            pw[i] = reinterpret_cast<const char*>(W + (int64_t)gw * ldw) + (((lane & 7) ^ key_w(row)) << 4);
        }
    };
    int a_tile = blockIdx.x, a_kt = 0, a_g = 0;       // next A K-tile to issue (global index a_g)
    int w_tile = blockIdx.x, w_kt = 0, w_g = 0;
    auto a_piece = [&](int sa, int p) {               // piece p of A(a_g) -> A stage sa
        if (a_g < G) {
            const int kta = ksplit ? ksplit_a(a_kt, ksplit) : a_kt;                       // wrap-around K cursor
            const int64_t koa = a_blocked ? (int64_t)kta * 1024 : (int64_t)kta * ROWB;
            glds16(pa[p] + koa, smem + sa * A2_BYTES + (wave * 32 + p * 8) * ROWB);
        }
    };
    auto a_advance = [&]() {
        if (a_g < G) {
            ++a_g;
            if (++a_kt == nkt) {
                a_kt = 0;
                a_tile += gstride;
                if (a_tile < nblk) { int64_t m0; int n0; tile_coords(a_tile, m0, n0); set_pa(m0); }
            }
        }
    };
    auto w_pieces = [&](int sw, int p0) {             // pieces p0, p0 + 1 of W(w_g) -> W stage sw
        if (w_g < G) {
            const int64_t ko = (int64_t)(ksplit ? ksplit_w(w_kt, ksplit) : w_kt) * ROWB;
            glds16(pw[p0] + ko, smem + GEMM3_OFF_W + sw * W2_BYTES + (wave * 32 + p0 * 8) * ROWB);
            glds16(pw[p0 + 1] + ko, smem + GEMM3_OFF_W + sw * W2_BYTES + (wave * 32 + (p0 + 1) * 8) * ROWB);
        }
    };
    auto w_advance = [&]() {
        if (w_g < G) {
            ++w_g;
            if (++w_kt == nkt) {
                w_kt = 0;
                w_tile += gstride;
                if (w_tile < nblk) { int64_t m0; int n0; tile_coords(w_tile, m0, n0); set_pw(n0); }
            }
        }
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    int a_off[8], a_key[8], w_off[4], w_key[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + li;
        a_off[i] = ra * ROWB; a_key[i] = key_a(ra);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rw = wn * 64 + (li >> 2) * 16 + j * 4 + (li & 3);
        w_off[j] = GEMM3_OFF_W + rw * ROWB; w_key[j] = key_w(rw);
    }
    auto load_a = [&](int sa, int kk, int half, u32x4 (&a)[4]) {
        const char* base = smem + sa * A2_BYTES;
        const int chunk = kk * 4 + lg;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = *reinterpret_cast<const u32x4*>(base + a_off[half * 4 + i] + ((chunk ^ a_key[half * 4 + i]) << 4));
    };
    auto load_w = [&](int sw, int kk, u32x4 (&w)[4]) {
        const char* base = smem + sw * W2_BYTES;
        const int chunk = kk * 4 + lg;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const u32x4*>(base + w_off[j] + ((chunk ^ w_key[j]) << 4));
    };

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma = [&](int half, const u32x4 (&a)[4], const u32x4 (&w)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = Mma<T>::run(w[j], a[i], acc[half * 4 + i][j]);
    };
    auto epilogue = [&](int64_t m0, int n0) {       // lane: 16 consecutive columns nb .. nb+15 of 8 rows
        const int nb = n0 + wn * 64 + lg * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t m = m0 + wm * 128 + i * 16 + li;
            if (m >= M) continue;
            float lo[8], hi[8];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                lo[rr] = acc[i][0][rr]; lo[4 + rr] = acc[i][1][rr];
                hi[rr] = acc[i][2][rr]; hi[4 + rr] = acc[i][3][rr];
            }
            OutT* dst;
            int nlim = N, col = nb;
            if (C2 != nullptr) {                    // two-output form (in_proj): columns >= nsplit go to C2
                const bool second = nb >= nsplit;
                const int width = second ? N - nsplit : nsplit;
                col = second ? nb - nsplit : nb;
                nlim = width;
                OutT* base = second ? C2 : C;
                dst = base + (out_blocked ? blocked_off(m, (int64_t)col * sizeof(OutT), ((int64_t)width * sizeof(OutT)) >> 7) / (int64_t)sizeof(OutT)
                                          : m * (int64_t)width + col);
            } else {
                dst = C + m * ldc + nb;
            }
            if (col + 16 <= nlim) {
                store8<OutT>(dst, lo);
                store8<OutT>(dst + 8, hi);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (col + e < nlim) Elem<OutT>::store(dst + e, lo[e]);
                    if (col + 8 + e < nlim) Elem<OutT>::store(dst + 8 + e, hi[e]);
                }
            }
        }
    };

    int tile = blockIdx.x;
    int64_t m0; int n0;
    tile_coords(tile, m0, n0);
    set_pa(m0);
    set_pw(n0);
    // prologue: A(0) W(0) | A(1) W(1) | A(2) piece 0  -> wait for the first eight, leave the rest in flight
#pragma unroll
    for (int p = 0; p < 4; ++p) a_piece(0, p);
    a_advance();
    w_pieces(0, 0); w_pieces(0, 2); w_advance();
#pragma unroll
    for (int p = 0; p < 4; ++p) a_piece(1, p);
    a_advance();
    w_pieces(1, 0); w_pieces(1, 2); w_advance();
    a_piece(2, 0);
    if (G >= 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 a0[4], a1[4], w0[4], w1[4];
    load_a(0, 0, 0, a0);
    load_w(0, 0, w0);
    zero_acc();
    int sa = 0, sw = 0, kt = 0, g = 0;
    int sa_fill = 2;                      // A stage that A(a_g) is being filled into
    bool pending = false;
    int64_t pm0 = 0; int pn0 = 0;
    while (true) {
        if (pending) {                    // previous tile's result (all its MFMAs were issued in the last iteration)
            epilogue(pm0, pn0);
            zero_acc();
            pending = false;
        }
        const int sa_n = sa == 2 ? 0 : sa + 1;
        if (wm == 0) a_piece(sa_fill, 1);
        load_a(sa, 0, 1, a1);             // phase 1: (m lo, k lo)
        mma(0, a0, w0);
        if (wm != 0) a_piece(sa_fill, 1);

        if (wm == 0) a_piece(sa_fill, 2);
        load_a(sa, 1, 0, a0);             // phase 2: (m hi, k lo)
        load_w(sw, 1, w1);
        mma(1, a1, w0);
        if (wm != 0) a_piece(sa_fill, 2);
        // every wave's reads of W stage sw are complete -> refill it with W(g+2) during phases 3 and 4
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        if (wm == 0) { a_piece(sa_fill, 3); w_pieces(sw, 0); }
        load_a(sa, 1, 1, a1);             // phase 3: (m lo, k hi)
        mma(0, a0, w1);
        if (wm != 0) { a_piece(sa_fill, 3); w_pieces(sw, 0); }
        a_advance();
        // K-tile g+1 (A and W) must have landed; younger and allowed in flight: A(g+2) x4, W(g+2) x2
        if (g + 2 < G) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        sa_fill = sa;                     // A stage sa is free: A(g+3) starts filling it
        if (wm == 0) { w_pieces(sw, 2); a_piece(sa_fill, 0); }
        load_a(sa_n, 0, 0, a0);           // phase 4: (m hi, k hi), with the first fragments of K-tile g+1 in flight
        load_w(sw ^ 1, 0, w0);
        mma(1, a1, w1);
        if (wm != 0) { w_pieces(sw, 2); a_piece(sa_fill, 0); }
        w_advance();
        sa = sa_n;
        sw ^= 1;
        ++g;
        if (kt + 1 < nkt) {
            ++kt;
        } else {
            pending = true; pm0 = m0; pn0 = n0;
            tile += gstride;
            if (tile >= nblk) break;
            tile_coords(tile, m0, n0);
            kt = 0;
        }
    }
    epilogue(pm0, pn0);
}

// =====================================================================================================================
// 256x256 tile on FOUR waves (one per SIMD, up to 512 registers each): wave tile 128x128 = 8x8 MFMA tiles (256 accumulator
// registers), full tiles only (M, N multiples of 256).  Against the 8-wave kernels: 0.25 instead of 0.375 ds_read_b128 per
// MFMA, half as many waves at every barrier, ONE barrier per K-tile.  Same LDS ring as gemm256r (A x3, W x2), same
// fragment images.  A K-tile is two k-steps of 64 MFMAs; the 16 fragments of the next k-step are read while the current
// one computes (two register sets), and each k-step also carries 8 LDS-DMAs (A(g+2) in k-step 0, W(g+2) in k-step 1),
// issued unconditionally so that they sit in the same basic block as the MFMAs: beyond the last K-tile the cursors stop
// and the DMAs re-read the last K-tile into ring slots nobody reads again.
//   barrier (between the two k-steps of K-tile g): every wave's fragment reads of K-tile g are complete (stage free for
//   refill) and, by the counted vmcnt(8) in front of it, K-tile g+1 has landed (the 8 younger DMAs are A(g+2)).
constexpr int GEMMQ_THREADS = 256;

// LDS-DMA through a buffer descriptor: address = descriptor base + voff (per lane) + soff (wave-uniform), 16 B per lane
// to lds_wave_base + lane * 16.  (Device-only builtins live in __device__ helpers so the host pass still emits the stub.)
__device__ __forceinline__ void blds16(const void* base, uint32_t voff, uint32_t soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xfffffffcu, 0x00020000),
                                             (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 1);   // aux 1 = sc0: +2-3 % on in_proj in a same-box A/B (nt: -8 %)
}

// MFMA with the accumulator pinned to AGPRs: at 256 accumulator registers per wave the register allocator otherwise
// shuttles accumulators between the two halves of the register file around every MFMA.
template <typename T> struct MmaAcc;
template <> struct MmaAcc<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
    }
    static __device__ __forceinline__ void run0(const u32x4& w, const u32x4& a, f32x4& c) {      // c = w . a (no zeroing pass)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(a));
    }
};
template <> struct MmaAcc<float> {
    static __device__ __forceinline__ void run(const u32x4& w, const u32x4& a, f32x4& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(__uint_as_float(w[s])), "v"(__uint_as_float(a[s])));
    }
    static __device__ __forceinline__ void run0(const u32x4& w, const u32x4& a, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(c) : "v"(__uint_as_float(w[0])), "v"(__uint_as_float(a[0])));
#pragma unroll
        for (int s = 1; s < 4; ++s)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(__uint_as_float(w[s])), "v"(__uint_as_float(a[s])));
    }
};

// loads with a hand-managed wait: 16 bytes per lane straight into an accumulator-file register / 4 bytes into a VGPR, address =
// 64-bit scalar base + 32-bit lane offset + immediate.  The compiler does not know these are memory operations: every consumer
// sits behind an explicit s_waitcnt (see the call sites).
template <int OFF> __device__ __forceinline__ void gload_a128(f32x4& dst, uint32_t vo, const float* sb) {
    // nt: every residual element is read once and written once per launch (a 2 GiB tensor): the streaming hint keeps it from
    // displacing the W tile and the A lines other CUs re-use (out_proj + residual 2.055 -> 2.015 ms in six interleaved pairs, r04k / r04l)
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=a"(dst) : "v"(vo), "s"(sb), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The same wait TIED to the registers it guards: the asm formally rewrites them, so every later reader (the MFMAs that accumulate
// into an accumulator row, the epilogue's multiplies by the row factors) depends on the wait and cannot be scheduled, copied or
// spilled across it; tools/isa_census.py --check-res-waits verifies the emitted order after every build.
template <int N> __device__ __forceinline__ void wait_vmcnt_row(f32x4 (&c)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])
                 : "n"(N)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt_regs(float (&v)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N)
                 : "memory");
}
// EPI_RES's first k-step: the 64 residual loads of a tile are issued in accumulator-row order (8 per row: load_res_group x2) right
// before it, and every row of the k-step issues exactly ONE LDS-DMA piece (slot kDmaSlot); when row r is about to accumulate into
// acc[r][*], younger than ITS eight loads are the loads of rows r+1..7 and the DMA pieces of rows 0..r-1 (vector memory
// operations of one type return in order on gfx950), hence:
constexpr int kResLoadsPerRow = 8;      // load_res_group(tb, i, 0) + load_res_group(tb, i, 1): 2 x 4 global_load_dwordx4
constexpr int kDmaPerRow = 1;           // kstep: `if (k == kDmaSlot) dma(r)`
constexpr int kDmaSlot = 1;
constexpr int res_row_wait(int r) { return (7 - r) * kResLoadsPerRow + r * kDmaPerRow; }
static_assert(res_row_wait(0) == 56 && res_row_wait(1) == 49 && res_row_wait(6) == 14 && res_row_wait(7) == 7, "EPI_RES row waits");
static_assert(res_row_wait(0) < 64, "s_waitcnt vmcnt is a 6-bit field");
template <int OFF> __device__ __forceinline__ void gload_v32(float& dst, uint32_t vo, const float* sb) {
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(vo), "s"(sb), "n"(OFF));
}

// Fused epilogues of the 4-wave kernel (the norm-folded layer form, api.hip "norm_fold"; reference semantics: rms_norm_fn(...,
// prenorm=True, residual_in_fp32=True) between out_proj of one block and in_proj of the next):
//   EPI_SCALE  C = (A . W^T) * rscale[row]                 in_proj on the UN-normalised residual with W_in . diag(w_norm) folded at
//              bind time: rscale = rstd of the row.  The 8 factors a lane needs are fetched when the tile starts.
//   EPI_RES    res += A . W^T (fp32, in place, fragment layout: common.hpp res_frag_off);  C = round(res) (plain rows);
//              ssq[row][n / 128] = sum of res^2 over the wave's 128 columns.
//              The accumulators START as the tile's residual values (loaded straight into the accumulator-file registers when
//              the tile begins; the first k-step waits row by row), so the read-modify-write of the fp32 stream costs no
//              register and no separate pass; the per-row partial sums of squares are written without atomics
//              (deterministic), one per 128-column wave tile, and reduced by rstd_kernel (norm.hip).
enum { EPI_NONE = 0, EPI_SCALE = 1, EPI_RES = 2 };
struct GemmEpi {
    const float* rscale;    // EPI_SCALE: [M]
    float* res;             // EPI_RES: [M, N] fp32 in the fragment layout (common.hpp res_frag_off)
    float* ssq;             // EPI_RES: [M, N / 128]
};

template <typename T, typename OutT, int EPI>
__global__ __launch_bounds__(GEMMQ_THREADS, 1) void gemm256q_kernel(const T* __restrict__ A, int64_t lda,
                                                                    const T* __restrict__ W, int64_t ldw,
                                                                    OutT* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                    int K, int tiles_m, int tiles_n, int a_blocked,
                                                                    OutT* __restrict__ C2, int nsplit, int out_blocked, int epi_swap,
                                                                    GemmEpi epi, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // 0..3
    const int gstride = (int)gridDim.x;
    const int nblk = tiles_m * tiles_n;
    const int nkt = (K * (int)sizeof(T)) / ROWB;
    if ((int)blockIdx.x >= nblk) return;
    const int my_tiles = (nblk - (int)blockIdx.x + gstride - 1) / gstride;
    const int G = my_tiles * nkt;                                 // K-tiles this block walks

    auto tile_coords = [&](int tile, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int xcd = tile & 7, idx = tile >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (walk & 256) {                  // n-fastest: the tiles an XCD runs at one time share as few A panels as possible
            const int tm = logical / tiles_n;
            m0 = (int64_t)tm * BM2;
            n0 = (logical - tm * tiles_n) * BN2;
            return;
        }
        const int group_m = walk & 255;
        const int gsz = group_m * tiles_n;
        const int g = logical / gsz;
        const int first_m = g * group_m;
        const int gm = min(group_m, tiles_m - first_m);
        const int in_g = logical - g * gsz;
        m0 = (int64_t)(first_m + in_g % gm) * BM2;
        n0 = (in_g / gm) * BN2;
    };

    // staging: wave stages rows [wave*64, +64) of the A tile and of the W tile, 8 pieces of 8 rows each, by
    // buffer_load_dwordx4 ... lds: address = descriptor base + per-lane offset (VGPR, constant for the whole kernel: full
    // tiles, so the lane part does not depend on the tile) + wave-uniform tile/K-tile offset (SGPR): no vector address
    // arithmetic per DMA, only the M0 (LDS destination) update.  Offsets are 32-bit: the launcher checks the tensor sizes.
    const int64_t a_pieces = (lda * (int64_t)sizeof(T)) >> 7;
    uint32_t a_lo[8], w_lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 64 + i * 8 + (lane >> 3);
        const int64_t ab = a_blocked ? ((int64_t)(row >> 3) * a_pieces << 10) + ((row & 7) << 7) : (int64_t)row * lda * (int64_t)sizeof(T);
        a_lo[i] = (uint32_t)(ab + (((lane & 7) ^ key_a(row)) << 4));
        w_lo[i] = (uint32_t)((int64_t)row * ldw * (int64_t)sizeof(T) + (((lane & 7) ^ key_w(row)) << 4));
    }
    uint32_t a_base, w_base;                          // wave-uniform byte offset of the cursor's tile
    auto set_pa = [&](int64_t m0) __attribute__((always_inline)) {
        a_base = (uint32_t)(a_blocked ? ((m0 >> 3) * a_pieces << 10) : m0 * lda * (int64_t)sizeof(T));
    };
    auto set_pw = [&](int n0) __attribute__((always_inline)) { w_base = (uint32_t)((int64_t)n0 * ldw * (int64_t)sizeof(T)); };
    int a_tile = blockIdx.x, a_kt = 0, a_g = 0;
    int w_tile = blockIdx.x, w_kt = 0, w_g = 0;
    // Wrap-around K cursor (KSPLIT: the bf16 -> fp32 instantiation only, i.e. the split-bf16 GEMMs of the fp32 model, api.hip
    // "f32_gemm_split"): an fp32 product a . w is three bf16 products a_hi w_hi + a_lo w_hi + a_hi w_lo.  Both operands are stored ONCE
    // as [hi | lo] (2 Ko columns, ksplit = Ko / 64 K-tiles per part) and the K-tile cursor visits 3 ksplit tiles: A reads hi, lo, hi
    // (back to column 0 after 2 ksplit tiles), W reads hi, hi, lo (back to column 0 after ksplit tiles).  Against the concatenated
    // [hi | lo | hi] x [hi | hi | lo] form the producers write a third less, the operands take a third less memory and HBM traffic
    // (the second visit of a part is served by the L2 / Infinity Cache), and the in-tensor offsets allow 1.5x larger chunks.
    // The memory K-tile index is derived from the (wave-uniform, scalar) cursor a_kt / w_kt at every use - a separately carried index
    // was placed in a VGPR by the compiler and every LDS-DMA then ran in a waterfall loop (-15 % on the whole GEMM).
    constexpr bool KSPLIT = !std::is_same<T, OutT>::value;
    auto a_piece = [&](int sa, int p) __attribute__((always_inline)) {               // piece p of A(a_g) -> A stage sa
        const int a_kx = KSPLIT && ksplit ? ksplit_a(a_kt, ksplit) : a_kt;
        const uint32_t soff = a_base + (uint32_t)a_kx * (a_blocked ? 1024u : (uint32_t)ROWB);
        blds16(A, a_lo[p], soff, smem + sa * A2_BYTES + (wave * 64 + p * 8) * ROWB);
    };
    auto a_issue = [&](int sa) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 8; ++p) a_piece(sa, p);
    };
    auto w_piece = [&](int sw, int p) __attribute__((always_inline)) {
        const int w_kx = KSPLIT && ksplit ? ksplit_w(w_kt, ksplit) : w_kt;
        const uint32_t soff = w_base + (uint32_t)w_kx * (uint32_t)ROWB;
        blds16(W, w_lo[p], soff, smem + GEMM3_OFF_W + sw * W2_BYTES + (wave * 64 + p * 8) * ROWB);
    };
    auto w_issue = [&](int sw) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 8; ++p) w_piece(sw, p);
    };
    auto a_advance = [&]() __attribute__((always_inline)) {
        if (a_g + 1 < G) {
            ++a_g;
            if (++a_kt == nkt) {
                a_kt = 0;
                a_tile += gstride;
                int64_t m0; int n0;
                tile_coords(a_tile, m0, n0);
                set_pa(m0);
            }
        }
    };
    auto w_advance = [&]() __attribute__((always_inline)) {
        if (w_g + 1 < G) {
            ++w_g;
            if (++w_kt == nkt) {
                w_kt = 0;
                w_tile += gstride;
                int64_t m0; int n0;
                tile_coords(w_tile, m0, n0);
                set_pw(n0);
            }
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    int a_fo[8], w_fo[8];                             // fragment byte offsets inside a stage for k-step 0 (swizzle applied);
#pragma unroll                                        // k-step 1 is the same address with bit 6 flipped
    for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + li;
        a_fo[i] = ra * ROWB + ((lg ^ key_a(ra)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rw = wn * 128 + (j >> 2) * 64 + (li >> 2) * 16 + (j & 3) * 4 + (li & 3);
        w_fo[j] = GEMM3_OFF_W + rw * ROWB + ((lg ^ key_w(rw)) << 4);
    }
    auto load_frags = [&](int sa, int sw, int kk, u32x4 (&a)[8], u32x4 (&w)[8]) __attribute__((always_inline)) {
        const int ab = sa * A2_BYTES, wb = sw * W2_BYTES, kx = kk << 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(smem + ((wb + w_fo[j]) ^ kx));
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const u32x4*>(smem + ((ab + a_fo[i]) ^ kx));
    };

    f32x4 acc[8][8];
    // One k-step: 8 rows of 8 MFMAs on (ac, wc).  With one wave per SIMD nothing else covers a burst of non-MFMA
    // instructions, so the 16 fragment reads of the NEXT k-step and the 8 LDS-DMA pieces are pinned one at a time behind
    // every second MFMA (an MFMA holds the pipe for 16 cycles = 4 issue slots; a read or DMA with its address math is 3-4):
    //   rows 0-3: next W fragments (a second register set); rows 4-6: next A fragments 6, 7 (spare registers) and 0..5 into
    //   the registers of rows that are already done; every row: one DMA piece; row 7: no reads, so every read was
    //   issued at least 8 MFMAs before the k-step ends.
    // slot_tag (PCAD_GEMM_STAGGER): the slot of a row in which THIS wave issues its LDS-DMA piece.  The four waves of the block run in
    // lockstep behind the per-K-tile barrier, so with one common slot all four hand a 1 KiB request to the CU's single address unit in
    // the same cycle and three of them wait at issue (16 cycles of address processing each) with nothing else to issue; with slot =
    // wave index each wave meets a free unit.  The fragment read that owned the slot moves to slot kDmaSlot.
    auto kstep = [&](const u32x4 (&ac)[8], const u32x4 (&wc)[8], u32x4 (&an)[8], u32x4 (&wn)[8], int nsa, int nsw, int nkk,
                     bool dma_a, int dma_stage, auto first_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int MODE = (int)decltype(first_tag)::value;       // 1: first k-step of an output tile: acc = product;
        constexpr bool FIRST = MODE == 1;                           // 2 (EPI_RES): the accumulators are arriving from memory (row waits)
        constexpr int SLOT = (int)decltype(slot_tag)::value;
        const int abn = nsa * A2_BYTES, wbn = nsw * W2_BYTES, kx = nkk << 6;
        auto lda = [&](int f) __attribute__((always_inline)) { an[f] = *reinterpret_cast<const u32x4*>(smem + ((abn + a_fo[f]) ^ kx)); };
        auto ldw = [&](int f) __attribute__((always_inline)) { wn[f] = *reinterpret_cast<const u32x4*>(smem + ((wbn + w_fo[f]) ^ kx)); };
        auto dma = [&](int p) __attribute__((always_inline)) { if (dma_a) a_piece(dma_stage, p); else w_piece(dma_stage, p); };
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (MODE == 2) {
                // row r multiplies into acc[r][0..7], whose 8 residual loads were issued in row order just before this k-step:
                // younger than them are the loads of rows r+1..7 (8 each) and this k-step's DMA pieces 0..r-1
                switch (r) {
                    case 0: wait_vmcnt_row<res_row_wait(0)>(acc[0]); break; case 1: wait_vmcnt_row<res_row_wait(1)>(acc[1]); break;
                    case 2: wait_vmcnt_row<res_row_wait(2)>(acc[2]); break; case 3: wait_vmcnt_row<res_row_wait(3)>(acc[3]); break;
                    case 4: wait_vmcnt_row<res_row_wait(4)>(acc[4]); break; case 5: wait_vmcnt_row<res_row_wait(5)>(acc[5]); break;
                    case 6: wait_vmcnt_row<res_row_wait(6)>(acc[6]); break; default: wait_vmcnt_row<res_row_wait(7)>(acc[7]); break;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int j = 2 * ks; j < 2 * ks + 2; ++j) {
                    if (FIRST) MmaAcc<T>::run0(wc[j], ac[r], acc[r][j]);
                    else MmaAcc<T>::run(wc[j], ac[r], acc[r][j]);
                }
                const int k = ks == SLOT ? kDmaSlot : (ks == kDmaSlot ? SLOT : ks);      // logical slot: the DMA's and the wave's own slot swapped
                // slot k of row r: W' in rows 0-3, A' in rows 4-6, ONE DMA piece per row (4 waves x 1 KiB every 8 MFMAs keeps
                // the CU's L1 half busy; all eight pieces within rows 0-3 saturated it and stalled the issue: TA stalled-by-TC x7)
                if (k == kDmaSlot) dma(r);
                else if (r < 4) {
                    if (k == 0) ldw(2 * r);
                    else if (k == 2) ldw(2 * r + 1);
                } else if (r == 4) {
                    if (k == 0) lda(6); else if (k == 2) lda(7);
                } else if (r == 5) {
                    if (k == 0) lda(0); else if (k == 2) lda(1); else lda(2);
                } else if (r == 6) {
                    if (k == 0) lda(3); else if (k == 2) lda(4); else lda(5);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // lane: 2 x 16 consecutive columns (jg) of 8 rows (i).  Every destination is lane_base + i * row_step + jg * 128 bytes
    // (plain and blocked layouts alike; in the two-output form the 64-column group jg lies wholly in one output because
    // nsplit is a multiple of 64), so the address math is one 64-bit lane value per tile plus wave-uniform steps.
    // byte correction of the lane's destination for the swapped epilogue: the lane's piece moves from column 16 lg to
    // (lg & 1) * 16 + (lg >> 1) * 8 of the 64-column group (2-byte elements)
    const int swap_fix = (((lg & 1) * 16 + (lg >> 1) * 8) - lg * 16) * 2;
    // ---- fused-epilogue state -------------------------------------------------------------------------------------------
    float rs[8];                                            // EPI_SCALE: row factors of the current tile (rows mrow + 16 i)
    auto load_rscale = [&](int64_t m0) __attribute__((always_inline)) {
        if constexpr (EPI == EPI_SCALE) {
            // asm loads: the values are consumed a whole tile later, behind >= 1 counted "s_waitcnt vmcnt(8)" with 8 younger
            // DMAs in flight, so they have landed by construction and no wait is ever issued for them
            const float* sb = epi.rscale + (m0 + wm * 128);
            const uint32_t vo = (uint32_t)li * 4u;
            gload_v32<0>(rs[0], vo, sb);   gload_v32<64>(rs[1], vo, sb);  gload_v32<128>(rs[2], vo, sb); gload_v32<192>(rs[3], vo, sb);
            gload_v32<256>(rs[4], vo, sb); gload_v32<320>(rs[5], vo, sb); gload_v32<384>(rs[6], vo, sb); gload_v32<448>(rs[7], vo, sb);
        }
    };
    // EPI_RES: residual values of tile (m0, n0) -> accumulator group (i, jg): 4 x 16 bytes per lane, straight into the AGPRs.
    // The residual tensor is in the fragment layout (common.hpp res_frag_off): the wave's 128 x 128 part of a tile is 64 KiB
    // contiguous, group (i, jg) quad k = 1 KiB = lane-linear 16-byte pieces, so every access instruction moves whole lines.
    const uint32_t res_vo = (uint32_t)lane * 16u;
    auto res_tile_base = [&](int64_t m0, int n0) __attribute__((always_inline)) -> float* {
        return epi.res + ((((m0 >> 8) * (int64_t)(N >> 8) + (n0 >> 8)) * 4 + wave) << 14);
    };
    auto load_res_group = [&](const float* tb, int i, int jg) __attribute__((always_inline)) {
        const float* sb = tb + (i * 2 + jg) * 1024;
        gload_a128<0>(acc[i][jg * 4 + 0], res_vo, sb);    gload_a128<1024>(acc[i][jg * 4 + 1], res_vo, sb);
        gload_a128<2048>(acc[i][jg * 4 + 2], res_vo, sb); gload_a128<3072>(acc[i][jg * 4 + 3], res_vo, sb);
    };
    auto epilogue = [&](int64_t m0, int n0) __attribute__((always_inline)) {
        const int64_t mrow = m0 + wm * 128 + li;
        float ss[8];
        float* res_tb = nullptr;
        if constexpr (EPI == EPI_SCALE) {
            // the row factors were requested when the tile started, >= one K-tile ago: behind every counted `vmcnt(8)` of the mainloop
            // they have landed (at most the 16 youngest operations - this K-tile's LDS-DMAs - are in flight here), so this wait never stalls; it exists to
            // make the multiplies below DEPEND on a wait that covers their operands
            wait_vmcnt_regs<16>(rs);
        }
        if constexpr (EPI == EPI_RES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ss[i] = 0.f;
            res_tb = res_tile_base(m0, n0);
        }
#pragma unroll
        for (int jg = 0; jg < 2; ++jg) {
            const int nbase = n0 + wn * 128 + jg * 64;                 // wave-uniform
            char* lane_dst;
            int64_t row_step;                                          // bytes between rows m and m + 16
            if (C2 != nullptr) {                                       // two-output form (in_proj): columns >= nsplit go to C2
                const bool second = nbase >= nsplit;
                const int width = second ? N - nsplit : nsplit;
                const int col = (second ? nbase - nsplit : nbase) + lg * 16;
                char* base = reinterpret_cast<char*>(second ? C2 : C);
                if (out_blocked) {
                    lane_dst = base + blocked_off(mrow, (int64_t)col * sizeof(OutT), ((int64_t)width * sizeof(OutT)) >> 7);
                    row_step = 2 * ((((int64_t)width * sizeof(OutT)) >> 7) << 10);
                } else {
                    lane_dst = base + (mrow * width + col) * (int64_t)sizeof(OutT);
                    row_step = 16 * (int64_t)width * sizeof(OutT);
                }
            } else {
                lane_dst = reinterpret_cast<char*>(C) + (mrow * ldc + nbase + lg * 16) * (int64_t)sizeof(OutT);
                row_step = 16 * ldc * (int64_t)sizeof(OutT);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // the AGPR -> VGPR copies of this group only, here (otherwise all 256 are hoisted to the top of the epilogue)
                f32x4 t0 = acc[i][jg * 4 + 0], t1 = acc[i][jg * 4 + 1], t2 = acc[i][jg * 4 + 2], t3 = acc[i][jg * 4 + 3];
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                if constexpr (EPI == EPI_RES) {
                    float* rdst = res_tb + (i * 2 + jg) * 1024 + lane * 4;
                    __builtin_nontemporal_store(t0, reinterpret_cast<f32x4*>(rdst));       __builtin_nontemporal_store(t1, reinterpret_cast<f32x4*>(rdst + 256));
                    __builtin_nontemporal_store(t2, reinterpret_cast<f32x4*>(rdst + 512)); __builtin_nontemporal_store(t3, reinterpret_cast<f32x4*>(rdst + 768));
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        ss[i] = __builtin_fmaf(t0[rr], t0[rr], __builtin_fmaf(t1[rr], t1[rr], __builtin_fmaf(t2[rr], t2[rr], __builtin_fmaf(t3[rr], t3[rr], ss[i]))));
                    asm volatile("" : "+v"(ss[i]));             // computed HERE (otherwise the chain is sunk to the end of the epilogue and every group's values stay live: spills)
                }
                if constexpr (EPI == EPI_SCALE) { t0 *= rs[i]; t1 *= rs[i]; t2 *= rs[i]; t3 *= rs[i]; }
                float lo[8], hi[8];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    lo[rr] = t0[rr]; lo[4 + rr] = t1[rr];
                    hi[rr] = t2[rr]; hi[4 + rr] = t3[rr];
                }
                OutT* dst = reinterpret_cast<OutT*>(lane_dst + i * row_step);
                if constexpr (sizeof(OutT) == 2) {
                    if (epi_swap) {
                        // lane group lg (= lane / 16) holds columns [16 lg, 16 lg + 16) of the 64-column group: `lo` of the four groups
                        // are four separate 16-byte runs of a 128-byte line.  Exchanging the upper-half lanes of `lo` with the
                        // lower-half lanes of `hi` (v_permlane32_swap) leaves `lo` with the line's first 64 bytes and `hi` with the
                        // second (lane group -> 16-byte piece 0, 2, 1, 3), so each store instruction writes whole 64-byte halves.
                        u32x4 L, H;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(pack_bf16x2(lo[2 * d], lo[2 * d + 1]), pack_bf16x2(hi[2 * d], hi[2 * d + 1]), false, false);
                            L[d] = sw[0]; H[d] = sw[1];
                        }
                        char* db = reinterpret_cast<char*>(dst) + swap_fix;          // from column lg * 16 to ((lg & 1) * 16 + (lg >> 1) * 8)
                        *reinterpret_cast<u32x4*>(db) = L;
                        *reinterpret_cast<u32x4*>(db + 64) = H;
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                }
                store8<OutT>(dst, lo);
                store8<OutT>(dst + 8, hi);
                __builtin_amdgcn_sched_barrier(0);      // keep the epilogue's register footprint at one 16-column group
            }
        }
        if constexpr (EPI == EPI_RES) {
            // sum of squares of the wave's 128 columns of row mrow + 16 i: the four 16-lane groups hold 32 columns each
            const int np = N >> 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = ss[i];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lg == 0) epi.ssq[(mrow + i * 16) * np + (n0 >> 7) + wn] = v;
            }
        }
    };

    int tile = blockIdx.x;
    int64_t m0; int n0;
    tile_coords(tile, m0, n0);
    set_pa(m0);
    set_pw(n0);
    load_rscale(m0);
    // prologue: A(0) W(0) A(1) W(1); wait for the first 16 pieces
    a_issue(0); a_advance();
    w_issue(0); w_advance();
    a_issue(1); a_advance();
    w_issue(1); w_advance();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    u32x4 fa0[8], fw0[8], fa1[8], fw1[8];
    load_frags(0, 0, 0, fa0, fw0);
    int sa = 0, sw = 0, kt = 0;
    auto mainloop = [&](auto slot) __attribute__((always_inline)) {
    while (true) {
        const int sa_n = sa == 2 ? 0 : sa + 1;
        const int sa_f = sa == 0 ? 2 : sa - 1;          // stage of K-tile g-1 == stage of K-tile g+2
        // k-step 0: MFMAs on (g, k-step 0); reads (g, k-step 1); DMAs A(g+2)
        if constexpr (EPI == EPI_RES) {
            if (kt == 0) {
                // the accumulators START as the tile's residual values: 64 loads of 16 bytes per lane straight into the accumulator
                // registers, row by row, and the first k-step waits for each row's eight just before it multiplies into them
                const float* tb = res_tile_base(m0, n0);
#pragma unroll
                for (int i = 0; i < 8; ++i) { load_res_group(tb, i, 0); load_res_group(tb, i, 1); }
                kstep(fa0, fw0, fa1, fw1, sa, sw, 1, true, sa_f, std::integral_constant<int, 2>{}, slot);
            } else {
                kstep(fa0, fw0, fa1, fw1, sa, sw, 1, true, sa_f, std::integral_constant<int, 0>{}, slot);
            }
        } else {
            if (kt == 0) kstep(fa0, fw0, fa1, fw1, sa, sw, 1, true, sa_f, std::integral_constant<int, 1>{}, slot);
            else kstep(fa0, fw0, fa1, fw1, sa, sw, 1, true, sa_f, std::integral_constant<int, 0>{}, slot);
        }
        a_advance();
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // k-step 1: MFMAs on (g, k-step 1); reads (g+1, k-step 0); DMAs W(g+2) into the W stage just freed
        kstep(fa1, fw1, fa0, fw0, sa_n, sw ^ 1, 0, false, sw, std::integral_constant<int, 0>{}, slot);
        w_advance();
        sa = sa_n;
        sw ^= 1;
        if (kt + 1 < nkt) {
            ++kt;
        } else {
            epilogue(m0, n0);
            tile += gstride;
            if (tile >= nblk) break;
            tile_coords(tile, m0, n0);
            load_rscale(m0);
            kt = 0;
        }
    }
    };
#ifdef PCAD_GEMM_STAGGER
    // one copy of the loop per wave, differing only in the DMA slot (wave-uniform branch, taken once)
    if (wave == 0) mainloop(std::integral_constant<int, 0>{});
    else if (wave == 1) mainloop(std::integral_constant<int, 1>{});
    else if (wave == 2) mainloop(std::integral_constant<int, 2>{});
    else mainloop(std::integral_constant<int, 3>{});
#else
    mainloop(std::integral_constant<int, kDmaSlot>{});
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing (unused) DMAs must not outlive the block's LDS
}

// persistent launch: 1 resident block per CU (LDS-limited), a multiple of 8 so block b stays on XCD b & 7
static int persistent_grid(int nblk) {
    const int cap = device_cu_count() / 8 * 8;
    return nblk < cap ? nblk : cap;
}

template <typename T, typename OutT, bool ROUND>
static hipError_t launch_gemm_t(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                int64_t M, int N, int K, hipStream_t s, bool a_blocked, int ksplit = 0) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    const bool vec = ((ldc * (int64_t)sizeof(OutT)) % 16 == 0) && (((uintptr_t)C) % 16 == 0);
    dim3 grid((unsigned)persistent_grid(tiles_m * tiles_n)), block(GEMM_THREADS);
    if (vec) {
        auto kfn = gemm_nt_kernel<T, OutT, ROUND, true, false>;
        if (hipError_t ae = ensure_dynamic_lds((const void*)kfn, GEMM_LDS)) return ae;
        hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K,
                           tiles_m, tiles_n, (float*)nullptr, (int64_t)0, 0, (int)a_blocked, ksplit);
    } else {
        auto kfn = gemm_nt_kernel<T, OutT, ROUND, false, false>;
        if (hipError_t ae = ensure_dynamic_lds((const void*)kfn, GEMM_LDS)) return ae;
        hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K,
                           tiles_m, tiles_n, (float*)nullptr, (int64_t)0, 0, (int)a_blocked, ksplit);
    }
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_gemm_split_t(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                      float* C2, int64_t ldc2, int nsplit, int64_t M, int N, int K, hipStream_t s,
                                      bool a_blocked) {
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
    dim3 grid((unsigned)persistent_grid(tiles_m * tiles_n)), block(GEMM_THREADS);
    auto kfn = gemm_nt_kernel<T, T, false, true, true>;
    if (hipError_t ae = ensure_dynamic_lds((const void*)kfn, GEMM_LDS)) return ae;
    hipLaunchKernelGGL(kfn, grid, block, GEMM_LDS, s, (const T*)A, lda, (const T*)W, ldw, (T*)C, ldc, M, N, K, tiles_m,
                       tiles_n, C2, ldc2, nsplit, (int)a_blocked, 0);
    return hipGetLastError();
}

hipError_t launch_gemm_nt_split(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, float* C2,
                                int64_t ldc2, int nsplit, int64_t M, int N, int K, int dt, hipStream_t s, bool a_blocked) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB || nsplit % 16 || (N - nsplit) % 16) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16) return hipErrorInvalidValue;
    if ((ldc * esz) % 16 || ((uintptr_t)C) % 16 || (ldc2 * 4) % 16 || ((uintptr_t)C2) % 16) return hipErrorInvalidValue;
    if (a_blocked && (lda * esz) % 128) return hipErrorInvalidValue;
    if (dt == BF16) return launch_gemm_split_t<bf16_t>(A, lda, W, ldw, C, ldc, C2, ldc2, nsplit, M, N, K, s, a_blocked);
    return launch_gemm_split_t<float>(A, lda, W, ldw, C, ldc, C2, ldc2, nsplit, M, N, K, s, a_blocked);
}

// shapes the 4-wave kernel takes: whole 256 x 256 tiles, unsigned 32-bit buffer offsets
template <typename T>
static bool quad_ok(int64_t lda, int64_t ldw, int64_t M, int N, bool two, int nsplit) {
    const int64_t esz_ = (int64_t)sizeof(T);
    return M % BM2 == 0 && N % BN2 == 0 && (!two || nsplit % 64 == 0) && M * lda * esz_ < ((int64_t)1 << 32) - 65536 &&
           (int64_t)N * ldw * esz_ < ((int64_t)1 << 32) - 65536;
}

// OutT != T only as <bf16_t, float>: bf16 operands, fp32 result stored as fp32 (the split-bf16 GEMMs of the fp32 model, api.hip
// "f32_gemm_split"); the fused epilogues exist for OutT == T only.
template <typename T, typename OutT = T>
static hipError_t launch_gemm256_t(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                   int N, int K, hipStream_t s, bool a_blocked, void* C2 = nullptr, int nsplit = 0,
                                   bool out_blocked = false, int epi_kind = EPI_NONE, GemmEpi epi = GemmEpi{nullptr, nullptr, nullptr},
                                   int ksplit = 0) {
    const int tiles_m = (int)((M + BM2 - 1) / BM2), tiles_n = (N + BN2 - 1) / BN2;
    dim3 grid((unsigned)persistent_grid(tiles_m * tiles_n)), block(GEMM_THREADS);
    static const bool quad = dev_env("PCAD_GEMM_NOQUAD") == nullptr;   // PCAD_DEV=1 only: the 8-wave kernel for A/B runs
    static const bool epi_swap = dev_env("PCAD_GEMM_EPI_PLAIN") == nullptr;  // 64-byte-contiguous epilogue stores (PCAD_DEV=1 PCAD_GEMM_EPI_PLAIN=1: the interleaved form, for A/B)
    const bool qok = quad_ok<T>(lda, ldw, M, N, C2 != nullptr, nsplit);
    if (epi_kind != EPI_NONE && !qok) return hipErrorInvalidValue;             // the fused epilogues exist on the 4-wave kernel only
#define PCAD_LAUNCH_Q(EPIK)                                                                                                     \
    do {                                                                                                                        \
        auto kq = gemm256q_kernel<T, OutT, EPIK>;                                                                               \
        if (hipError_t ae = ensure_dynamic_lds((const void*)kq, GEMM3_LDS)) return ae;                                          \
        hipLaunchKernelGGL(kq, grid, dim3(GEMMQ_THREADS), GEMM3_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K, \
                           tiles_m, tiles_n, (int)a_blocked, (OutT*)C2, nsplit, (int)out_blocked, (int)epi_swap, epi, ksplit); \
                                                                                             \
        return hipGetLastError();                                                                                               \
    } while (0)
    if constexpr (std::is_same<T, OutT>::value) {
        if (epi_kind == EPI_SCALE) PCAD_LAUNCH_Q(EPI_SCALE);
        if (epi_kind == EPI_RES) PCAD_LAUNCH_Q(EPI_RES);
    } else if (epi_kind != EPI_NONE) {
        return hipErrorInvalidValue;
    }
    if (quad && qok) PCAD_LAUNCH_Q(EPI_NONE);
#undef PCAD_LAUNCH_Q
    auto kr = gemm256r_kernel<T, OutT>;
    if (hipError_t ae = ensure_dynamic_lds((const void*)kr, GEMM3_LDS)) return ae;
    hipLaunchKernelGGL(kr, grid, block, GEMM3_LDS, s, (const T*)A, lda, (const T*)W, ldw, (OutT*)C, ldc, M, N, K, tiles_m,
                       tiles_n, (int)a_blocked, (OutT*)C2, nsplit, (int)out_blocked, ksplit);
    return hipGetLastError();
}

// in_proj form on the 256x256 kernel: columns [0, nsplit) -> C1, [nsplit, N) -> C2: two separate tensors of nsplit and
// N - nsplit columns, plain (contiguous rows) or both in the blocked layout.
hipError_t launch_gemm_nt_two(const void* A, int64_t lda, const void* W, int64_t ldw, void* C1, void* C2, int nsplit,
                              bool out_blocked, int64_t M, int N, int K, int dt, hipStream_t s, const float* rscale, int out_dt, int ksplit) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (out_dt < 0) out_dt = dt;
    const int osz = out_dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB || nsplit % 16 || N % 16 || nsplit <= 0 || nsplit >= N) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16) return hipErrorInvalidValue;
    if (((uintptr_t)C1) % 16 || ((uintptr_t)C2) % 16) return hipErrorInvalidValue;
    if (out_blocked && ((nsplit * osz) % 128 || ((N - nsplit) * osz) % 128)) return hipErrorInvalidValue;
    const int ek = rscale ? EPI_SCALE : EPI_NONE;
    const GemmEpi epi{rscale, nullptr, nullptr};
    if (out_dt != dt) {                    // bf16 operands -> fp32 outputs (split-bf16 in_proj of the fp32 model)
        if (dt != BF16 || out_dt != F32 || rscale) return hipErrorInvalidValue;
        if (ksplit && (ksplit < 0 || K != 3 * ksplit * (ROWB / 2))) return hipErrorInvalidValue;     // K = 3 Ko, ksplit = Ko / 64 K-tiles per part
        return launch_gemm256_t<bf16_t, float>(A, lda, W, ldw, C1, nsplit, M, N, K, s, false, C2, nsplit, out_blocked, EPI_NONE, epi, ksplit);
    }
    if (ksplit) return hipErrorInvalidValue;
    if (dt == BF16) return launch_gemm256_t<bf16_t>(A, lda, W, ldw, C1, nsplit, M, N, K, s, false, C2, nsplit, out_blocked, ek, epi);
    return launch_gemm256_t<float>(A, lda, W, ldw, C1, nsplit, M, N, K, s, false, C2, nsplit, out_blocked, ek, epi);
}

bool gemm_fold_shapes_ok(int64_t M, int D, int E, int dt) {
    const int64_t esz = dt == BF16 ? 2 : 4;
    const int Dp = fold_padded_width(D);
    // in_proj [M, D (rows Dp apart)] x [2E, D]^T (two blocked outputs), out_proj [M, E] x [Dp, E]^T (weight rows past D are zero);
    // every tensor below 2^32 bytes
    const bool in_ok = dt == BF16 ? quad_ok<bf16_t>(Dp, D, M, 2 * E, true, E) : quad_ok<float>(Dp, D, M, 2 * E, true, E);
    const bool out_ok = dt == BF16 ? quad_ok<bf16_t>(E, E, M, Dp, false, 0) : quad_ok<float>(E, E, M, Dp, false, 0);
    return in_ok && out_ok && (D * esz) % ROWB == 0 && (E * esz) % ROWB == 0 && M * (int64_t)Dp * 4 < ((int64_t)1 << 32);
}

hipError_t launch_gemm_nt_res(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, float* res, float* ssq, int64_t M,
                              int N, int K, int dt, hipStream_t s, bool a_blocked) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB || !res || !ssq || !C) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16 || ((uintptr_t)res) % 16) return hipErrorInvalidValue;
    if (a_blocked && (lda * esz) % 128) return hipErrorInvalidValue;
    // (Tried and removed, profiles/r04_ab_runs.txt r04d / r04e / r04g: delaying block b by ((b / 8) % 8) eighths of a tile so that an
    // eighth of the CUs is in its epilogue at a time - no effect with the fragment layout, 2.096 vs 2.097 ms.  Ablations: residual
    // loads from cache 1.94 ms, no write-back 1.90, neither 1.71 = the plain out_proj; i.e. the 4.3 GB of extra traffic costs
    // 0.37 ms, about half of its HBM time, the rest is hidden behind the mainloops.  Also removed: pulling the NEXT tile's residual
    // towards L2 with 8 throw-away loads per wave during a tile's last K-tile (so that the tile-start loads hit) - out_proj + residual
    // 2.05 -> 2.29 ms: the extra requests in the mainloop cost more than the tile-start latency they hide, r04l.)
    const GemmEpi epi{nullptr, res, ssq};
    if (dt == BF16) return launch_gemm256_t<bf16_t>(A, lda, W, ldw, C, N, M, N, K, s, a_blocked, nullptr, 0, false, EPI_RES, epi);
    return launch_gemm256_t<float>(A, lda, W, ldw, C, N, M, N, K, s, a_blocked, nullptr, 0, false, EPI_RES, epi);
}

hipError_t launch_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M,
                          int N, int K, int dt, int out_dt, bool round_bf16, hipStream_t s, bool a_blocked, int ksplit) {
    if (M <= 0 || N <= 0) return hipSuccess;
    const int esz = dt == BF16 ? 2 : 4;
    if (K <= 0 || (K * esz) % ROWB) return hipErrorInvalidValue;
    // wrap-around K cursor: bf16 [hi | lo] operands (2 Ko columns), fp32 result, K = 3 Ko, ksplit = Ko / 64
    if (ksplit && (ksplit < 0 || dt != BF16 || out_dt != F32 || round_bf16 || K != 3 * ksplit * (ROWB / 2))) return hipErrorInvalidValue;
    if ((lda * esz) % 16 || (ldw * esz) % 16 || ((uintptr_t)A) % 16 || ((uintptr_t)W) % 16)
        return hipErrorInvalidValue;
    if (a_blocked && (lda * esz) % 128) return hipErrorInvalidValue;
    static const bool no256 = dev_env("PCAD_GEMM_NO256") != nullptr;      // PCAD_DEV=1 only: force the 256x128 kernel
    const int osz = out_dt == BF16 ? 2 : 4;
    const bool big = !no256 && M >= 2048 && N >= 512 && N % 16 == 0 && (out_dt == dt || (dt == BF16 && out_dt == F32 && !round_bf16)) &&
                     (ldc * osz) % 16 == 0 && ((uintptr_t)C) % 16 == 0;
    if (big) {
        if (dt == BF16 && out_dt == F32)
            return launch_gemm256_t<bf16_t, float>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked, nullptr, 0, false, EPI_NONE, GemmEpi{nullptr, nullptr, nullptr}, ksplit);
        if (dt == BF16) return launch_gemm256_t<bf16_t>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked);
        return launch_gemm256_t<float>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked);
    }
    if (dt == BF16 && out_dt == BF16)
        return launch_gemm_t<bf16_t, bf16_t, false>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked);
    if (dt == BF16 && out_dt == F32)
        return round_bf16 ? launch_gemm_t<bf16_t, float, true>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked)
                          : launch_gemm_t<bf16_t, float, false>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked, ksplit);
    if (dt == F32 && out_dt == F32)
        return launch_gemm_t<float, float, false>(A, lda, W, ldw, C, ldc, M, N, K, s, a_blocked);
    return hipErrorInvalidValue;
}

}  // namespace pcad
