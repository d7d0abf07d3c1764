// Host-side launcher declarations (one per kernel family). All launch on the given stream; no sync.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace pcad {

enum DType { F32 = 0, BF16 = 1 };

// Developer A/B switches are environment variables that are honoured ONLY when PCAD_DEV=1 is also set; a production
// process never changes behaviour because of a stray variable.  (Numerics / sizing options of the ABI: pcad_set_option.)
inline const char* dev_env(const char* name) {
    static const bool dev = [] { const char* d = getenv("PCAD_DEV"); return d && d[0] == '1'; }();
    return dev ? getenv(name) : nullptr;
}

// Per-device launch state.  A handle is bound to the device that was current when it was created and several handles on several
// devices may live in one process (include/pcad.h "Threading / streams"), so nothing about a device is cached process-wide:
// the CU count and the "this kernel may use > 64 KiB of dynamic LDS" attribute are kept per (device ordinal[, kernel]).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);   // hipFuncSetAttribute once per (current device, kernel)
int device_cu_count();                                          // multiProcessorCount of the current device (256 on MI355X)

struct Positions {            // by-value kernel argument: the positions evaluated by the head kernel
    int n;                    // 0 => all L positions
    int p[16];
};

// norm.hip --------------------------------------------------------------------------------------
// fused residual add + RMSNorm (rms_norm_fn prenorm=True).  x/y dtype `dt`; residuals `rdt`.
// split_y (dt == F32 only): y is written as a bf16 tensor [rows, 2 D] = [hi | lo] (pack.hip launch_split_rows' format).
hipError_t launch_add_rmsnorm(const void* x, const void* res_in, const float* w, void* y, void* res_out,
                              int64_t rows, int D, float eps, int dt, int rdt, hipStream_t s, bool split_y = false);
// layer-0 variant: x = Emb[strand token] gathered on the fly (RCPS strands by index arithmetic).
// rstd_out != nullptr: the norm-folded form - y (may be nullptr) = the un-normalised embedding row, rstd_out[row] = its rstd, and
// res_out (fp32) is written in the 4-wave GEMM's fragment layout (common.hpp res_frag_off; 2 B L % 256 == 0) as a tensor of
// Dp = round_up(D, 256) columns whose padding columns are zeroed; y rows are Dp elements apart.
hipError_t launch_embed_rmsnorm(const int32_t* ids, const void* emb, const int32_t* comp8, const float* w,
                                void* y, void* res_out, int B, int L, int D, float eps, int dt, int rdt,
                                hipStream_t s, float* rstd_out = nullptr, int Dp = 0,     // Dp: padded width of res / y rows (0: D)
                                bool split_y = false);
// rstd[row] = rsqrt(sum of the np partial sums of squares of the row / D + eps)
hipError_t launch_rstd(const float* ssq, float* rstd, int64_t rows, int np, int D, float eps, hipStream_t s);
// final add + norm_f + RC re-assembly + tied RCPS LM head, only at the requested positions (a shared list `pos`,
// or one position per window from the device array `pos_per_seq` [B]).  h_compact: h holds only the evaluated rows,
// [(strand * P + q), D] as launch_gather_rows orders them (the residual stream `res` is always the full tensor).
// ids [B, L] + status (device word, or nullptr): token ids outside [0, 8) / per-window positions outside [0, L) set bits 1 / 2.
hipError_t launch_final_head(const void* h, const void* res, const float* w, const void* emb,
                             const float* emb_f32, const int32_t* comp8, void* hidden_out, float* logits_out,
                             int B, int L, int D, float eps, Positions pos, const int32_t* pos_per_seq, int dt, int rdt,
                             hipStream_t s, bool h_compact = false, const int32_t* ids = nullptr, int32_t* status = nullptr,
                             int res_frag = 0);            // res_frag != 0: fp32 residual in the fragment layout (common.hpp res_frag_off) of that (padded) width
// hidden_states[i] (block input = previous mixer output / embedding) assembled in RCPS layout.
hipError_t launch_assemble_hidden(const void* h, void* out, int B, int L, int D, int dt, hipStream_t s);
hipError_t launch_embed_only(const int32_t* ids, const void* emb, const int32_t* comp8, void* h, int B, int L,
                             int D, int dt, hipStream_t s);
// a[i] = round_dt(a[i] + b[i]), n % 8 == 0: the sum of the two directions' out_proj outputs (strict reference order only)
hipError_t launch_add_round(void* a, const void* b, int64_t n, int dt, hipStream_t s);

// gemm.hip --------------------------------------------------------------------------------------
// C[M,N] = A[M,K] W[N,K]^T.  K multiple of 128 bytes; lda/ldw multiples of 16 bytes.
// out_dt: F32 or == dt.  round_bf16: round the fp32 result to bf16 precision before an F32 store.
// a_blocked: A is in the blocked layout with rows of lda elements.
// ksplit != 0 (dt == BF16, out_dt == F32 only): the wrap-around K cursor of the split-bf16 GEMMs (api.hip "f32_gemm_split") - both
// operands are bf16 [hi | lo] tensors of 2 Ko columns, K = 3 Ko, ksplit = Ko / 64 (K-tiles per part): the cursor reads A as hi, lo, hi
// and W as hi, hi, lo, i.e. C = a_hi w_hi + a_lo w_hi + a_hi w_lo accumulated in fp32 in that order.
hipError_t launch_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                          int64_t M, int N, int K, int dt, int out_dt, bool round_bf16, hipStream_t s,
                          bool a_blocked = false, int ksplit = 0);
// x_proj form: columns [0, nsplit) -> C (dtype dt, ld ldc); columns [nsplit, N) -> C2 (fp32, rounded to dt's
// precision, ld ldc2).  nsplit % 16 == 0.
hipError_t launch_gemm_nt_split(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                float* C2, int64_t ldc2, int nsplit, int64_t M, int N, int K, int dt, hipStream_t s,
                                bool a_blocked = false);

// in_proj form on the 256x256 kernel: columns [0, nsplit) -> C1, [nsplit, N) -> C2 (separate tensors of nsplit and
// N - nsplit columns; both plain or both blocked).
// rscale (or nullptr): per-row factor [M] applied to the result before it is rounded (the norm-folded in_proj: rstd of the row).
// out_dt: -1 = dt; F32 with dt == BF16: bf16 operands, fp32 outputs (no rscale) - the split-bf16 in_proj of the fp32 model, with
// ksplit = Ko / 64 and [hi | lo] operands as in launch_gemm_nt.
hipError_t launch_gemm_nt_two(const void* A, int64_t lda, const void* W, int64_t ldw, void* C1, void* C2, int nsplit,
                              bool out_blocked, int64_t M, int N, int K, int dt, hipStream_t s, const float* rscale = nullptr,
                              int out_dt = -1, int ksplit = 0);

// out_proj of the norm-folded layer form, on the 4-wave kernel only (gemm_fold_shapes_ok):
//   res [M, N] fp32 (FRAGMENT layout, common.hpp res_frag_off) += A . W^T (in place);  C [M, N] (plain rows, dtype dt; unused for
//   fp32: see api.hip) = round(res);  ssq [M, N / 128] = per-row partial
//   sums of squares of the updated residual, one per 128-column wave tile (deterministic; reduced by launch_rstd).
hipError_t launch_gemm_nt_res(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, float* res, float* ssq, int64_t M,
                              int N, int K, int dt, hipStream_t s, bool a_blocked);
// whether a chunk of M token-rows of a (D, E) model can run the folded form (whole 256 x 256 tiles for both projections)
bool gemm_fold_shapes_ok(int64_t M, int D, int E, int dt);
inline int fold_padded_width(int D) { return (D + 255) / 256 * 256; }     // width of res / u in the folded form (l20: 384 -> 512)

// conv.hip --------------------------------------------------------------------------------------
// out_blocked: yf / yr in the blocked layout (common.hpp::blocked_off), buffers padded to a multiple of 8 rows.
// in_blocked: x is a [S*L, E] tensor in the blocked layout (ldx ignored).
hipError_t launch_conv_bidir(const void* x, int64_t ldx, const float* wf, const float* bf, const float* wr,
                             const float* br, void* yf, void* yr, int S, int L, int E, int dt, bool out_blocked,
                             hipStream_t s, bool in_blocked = false);

// convx.hip -------------------------------------------------------------------------------------
// Fused conv1d+SiLU (both directions) + x_proj (both directions), Rp == 64 or 96: x [S*L, E] blocked -> xc0 / xc1 [S*L, E]
// blocked, dtl_d [S*L, Rp] (dtype dt), bc_d [S*L, 32] fp32; Wx_d [Rp + 32, E].  convw: taps packed by launch_pack_convw.
size_t convx_packed_bytes(int E, int dt);
hipError_t launch_pack_convw(const float* wf, const float* bf, const float* wr, const float* br, float* out, int E, int dt,
                             hipStream_t s);
hipError_t launch_convx(const void* x, const float* convw, const void* Wx0, void* xc0, void* dtl0, float* bc0,
                        const void* Wx1, void* xc1, void* dtl1, float* bc1, int S, int L, int E, int dt, hipStream_t s, int Rp = 64,
                        bool dtl_split = false,       // dtl_split (dt == F32): dtl_d is written as bf16 [S*L, 2 Rp] = [hi | lo]
                        bool w_split = false,         // w_split (dt == F32): Wx_d is the bf16 [Rp + 32, 2E] copy of launch_pack_convx_wsplit and
                                                      // x_proj runs as three bf16 MFMA products per fp32 product
                        float* part_ws = nullptr,     // scratch of convx_split_bytes(): small launches split the channel walk over several
                                                      // blocks per row tile (convx_ksplit) and a second tiny kernel adds their partial x_dbl
                        int policy_S = 0);            // strands the K-split policy is evaluated for (0: S; see scan_segment_bytes)
int convx_ksplit(int S, int L, int E, int dt);        // K-split factor for this launch shape (1: none)
size_t convx_split_bytes(int S, int L, int E, int dt, int Rp, int policy_S = 0);
hipError_t launch_pack_convx_wsplit(const float* src, int64_t ld, void* dst, int rows, int E, hipStream_t s);
// dt_rank padded to the K granule of the fused kernels: 64 up to dt_rank 64 (every PlantCaduceus size), else the next multiple of 32
// (PlantCAD2 Large: dt_rank 96 -> 96)
inline int padded_dt_rank(int R) { return R <= 64 ? 64 : (R + 31) / 32 * 32; }

// scan.hip --------------------------------------------------------------------------------------
// Selective scan of one direction.  delta == nullptr: fused dt_proj (delta tile = dt_low[rows, Rp] . Wdt[E, Rp]^T on
// MFMA inside the kernel, Rp % 32 == 0, zero-padded K);  delta != nullptr: delta [rows, E] read from memory.
// bc: fp32 [rows, 32] = B_t | C_t.  The recurrence uses A2 * a_scale as the base-2 decay rate: pass
// (A * log2(e), 1) or (A, log2(e)).
// uy_blocked: u and y are in the blocked layout (whole-tensor row index s*L + t, buffers padded to 8 rows).
// z_blocked (needs uy_blocked): z is a separate [S*L, E] tensor in the same blocked layout (ldz ignored).
// seg_ws (fused dt_proj form only): scratch of scan_segment_bytes(S, L, E) bytes; when given and scan_segments() > 1 the walk of
// every strand is cut into segments that run as separate workgroups (long sequences with few strands: PlantCAD2's 8 192-bp windows).
// walk_len (0 or >= L: the whole strand): only the first walk_len steps of the walk are run (forward: rows [0, walk_len); reverse:
// rows [L - walk_len, L)); the other rows of y are not written.  Ignored when the walk is cut into segments.
hipError_t launch_scan(const void* u, const void* z, int64_t ldz, const void* delta, const void* dt_low, int64_t lddt,
                       const void* Wdt, int Rp, const float* bc, const float* A2, float a_scale, const float* Dskip,
                       const float* dbias, void* y, int S, int L, int E, bool reverse, int accumulate, int dt,
                       hipStream_t s, bool uy_blocked = false, bool z_blocked = false, float* seg_ws = nullptr, int walk_len = 0,
                       void* ysplit = nullptr, bool dt_split = false, int policy_S = 0);
// dt_split (dt == F32, fused dt_proj): dt_low is bf16 [rows, lddt >= 2 Rp] = [hi | lo] and Wdt bf16 [E, 2 Rp] = [hi | lo] (Rp = the padded
// dt_rank, <= 96): the fp32 model's dt_proj as three bf16 MFMA products per fp32 product ("f32_gemm_split"; K walk hi.hi, lo.hi, hi.lo).
// ysplit (fp32 engine layouts only: dt == F32, fused dt_proj, blocked u / y / z, L % 8 == 0; reverse gating launch, unsegmented, whole
// walk): the output is written NOT to y but as out_proj's split-bf16 operand, bf16 [rows8, 2E] blocked = [hi | lo] (pack.hip).

// Segments per strand for the scan of S strands of L steps over E channels.  Pass A + pass B cost ~1.8x the arithmetic of one
// walk, and a single wave per SIMD already keeps the VALU ~65 % busy, so cutting only pays when most SIMDs would otherwise idle
// (measured: 1 536 waves at L = 8 192 are 15 % FASTER uncut).  Cut when the launch has at most 768 waves (0.75 per SIMD): into
// enough segments for ~2 300 waves, at most 8 (16 for short strands); long strands (L >= 2 048) into segments of at least 512 steps (16 blocks of 32),
// short ones (the reference's 512-bp windows in batches of at most 8 at l32: notebooks/examples.ipynb:141-170 runs B = 1) into
// segments of at least 64 steps when the launch has at most 512 waves, where the chip is so empty that two 64-step passes beat one
// 512-step walk (profiles/r04j_small_batch.txt, seq/s without -> with: l32 B = 1 61 -> 107, B = 8 455 -> 548; l20 B = 1 119 -> 257,
// B = 8 929 -> 1682; at 768 waves - l20 B = 32 - it loses 13 %, hence the lower bound for short strands).  Round 5: short strands into up
// to 16 segments of at least 32 steps (one delta tile): l32 B = 1 107 -> 118 seq/s, l20 B = 1 / 2 / 4 243 / 474 / 918 -> 283 / 546 / 996,
// B >= 8 unchanged (profiles/r05g_small_batch.txt, same box).
inline int scan_segments(int S, int L, int E, int* seg_blocks) {
    const int64_t waves = (int64_t)S * (E / 64);
    const int nblk = (L + 31) / 32;
    const int min_blocks = L >= 2048 ? 16 : 1;
    int G = 1;
    if (L >= 256 && waves > 0 && waves <= (L >= 2048 ? 768 : 512)) {
        G = (int)((2304 + waves - 1) / waves);
        if (G > (L >= 2048 ? 8 : 16)) G = L >= 2048 ? 8 : 16;
        if (G > nblk / min_blocks) G = nblk / min_blocks;
        if (G < 3) G = 1;                              // below ~3x the waves the second pass is not paid for
    }
    const int sb = (nblk + G - 1) / G;
    G = (nblk + sb - 1) / sb;                          // no empty segment
    if (seg_blocks) *seg_blocks = sb;
    return G;
}
// policy_S (here and in launch_scan / launch_convx; 0: S): the strand count the small-launch policy is evaluated for.  The engine
// passes the strands of the WHOLE pcad_forward batch, so that every chunk of a call runs the same form and results do not depend on
// how the batch was cut into chunks, bit for bit; the scratch is sized for the S strands of the launch.
inline size_t scan_segment_bytes(int S, int L, int E, int policy_S = 0) {
    const int G = scan_segments(policy_S > 0 ? policy_S : S, L, E, nullptr);
    return G > 1 ? (size_t)S * G * E * 17 * sizeof(float) : 0;        // [S][G][E][16] states + [S][G][E] delta sums
}

// "Pair" walks (launches with few waves - PlantCAD2's 8 192-bp windows in batches of a few dozen, 512-bp windows in batches of a few
// dozen: at most 3.5 waves per SIMD; at 2 the VALU idles a third of the time): both directions of the layer in ONE launch, each
// walking half of its strand per launch - launch 1: forward rows [0, L/2) and reverse rows [L/2, L) from zero states, outputs ungated
// (or each gated, "gate_each"), end states kept; launch 2: the second halves from those states, each adding what the OTHER direction's
// first launch left in y and gating.  Twice the waves per launch, the same arithmetic, no extra pass.  Against the plain two-launch
// form the bf16 model's gate-once sum is rounded at the other direction's partial on half of the rows (y_rev stored, y_fwd added in
// fp32, instead of the reverse): results of a pair walk and of a plain walk agree to bf16 rounding of one addend (fp32 model: to fp32
// summation order), exactly as the segmented form does - and the form is chosen like it: from the strands of the whole pcad_forward
// call, governed by "scan_segments".
struct ScanDirection { const void *u, *dt_low, *Wdt; const float *bc, *A2, *Dskip, *dbias; };
inline bool scan_pair_wanted(int S, int L, int E) {
    const int64_t waves = (int64_t)S * (E / 64);
    static const int64_t max_waves = [] { const char* v = dev_env("PCAD_PAIR_MAX_WAVES"); return v ? (int64_t)atoll(v) : (int64_t)3584; }();   // PCAD_DEV=1 A/B knob
    return L % 64 == 0 && L >= 128 && waves > 0 && waves <= max_waves && scan_segments(S, L, E, nullptr) == 1;
}
inline size_t scan_pair_bytes(int S, int E) { return (size_t)2 * S * 2 * E * 16 * sizeof(float); }     // [dir][S][2][E][16] states
// u / y / z blocked [S*L (8-row padded), E]; dt_low [S*L, lddt]; Wdt [E, Rp] (dt_split: bf16 [hi | lo] operands as in launch_scan);
// A2 pre-scaled by log2(e); ws: scan_pair_bytes(S, E); ysplit (fp32 + f32_gemm_split): the gated output as out_proj's [hi | lo] operand.
hipError_t launch_scan_pair(const ScanDirection& fwd, const ScanDirection& rev, const void* z, int64_t lddt, int Rp, void* y, int S, int L, int E,
                            bool gate_each, int dt, hipStream_t s, float* ws, void* ysplit = nullptr, bool dt_split = false,
                            int phases = 3);       // bit 0: the first-half launch, bit 1: the second-half launch

// pack.hip --------------------------------------------------------------------------------------
// rows (strand b, p_q) and (strand B + b, L - 1 - p_q) of a [2B*L, E] activation tensor (plain or blocked) -> out[(strand * P + q), E]
hipError_t launch_gather_rows(const void* src, void* out, int B, int L, int E, Positions pos, int dt, bool blocked,
                              hipStream_t s);
// generic 2-D copy/convert with zero padding: dst[r, c] (dst_dt, ld = dst_ld) = src[r, c] for r < rows, c < cols else 0
hipError_t launch_pack2d(const void* src, int src_dt, int64_t src_ld, void* dst, int dst_dt, int64_t dst_ld,
                         int rows, int cols, int dst_rows, int dst_cols, hipStream_t s);
// dst[r, c] (dst_dt) = src[r, c] * scale[c]: in_proj weight with the RMSNorm weight folded into its columns (norm-folded form)
hipError_t launch_pack_scale_cols(const void* src, int src_dt, int64_t src_ld, const float* scale, void* dst, int dst_dt,
                                  int64_t dst_ld, int rows, int cols, hipStream_t s);
// Layer 0 of the norm-folded form.  Its in_proj operand is one of the V embedding rows, so its output is one of V rows:
//   tab [V, N2] (dtype dt) = round((emb [V, D] . Wf [N2, D]^T) * rstd(emb row))                 (bind time)
//   x, z [2 B L, E] blocked = rows of tab gathered by token (rc strand by index arithmetic)    (instead of the layer-0 in_proj launch)
hipError_t launch_embed_inproj_table(const void* emb, const void* Wf, void* tab, int V, int D, int N2, float eps, int dt, hipStream_t s);
hipError_t launch_embed_xz_gather(const int32_t* ids, const int32_t* comp8, const void* tab, void* x, void* z, int B, int L, int E, int dt,
                                  hipStream_t s);
// split-bf16 operands of the fp32 model's big GEMMs ("f32_gemm_split"): v = hi + lo, hi = bf16(v), lo = bf16(v - hi);
//   weights     [rows, cols] (fp32 / bf16 source) -> bf16 [rows, 2 cols] = [hi | lo]                    (bind time)
//   activations fp32 [rows, K] (plain or blocked) -> bf16 [rows, 2 K]    = [hi | lo] (plain or blocked), K % 64 == 0
// so that a_hi w_hi + a_lo w_hi + a_hi w_lo is ONE bf16 GEMM of 3 K / 64 K-tiles whose cursor wraps around both operands
// (gemm.hip "wrap-around K cursor": launch_gemm_nt(..., K = 3 K, ksplit = K / 64)) with an fp32 result.
hipError_t launch_pack_split_w(const void* src, int src_dt, int64_t src_ld, void* dst, int rows, int cols, hipStream_t s);
hipError_t launch_split_rows(const float* src, int64_t src_ld, void* dst, int64_t rows, int K, bool src_blocked, bool dst_blocked,
                             hipStream_t s);       // src_ld: elements between plain source rows (ignored for a blocked source)
// A2[e, n] = -exp(A_log[e, n]) * log2(e)   (A_log read through its storage dtype)
hipError_t launch_pack_A(const void* A_log, int src_dt, float* A2, int64_t n, float scale, hipStream_t s);

}  // namespace pcad
