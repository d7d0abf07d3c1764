/*
 * pcad.h — C ABI of the MI355X-native PlantCaduceus forward engine (libpcad.so).
 *
 * Drop-in boundary for ONE path of kuleshov-group/PlantCaduceus: the masked-LM forward that
 * `src/zero_shot_score.py:115` (`model(input_ids=curIDs)`) and `src/train_XGBoost.py:104`
 * (`model(input_ids=..., output_hidden_states=True)`) run, i.e. the third-party operators
 *   mamba_ssm.ops.selective_scan_interface.selective_scan_fn / mamba_inner_fn   (mamba-ssm==2.2.2)
 *   causal_conv1d.causal_conv1d_fn                                              (causal-conv1d==1.4.0)
 *   mamba_ssm.ops.triton.layer_norm.rms_norm_fn                                 (Triton fused add-norm)
 *   CaduceusMixer / RCPS wrappers                                               (HF-hub remote code)
 * pinned by reference env/requirements.txt:9-10 and env/environment.yml:12.
 *
 * Conventions
 *   - plain C: pointers and sizes only; no torch / HIP types in any signature (`pcad_stream` is a
 *     hipStream_t passed as void*).
 *   - every buffer and every stream is OWNED BY THE CALLER (ids, outputs, weights, weight arena, workspace); the library
 *     owns only the opaque handle (plus the HIP events of the optional profiling mode, freed with it).
 *     No allocation, no host synchronisation inside pcad_forward.
 *   - all work is enqueued on the caller's stream; a handle is bound to the current device and is not
 *     thread-safe; distinct handles on distinct devices are independent (one process per GPU).
 *   - status: 0 = OK, negative = error (enum below); message via thread-local pcad_last_error().
 *   - activation layout is token-major: [strand, position, channel] (channel contiguous).
 */
#ifndef PCAD_H
#define PCAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCAD_VERSION 100 /* 0.1.0 */
#define PCAD_MAX_VOCAB 8
#define PCAD_MAX_POSITIONS 16

typedef void* pcad_stream;            /* hipStream_t */
typedef struct pcad_engine* pcad_handle;

enum pcad_status {
    PCAD_OK = 0,
    PCAD_ERR_INVALID = -1,      /* bad argument / unsupported shape */
    PCAD_ERR_UNBOUND = -2,      /* weights not bound */
    PCAD_ERR_WORKSPACE = -3,    /* workspace or arena too small / misaligned */
    PCAD_ERR_MISSING = -4,      /* a required tensor name was not supplied */
    PCAD_ERR_HIP = -5           /* HIP runtime error (launch failure, ...) */
};

enum pcad_dtype { PCAD_F32 = 0, PCAD_BF16 = 1 };

/* Model geometry (HF config.json of the snapshot; replaces CaduceusConfig -> Mamba(**ssm_cfg)). */
typedef struct pcad_config {
    int32_t d_model;            /* D */
    int32_t n_layer;
    int32_t d_state;            /* N, must be 16 */
    int32_t d_conv;             /* W, must be 4 */
    int32_t expand;             /* E = expand * D */
    int32_t dt_rank;            /* R */
    int32_t vocab;              /* padded vocabulary, must be 8 */
    float   eps;                /* RMSNorm epsilon (1e-5) */
    int32_t dtype;              /* pcad_dtype of activations / GEMM operands (torch_dtype of the model) */
    int32_t residual_in_fp32;   /* residual stream kept in fp32 (reference default) */
    int32_t complement[PCAD_MAX_VOCAB]; /* token id -> complementary base id (RCPS) */
} pcad_config;

/* A named tensor in the reference's state-dict naming, device memory, row-major. */
typedef struct pcad_tensor {
    const char* name;
    const void* data;
    int32_t     dtype;          /* pcad_dtype */
    int32_t     ndim;
    int64_t     shape[4];
} pcad_tensor;

int         pcad_version(void);
const char* pcad_last_error(void);
/* sha1 (16 hex digits) over the kernel sources the library was built from (csrc/source_hash.py): a loader that has the
 * sources beside it compares the two and refuses a stale binary (plantcaduceus_amd/engine.py, bench.py). */
const char* pcad_build_hash(void);

/* Replaces: AutoModelForMaskedLM.from_pretrained(...).to(device)  (src/zero_shot_score.py:91-97) */
int    pcad_create(const pcad_config* cfg, pcad_handle* out);
void   pcad_destroy(pcad_handle h);

/* Options (call before pcad_workspace_bytes / pcad_forward):
 *   "chunk_seqs"  windows per pass through the layer stack (0 = default: as many as the kernels' unsigned 32-bit in-tensor
 *                 offsets allow, (2^32 - 2 MiB) / (d_inner * elem) token-rows = 1 023 windows of 512 bp at l32 bf16, the batch
 *                 split evenly into the fewest such chunks - or into up to twice as many when that makes the chunk's token-rows a
 *                 multiple of 16 384, i.e. whole rounds of the persistent GEMMs: the fp32 model's 1 024 windows run as 4 x 256, not
 *                 3 x 342).  Results do not depend on it (rows are independent) - bit for bit
 *                 (the small-launch forms of "scan_segments" are chosen for the whole batch of the call, not per chunk)
 *                 whenever every chunking runs the same layer form, which holds for window lengths that are multiples of 128
 *                 (every shipped use); for other lengths "norm_fold" engages per chunk (whole 256-row tiles only), so two
 *                 chunkings of a bf16 batch can differ by bf16 rounding; the
 *                 workspace does: ~30 MB per window at l32 bf16, i.e. up to ~30 GB for one 1 023-window chunk (a batch of
 *                 1 024 runs as two chunks of 512 in a 15.3 GB workspace) - always size it with pcad_workspace_bytes.
 *   "workspace_limit_mb"  N > 0: chunks are sized so that pcad_workspace_bytes stays at or below N MiB whatever the batch (a server
 *                 holding several models on one GPU bounds each one's slab; the default chunking takes up to ~30 GB for a
 *                 1 023-window chunk at l32 bf16); smaller chunks cost throughput (1 024 windows as 16 chunks instead of 2: -7 %),
 *                 never results.  One window always runs, even if its workspace exceeds the limit.  0 (default): no limit.
 *   "last_layer_shortcut"  1 (default): when pcad_forward is given a list of positions, the LAST layer's scans stop at the
 *                 furthest evaluated row and its out_proj runs on the evaluated rows only (bit-identical outputs);  0: full layer.
 *   "gate_each"   1: SiLU(z) applied to each direction's scan output, each rounded, then summed — the reference's order
 *                 (two selective_scan_fn calls);  0 (default): applied once to the sum of both directions (same value in
 *                 exact arithmetic, one rounding fewer, faster).
 *   "norm_fold"   1 (default for the bf16 model; -1 restores "default"): the fused add + RMSNorm launch between two blocks is folded
 *                 into the GEMMs around it: out_proj's epilogue
 *                 adds its fp32 result to the fp32 residual stream in place and writes the rounded sum plus per-row partial sums of
 *                 squares (deterministic, no atomics); in_proj runs on that un-normalised operand with W_in . diag(w_norm) (folded
 *                 when the weights are bound) and multiplies by the row's rstd before it rounds.  Same value in exact arithmetic
 *                 as rms_norm_fn(prenorm=True, residual_in_fp32=True); rounding points move (the mixer output is not rounded
 *                 before it is added; in_proj's operand is round(res) instead of round(res * rstd * w)).  Used for chunks whose
 *                 token-rows are whole 256-row tiles (a d_model that is not a multiple of 256 - PlantCaduceus_l20's 384 - is padded
 *                 to the next one inside the workspace) with an fp32 residual stream,
 *                 never by pcad_forward_all_hidden (+4.5 % end to end, same-box A/B, profiles/r04_ab_runs.txt).  Decided once
 *                 per pcad_forward call: every chunk of the batch runs folded or none does.  The form needs extra weight copies
 *                 (+37 % of the arena at l32) that are carved and packed only if the options in force at
 *                 pcad_weight_arena_bytes / pcad_bind_weights time ask for it: set "norm_fold" 1 on an fp32 model BEFORE those
 *                 calls (afterwards pcad_forward refuses); turning it off later is always possible;
 *                 0 (default for the fp32 model, whose 1e-4 parity budget would pay for accumulating onto the residual: 2.2e-5 of
 *                 max after 32 layers instead of 1.3e-6): the reference's operation order (one add + RMSNorm launch per block).
 *   "f32_gemm_split"  1: the fp32 model's two big projections (in_proj, out_proj: 3/4 of its time on the fp32 MFMA instructions) run
 *                 as split-bf16 GEMMs (pcad_gemm_nt_split below: three bf16 products per fp32 product, fp32 accumulation and
 *                 result); everything else of the fp32 model is unchanged.  Measured 4e-7 of the logits' range against the fp32
 *                 oracle after 32 layers (the plain fp32 GEMMs: 1e-6), arg-max exact - inside north_star's 1e-4 - at about twice
 *                 the fp32 model's speed.  0 (default): fp32 MFMA.  Needs bf16 [hi | lo] weight copies (the size of the fp32 weights
 *                 again) packed at bind time: set it before pcad_weight_arena_bytes / pcad_bind_weights.  Ignored by the bf16 model
 *                 and while "norm_fold" 1 is forced.  Operands are stored once as [hi | lo] (the byte size of the fp32 tensor) and
 *                 the GEMMs' K-tile cursor wraps around them (hi.hi, lo.hi, hi.lo), so the chunk cap is the fp32 model's own,
 *                 (2^32 - 2 MiB) / (4 d_inner) token-rows.  "last_layer_shortcut" runs the evaluated rows through the same
 *                 split-bf16 product (bit-identical to the full layer).
 *   "reference_order"  one switch over the options above, for users who want every rounding point where the reference has it
 *                 (BiMambaWrapper "add" of two Mamba calls, rms_norm_fn(prenorm=True, residual_in_fp32=True), mamba_inner_fn):
 *                 0 (default): the engine's defaults ("gate_each" 0, "norm_fold" default, layer 0's in_proj as a table);
 *                 1: "gate_each" 1 + "norm_fold" 0 (hence no layer-0 table) - the only reordering left is the tied out_proj applied
 *                    once to y_fwd + y_rev (linearity) instead of once per direction;
 *                 2: strict - additionally each direction's scan output gets its own tied out_proj launch, each result is stored in
 *                    the model dtype, and the two are added and rounded, exactly BiMambaWrapper's `out + out_rev`: the rounding
 *                    points of oracle/c "ref_order".  What remains different from the reference is fp32 summation order inside the
 *                    GEMMs / scan and the exp / log implementations - the noise any two BLAS libraries have between them.
 *                 Cost at l32 bf16 (same box, profiles/r05_*): see INTEGRATION.md "Operation order".  Setting "gate_each" /
 *                 "norm_fold" afterwards overrides the respective part; level 2 is ignored while "norm_fold" is forced to 1.
 *   "scan_segments"  1 (default): when a launch has few scan waves (at most 768 for L >= 2 048: PlantCAD2's 8 192-bp windows in small
 *                 batches; at most 512 for shorter windows: up to 8 windows of 512 bp at l32) the scan of every strand is cut into up to 8 (short windows: 16, of at least 32 steps) segments that run as
 *                 separate workgroups (zero-state pass, carry, real pass: ~1.8x the arithmetic for up to 8x the parallelism;
 *                 results equal up to fp32 rounding of the carried decay product);  0: one workgroup walks the whole strand.
 *                 Launches with more waves than that but at most 3 584 per direction (3.5 per SIMD: 32 windows of 8 192 bp at the
 *                 PlantCAD2 Medium / Large widths, up to 56 windows of 512 bp at l32) take the PAIR form instead (round 6): both
 *                 directions of a layer run in one launch and walk half a strand per launch - forward rows [0, L/2) with reverse rows
 *                 [L/2, L), then the other halves from the kept states, each adding what the other direction's first launch left and
 *                 gating: twice the waves per launch, the same arithmetic, no extra pass (L % 64 == 0; never the last layer, whose
 *                 walks "last_layer_shortcut" shortens).  Against the plain two-launch form the bf16 model's gate-once sum is rounded
 *                 at the other direction's partial on half of the rows: equal to bf16 rounding of one addend (fp32: summation order;
 *                 bf16 with "gate_each" / "reference_order" >= 1: bit-identical).  +13 % at 32 x 8 192 bp (Medium), +12 % at 16 x 512 bp.
 *                 The same switch governs the K-split of the fused conv + x_proj kernel for launches of at most 64 row tiles (up to 8
 *                 windows of 512 bp): several blocks per row tile each walk a share of the channels and a second tiny kernel adds
 *                 their partial x_proj sums in a fixed order (deterministic; another fp32 summation order than the unsplit walk).
 *                 Both forms are selected from the batch size of the pcad_forward call (not from the chunk), so two calls with the
 *                 same batch agree bit for bit whatever "chunk_seqs" is; calls with different batch sizes may differ by fp32 rounding
 *                 of these sums when one of them is small enough to take a split form.
 *                 Never used by the benchmark batch (1 024 windows: 65 536 waves).
 *   "debug_repeat_class" / "debug_repeat"  measurement aid (tools/power_probe.py): every idempotent launch of ONE pcad_kernel_class
 *                 (in_proj, conv + x_proj, the forward-direction scan, the reference-order out_proj) is issued `debug_repeat` times
 *                 back to back, so that a forward is seconds of that kernel at the engine's own launch sizes while board power is
 *                 sampled; outputs are unchanged.  Default: class -1 (none), repeat 1.
 *   "poison_workspace"  1: debug aid — the workspace is filled with 0xFF bytes (NaN in every dtype) before each forward, so a
 *                 read of anything this forward did not write shows up as NaN outputs (tests/test_gpu_model.py).
 * The library reads NO environment variables unless PCAD_DEV=1 is set (developer A/B switches, see csrc/kernels.hpp). */
int    pcad_set_option(pcad_handle h, const char* key, int64_t value);

/* Asynchronous input validation.  The reference raises (nn.Embedding index error / tensor index error) for a token id
 * outside the vocabulary or an evaluated position outside the window; pcad_forward* never synchronise, so they report these
 * through a caller-owned DEVICE word: bit PCAD_STATUS_BAD_TOKEN / PCAD_STATUS_BAD_POSITION is OR-ed into *status by the
 * forward's last kernel (such an id is embedded as id & 7, such a position is clamped, so nothing is read out of bounds and
 * the affected rows are simply wrong).  The caller zeroes the word, and reads it at its next synchronisation point.
 * NULL (default): no reporting. */
enum pcad_status_bits { PCAD_STATUS_BAD_TOKEN = 1, PCAD_STATUS_BAD_POSITION = 2 };
int    pcad_set_status_buffer(pcad_handle h, int32_t* status);

/* Bytes of caller-owned device memory that pcad_bind_weights packs the model into (depends on "norm_fold" / "reference_order" as set
 * when it is called: query it after the options, right before binding). */
size_t pcad_weight_arena_bytes(pcad_handle h);

/* Pack the reference-named tensors (fp32 or bf16, device pointers) into `arena` in the engine's layouts
 * (padded x_proj / dt_proj, pre-exponentiated A, tied in_proj/out_proj stored once).  Tied duplicates
 * (`mamba_rev.in_proj/out_proj`, `lm_head`) may be absent.  The arena must outlive the handle. */
int    pcad_bind_weights(pcad_handle h, const pcad_tensor* tensors, int n,
                         void* arena, size_t arena_bytes, pcad_stream stream);

/* Workspace needed by pcad_forward for `batch` sequences of `seqlen` tokens (both strands). */
size_t pcad_workspace_bytes(pcad_handle h, int batch, int seqlen);

/* Replaces: outputs = model(input_ids=ids[, output_hidden_states=True])
 *           (src/zero_shot_score.py:115; src/train_XGBoost.py:104).
 *   ids        device int32 [B, L] token ids (0..7)
 *   positions  HOST int32 [P] sequence positions to evaluate, or NULL with P == 0 for all L positions
 *              (P <= PCAD_MAX_POSITIONS).  Let Q = P ? P : L.
 *   hidden_out device [B, Q, 2*D] in cfg.dtype, or NULL: hidden_states[-1] rows
 *              (= cat(H(ids)[p], reverse_channels(H(rc ids)[L-1-p])))
 *   logits_out device fp32 [B, Q, vocab], or NULL: RCPS LM-head logits (`.logits.float()`)
 */
int    pcad_forward(pcad_handle h, const int32_t* ids, int B, int L,
                    const int32_t* positions, int P,
                    void* hidden_out, float* logits_out,
                    void* workspace, size_t workspace_bytes, pcad_stream stream);

/* In-silico-mutagenesis form (reference pipelines/in-silico-mutagenesis -> src/zero_shot_score.py -input-vcf, one
 * masked forward per variant position): window b is evaluated at its own position pos_per_seq[b] (device int32 [B]; a value
 * outside [0, L) is clamped AND reported through pcad_set_status_buffer).  hidden_out [B, 1, 2*D] / logits_out [B, 1, vocab],
 * either may be NULL. */
int    pcad_forward_at(pcad_handle h, const int32_t* ids, int B, int L, const int32_t* pos_per_seq,
                       void* hidden_out, float* logits_out,
                       void* workspace, size_t workspace_bytes, pcad_stream stream);

/* Per-layer mixer outputs for the last pcad_forward are not kept; this variant additionally writes
 * hidden_states[0..n_layer-1] (the inputs of every block) to `all_hidden` ([n_layer, B, L, 2*D],
 * cfg.dtype) — the `output_hidden_states=True` tuple of the reference minus its last entry. */
int    pcad_forward_all_hidden(pcad_handle h, const int32_t* ids, int B, int L,
                               void* all_hidden, void* hidden_out, float* logits_out,
                               void* workspace, size_t workspace_bytes, pcad_stream stream);

/* ---- measurement: per-kernel-class timing with HIP events on the caller's stream -------------------- */
enum pcad_kernel_class {
    PCAD_K_NORM = 0, PCAD_K_GEMM_IN, PCAD_K_CONV, PCAD_K_GEMM_X, PCAD_K_SCAN, PCAD_K_GEMM_OUT, PCAD_K_HEAD,
    PCAD_K_GEMM_OUT_RES,   /* "norm_fold": out_proj + residual add + row statistics in one launch */
    PCAD_K_RSTD,           /* "norm_fold": the per-layer reduction of the row statistics (the layer-0 embedding kernel counts as PCAD_K_NORM) */
    PCAD_NUM_KERNEL_CLASSES
};
typedef struct pcad_kernel_stat {
    char    name[32];
    int64_t launches;
    double  total_ms;           /* sum of (stop - start) event times of this class's launches */
} pcad_kernel_stat;
/* on = 0: off; on = 1: every launch of pcad_forward is bracketed by a pair of hipEvents recorded on its stream;
 * on = N > 1: every N-th launch of each kernel class is (a sample over the same region at 1/N of the event overhead). */
int pcad_profile_enable(pcad_handle h, int on);
/* Waits for the recorded events, writes one entry per kernel class (<= max_out), resets the counters.
 * Returns the number of entries written (>= 0) or a negative status. */
int pcad_profile_read(pcad_handle h, pcad_kernel_stat* out, int max_out);

/* ---- per-operator entry points (unit parity against the operators they replace) ------------------ */

/* rms_norm_fn(x, weight, None, residual=residual, eps, prenorm=True, residual_in_fp32)
 *   x [rows, D] dtype; residual_in [rows, D] res_dtype or NULL; weight fp32 [D];
 *   y [rows, D] dtype; residual_out [rows, D] res_dtype or NULL.  D % 8 == 0, D <= 2048. */
int pcad_add_rmsnorm(const void* x, const void* residual_in, const float* weight,
                     void* y, void* residual_out, int64_t rows, int D, float eps,
                     int dtype, int res_dtype, pcad_stream stream);

/* causal_conv1d_fn(x, weight, bias, activation="silu"), both directions in one pass, token-major:
 *   x [S, L, ldx>=E] dtype (channel-contiguous rows); w_fwd/w_rev fp32 [E, 4]; b_fwd/b_rev fp32 [E]
 *   y_fwd[t] = silu(b + sum_k w[k] x[t-3+k])   (causal),  y_rev[t] = silu(b + sum_k w[k] x[t+3-k]) (anti-causal)
 *   y_fwd / y_rev [S, L, E] dtype; either may be NULL. */
int pcad_causal_conv1d_silu(const void* x, int64_t ldx, const float* w_fwd, const float* b_fwd,
                            const float* w_rev, const float* b_rev, void* y_fwd, void* y_rev,
                            int S, int L, int E, int dtype, pcad_stream stream);

/* The form the engine runs for the head of mamba_inner_fn — causal_conv1d_fn(x, w, b, "silu") of BOTH directions fused with
 * BOTH `x_dbl = x_proj(conv_out)` GEMMs, x read once (one kernel instead of causal_conv1d_fwd x2 + cuBLAS x2):
 *   x        [rows8, E] dtype in the BLOCKED layout: byte offset of (row r, byte cb of the row) =
 *            (((r >> 3) * (E*esz/128) + (cb >> 7)) << 10) + ((r & 7) << 7) + (cb & 127); rows8 = S*L rounded up to 8;
 *            E*esz a multiple of 128 bytes; (S*L + 16) * E * esz < 2^32
 *   w_*, b_* fp32 [E, 4] / [E] conv taps and bias per direction (fwd: causal, rev: anti-causal on the same rows)
 *   Rp       dt_rank padded: 64 (dt_rank R <= 64: every PlantCaduceus size, PlantCAD2 Small / Medium) or 96 (R in 65..96:
 *            PlantCAD2 Large)
 *   Wx_*     [Rp + 32, E] dtype: x_proj.weight packed as rows [0, R) = dt rows, zero rows [R, Rp), rows [Rp, Rp + 16) = B,
 *            [Rp + 16, Rp + 32) = C
 *   scratch  device buffer of pcad_conv_xproj_scratch_bytes(E, dtype) bytes (packed taps; written by this call)
 *   xc_*     [rows8, E] dtype, blocked: silu(conv) per direction
 *   dtl_*    [S*L, Rp] dtype: x_dbl[:, :R] zero-padded to Rp;   bc_* fp32 [S*L, 32] = B_t | C_t rounded to dtype */
size_t pcad_conv_xproj_scratch_bytes(int E, int dtype);
int pcad_conv_xproj_bidir(const void* x, const float* w_fwd, const float* b_fwd, const float* w_rev, const float* b_rev,
                          const void* Wx_fwd, const void* Wx_rev, void* scratch,
                          void* xc_fwd, void* dtl_fwd, float* bc_fwd, void* xc_rev, void* dtl_rev, float* bc_rev,
                          int S, int L, int E, int Rp, int dtype, pcad_stream stream);

/* selective_scan_fn(u, delta, A, B, C, D, z, delta_bias, delta_softplus=True), token-major:
 *   u, delta [S, L, E] dtype; z [S, L, ldz>=E] dtype or NULL; bc fp32 [S*L, 32] = B_t (16) | C_t (16) per token;
 *   A fp32 [E, 16] (negative real, NOT pre-scaled); Dskip, delta_bias fp32 [E];
 *   reverse != 0 walks t = L-1..0;  accumulate: 0 y = out;  1 y = round(out) + y (bi-directional "add" strategy: each
 *   direction gated and rounded, as the reference's two Mamba calls are);  2 y = (y + out_ungated) * silu(z), z required
 *   (the sum of both directions gated once - what the engine runs).  y [S, L, E] dtype. */
int pcad_selective_scan(const void* u, const void* delta, const void* z, int64_t ldz, const float* bc,
                        const float* A, const float* Dskip, const float* delta_bias,
                        void* y, int S, int L, int E, int reverse, int accumulate,
                        int dtype, pcad_stream stream);

/* The form the engine runs — mamba_inner_fn's `delta = dt_proj.weight @ x_dbl[:, :R]` fused into the scan:
 *   delta[t, c] = round_dtype(sum_k dt_low[t, k] * Wdt[c, k]) computed on MFMA inside the kernel (never stored);
 *   dt_low [S*L, lddt>=Rp] dtype and Wdt [E, Rp] dtype, Rp % 64 == 0 with K zero-padded past dt_rank. */
int pcad_selective_scan_dtproj(const void* u, const void* dt_low, int64_t lddt, const void* Wdt, int Rp,
                               const void* z, int64_t ldz, const float* bc,
                               const float* A, const float* Dskip, const float* delta_bias,
                               void* y, int S, int L, int E, int reverse, int accumulate,
                               int dtype, pcad_stream stream);

/* F.linear(a, w): C[M,N] = A[M,K] . W[N,K]^T on MFMA.  lda/ldw/ldc in elements; K % (128/sizeof(elem)) == 0,
 * lda, ldw multiples of 16 bytes.  out_dtype: PCAD_F32 or `dtype`. */
int pcad_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                 int64_t M, int N, int K, int dtype, int out_dtype, pcad_stream stream);

/* F.linear(a, w) of fp32 tensors on the bf16 matrix pipes - the form the fp32 model's in_proj / out_proj take with
 * pcad_set_option("f32_gemm_split", 1): each operand is carried as two bf16 values (hi = bf16(v), lo = bf16(v - hi): 16 mantissa
 * bits) and C = A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T is ONE bf16 GEMM of 3 K / 64 K-tiles whose cursor wraps around operands
 * stored once as [hi | lo] (A: hi, lo, hi;  W: hi, hi, lo), fp32 accumulation in that order, fp32 result (3/16 of the fp32-MFMA
 * cost per flop; operand error 2^-17, dropped term 2^-16 relative).
 *   A [M, K] (lda), W [N, K] (ldw), C [M, N] (ldc) fp32, K % 64 == 0; scratch: device buffer of
 *   pcad_gemm_nt_split_scratch_bytes(M, N, K) bytes (the split operands; the engine keeps the weights' split copy in its arena). */
size_t pcad_gemm_nt_split_scratch_bytes(int64_t M, int N, int K);
int pcad_gemm_nt_split(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N, int K,
                       void* scratch, size_t scratch_bytes, pcad_stream stream);

/* out_proj of the "norm_fold" layer form as one operator (MFMA, 256 x 256 tiles only: M % 256 == 0, N % 256 == 0, K * elem a
 * multiple of 128 bytes, every tensor < 4 GiB):
 *   res [M, N] fp32 += A [M, K] . W [N, K]^T   (in place; the mixer output added to the fp32 residual stream).  res is in the
 *                       kernel's FRAGMENT layout - the order in which the GEMM's lanes hold a 256 x 256 tile, so that the
 *                       read-modify-write moves whole cache lines: element (row, col) lives at float offset
 *                         ((((tile * 4 + wave) * 8 + i) * 2 + jg) * 4 + k) * 256 + (16 lg + li) * 4 + r     with
 *                         tile = (row / 256) * (N / 256) + col / 256, wave = 2 ((row / 128) % 2) + (col / 128) % 2,
 *                         i = (row / 16) % 8, li = row % 16, jg = (col / 64) % 2, lg = (col / 16) % 4, k = (col / 4) % 4, r = col % 4
 *                       (csrc/common.hpp res_frag_off; plantcaduceus_amd.ops.to_res_fragment / from_res_fragment)
 *   C   [M, N] dtype  = round(res)              (plain rows: the next in_proj's operand)
 *   ssq [M, N / 128]  = per-row sums of squares of the updated res over each 128-column slab (deterministic partials whose
 *                       sum / N gives the next block's RMSNorm statistic)
 * Replaces: out_proj (F.linear) + the residual add of rms_norm_fn(..., prenorm=True, residual_in_fp32=True). */
int pcad_gemm_nt_residual(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, float* res, float* ssq,
                          int64_t M, int N, int K, int dtype, pcad_stream stream);

/* Rows of a [2B * L, E] activation tensor (plain rows) that a forward evaluated at positions p_0..p_{P-1} consumes after the
 * last mixer: strand b row p_q and strand B + b row L - 1 - p_q  ->  out [(strand * P + q), E].  positions: HOST int32 [P],
 * P <= PCAD_MAX_POSITIONS.  (The last-layer shortcut of pcad_forward; reference callers read one position:
 * src/zero_shot_score.py:117, src/train_XGBoost.py:105.) */
int pcad_gather_rows(const void* src, void* out, int B, int L, int E, const int32_t* positions, int P, int dtype,
                     pcad_stream stream);

/* The forward's last kernel as one operator: res + h -> norm_f -> RC re-assembly of hidden_states[-1] -> tied RCPS LM head, at
 * the shared positions (HOST int32 [P], P = 0: all L) or one position per window (pos_per_seq, DEVICE int32 [B]; then
 * positions must be NULL / P = 0).  h / res: [2B * L, D] (h: dtype, res: res_dtype); h_compact != 0: h holds only the evaluated
 * rows as pcad_gather_rows orders them; res_fragment_layout != 0: res (fp32) is in the fragment layout of pcad_gemm_nt_residual.  emb_f32: [vocab, D] fp32 (the dtype-rounded tied embedding / LM-head weight);
 * hidden_out [B, Q, 2D] dtype and logits_out [B, Q, vocab] fp32, either may be NULL.  ids (DEVICE [B, L]) and status (DEVICE
 * word) may be NULL: input validation as in pcad_set_status_buffer.
 * Replaces: norm_f (rms_norm_fn) x2, the flips / cats of RCPSWrapper's output and RCPSLMHead. */
int pcad_final_head(const void* h, const void* res, const float* norm_weight, const float* emb_f32, const int32_t* complement,
                    void* hidden_out, float* logits_out, int B, int L, int D, float eps, const int32_t* positions, int P,
                    const int32_t* pos_per_seq, int h_compact, const int32_t* ids, int32_t* status, int dtype, int res_dtype,
                    int res_fragment_layout, pcad_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PCAD_H */
