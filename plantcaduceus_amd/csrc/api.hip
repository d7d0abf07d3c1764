// C-ABI of libpcad.so (see include/pcad.h): handle, weight binding, the per-layer launch sequence of the
// PlantCaduceus forward on the 2B-strand batch, and the per-operator entry points.
//
// Forward = CaduceusForMaskedLM.forward restated per SURVEY.md Appendix A ("2B-strand form"): the RCPS
// network equals a plain bi-directional Mamba stack applied to [ids ; reverse_complement(ids)], so no flip
// or concatenation kernel exists here; the tied in_proj / out_proj run once per strand-layer.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pcad.h"
#include "build_hash.h"
#include "kernels.hpp"

using namespace pcad;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(PCAD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

struct DirWeights {
    float *conv_w, *conv_b;   // [E,4], [E]
    void* Wx;                 // [XP, E] dtype
    void* Wx_s;               // [XP, 2E] bf16, per 32-channel K-tile [hi | lo] ("f32_gemm_split": x_proj inside the fused conv kernel), else nullptr
    void* Wdt;                // [E, Rp] dtype
    void* Wdt_s;              // [E, 2 Rp] bf16 = [hi | lo] of Wdt ("f32_gemm_split": the fp32 model's dt_proj on the bf16 pipes), else nullptr
    float *dt_bias, *A2, *Dskip;
};

struct LayerWeights {
    float* convw;    // conv taps of both directions packed per K-tile for the fused conv+x_proj kernel
    float* norm_w;   // [D]
    void* W_in;      // [2E, D]
    void* W_in_f;    // [2E, D] = W_in . diag(norm_w), rounded once from the source precision: in_proj of the norm-folded form
    void* W_out;     // [D, E]
    void* W_out_p;   // [Dp, E]: W_out with zero rows up to Dp = round_up(D, 256) for the folded out_proj (== W_out when D % 256 == 0)
    void* W_in_s;    // [2E, 2D] bf16 = [hi | lo] of W_in: split-bf16 in_proj of the fp32 model ("f32_gemm_split"), else nullptr
    void* W_out_s;   // [D, 2E] bf16 = [hi | lo] of W_out
    DirWeights dir[2];
};

}  // namespace

struct pcad_engine {
    pcad_config cfg;
    int D, E, N, R, Rp, XP, V, nl;
    int esz;        // bytes per activation element
    int rdt;        // residual dtype
    int chunk;      // PCAD_CHUNK_SEQS override: sequences per pass through the layer stack (0: derive from chunk_rows)
    int64_t chunk_rows;   // token-rows (2 strands x L per window) per pass through the layer stack
    bool gate_once; // SiLU(z) applied once to y_fwd + y_rev (reverse scan) instead of once per direction
    bool convx;     // conv + x_proj of both directions in one kernel (needs xzsplit and Rp == 64 or 96: dt_rank <= 96); PCAD_NO_CONVX=1: off
    bool xzsplit;   // in_proj writes x and z as two blocked tensors (needs `blocked`); PCAD_PLAIN_XZ=1 turns it off (A/B knob)
    bool blocked;   // xc and y in the blocked layout (common.hpp::blocked_off); PCAD_PLAIN_LAYOUT=1 turns it off (A/B knob)
    bool segments = true;  // pcad_set_option("scan_segments", 0): never cut the scan of long strands into segments
    bool shortcut = true;  // pcad_set_option("last_layer_shortcut", 0): run the last layer in full even when only a few positions are evaluated
    int ref_order = 0;      // pcad_set_option("reference_order", 0 / 1 / 2): see include/pcad.h; 2 = each direction's tied out_proj on its own
    int norm_fold = -1;     // pcad_set_option("norm_fold", 0 / 1); -1 (default): on for the bf16 model, off for the fp32 model (forward_impl)
    int rep_class = -1, rep_count = 1;   // pcad_set_option("debug_repeat_class" / "debug_repeat"): measurement aid, see forward_impl
    bool poison = false;   // pcad_set_option("poison_workspace", 1): debug — fill the workspace with 0xFF (NaN patterns) before every forward
    bool bound = false;
    int64_t ws_limit = 0;       // pcad_set_option("workspace_limit_mb"): chunks are sized so that the workspace stays below it (0: no limit)
    bool f32_split = false;     // pcad_set_option("f32_gemm_split", 1): the fp32 model's in_proj / out_proj as split-bf16 GEMMs (split_wanted)
    bool split_packed = false;  // ... and their [hi | lo] weight copies exist in the arena (decided like fold_packed)
    bool fold_packed = false;   // the norm-folded form's extra weight copies (W_in_f, xz_tab0, padded W_out) exist in the arena: decided
                                // from the options in force when pcad_weight_arena_bytes / pcad_bind_weights run (fold_wanted)
    int32_t* status = nullptr;   // caller-owned device word for asynchronous input-validation flags (pcad_set_status_buffer)
    std::vector<LayerWeights> layers;
    void* xz_tab0 = nullptr;    // [V, 2E] dtype: layer 0's in_proj output per token id (norm-folded form), built at bind time
    void* emb = nullptr;        // [V, D] dtype
    float* emb_f32 = nullptr;   // [V, D] fp32 copy of the dtype-rounded table
    float* normf_w = nullptr;
    int32_t* comp = nullptr;    // [8] device
    // optional per-kernel-class timing with HIP events recorded on the caller's stream
    bool prof = false;
    int prof_stride = 1;                                  // bracket every prof_stride-th launch of a class
    int64_t prof_seen[PCAD_NUM_KERNEL_CLASSES] = {0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[PCAD_NUM_KERNEL_CLASSES];
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[PCAD_NUM_KERNEL_CLASSES] = {0};
    int64_t prof_n[PCAD_NUM_KERNEL_CLASSES] = {0};
};

namespace {

// ---- arena carving (identical walk for size query and binding) ------------------------------------
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    }
};

// whether the options ask for the norm-folded layer form on this model at all (per-forward shape conditions come on top)
bool fold_wanted(const pcad_engine* e) {
    const bool want = e->norm_fold == 1 || (e->norm_fold < 0 && e->cfg.dtype == PCAD_BF16);
    return want && e->rdt == F32 && e->xzsplit && e->blocked;
}

// "f32_gemm_split": fp32 model only, and never together with the norm-folded form (whose GEMM epilogues are fp32-in / fp32-out)
bool split_wanted(const pcad_engine* e) {
    return e->f32_split && e->cfg.dtype == PCAD_F32 && !fold_wanted(e) && e->xzsplit && e->blocked && e->D % 64 == 0 && e->E % 64 == 0;
}

// token-rows per pass through the layer stack: the kernels address their tensors with unsigned 32-bit byte offsets; the widest
// per-row tensor is E * esz bytes (x, z, xc, y; with the split-bf16 GEMMs out_proj's operand is 2 E bf16 columns = the same 4 E bytes)
int64_t chunk_row_limit(const pcad_engine* e) {
    const int64_t per_row = (int64_t)e->E * e->esz;
    return ((((int64_t)1 << 32) - ((int64_t)2 << 20)) / per_row) & ~(int64_t)7;
}

void carve_weights(pcad_engine* e, Carver& c) {
    const size_t D = e->D, E = e->E, N = e->N, V = e->V, esz = e->esz;
    // the folded form's copies (a second in_proj weight per layer, the layer-0 table, out_proj padded to 256 rows) are carved only
    // when the fold can engage: +37 % of the arena at l32 that an fp32 model or "norm_fold" 0 / "reference_order" never reads
    const bool pf = fold_wanted(e);
    const bool ps = split_wanted(e);
    e->emb = c.take(V * D * esz);
    e->emb_f32 = (float*)c.take(V * D * 4);
    e->normf_w = (float*)c.take(D * 4);
    e->comp = (int32_t*)c.take(8 * 4);
    e->xz_tab0 = pf ? c.take(V * 2 * E * esz) : nullptr;
    e->layers.resize(e->nl);
    for (auto& L : e->layers) {
        L.norm_w = (float*)c.take(D * 4);
        L.convw = (float*)c.take(convx_packed_bytes((int)E, e->cfg.dtype));
        L.W_in = c.take(2 * E * D * esz);
        L.W_in_f = pf ? c.take(2 * E * D * esz) : nullptr;
        L.W_out = c.take(D * E * esz);
        L.W_out_p = pf && (size_t)fold_padded_width((int)D) != D ? c.take((size_t)fold_padded_width((int)D) * E * esz) : L.W_out;
        L.W_in_s = ps ? c.take(2 * E * 2 * D * 2) : nullptr;
        L.W_out_s = ps ? c.take(D * 2 * E * 2) : nullptr;
        for (int d = 0; d < 2; ++d) {
            DirWeights& w = L.dir[d];
            w.conv_w = (float*)c.take(E * 4 * 4);
            w.conv_b = (float*)c.take(E * 4);
            w.Wx = c.take((size_t)e->XP * E * esz);
            w.Wx_s = ps && e->convx ? c.take((size_t)e->XP * 2 * E * 2) : nullptr;
            w.Wdt = c.take(E * (size_t)e->Rp * esz);
            w.Wdt_s = ps && e->convx ? c.take(E * (size_t)e->Rp * 2 * 2) : nullptr;
            w.dt_bias = (float*)c.take(E * 4);
            w.A2 = (float*)c.take(E * N * 4);
            w.Dskip = (float*)c.take(E * 4);
        }
    }
}

struct Workspace {
    void *res, *u, *h, *xz, *zb, *xc[2], *dtl[2], *y;
    void* ys;        // "f32_gemm_split": out_proj's operand, bf16 [rows8, 2E] blocked = [hi | lo] of y; else nullptr
    float* bc[2];
    float *rstd, *ssq;   // norm-folded form: rstd [rows]; partial sums of squares [rows, D / 128]
    float* seg;      // segmented-scan scratch (long sequences with few strands), or nullptr
    float* pair;     // state hand-over of the pair walks (kernels.hpp scan_pair_wanted), or nullptr
    float* cxp;      // K-split scratch of the fused conv + x_proj kernel (small launches), or nullptr
    size_t bytes;
};

// Bpol: windows of the whole pcad_forward call - the small-launch forms (segmented scan, conv + x_proj K-split) are chosen for the
// call, not per chunk, so that results never depend on the chunking
Workspace carve_workspace(const pcad_engine* e, void* base, int Bc, int L, int Bpol) {
    Carver c(base);
    const size_t rows = (size_t)2 * Bc * L;
    const size_t D = e->D, E = e->E, esz = e->esz;
    Workspace w;
    const size_t Dp = fold_padded_width((int)D);       // the folded form keeps res / u Dp = round_up(D, 256) columns wide
    w.res = c.take(rows * Dp * (e->rdt == F32 ? 4 : esz));
    const bool sp = split_wanted(e);
    w.u = c.take(rows * Dp * esz);                     // split: bf16 [rows, 2D] = [hi | lo] (the same 4 D bytes per row)
    w.h = c.take(rows * D * esz);
    const size_t rows8z = (rows + 7) / 8 * 8;
    // in_proj output: plain xz [rows, 2E]; or (xzsplit) x [rows8, E] in `xz` and z [rows8, E] in `zb`, both blocked
    w.xz = c.take((e->xzsplit ? rows8z : rows * 2) * E * esz);
    w.zb = e->xzsplit ? c.take(rows8z * E * esz) : nullptr;
    const size_t rows8 = (rows + 7) / 8 * 8;   // xc and y use the blocked layout: whole 8-row blocks
    w.xc[0] = c.take(rows8 * E * esz);
    w.xc[1] = c.take(rows8 * E * esz);
    // dt_low (x_proj columns [0, Rp), zero padded past R); split: bf16 [rows, 2 Rp] = [hi | lo]
    const size_t dtl_bytes = sp && e->convx ? rows * e->Rp * 2 * 2 : rows * e->Rp * esz;
    w.dtl[0] = c.take(dtl_bytes);
    w.dtl[1] = c.take(dtl_bytes);
    w.bc[0] = (float*)c.take(rows * 2 * e->N * 4);   // B_t | C_t rows, fp32 (values rounded to the model dtype)
    w.bc[1] = (float*)c.take(rows * 2 * e->N * 4);
    w.y = c.take(rows8 * E * esz);
    w.ys = sp ? c.take(rows8 * 2 * E * 2) : nullptr;
    w.rstd = (float*)c.take(rows * 4);
    w.ssq = (float*)c.take(rows * (Dp / 128) * 4);
    const size_t segb = e->segments ? scan_segment_bytes(2 * Bc, L, (int)E, 2 * Bpol) : 0;
    w.seg = segb ? (float*)c.take(segb) : nullptr;
    const size_t pairb = e->segments && e->convx && scan_pair_wanted(2 * Bpol, L, (int)E) ? scan_pair_bytes(2 * Bc, (int)E) : 0;
    w.pair = pairb ? (float*)c.take(pairb) : nullptr;
    // small launches: the conv + x_proj kernel splits its channel walk over several blocks per row tile ("scan_segments" 0 turns this
    // off together with the segmented scan: both trade a different fp32 summation order for parallelism on an otherwise empty chip)
    const size_t cxb = e->segments && e->convx ? convx_split_bytes(2 * Bc, L, (int)E, e->cfg.dtype, e->Rp, 2 * Bpol) : 0;
    w.cxp = cxb ? (float*)c.take(cxb) : nullptr;
    w.bytes = c.off;
    return w;
}

const char* const kClassNames[PCAD_NUM_KERNEL_CLASSES] = {
    "add_rmsnorm", "gemm_in_proj", "conv1d_bidir", "gemm_x_proj", "selective_scan", "gemm_out_proj", "final_head",
    "gemm_out_proj_res", "rstd_reduce"};

constexpr size_t kProfCap = 1 << 16;

hipEvent_t prof_event(pcad_engine* e) {
    if (!e->prof_pool.empty()) {
        hipEvent_t ev = e->prof_pool.back();
        e->prof_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct ProfScope {   // records start/stop events around one launch when profiling is on
    pcad_engine* e; int cls; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(pcad_engine* e_, int cls_, hipStream_t s_) : e(e_), cls(cls_), s(s_) {
        // at most kProfCap un-read event pairs per class: a caller that never calls pcad_profile_read cannot grow the lists
        if (e->prof && (e->prof_seen[cls]++ % e->prof_stride) == 0 && e->prof_ev[cls].size() < kProfCap) {
            a = prof_event(e); b = prof_event(e);
            if (a) (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (a && b) { (void)hipEventRecord(b, s); e->prof_ev[cls].push_back({a, b}); }
    }
};

const pcad_tensor* find(const std::map<std::string, const pcad_tensor*>& m, const std::string& k) {
    auto it = m.find(k);
    return it == m.end() ? nullptr : it->second;
}

int64_t numel(const pcad_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

}  // namespace

extern "C" {

int pcad_version(void) { return PCAD_VERSION; }
const char* pcad_build_hash(void) { return PCAD_BUILD_HASH; }
const char* pcad_last_error(void) { return g_err; }

int pcad_create(const pcad_config* cfg, pcad_handle* out) {
    if (!cfg || !out) return fail(PCAD_ERR_INVALID, "pcad_create: null argument");
    if (cfg->d_state != 16) return fail(PCAD_ERR_INVALID, "d_state=%d unsupported (16 only)", cfg->d_state);
    if (cfg->d_conv != 4) return fail(PCAD_ERR_INVALID, "d_conv=%d unsupported (4 only)", cfg->d_conv);
    if (cfg->vocab != 8) return fail(PCAD_ERR_INVALID, "padded vocab=%d unsupported (8 only)", cfg->vocab);
    if (cfg->d_model <= 0 || cfg->d_model % 64 || cfg->d_model > 2048)
        return fail(PCAD_ERR_INVALID, "d_model=%d must be a multiple of 64, <= 2048", cfg->d_model);
    if (cfg->expand < 1 || cfg->n_layer < 1 || cfg->dt_rank < 1 || cfg->dt_rank > 256)
        return fail(PCAD_ERR_INVALID, "bad expand/n_layer/dt_rank");
    if (cfg->dtype != PCAD_F32 && cfg->dtype != PCAD_BF16) return fail(PCAD_ERR_INVALID, "bad dtype %d", cfg->dtype);
    for (int i = 0; i < 8; ++i)
        if (cfg->complement[i] < 0 || cfg->complement[i] > 7) return fail(PCAD_ERR_INVALID, "bad complement map");
    pcad_engine* e = new pcad_engine();
    e->cfg = *cfg;
    e->D = cfg->d_model; e->E = cfg->expand * cfg->d_model; e->N = 16; e->R = cfg->dt_rank; e->V = 8;
    e->nl = cfg->n_layer;
    e->Rp = padded_dt_rank(e->R);        // K of dt_proj: 64 up to dt_rank 64, else the next multiple of 32 (PlantCAD2 Large: 96)
    e->XP = e->Rp + 2 * e->N;            // x_proj rows: [dt (R) | 0-pad | B (16) | C (16)]
    e->esz = cfg->dtype == PCAD_BF16 ? 2 : 4;
    e->rdt = (cfg->residual_in_fp32 || cfg->dtype == PCAD_F32) ? F32 : BF16;
    const char* ck = dev_env("PCAD_CHUNK_SEQS");       // PCAD_DEV=1 only; the ABI's knob is pcad_set_option("chunk_seqs")
    // Token-rows per chunk: bounded by the kernels' unsigned 32-bit in-tensor byte offsets (rows * E * esz < 2^32 in the scan,
    // the fused conv+x_proj kernel and the 4-wave GEMM): (2^32 - 2 MiB) / (E * esz) rows = 1023 windows of 512 bp at l32 bf16
    // (30 GB of workspace), 85 windows of 8 192 bp at the l28 width.  Fewer, larger launches: 1024 windows as 4 chunks instead of
    // 16 measured +6 % (each of the 193 launches per chunk pays a fill/drain of the chip), and for long windows the chunk is what
    // sets the scan's wave count (strands x E/64): 80 windows of 8 192 bp as one chunk instead of two, +20 %.  Floor: one launch
    // of the scan should fill the chip's 4096 wave slots (2 strands x E/64 waves per window).
    e->chunk = ck ? atoi(ck) : 0;
    if (e->chunk < 0) e->chunk = 0;
    e->chunk_rows = 0;        // derived per call from the options in force: chunk_row_limit()
    // developer A/B switches (honoured only with PCAD_DEV=1): plain layouts / unfused conv
    e->blocked = dev_env("PCAD_PLAIN_LAYOUT") == nullptr && (e->E * e->esz) % 128 == 0;
    e->xzsplit = e->blocked && dev_env("PCAD_PLAIN_XZ") == nullptr && e->E % 16 == 0;
    e->convx = e->xzsplit && (e->Rp == 64 || e->Rp == 96) && dev_env("PCAD_NO_CONVX") == nullptr;
    e->gate_once = true;                  // pcad_set_option("gate_each", 1) restores the per-direction gate
    *out = e;
    return PCAD_OK;
}

void pcad_destroy(pcad_handle h) {
    if (!h) return;
    for (int c = 0; c < PCAD_NUM_KERNEL_CLASSES; ++c)
        for (auto& pr : h->prof_ev[c]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto ev : h->prof_pool) (void)hipEventDestroy(ev);
    delete h;
}

int pcad_set_option(pcad_handle h, const char* key, int64_t value) {
    if (!h || !key) return fail(PCAD_ERR_INVALID, "pcad_set_option: null argument");
    const std::string k(key);
    if (k == "chunk_seqs") {
        if (value < 0 || value > (1 << 20)) return fail(PCAD_ERR_INVALID, "chunk_seqs=%lld out of range", (long long)value);
        h->chunk = (int)value;
    } else if (k == "gate_each") {
        h->gate_once = value == 0;
    } else if (k == "norm_fold") {
        h->norm_fold = value < 0 ? -1 : (value != 0 ? 1 : 0);
    } else if (k == "workspace_limit_mb") {
        if (value < 0 || value > ((int64_t)1 << 30)) return fail(PCAD_ERR_INVALID, "workspace_limit_mb=%lld out of range", (long long)value);
        h->ws_limit = value << 20;
    } else if (k == "f32_gemm_split") {
        h->f32_split = value != 0;
    } else if (k == "reference_order") {
        // one switch for "every rounding point where the reference has it" (BiMambaWrapper + rms_norm_fn + mamba_inner_fn):
        //   1  = gate_each 1 + norm_fold 0 (which also means no layer-0 in_proj table): only the tied out_proj fold remains
        //   2  = 1 + each direction's out_proj computed and stored in the model dtype, then summed and rounded ("add" strategy)
        //   0  = the engine's defaults
        if (value < 0 || value > 2) return fail(PCAD_ERR_INVALID, "reference_order=%lld out of range (0, 1, 2)", (long long)value);
        h->ref_order = (int)value;
        h->gate_once = value == 0;
        h->norm_fold = value == 0 ? -1 : 0;
    } else if (k == "debug_repeat_class") {
        if (value < -1 || value >= PCAD_NUM_KERNEL_CLASSES) return fail(PCAD_ERR_INVALID, "debug_repeat_class=%lld out of range", (long long)value);
        h->rep_class = (int)value;
    } else if (k == "debug_repeat") {
        if (value < 1 || value > 10000) return fail(PCAD_ERR_INVALID, "debug_repeat=%lld out of range", (long long)value);
        h->rep_count = (int)value;
    } else if (k == "poison_workspace") {
        h->poison = value != 0;
    } else if (k == "scan_segments") {
        h->segments = value != 0;
    } else if (k == "last_layer_shortcut") {
        h->shortcut = value != 0;
    } else {
        return fail(PCAD_ERR_INVALID, "pcad_set_option: unknown option '%s'", key);
    }
    return PCAD_OK;
}

int pcad_set_status_buffer(pcad_handle h, int32_t* status) {
    if (!h) return fail(PCAD_ERR_INVALID, "pcad_set_status_buffer: null handle");
    if (((uintptr_t)status) % 4) return fail(PCAD_ERR_INVALID, "pcad_set_status_buffer: misaligned pointer");
    h->status = status;
    return PCAD_OK;
}

size_t pcad_weight_arena_bytes(pcad_handle h) {
    if (!h) return 0;
    pcad_engine tmp = *h;
    Carver c(nullptr);
    carve_weights(&tmp, c);
    return c.off;
}

int pcad_bind_weights(pcad_handle h, const pcad_tensor* tensors, int n, void* arena, size_t arena_bytes,
                      pcad_stream stream) {
    if (!h || !tensors || !arena) return fail(PCAD_ERR_INVALID, "pcad_bind_weights: null argument");
    if (((uintptr_t)arena) % 256) return fail(PCAD_ERR_WORKSPACE, "weight arena must be 256-byte aligned");
    if (arena_bytes < pcad_weight_arena_bytes(h))
        return fail(PCAD_ERR_WORKSPACE, "weight arena too small: %zu < %zu", arena_bytes, pcad_weight_arena_bytes(h));
    hipStream_t s = (hipStream_t)stream;
    pcad_engine* e = h;
    Carver c(arena);
    carve_weights(e, c);
    std::map<std::string, const pcad_tensor*> m;
    for (int i = 0; i < n; ++i) {
        if (!tensors[i].name || !tensors[i].data) return fail(PCAD_ERR_INVALID, "tensor %d has null name/data", i);
        if (tensors[i].dtype != PCAD_F32 && tensors[i].dtype != PCAD_BF16)
            return fail(PCAD_ERR_INVALID, "tensor %s: bad dtype", tensors[i].name);
        m[tensors[i].name] = &tensors[i];
    }
    const int D = e->D, E = e->E, N = e->N, R = e->R, Rp = e->Rp, V = e->V, dt = e->cfg.dtype;
    const std::string pre = "caduceus.backbone.";

    auto need = [&](const std::string& k, int64_t expect) -> const pcad_tensor* {
        const pcad_tensor* t = find(m, k);
        if (!t) { fail(PCAD_ERR_MISSING, "missing tensor %s", k.c_str()); return nullptr; }
        if (numel(t) != expect) {
            fail(PCAD_ERR_INVALID, "tensor %s has %lld elements, expected %lld", k.c_str(), (long long)numel(t),
                 (long long)expect);
            return nullptr;
        }
        return t;
    };
#define NEED(var, key, cnt)                      \
    const pcad_tensor* var = need((key), (cnt)); \
    if (!var) return g_err[0] == 'm' ? PCAD_ERR_MISSING : PCAD_ERR_INVALID

    NEED(t_emb, pre + "embeddings.word_embeddings.embedding.weight", (int64_t)V * D);
    HIP_TRY(launch_pack2d(t_emb->data, t_emb->dtype, D, e->emb, dt, D, V, D, V, D, s));
    // fp32 copy of the dtype-rounded table (what F.linear sees through the tied lm_head weight)
    HIP_TRY(launch_pack2d(e->emb, dt, D, e->emb_f32, F32, D, V, D, V, D, s));
    NEED(t_nf, pre + "norm_f.weight", (int64_t)D);
    HIP_TRY(launch_pack2d(t_nf->data, t_nf->dtype, D, e->normf_w, F32, D, 1, D, 1, D, s));
    HIP_TRY(hipMemcpyAsync(e->comp, e->cfg.complement, 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // the source array lives in the handle, so the async copy's host buffer stays valid

    for (int i = 0; i < e->nl; ++i) {
        LayerWeights& L = e->layers[i];
        const std::string lp = pre + "layers." + std::to_string(i) + ".";
        NEED(t_norm, lp + "norm.weight", (int64_t)D);
        HIP_TRY(launch_pack2d(t_norm->data, t_norm->dtype, D, L.norm_w, F32, D, 1, D, 1, D, s));
        const std::string mf = lp + "mixer.submodule.mamba_fwd.";
        NEED(t_in, mf + "in_proj.weight", (int64_t)2 * E * D);
        HIP_TRY(launch_pack2d(t_in->data, t_in->dtype, D, L.W_in, dt, D, 2 * E, D, 2 * E, D, s));
        if (L.W_in_f) HIP_TRY(launch_pack_scale_cols(t_in->data, t_in->dtype, D, L.norm_w, L.W_in_f, dt, D, 2 * E, D, s));
        NEED(t_out, mf + "out_proj.weight", (int64_t)D * E);
        HIP_TRY(launch_pack2d(t_out->data, t_out->dtype, E, L.W_out, dt, E, D, E, D, E, s));
        if (L.W_out_p != L.W_out) HIP_TRY(launch_pack2d(t_out->data, t_out->dtype, E, L.W_out_p, dt, E, D, E, fold_padded_width(D), E, s));
        if (L.W_in_s) HIP_TRY(launch_pack_split_w(t_in->data, t_in->dtype, D, L.W_in_s, 2 * E, D, s));
        if (L.W_out_s) HIP_TRY(launch_pack_split_w(t_out->data, t_out->dtype, E, L.W_out_s, D, E, s));
        for (int d = 0; d < 2; ++d) {
            DirWeights& w = L.dir[d];
            const std::string mp = lp + "mixer.submodule.mamba_" + (d == 0 ? "fwd." : "rev.");
            NEED(t_cw, mp + "conv1d.weight", (int64_t)E * 4);
            HIP_TRY(launch_pack2d(t_cw->data, t_cw->dtype, 4, w.conv_w, F32, 4, E, 4, E, 4, s));
            NEED(t_cb, mp + "conv1d.bias", (int64_t)E);
            HIP_TRY(launch_pack2d(t_cb->data, t_cb->dtype, E, w.conv_b, F32, E, 1, E, 1, E, s));
            NEED(t_x, mp + "x_proj.weight", (int64_t)(R + 2 * N) * E);
            // rows [0,R) -> [0,R); zero rows [R,Rp); rows [R, R+2N) -> [Rp, Rp+2N)
            const size_t esz_src = t_x->dtype == PCAD_BF16 ? 2 : 4;
            HIP_TRY(launch_pack2d(t_x->data, t_x->dtype, E, w.Wx, dt, E, R, E, Rp, E, s));
            HIP_TRY(launch_pack2d((const char*)t_x->data + (size_t)R * E * esz_src, t_x->dtype, E,
                                  (char*)w.Wx + (size_t)Rp * E * e->esz, dt, E, 2 * N, E, 2 * N, E, s));
            if (w.Wx_s) HIP_TRY(launch_pack_convx_wsplit((const float*)w.Wx, E, w.Wx_s, e->XP, E, s));     // from the padded fp32 copy
            NEED(t_dw, mp + "dt_proj.weight", (int64_t)E * R);
            HIP_TRY(launch_pack2d(t_dw->data, t_dw->dtype, R, w.Wdt, dt, Rp, E, R, E, Rp, s));
            if (w.Wdt_s) HIP_TRY(launch_pack_split_w(w.Wdt, dt, Rp, w.Wdt_s, E, Rp, s));         // from the zero-padded fp32 copy
            NEED(t_db, mp + "dt_proj.bias", (int64_t)E);
            HIP_TRY(launch_pack2d(t_db->data, t_db->dtype, E, w.dt_bias, F32, E, 1, E, 1, E, s));
            NEED(t_A, mp + "A_log", (int64_t)E * N);
            HIP_TRY(launch_pack_A(t_A->data, t_A->dtype, w.A2, (int64_t)E * N, 1.4426950408889634f, s));
            NEED(t_D, mp + "D", (int64_t)E);
            HIP_TRY(launch_pack2d(t_D->data, t_D->dtype, E, w.Dskip, F32, E, 1, E, 1, E, s));
        }
        if ((E * e->esz) % 128 == 0)
            HIP_TRY(launch_pack_convw(L.dir[0].conv_w, L.dir[0].conv_b, L.dir[1].conv_w, L.dir[1].conv_b, L.convw, E, dt, s));
    }
#undef NEED
    // layer 0's in_proj (norm-folded form) as a table over the V token ids: emb and layer 0's folded in_proj weight are packed above
    if (e->xz_tab0) HIP_TRY(launch_embed_inproj_table(e->emb, e->layers[0].W_in_f, e->xz_tab0, V, D, 2 * E, e->cfg.eps, dt, s));
    e->fold_packed = fold_wanted(e);
    e->split_packed = split_wanted(e);
    e->bound = true;
    return PCAD_OK;
}

// windows per chunk for a batch of B windows of L positions: the fewest chunks within the row limit, evenly sized (no small
// tail chunk)
static int chunk_for(const pcad_engine* e, int B, int L) {
    int64_t cap = chunk_row_limit(e) / (2 * (int64_t)L);
    if (e->chunk > 0 && e->chunk < cap) cap = e->chunk;
    if (cap < 1) cap = 1;
    if (cap > B) cap = B;
    if (e->ws_limit > 0 && carve_workspace(e, nullptr, (int)cap, L, B).bytes > (size_t)e->ws_limit) {
        int64_t lo = 1, hi = cap;                     // largest chunk whose workspace fits (the size is monotone in the chunk)
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) / 2;
            if (carve_workspace(e, nullptr, (int)mid, L, B).bytes <= (size_t)e->ws_limit) lo = mid; else hi = mid - 1;
        }
        cap = lo;                                     // one window always runs, whatever the limit
    }
    int64_t n = (B + cap - 1) / cap;
    // Whole rounds of the persistent GEMMs: a chunk whose token-rows are a multiple of 16 384 (64 m-tiles of 256 rows) gives every CU
    // the same number of output tiles and every XCD whole groups of the tile walk.  When the fewest-chunks split misses that (the fp32
    // model's 1 024-window batch: 3 chunks of 342 windows = 1 368 m-tiles) and a split into up to twice as many chunks hits it
    // (4 x 256 windows), take that one: +1.6 % on the fp32 + f32_gemm_split model (profiles/r06_f32_split_ab.txt r06v).  Never when
    // the caller set "chunk_seqs" / "workspace_limit_mb"; results do not depend on the chunking.
    if (e->chunk == 0 && e->ws_limit == 0) {
        auto whole = [&](int64_t m) { const int64_t c = (B + m - 1) / m; return (2 * c * (int64_t)L) % 16384 == 0; };
        if (!whole(n))
            for (int64_t m = n + 1; m <= 2 * n && m <= B; ++m)
                if (whole(m)) { n = m; break; }
    }
    return (int)((B + n - 1) / n);
}

size_t pcad_workspace_bytes(pcad_handle h, int batch, int seqlen) {
    if (!h || batch <= 0 || seqlen <= 0) return 0;
    const int Bc = chunk_for(h, batch, seqlen);
    const int nchunks = (batch + Bc - 1) / Bc;
    (void)nchunks;                                                   // chunks run one after the other in ONE workspace slab
    return carve_workspace(h, nullptr, Bc, seqlen, batch).bytes;
}

static int forward_impl(pcad_handle h, const int32_t* ids, int B, int L, const int32_t* positions, int P,
                        const int32_t* pos_per_seq, void* all_hidden, void* hidden_out, float* logits_out, void* workspace, size_t ws_bytes,
                        pcad_stream stream) {
    if (!h) return fail(PCAD_ERR_INVALID, "pcad_forward: null handle");
    pcad_engine* e = h;
    if (!e->bound) return fail(PCAD_ERR_UNBOUND, "pcad_forward: weights not bound");
    if (B < 0 || L <= 0) return fail(PCAD_ERR_INVALID, "pcad_forward: bad B=%d L=%d", B, L);
    if (B == 0) return PCAD_OK;
    if (!ids || !workspace) return fail(PCAD_ERR_INVALID, "pcad_forward: null ids/workspace");
    if (P < 0 || P > PCAD_MAX_POSITIONS || (P > 0 && !positions))
        return fail(PCAD_ERR_INVALID, "pcad_forward: bad positions (P=%d)", P);
    Positions pos;
    pos.n = P;
    for (int i = 0; i < 16; ++i) pos.p[i] = 0;
    for (int i = 0; i < P; ++i) {
        if (positions[i] < 0 || positions[i] >= L)
            return fail(PCAD_ERR_INVALID, "pcad_forward: position %d out of range [0,%d)", positions[i], L);
        pos.p[i] = positions[i];
    }
    if (((uintptr_t)workspace) % 256) return fail(PCAD_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    const size_t need = pcad_workspace_bytes(h, B, L);
    if (ws_bytes < need) return fail(PCAD_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);

    hipStream_t cs = (hipStream_t)stream;
    const int D = e->D, E = e->E, N = e->N, Rp = e->Rp, XP = e->XP, dt = e->cfg.dtype, rdt = e->rdt;
    const int Dp = fold_padded_width(D);        // width of res / u while a chunk runs in the norm-folded form
    const size_t esz = e->esz;
    const float eps = e->cfg.eps;
    const int Q = pos_per_seq ? 1 : (P ? P : L);
    const int chunk = chunk_for(e, B, L);
    const int nchunks = (B + chunk - 1) / chunk;

    // One chunk = up to `chunk` windows (2x strands) walking the whole layer stack, chunks one after the other, everything on
    // the caller's stream.  (Multi-stream schedules were built and measured twice and removed: chunks alternating between two
    // streams gain nothing because the big kernels each fill the CUs, +1 %; the add+norm kernels on a side stream beside the other
    // chunk's GEMM are zero-sum, in_proj stretches by the norm's duration, -4 %: DESIGN.md §8.)
    struct Lane { Workspace w; int b0, Bc; bool fold; };
    // debug aid (race / uninitialised-read screen): every byte of the workspace starts as 0xFF, so a kernel that consumes a
    // value no kernel of THIS forward produced turns the outputs into NaN instead of silently reusing the previous call's data
    if (e->poison) HIP_TRY(hipMemsetAsync(workspace, 0xFF, need, cs));

    // Norm-folded layer form (pcad_set_option("norm_fold", 0) restores the reference's order).  The reference's block is
    //     res = h + res (fp32);  u = round(res * rstd(res) * w_norm);  xz = round(u . W_in^T);  ...;  h = round(y . W_out^T)
    // (rms_norm_fn(..., prenorm=True, residual_in_fp32=True)).  Folded: out_proj's epilogue does
    // res += y . W_out^T in fp32 (the accumulators start as the residual values), writes round(res) and per-row partial sums of
    // squares; in_proj runs on round(res) with W_in . diag(w_norm) (folded at bind time) and multiplies by rstd[row] before it
    // rounds.  The add + norm launch and its read of h / write of u disappear; what moves is rounding: h is not rounded before it
    // is added, and the operand of in_proj is round(res) instead of round(res * rstd * w).  While a chunk runs in this form its
    // fp32 residual tensor is kept in the GEMM's fragment layout (common.hpp res_frag_off) so that the epilogue's
    // read-modify-write moves whole lines; only the embedding kernel, the folded out_proj and the head kernel touch it.
    // Default: on for the bf16 model only.  Accumulating the K products onto the (large) residual value instead of onto zero costs
    // the fp32 model precision it can see - hidden states 2.2e-5 of max after 32 layers against 1.3e-6 with the separate add
    // (profiles/r04h_gpu_tests.log; still inside north_star's 1e-4) - while under bf16 storage the difference is far below the
    // rounding noise (probabilities 8.1e-3 vs 8.6e-3 from the reference-order emulation).
    // Used when every GEMM of the chunk
    // runs on the 4-wave kernel (whole 256 x 256 tiles: token-rows % 256 == 0; a d_model that is not a multiple of 256 - l20's 384 -
    // is padded to the next one with zero out_proj weight rows and zero residual columns) and the residual stream is fp32; never
    // for pcad_forward_all_hidden
    // (hidden_states[i] are the mixer outputs h, which the folded form never materialises).
    // Decided ONCE per forward - every chunk folds or none does - so that a result never depends on how the batch was cut
    // into chunks (an uneven split of odd-length windows could otherwise give one chunk whole 256-row tiles and another not).
    // (also when the fold is only the DEFAULT of this model - a bf16 engine bound under "norm_fold" 0 / "reference_order" >= 1 and switched
    // back afterwards: running unfolded would silently cost 4.5 % and differ from a freshly bound engine; like "f32_gemm_split" it is refused)
    if (fold_wanted(e) && !e->fold_packed)
        return fail(PCAD_ERR_INVALID, "pcad_forward: the norm-folded layer form (\"norm_fold\" %s) was enabled after pcad_bind_weights, under options that "
                                      "did not ask for it; its weight copies are packed at bind time - set \"norm_fold\" / \"reference_order\" before "
                                      "pcad_weight_arena_bytes / pcad_bind_weights (turning the form OFF afterwards is always possible)",
                    e->norm_fold == 1 ? "1" : "default");
    if (split_wanted(e) && !e->split_packed)
        return fail(PCAD_ERR_INVALID, "pcad_forward: \"f32_gemm_split\" 1 was set after pcad_bind_weights; the split weight copies are packed at "
                                      "bind time - set the option before pcad_weight_arena_bytes / pcad_bind_weights");
    // Split-bf16 GEMMs of the fp32 model ("f32_gemm_split"; pack.hip): in_proj and out_proj - 3/4 of the fp32 model's time on the
    // fp32 MFMA instructions - run as bf16 GEMMs of 3 K / 64 K-tiles on [hi | lo] x [hi | lo] operands (wrap-around K cursor: hi.hi,
    // lo.hi, hi.lo; gemm.hip) with an fp32 result: operand error 2^-17, measured 4e-7 of the logits' range after 32 layers (fp32 MFMA:
    // 1e-6 from summation order alone).
    const bool sp = split_wanted(e) && e->split_packed;
    bool fold_all = fold_wanted(e) && e->fold_packed && !all_hidden;
    for (int ck = 0; ck < nchunks && fold_all; ++ck) {
        const int Bc = (B - ck * chunk) < chunk ? (B - ck * chunk) : chunk;
        fold_all = gemm_fold_shapes_ok((int64_t)2 * Bc * L, D, E, dt);
    }
    auto fold_for = [&](const Lane&) -> bool { return fold_all; };
    // Measurement aid (tools/power_probe.py): every launch of ONE kernel class is issued `debug_repeat` times back to back, so a
    // forward becomes seconds of that kernel - the engine's own instantiation, layouts and launch sizes - while the host samples
    // board power and clocks.  Only launches that are idempotent are repeated (in_proj, conv + x_proj, the forward-direction scan,
    // the reference-order out_proj); outputs are unchanged.
    auto reps = [&](int cls) -> int { return e->rep_class == cls ? e->rep_count : 1; };
    static const bool tab0 = dev_env("PCAD_NO_TAB0") == nullptr;     // layer 0's in_proj as a table look-up (phase_P); PCAD_DEV=1 A/B switch
    auto phase_N = [&](Lane& c, int li) -> int {        // residual add + norm (layer 0: RCPS embedding + norm)
        hipStream_t s = cs;
        const LayerWeights& W = e->layers[li];
        const int S = 2 * c.Bc;
        const int64_t rows = (int64_t)S * L;
        const int32_t* ids_c = ids + (int64_t)c.b0 * L;
        if (c.fold) {
            if (li == 0) {      // res = Emb[token] (fp32, fragment layout) [+ u = the same rows in the model dtype and rstd when layer 0's in_proj runs as a GEMM]
                ProfScope ps(e, PCAD_K_NORM, s);
                HIP_TRY(launch_embed_rmsnorm(ids_c, e->emb, e->comp, W.norm_w, tab0 ? nullptr : c.w.u, c.w.res, c.Bc, L, D, eps, dt, rdt, s, c.w.rstd, Dp));
            }
            return PCAD_OK;     // later layers: the previous out_proj's epilogue already produced res, round(res) and rstd
        }
        if (li == 0) {
            if (all_hidden) {   // hidden_states[0] = RCPSEmbedding output
                HIP_TRY(launch_embed_only(ids_c, e->emb, e->comp, c.w.h, c.Bc, L, D, dt, s));
                HIP_TRY(launch_assemble_hidden(c.w.h, (char*)all_hidden + ((size_t)c.b0 * L * 2 * D) * esz, c.Bc, L, D, dt, s));
            }
            ProfScope ps(e, PCAD_K_NORM, s);
            HIP_TRY(launch_embed_rmsnorm(ids_c, e->emb, e->comp, W.norm_w, c.w.u, c.w.res, c.Bc, L, D, eps, dt, rdt, s, nullptr, 0, sp));
        } else {
            ProfScope ps(e, PCAD_K_NORM, s);
            HIP_TRY(launch_add_rmsnorm(c.w.h, c.w.res, W.norm_w, c.w.u, c.w.res, rows, D, eps, dt, rdt, s, sp));
        }
        return PCAD_OK;
    };
    auto phase_P = [&](Lane& c, int li) -> int {        // in_proj, conv + x_proj (both directions)
        hipStream_t s = cs;
        const LayerWeights& W = e->layers[li];
        const int S = 2 * c.Bc;
        const int64_t rows = (int64_t)S * L;
        // in_proj (tied between directions: once per strand)
        // Layer 0 of the norm-folded form: the operand rows are the V = 8 embedding rows themselves, so in_proj's output is a look-up
        // (table built at bind time): one copy kernel instead of 1 / n_layer of the in_proj GEMMs.  PCAD_DEV=1 PCAD_NO_TAB0=1: the GEMM.
        if (c.fold && li == 0 && tab0) {
            ProfScope ps(e, PCAD_K_NORM, s);
            HIP_TRY(launch_embed_xz_gather(ids + (int64_t)c.b0 * L, e->comp, e->xz_tab0, c.w.xz, c.w.zb, c.Bc, L, E, dt, s));
        } else
        for (int rep = 0; rep < reps(PCAD_K_GEMM_IN); ++rep)
        { ProfScope ps(e, PCAD_K_GEMM_IN, s);
        if (c.fold) HIP_TRY(launch_gemm_nt_two(c.w.u, Dp, W.W_in_f, D, c.w.xz, c.w.zb, E, true, rows, 2 * E, D, dt, s, c.w.rstd));
        else if (sp) HIP_TRY(launch_gemm_nt_two(c.w.u, 2 * D, W.W_in_s, 2 * D, c.w.xz, c.w.zb, E, true, rows, 2 * E, 3 * D, BF16, s, nullptr, F32, D / 64));
        else if (e->xzsplit) HIP_TRY(launch_gemm_nt_two(c.w.u, D, W.W_in, D, c.w.xz, c.w.zb, E, true, rows, 2 * E, D, dt, s));
        else HIP_TRY(launch_gemm_nt(c.w.u, D, W.W_in, D, c.w.xz, 2 * E, rows, 2 * E, D, dt, dt, false, s)); }
        // conv1d + SiLU, causal and anti-causal from one read of x (fused with x_proj of both directions when possible)
        const bool convx = e->convx && ((int64_t)rows + 16) * E * esz < ((int64_t)1 << 32);      // the fused kernel's 32-bit offsets
        if (convx) for (int rep = 0; rep < reps(PCAD_K_CONV); ++rep) {
            ProfScope ps(e, PCAD_K_CONV, s);
            HIP_TRY(launch_convx(c.w.xz, W.convw, sp ? W.dir[0].Wx_s : W.dir[0].Wx, c.w.xc[0], c.w.dtl[0], c.w.bc[0], sp ? W.dir[1].Wx_s : W.dir[1].Wx, c.w.xc[1],
                                 c.w.dtl[1], c.w.bc[1], S, L, E, dt, s, Rp, sp, sp, c.w.cxp, 2 * B));      // sp: dt_low as bf16 [hi | lo] for the scan's split dt_proj
        } else {
            ProfScope ps(e, PCAD_K_CONV, s);
            HIP_TRY(launch_conv_bidir(c.w.xz, e->xzsplit ? E : 2 * E, W.dir[0].conv_w, W.dir[0].conv_b, W.dir[1].conv_w,
                                      W.dir[1].conv_b, c.w.xc[0], c.w.xc[1], S, L, E, dt, e->blocked, s, e->xzsplit));
        }
        return PCAD_OK;
    };
    // Last-layer shortcut (reference callers read ONE position: src/zero_shot_score.py:117,
    // src/train_XGBoost.py:105): with a shared list of P evaluated positions only rows p_q of the forward strands and L - 1 - p_q of
    // the reverse-complement strands of the LAST mixer's output are consumed.  The left-to-right scan stops after the furthest of
    // them, the right-to-left scan likewise (walk_len steps each), and the tied out_proj runs on the 2B * P gathered rows.  Same
    // arithmetic on the consumed rows (sequential walks, row-independent GEMM): results are bit-identical to the full layer.
    int walk_len = 0;
    if (e->shortcut && P > 0 && !pos_per_seq && !all_hidden && (int64_t)P * E <= (int64_t)L * D) {
        int pmin = pos.p[0], pmax = pos.p[0];
        for (int i = 1; i < P; ++i) { pmin = pos.p[i] < pmin ? pos.p[i] : pmin; pmax = pos.p[i] > pmax ? pos.p[i] : pmax; }
        const int need = (pmax + 1 > L - pmin) ? pmax + 1 : L - pmin;      // forward strands need row pmax, rc strands row L - 1 - pmin
        walk_len = (need + 7) / 8 * 8;                                   // whole 8-step groups (two prefetch chunks)
        if (walk_len > L) walk_len = L;
    }
    auto phase_V = [&](Lane& c, int li) -> int {        // x_proj + fused dt_proj/scan, both directions; out_proj
        hipStream_t s = cs;
        const LayerWeights& W = e->layers[li];
        const int S = 2 * c.Bc;
        const int64_t rows = (int64_t)S * L;
        const bool last_short = walk_len > 0 && li + 1 == e->nl;
        // split-bf16 dt_proj inside the scan ("f32_gemm_split"): the fused conv + x_proj kernel wrote dt_low as bf16 [rows, 3 Rp]
        const bool dts = sp && e->convx && ((int64_t)rows + 16) * E * esz < ((int64_t)1 << 32);
        // strict reference order ("reference_order" 2; never with norm_fold): the reverse direction's gated output goes to its own
        // tensor (xc[0]: the forward scan, its only reader, has run) and each direction gets its own tied out_proj below
        const bool strict = e->ref_order == 2 && !c.fold;
        void* y_rev = strict ? c.w.xc[0] : c.w.y;
        // "f32_gemm_split": out_proj's [hi | lo] operand is written by the gating (reverse) scan itself where it can (whole walk,
        // unsegmented, L % 8 == 0, one out_proj for both directions), instead of fp32 y + a conversion pass
        const bool ys_from_scan = sp && !strict && !last_short && L % 8 == 0 && e->blocked && e->xzsplit &&
                                  !(c.w.seg && scan_segments(2 * B, L, E, nullptr) > 1);
        // the full-size tied out_proj of one [rows, E] tensor (y, or in the strict order each direction's own): fp32 / bf16 GEMM, or the
        // split-bf16 form (operand conversion unless the scan wrote it + bf16 GEMM with K' = 3E, fp32 result)
        auto out_proj_full = [&](const void* ysrc, void* dst) -> hipError_t {
            if (sp) {
                if (!(ys_from_scan && ysrc == c.w.y)) {
                    if (hipError_t er = launch_split_rows((const float*)ysrc, E, c.w.ys, rows, E, e->blocked, e->blocked, s)) return er;
                }
                return launch_gemm_nt(c.w.ys, 2 * E, W.W_out_s, 2 * E, dst, D, rows, D, 3 * E, BF16, F32, false, s, e->blocked, E / 64);
            }
            return launch_gemm_nt(ysrc, E, W.W_out, E, dst, D, rows, D, E, dt, dt, false, s, e->blocked);
        };
        // Pair walks (kernels.hpp scan_pair_wanted: few waves per launch - long windows in small batches): both directions in one
        // launch, half a strand each, twice; chosen from the strands of the whole call like the segmented form
        // (never the LAST layer: with a list of positions its walks are shortened plain walks - "last_layer_shortcut" - and the full
        // layer must stay bit-identical to them on the evaluated rows)
        const bool pair = c.w.pair && !strict && li + 1 < e->nl && e->convx && ((int64_t)rows + 16) * E * esz < ((int64_t)1 << 32) &&
                          scan_pair_wanted(2 * B, L, E) && reps(PCAD_K_SCAN) == 1;
        if (pair) {
            const DirWeights &d0 = W.dir[0], &d1 = W.dir[1];
            const ScanDirection f{c.w.xc[0], c.w.dtl[0], dts ? d0.Wdt_s : d0.Wdt, c.w.bc[0], d0.A2, d0.Dskip, d0.dt_bias};
            const ScanDirection r{c.w.xc[1], c.w.dtl[1], dts ? d1.Wdt_s : d1.Wdt, c.w.bc[1], d1.A2, d1.Dskip, d1.dt_bias};
            for (int ph = 1; ph <= 2; ++ph) {
                ProfScope ps(e, PCAD_K_SCAN, s);
                HIP_TRY(launch_scan_pair(f, r, c.w.zb, dts ? 2 * Rp : Rp, Rp, c.w.y, S, L, E, !e->gate_once, dt, s, c.w.pair,
                                         ys_from_scan ? c.w.ys : nullptr, dts, ph));
            }
        }
        for (int d = 0; d < 2 && !pair; ++d) {
            const DirWeights& dw = W.dir[d];
            // x_proj -> dt_low [rows, Rp] (model dtype, zero padded) and B_t | C_t [rows, 32] (fp32 side output)
            if (!(e->convx && ((int64_t)rows + 16) * E * esz < ((int64_t)1 << 32))) { ProfScope ps(e, PCAD_K_GEMM_X, s);
            HIP_TRY(launch_gemm_nt_split(c.w.xc[d], E, dw.Wx, E, c.w.dtl[d], Rp, c.w.bc[d], 2 * N, Rp, rows, XP, E, dt, s,
                                         e->blocked)); }
            // dt_proj (on MFMA inside the scan) + bias + softplus + recurrence + D skip + SiLU(z) gate
            for (int rep = 1; rep < (d == 0 ? reps(PCAD_K_SCAN) : 1); ++rep)         // measurement aid: the forward-direction launch is idempotent
                HIP_TRY(launch_scan(c.w.xc[d], nullptr, e->xzsplit ? E : 2 * E, nullptr, c.w.dtl[d], dts ? 2 * Rp : Rp, dts ? dw.Wdt_s : dw.Wdt, Rp, c.w.bc[d], dw.A2, 1.0f,
                                    dw.Dskip, dw.dt_bias, c.w.y, S, L, E, false, 0, dt, s, e->blocked, e->xzsplit, c.w.seg, 0, nullptr, dts, 2 * B));
            ProfScope ps(e, PCAD_K_SCAN, s);
            const void* zp = e->xzsplit ? c.w.zb : (const void*)((const char*)c.w.xz + (size_t)E * esz);
            // gate_once: the forward scan stores its ungated output, the reverse scan adds its own and applies SiLU(z)
            // to the sum (one SiLU per element instead of two, z read once; a rounding-order difference from
            // y_f*g + y_r*g, like the out_proj fold below).  PCAD_GATE_EACH=1: each direction gated and rounded.
            const bool gated = strict || !e->gate_once || d == 1;
            HIP_TRY(launch_scan(c.w.xc[d], gated ? zp : nullptr, e->xzsplit ? E : 2 * E, nullptr, c.w.dtl[d], dts ? 2 * Rp : Rp, dts ? dw.Wdt_s : dw.Wdt, Rp,
                                c.w.bc[d], dw.A2, 1.0f, dw.Dskip, dw.dt_bias, d == 1 ? y_rev : c.w.y, S, L, E, d == 1,
                                strict ? 0 : (d == 1 ? (e->gate_once ? 2 : 1) : 0), dt, s, e->blocked, e->xzsplit, c.w.seg, last_short ? walk_len : 0,
                                d == 1 && ys_from_scan ? c.w.ys : nullptr, dts, 2 * B));
        }
        if (strict) {
            // out = round(out_proj(y_fwd)) + round(out_proj(y_rev)), rounded: BiMambaWrapper's "add" of two Mamba calls that each end
            // in their own (tied) out_proj.  Second output: u (dead since in_proj); last-layer shortcut: the gathered rows go
            // through u, the second small output to xz (dead since the scans).
            if (last_short) {
                ProfScope ps(e, PCAD_K_HEAD, s);
                // the gathered rows of one direction (in u) through the tied out_proj: the same product as the full-size launch
                // (split-bf16 with "f32_gemm_split": bit-identical rows)
                auto out_proj_rows = [&](void* dst) -> hipError_t {
                    if (sp) {
                        if (hipError_t er = launch_split_rows((const float*)c.w.u, E, c.w.ys, (int64_t)S * P, E, false, false, s)) return er;
                        return launch_gemm_nt(c.w.ys, 2 * E, W.W_out_s, 2 * E, dst, D, (int64_t)S * P, D, 3 * E, BF16, F32, false, s, false, E / 64);
                    }
                    return launch_gemm_nt(c.w.u, E, W.W_out, E, dst, D, (int64_t)S * P, D, E, dt, dt, false, s, false);
                };
                HIP_TRY(launch_gather_rows(c.w.y, c.w.u, c.Bc, L, E, pos, dt, e->blocked, s));
                HIP_TRY(out_proj_rows(c.w.h));
                HIP_TRY(launch_gather_rows(y_rev, c.w.u, c.Bc, L, E, pos, dt, e->blocked, s));
                HIP_TRY(out_proj_rows(c.w.xz));
                HIP_TRY(launch_add_round(c.w.h, c.w.xz, (int64_t)S * P * D, dt, s));
                return PCAD_OK;
            }
            { ProfScope ps(e, PCAD_K_GEMM_OUT, s);
            HIP_TRY(out_proj_full(c.w.y, c.w.h)); }
            { ProfScope ps(e, PCAD_K_GEMM_OUT, s);
            HIP_TRY(out_proj_full(y_rev, c.w.u)); }
            { ProfScope ps(e, PCAD_K_NORM, s);
            HIP_TRY(launch_add_round(c.w.h, c.w.u, rows * D, dt, s)); }
            if (all_hidden && li + 1 < e->nl) {
                char* dst = (char*)all_hidden + ((size_t)(li + 1) * B * L * 2 * D + (size_t)c.b0 * L * 2 * D) * esz;
                HIP_TRY(launch_assemble_hidden(c.w.h, dst, c.Bc, L, D, dt, s));
            }
            return PCAD_OK;
        }
        if (last_short) {       // out_proj on the evaluated rows only: gather (-> u, dead since in_proj) and a small GEMM (-> first rows of h)
            ProfScope ps(e, PCAD_K_HEAD, s);          // counted with the head: not a full-size out_proj launch
            HIP_TRY(launch_gather_rows(c.w.y, c.w.u, c.Bc, L, E, pos, dt, e->blocked, s));
            if (sp) {           // the same split-bf16 product as the full-size out_proj (same operand values, same K order: bit-identical rows)
                HIP_TRY(launch_split_rows((const float*)c.w.u, E, c.w.ys, (int64_t)S * P, E, false, false, s));
                HIP_TRY(launch_gemm_nt(c.w.ys, 2 * E, W.W_out_s, 2 * E, c.w.h, D, (int64_t)S * P, D, 3 * E, BF16, F32, false, s, false, E / 64));
                return PCAD_OK;
            }
            HIP_TRY(launch_gemm_nt(c.w.u, E, W.W_out, E, c.w.h, D, (int64_t)S * P, D, E, dt, dt, false, s, false));
            return PCAD_OK;
        }
        if (c.fold && li + 1 < e->nl) {     // out_proj + residual add + the next block's norm statistics in one launch
            { ProfScope ps(e, PCAD_K_GEMM_OUT_RES, s);
            HIP_TRY(launch_gemm_nt_res(c.w.y, E, W.W_out_p, E, c.w.u, (float*)c.w.res, c.w.ssq, rows, Dp, E, dt, s, e->blocked)); }
            ProfScope ps(e, PCAD_K_RSTD, s);
            HIP_TRY(launch_rstd(c.w.ssq, c.w.rstd, rows, Dp / 128, D, eps, s));
            return PCAD_OK;
        }
        // out_proj on (y_fwd + y_rev): the two tied out_proj calls folded by linearity
        for (int rep = 0; rep < reps(PCAD_K_GEMM_OUT); ++rep)
        { ProfScope ps(e, PCAD_K_GEMM_OUT, s);
        HIP_TRY(out_proj_full(c.w.y, c.w.h)); }
        if (all_hidden && li + 1 < e->nl) {
            char* dst = (char*)all_hidden + ((size_t)(li + 1) * B * L * 2 * D + (size_t)c.b0 * L * 2 * D) * esz;
            HIP_TRY(launch_assemble_hidden(c.w.h, dst, c.Bc, L, D, dt, s));
        }
        return PCAD_OK;
    };
    auto phase_head = [&](Lane& c) -> int {
        void* hout = hidden_out ? (char*)hidden_out + ((size_t)c.b0 * Q * 2 * D) * esz : nullptr;
        float* lout = logits_out ? logits_out + (size_t)c.b0 * Q * e->V : nullptr;
        if (hout || lout) {
            ProfScope ps(e, PCAD_K_HEAD, cs);
            HIP_TRY(launch_final_head(c.w.h, c.w.res, e->normf_w, e->emb, e->emb_f32, e->comp, hout, lout, c.Bc, L, D, eps,
                                      pos, pos_per_seq ? pos_per_seq + c.b0 : nullptr, dt, rdt, cs, walk_len > 0,
                                      ids + (int64_t)c.b0 * L, e->status, c.fold ? Dp : 0));
        }
        return PCAD_OK;
    };

    for (int ck = 0; ck < nchunks; ++ck) {
        Lane c;
        c.b0 = ck * chunk;
        c.Bc = (B - c.b0) < chunk ? (B - c.b0) : chunk;
        c.w = carve_workspace(e, workspace, c.Bc, L, B);
        c.fold = fold_for(c);
        for (int li = 0; li < e->nl; ++li) {
            if (int rc = phase_N(c, li)) return rc;
            if (int rc = phase_P(c, li)) return rc;
            if (int rc = phase_V(c, li)) return rc;
        }
        if (int rc = phase_head(c)) return rc;
    }
    return PCAD_OK;
}

int pcad_forward(pcad_handle h, const int32_t* ids, int B, int L, const int32_t* positions, int P, void* hidden_out,
                 float* logits_out, void* workspace, size_t workspace_bytes, pcad_stream stream) {
    return forward_impl(h, ids, B, L, positions, P, nullptr, nullptr, hidden_out, logits_out, workspace, workspace_bytes,
                        stream);
}

int pcad_forward_at(pcad_handle h, const int32_t* ids, int B, int L, const int32_t* pos_per_seq, void* hidden_out,
                    float* logits_out, void* workspace, size_t workspace_bytes, pcad_stream stream) {
    if (!pos_per_seq) return fail(PCAD_ERR_INVALID, "pcad_forward_at: null pos_per_seq");
    return forward_impl(h, ids, B, L, nullptr, 0, pos_per_seq, nullptr, hidden_out, logits_out, workspace, workspace_bytes,
                        stream);
}

int pcad_forward_all_hidden(pcad_handle h, const int32_t* ids, int B, int L, void* all_hidden, void* hidden_out,
                            float* logits_out, void* workspace, size_t workspace_bytes, pcad_stream stream) {
    if (!all_hidden) return fail(PCAD_ERR_INVALID, "pcad_forward_all_hidden: null all_hidden");
    return forward_impl(h, ids, B, L, nullptr, 0, nullptr, all_hidden, hidden_out, logits_out, workspace, workspace_bytes,
                        stream);
}

int pcad_profile_enable(pcad_handle h, int on) {
    if (!h) return fail(PCAD_ERR_INVALID, "pcad_profile_enable: null handle");
    h->prof = on != 0;
    h->prof_stride = on > 1 ? on : 1;
    for (int c = 0; c < PCAD_NUM_KERNEL_CLASSES; ++c) h->prof_seen[c] = 0;
    return PCAD_OK;
}

int pcad_profile_read(pcad_handle h, pcad_kernel_stat* out, int max_out) {
    if (!h || !out) return fail(PCAD_ERR_INVALID, "pcad_profile_read: null argument");
    for (int c = 0; c < PCAD_NUM_KERNEL_CLASSES; ++c) {
        for (auto& pr : h->prof_ev[c]) {
            HIP_TRY(hipEventSynchronize(pr.second));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            h->prof_ms[c] += ms;
            h->prof_n[c] += 1;
            h->prof_pool.push_back(pr.first);
            h->prof_pool.push_back(pr.second);
        }
        h->prof_ev[c].clear();
    }
    int n = 0;
    for (int c = 0; c < PCAD_NUM_KERNEL_CLASSES && n < max_out; ++c, ++n) {
        snprintf(out[n].name, sizeof(out[n].name), "%s", kClassNames[c]);
        out[n].launches = h->prof_n[c];
        out[n].total_ms = h->prof_ms[c];
        h->prof_n[c] = 0;
        h->prof_ms[c] = 0.0;
    }
    return n;
}

// ---- per-operator entry points -------------------------------------------------------------------
int pcad_add_rmsnorm(const void* x, const void* residual_in, const float* weight, void* y, void* residual_out,
                     int64_t rows, int D, float eps, int dtype, int res_dtype, pcad_stream stream) {
    if (!x || !weight || !y) return fail(PCAD_ERR_INVALID, "pcad_add_rmsnorm: null argument");
    if (rows < 0 || D <= 0 || D % 8 || D > 2048) return fail(PCAD_ERR_INVALID, "pcad_add_rmsnorm: bad rows/D");
    HIP_TRY(launch_add_rmsnorm(x, residual_in, weight, y, residual_out, rows, D, eps, dtype, res_dtype,
                               (hipStream_t)stream));
    return PCAD_OK;
}

int pcad_causal_conv1d_silu(const void* x, int64_t ldx, const float* w_fwd, const float* b_fwd, const float* w_rev,
                            const float* b_rev, void* y_fwd, void* y_rev, int S, int L, int E, int dtype,
                            pcad_stream stream) {
    if (!x || !w_fwd || !b_fwd || !w_rev || !b_rev) return fail(PCAD_ERR_INVALID, "pcad_causal_conv1d_silu: null argument");
    if (S < 0 || L < 0 || E <= 0 || E % 8 || ldx < E || ldx % 8)
        return fail(PCAD_ERR_INVALID, "pcad_causal_conv1d_silu: bad shape (E and ldx must be multiples of 8)");
    HIP_TRY(launch_conv_bidir(x, ldx, w_fwd, b_fwd, w_rev, b_rev, y_fwd, y_rev, S, L, E, dtype, false, (hipStream_t)stream));
    return PCAD_OK;
}

size_t pcad_conv_xproj_scratch_bytes(int E, int dtype) {
    if (E <= 0 || (dtype != PCAD_F32 && dtype != PCAD_BF16) || (E * (dtype == PCAD_BF16 ? 2 : 4)) % 128) return 0;
    return convx_packed_bytes(E, dtype);
}

int pcad_conv_xproj_bidir(const void* x, const float* w_fwd, const float* b_fwd, const float* w_rev, const float* b_rev,
                          const void* Wx_fwd, const void* Wx_rev, void* scratch, void* xc_fwd, void* dtl_fwd,
                          float* bc_fwd, void* xc_rev, void* dtl_rev, float* bc_rev, int S, int L, int E, int Rp, int dtype,
                          pcad_stream stream) {
    if (!x || !w_fwd || !b_fwd || !w_rev || !b_rev || !Wx_fwd || !Wx_rev || !scratch || !xc_fwd || !dtl_fwd || !bc_fwd ||
        !xc_rev || !dtl_rev || !bc_rev)
        return fail(PCAD_ERR_INVALID, "pcad_conv_xproj_bidir: null argument");
    if (dtype != PCAD_F32 && dtype != PCAD_BF16) return fail(PCAD_ERR_INVALID, "pcad_conv_xproj_bidir: bad dtype");
    if (Rp != 64 && Rp != 96) return fail(PCAD_ERR_INVALID, "pcad_conv_xproj_bidir: Rp must be 64 (dt_rank <= 64) or 96 (dt_rank 65..96)");
    const int64_t esz = dtype == PCAD_BF16 ? 2 : 4;
    if (S < 0 || L < 0 || E <= 0 || (E * esz) % 128)
        return fail(PCAD_ERR_INVALID, "pcad_conv_xproj_bidir: E * elem must be a multiple of 128 bytes");
    if (((int64_t)S * L + 16) * E * esz >= ((int64_t)1 << 32))
        return fail(PCAD_ERR_INVALID, "pcad_conv_xproj_bidir: (S*L + 16) * E * elem must be < 2^32 (32-bit in-tensor offsets)");
    if (S == 0 || L == 0) return PCAD_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_pack_convw(w_fwd, b_fwd, w_rev, b_rev, (float*)scratch, E, dtype, s));
    HIP_TRY(launch_convx(x, (const float*)scratch, Wx_fwd, xc_fwd, dtl_fwd, bc_fwd, Wx_rev, xc_rev, dtl_rev, bc_rev, S, L, E,
                         dtype, s, Rp));
    return PCAD_OK;
}

static int scan_args_ok(const void* u, const float* bc, const float* A, const float* Dskip, const float* delta_bias,
                        void* y, int S, int L, int E) {
    if (!u || !bc || !A || !Dskip || !delta_bias || !y) return fail(PCAD_ERR_INVALID, "pcad_selective_scan: null argument");
    if (S < 0 || L < 0 || E <= 0 || E % 64) return fail(PCAD_ERR_INVALID, "pcad_selective_scan: E must be a multiple of 64");
    if (((uintptr_t)bc) % 16) return fail(PCAD_ERR_INVALID, "pcad_selective_scan: bc must be 16-byte aligned");
    return PCAD_OK;
}

int pcad_selective_scan(const void* u, const void* delta, const void* z, int64_t ldz, const float* bc, const float* A,
                        const float* Dskip, const float* delta_bias, void* y, int S, int L, int E, int reverse,
                        int accumulate, int dtype, pcad_stream stream) {
    if (!delta) return fail(PCAD_ERR_INVALID, "pcad_selective_scan: null delta");
    if (int rc = scan_args_ok(u, bc, A, Dskip, delta_bias, y, S, L, E)) return rc;
    if (S == 0 || L == 0) return PCAD_OK;
    // raw A is scaled by log2(e) when the kernel loads it into registers (the engine passes pre-scaled A)
    HIP_TRY(launch_scan(u, z, ldz, delta, nullptr, 0, nullptr, 0, bc, A, 1.4426950408889634f, Dskip, delta_bias, y, S, L,
                        E, reverse != 0, accumulate, dtype, (hipStream_t)stream));
    return PCAD_OK;
}

int pcad_selective_scan_dtproj(const void* u, const void* dt_low, int64_t lddt, const void* Wdt, int Rp, const void* z,
                               int64_t ldz, const float* bc, const float* A, const float* Dskip,
                               const float* delta_bias, void* y, int S, int L, int E, int reverse, int accumulate,
                               int dtype, pcad_stream stream) {
    if (!dt_low || !Wdt) return fail(PCAD_ERR_INVALID, "pcad_selective_scan_dtproj: null dt_low / Wdt");
    if (Rp <= 0 || Rp % 32 || lddt < Rp || ((uintptr_t)dt_low) % 16 || ((uintptr_t)Wdt) % 16 ||
        (lddt * (dtype == PCAD_BF16 ? 2 : 4)) % 16)
        return fail(PCAD_ERR_INVALID, "pcad_selective_scan_dtproj: Rp must be a multiple of 32 (zero padded), rows 16-byte aligned");
    if (int rc = scan_args_ok(u, bc, A, Dskip, delta_bias, y, S, L, E)) return rc;
    if (S == 0 || L == 0) return PCAD_OK;
    HIP_TRY(launch_scan(u, z, ldz, nullptr, dt_low, lddt, Wdt, Rp, bc, A, 1.4426950408889634f, Dskip, delta_bias, y, S, L,
                        E, reverse != 0, accumulate, dtype, (hipStream_t)stream));
    return PCAD_OK;
}

int pcad_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int N, int K,
                 int dtype, int out_dtype, pcad_stream stream) {
    if (!A || !W || !C) return fail(PCAD_ERR_INVALID, "pcad_gemm_nt: null argument");
    hipError_t err = launch_gemm_nt(A, lda, W, ldw, C, ldc, M, N, K, dtype, out_dtype, false, (hipStream_t)stream);
    if (err == hipErrorInvalidValue)
        return fail(PCAD_ERR_INVALID, "pcad_gemm_nt: K*elem must be a multiple of 128 bytes; A/W 16-byte aligned rows");
    if (err != hipSuccess) return fail(PCAD_ERR_HIP, "pcad_gemm_nt: %s", hipGetErrorString(err));
    return PCAD_OK;
}

size_t pcad_gemm_nt_split_scratch_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return align_up((size_t)M * 2 * K * 2) + align_up((size_t)N * 2 * K * 2);       // bf16 [M, 2K] = [hi | lo] of A, bf16 [N, 2K] of W
}

int pcad_gemm_nt_split(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N, int K,
                       void* scratch, size_t scratch_bytes, pcad_stream stream) {
    if (!A || !W || !C || !scratch) return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_split: null argument");
    if (M < 0 || N <= 0 || K <= 0 || K % 64 || lda < K || ldw < K || ldc < N)
        return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_split: K must be a multiple of 64; lda, ldw >= K; ldc >= N");
    if (((uintptr_t)scratch) % 256 || scratch_bytes < pcad_gemm_nt_split_scratch_bytes(M, N, K))
        return fail(PCAD_ERR_WORKSPACE, "pcad_gemm_nt_split: scratch must be 256-byte aligned and pcad_gemm_nt_split_scratch_bytes large");
    if (M == 0) return PCAD_OK;
    hipStream_t s = (hipStream_t)stream;
    void* As = scratch;
    void* Ws = (char*)scratch + align_up((size_t)M * 2 * K * 2);
    HIP_TRY(launch_split_rows(A, lda, As, M, K, false, false, s));                // [hi | lo]
    HIP_TRY(launch_pack_split_w(W, PCAD_F32, ldw, Ws, N, K, s));                  // [hi | lo]
    // 3 K / 64 K-tiles, the cursor wrapping around both operands: a_hi w_hi + a_lo w_hi + a_hi w_lo
    hipError_t err = launch_gemm_nt(As, 2 * (int64_t)K, Ws, 2 * (int64_t)K, C, ldc, M, N, 3 * K, BF16, F32, false, s, false, K / 64);
    if (err != hipSuccess) return fail(PCAD_ERR_HIP, "pcad_gemm_nt_split: %s", hipGetErrorString(err));
    return PCAD_OK;
}

int pcad_gemm_nt_residual(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, float* res, float* ssq, int64_t M, int N,
                          int K, int dtype, pcad_stream stream) {
    if (!A || !W || !res || !ssq || !C) return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_residual: null argument");
    if (dtype != PCAD_F32 && dtype != PCAD_BF16) return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_residual: bad dtype");
    if (M < 0 || N <= 0 || K <= 0 || M % 256 || N % 256 || M * (int64_t)N * 4 >= ((int64_t)1 << 32))
        return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_residual: M and N must be multiples of 256 and M * N * 4 < 2^32");
    hipError_t err = launch_gemm_nt_res(A, lda, W, ldw, C, res, ssq, M, N, K, dtype, (hipStream_t)stream, false);
    if (err == hipErrorInvalidValue)
        return fail(PCAD_ERR_INVALID, "pcad_gemm_nt_residual: K*elem must be a multiple of 128 bytes; 16-byte aligned rows; tensors < 4 GiB");
    if (err != hipSuccess) return fail(PCAD_ERR_HIP, "pcad_gemm_nt_residual: %s", hipGetErrorString(err));
    return PCAD_OK;
}

static int positions_arg(const char* who, const int32_t* positions, int P, int L, Positions* pos) {
    if (P < 0 || P > PCAD_MAX_POSITIONS || (P > 0 && !positions)) return fail(PCAD_ERR_INVALID, "%s: bad positions (P=%d)", who, P);
    pos->n = P;
    for (int i = 0; i < 16; ++i) pos->p[i] = 0;
    for (int i = 0; i < P; ++i) {
        if (positions[i] < 0 || positions[i] >= L) return fail(PCAD_ERR_INVALID, "%s: position %d out of range [0,%d)", who, positions[i], L);
        pos->p[i] = positions[i];
    }
    return PCAD_OK;
}

int pcad_gather_rows(const void* src, void* out, int B, int L, int E, const int32_t* positions, int P, int dtype, pcad_stream stream) {
    if (!src || !out) return fail(PCAD_ERR_INVALID, "pcad_gather_rows: null argument");
    if (dtype != PCAD_F32 && dtype != PCAD_BF16) return fail(PCAD_ERR_INVALID, "pcad_gather_rows: bad dtype");
    if (B < 0 || L <= 0 || E <= 0 || (E * (dtype == PCAD_BF16 ? 2 : 4)) % 16 || P < 1)
        return fail(PCAD_ERR_INVALID, "pcad_gather_rows: bad shape (E * elem must be a multiple of 16 bytes, P >= 1)");
    Positions pos;
    if (int rc = positions_arg("pcad_gather_rows", positions, P, L, &pos)) return rc;
    if (B == 0) return PCAD_OK;
    HIP_TRY(launch_gather_rows(src, out, B, L, E, pos, dtype, false, (hipStream_t)stream));
    return PCAD_OK;
}

int pcad_final_head(const void* h, const void* res, const float* norm_weight, const float* emb_f32, const int32_t* complement,
                    void* hidden_out, float* logits_out, int B, int L, int D, float eps, const int32_t* positions, int P,
                    const int32_t* pos_per_seq, int h_compact, const int32_t* ids, int32_t* status, int dtype, int res_dtype,
                    int res_fragment_layout, pcad_stream stream) {
    if (!h || !res || !norm_weight || !emb_f32 || !complement) return fail(PCAD_ERR_INVALID, "pcad_final_head: null argument");
    if (res_fragment_layout && (res_dtype != PCAD_F32 || D % 256 || ((int64_t)2 * B * L) % 256))
        return fail(PCAD_ERR_INVALID, "pcad_final_head: the fragment layout needs an fp32 residual, D %% 256 == 0 and 2 B L %% 256 == 0");
    if ((dtype != PCAD_F32 && dtype != PCAD_BF16) || (res_dtype != PCAD_F32 && res_dtype != PCAD_BF16) || (dtype == PCAD_F32 && res_dtype != PCAD_F32))
        return fail(PCAD_ERR_INVALID, "pcad_final_head: bad dtype / res_dtype");
    if (B < 0 || L <= 0 || D <= 0 || D % 8 || D > 2048) return fail(PCAD_ERR_INVALID, "pcad_final_head: bad B / L / D");
    if (pos_per_seq && (positions || P)) return fail(PCAD_ERR_INVALID, "pcad_final_head: positions and pos_per_seq are exclusive");
    if (h_compact && (pos_per_seq || P == 0)) return fail(PCAD_ERR_INVALID, "pcad_final_head: h_compact needs a shared list of positions");
    Positions pos;
    if (int rc = positions_arg("pcad_final_head", positions, P, L, &pos)) return rc;
    if (B == 0) return PCAD_OK;
    HIP_TRY(launch_final_head(h, res, norm_weight, nullptr, emb_f32, complement, hidden_out, logits_out, B, L, D, eps, pos, pos_per_seq,
                              dtype, res_dtype, (hipStream_t)stream, h_compact != 0, ids, status, res_fragment_layout ? D : 0));
    return PCAD_OK;
}

}  // extern "C"
