/*
 * CPU ORACLE (C / OpenMP) — TEST INFRASTRUCTURE ONLY.  Never linked into or called by the product
 * path (libpcad.so / plantcaduceus_amd).  Used by tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg only, as the checker and as the reported host-CPU baseline ("port").
 *
 * fp32 restatement of the PlantCaduceus masked-LM forward in the 2B-strand form
 * (oracle/caduceus_oracle.py::forward_strands is the same algorithm in torch and
 * is what this file is validated against in tests/test_oracle.py).  The algorithm follows the
 * third-party packages the reference pins but does not vendor — mamba-ssm==2.2.2, causal-conv1d==1.4.0
 * (reference env/requirements.txt:9-10), Triton fused add-norm (env/environment.yml:12), HF-hub
 * modeling_caduceus.py / modeling_rcps.py — as called from reference src/zero_shot_score.py:115 and
 * src/train_XGBoost.py:104.  "Parity unpinned" by the reference's own tests (it has none); pinned by
 * the fixtures described in oracle/caduceus_oracle.py.
 *
 * Parallelisation: the 2B-strand batch walks the stack layer by layer and every operator is an OpenMP
 * parallel loop (GEMM: weight-panel x row-block tiles; conv/norm: token rows; scan: strand x 32-channel
 * blocks), so even a small sample uses all host cores.  Plain blocked loops the compiler vectorises, no
 * intrinsics: a scalar-source port.  The channel loops only vectorise with branch-free exp / log / softplus
 * and -fno-trapping-math (oracle/c/Makefile) - until round 4 they silently did not, and the scan ran scalar
 * (3x slower; outputs are bit-identical to that build).  The projections can be routed to the host BLAS
 * instead (oracle_set_gemm).  ORACLE_TIMING=1 prints the per-operator wall time of a forward.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <time.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NST 16
#define VL 16

typedef struct {
    const float* norm_w;   /* [D] */
    const float* in_proj;  /* [2E, D] */
    const float* out_proj; /* [D, E] */
    /* per direction (0 = mamba_fwd, 1 = mamba_rev) */
    const float* conv_w[2];  /* [E, 4] */
    const float* conv_b[2];  /* [E] */
    const float* x_proj[2];  /* [R+2N, E] */
    const float* dt_w[2];    /* [E, R] */
    const float* dt_b[2];    /* [E] */
    const float* A_log[2];   /* [E, N] */
    const float* Dskip[2];   /* [E] */
} oracle_layer;

typedef struct {
    int32_t d_model, n_layer, d_inner, dt_rank;
    float eps;
    int32_t emulate_bf16;   /* round to bf16 at the reference's tensor boundaries (model run with torch_dtype=bfloat16) */
    int32_t ref_order;      /* 1: the reference's order of the tied out_proj — out_proj(y_fwd) and out_proj(y_rev) each
                               computed and rounded, then summed and rounded (BiMambaWrapper "add");  0: the engine's fold
                               out_proj(y_fwd + y_rev).  Identical in exact arithmetic. */
    int32_t complement[8];
    const float* emb;     /* [8, D] (tied LM head) */
    const float* norm_f;  /* [D] */
    const oracle_layer* layers;
} oracle_model;

/* round-to-nearest-even fp32 -> bf16 -> fp32 */
static inline float rbf(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    v.u += 0x7fffu + ((v.u >> 16) & 1u);
    v.u &= 0xffff0000u;
    return v.f;
}
static void round_rows(float* x, size_t n, int on) {
    if (!on) return;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) x[i] = rbf(x[i]);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* expf with a Cephes-style polynomial, written branch-free (clamp + final select) so that the channel loops vectorise
 * (rel. error ~1e-7; exactly 0 below -87 like the branchy form it replaces) */
static inline float vexpf(float x) {
    const float xc = x > 88.0f ? 88.0f : (x < -87.0f ? -87.0f : x);
    const float fx = floorf(xc * 1.44269504088896341f + 0.5f);
    const float r = (xc - fx * 0.693359375f) - fx * -2.12194440e-4f;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    p = p * r * r + r + 1.0f;
    union { float f; int32_t i; } u;
    u.i = ((int32_t)fx + 127) << 23;
    const float v = p * u.f;
    return x < -87.0f ? 0.0f : v;
}

/* logf for normal positive x, Cephes polynomial (vectorisable; rel. error ~1e-7) */
static inline float vlogf(float x) {
    union { float f; int32_t i; } u;
    u.f = x;
    int32_t e = ((u.i >> 23) & 255) - 126;
    u.i = (u.i & 0x007fffff) | 0x3f000000;   /* mantissa in [0.5, 1) */
    const float m0 = u.f;
    const int lo = m0 < 0.707106781186547524f;
    e -= lo;
    const float m = lo ? m0 + m0 - 1.0f : m0 - 1.0f;
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = y * m - 1.1514610310E-1f;
    y = y * m + 1.1676998740E-1f;
    y = y * m - 1.2420140846E-1f;
    y = y * m + 1.4249322787E-1f;
    y = y * m - 1.6668057665E-1f;
    y = y * m + 2.0000714765E-1f;
    y = y * m - 2.4999993993E-1f;
    y = y * m + 3.3333331174E-1f;
    y = y * m * z;
    const float fe = (float)e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    return m + y + 0.693359375f * fe;
}

static inline float silu_f(float v) { return v / (1.0f + vexpf(-v)); }
/* softplus with torch's threshold 20; log1p(e) via Kahan's correction log(w) * e / (w - 1) */
static inline float softplus_f(float v) {
    const float e = vexpf(v > 20.0f ? 20.0f : v);
    const float w = 1.0f + e;
    const float d = w - 1.0f;
    const float l = d == 0.0f ? e : vlogf(w) * (e / d);
    return v > 20.0f ? v : l;
}

/* C[M,N] = A[M,K] . W[N,K]^T  (F.linear without bias).  Cache-blocked: a panel of NB weight rows stays in
 * L2 while all row blocks of A stream past it; inside, a 4x4 register block of VL-wide partial sums. */
static void linear_micro(const float* A, int lda, const float* W, int K, float* C, int ldc, int mi, int nj) {
    const int Kv = K - K % VL;
    if (mi == 4 && nj == 4) {
        float acc[4][4][VL] = {{{0.f}}};
        for (int k = 0; k < Kv; k += VL)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
#pragma omp simd
                    for (int l = 0; l < VL; ++l)
                        acc[i][j][l] += A[(size_t)i * lda + k + l] * W[(size_t)j * K + k + l];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                for (int l = 0; l < VL; ++l) s += acc[i][j][l];
                for (int kk = Kv; kk < K; ++kk) s += A[(size_t)i * lda + kk] * W[(size_t)j * K + kk];
                C[(size_t)i * ldc + j] = s;
            }
        return;
    }
    for (int i = 0; i < mi; ++i)
        for (int j = 0; j < nj; ++j) {
            const float* a = A + (size_t)i * lda;
            const float* w = W + (size_t)j * K;
            float part[VL] = {0.f};
            for (int k = 0; k < Kv; k += VL)
                for (int l = 0; l < VL; ++l) part[l] += a[k + l] * w[k + l];
            float s = 0.f;
            for (int l = 0; l < VL; ++l) s += part[l];
            for (int kk = Kv; kk < K; ++kk) s += a[kk] * w[kk];
            C[(size_t)i * ldc + j] = s;
        }
}

/* Optional host-BLAS hook for the four projections (oracle/c_oracle.py installs a numpy/OpenBLAS sgemm): the same
 * forward with library GEMMs, so the reported CPU baseline is not limited by this file's plain-C GEMM loops. */
typedef void (*oracle_gemm_fn)(const float* A, int lda, const float* W, int K, float* C, int ldc, int M, int N);
static oracle_gemm_fn g_gemm = 0;
void oracle_set_gemm(oracle_gemm_fn fn) { g_gemm = fn; }

static void linear_nt(const float* A, int lda, const float* W, int K, float* C, int ldc, int M, int N) {
    if (g_gemm) { g_gemm(A, lda, W, K, C, ldc, M, N); return; }
    /* one work item = MB rows x NB weight rows: the item's A block (MB*K floats) and weight panel (NB*K floats)
     * both stay in a core's L2 while its 4x4 micro-tiles are walked, so each is fetched once per item */
    enum { MB = 64, NB = 64 };
    const int nib = (M + MB - 1) / MB, njb = (N + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int ibi = 0; ibi < nib; ++ibi) {
        for (int jbi = 0; jbi < njb; ++jbi) {
            const int jb = jbi * NB, ib = ibi * MB;
            const int je = jb + NB < N ? jb + NB : N;
            const int ie = ib + MB < M ? ib + MB : M;
            for (int j0 = jb; j0 < je; j0 += 4)
                for (int i0 = ib; i0 < ie; i0 += 4)
                    linear_micro(A + (size_t)i0 * lda, lda, W + (size_t)j0 * K, K, C + (size_t)i0 * ldc + j0, ldc,
                                 ie - i0 < 4 ? ie - i0 : 4, je - j0 < 4 ? je - j0 : 4);
        }
    }
}

/* rms_norm_fn(prenorm=True) over `rows` token rows: res = x (+ res); y = res * rsqrt(mean(res^2)+eps) * w */
static void add_rmsnorm(const float* x, float* res, int has_res, const float* w, float* y, int64_t rows, int D,
                        float eps, int rb) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < rows; ++t) {
        float ss = 0.f;
        float* r = res + (size_t)t * D;
        const float* xr = x + (size_t)t * D;
        for (int c = 0; c < D; ++c) {
            const float v = has_res ? xr[c] + r[c] : xr[c];
            r[c] = v;
            ss += v * v;
        }
        const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
        for (int c = 0; c < D; ++c) { const float v = r[c] * rstd * w[c]; y[(size_t)t * D + c] = rb ? rbf(v) : v; }
    }
}

/* conv1d (width 4) + SiLU on x = xz[..., :E]; d = 1 is the anti-causal conv (reverse Mamba on unflipped rows) */
static void conv_silu(const float* xz, const float* cw, const float* cb, float* xc, int S, int L, int E, int d, int rb) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < L; ++t) {
            const float* base = xz + (size_t)s * L * 2 * E;
            float* o = xc + ((size_t)s * L + t) * E;
            /* taps outermost (same order of additions per channel: bias, then k = 0..3) so that the channel loops vectorise */
            for (int c = 0; c < E; ++c) o[c] = cb[c];
            for (int k = 0; k < 4; ++k) {
                const int tt = d == 0 ? t - 3 + k : t + 3 - k;
                if (tt < 0 || tt >= L) continue;
                const float* xr = base + (size_t)tt * 2 * E;
                for (int c = 0; c < E; ++c) o[c] += cw[c * 4 + k] * xr[c];
            }
            if (rb) for (int c = 0; c < E; ++c) o[c] = rbf(silu_f(o[c]));
            else for (int c = 0; c < E; ++c) o[c] = silu_f(o[c]);
        }
}

/* selective scan of one direction, accumulated into y (gated by silu(z)); sequential in t; parallel over (strand, block of CB
 * channels).  Rows of a strand are E floats apart, so a walk over t touches a new page per array per step: the per-(t, c) inputs of
 * TS steps are staged into a local tile first (independent loads, vectorised softplus / SiLU), the recurrence then runs out of
 * that tile, and the tile's outputs are written back - same operations in the same order per element as the plain loop nest. */
#define CB 32
#define TS 32
static void scan_dir(const float* xc, const float* delta, const float* dbl, const float* xz, const float* A,
                     const float* dt_b, const float* Dskip, float* y, int S, int L, int E, int R, int d, int rb, int acc) {
    const int XP = R + 2 * NST;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int c0 = 0; c0 < E; c0 += CB) {
            const size_t r0 = (size_t)s * L;
            const int nc = E - c0 < CB ? E - c0 : CB;
            float st[NST][CB], ab[NST][CB];
            float dvt[TS][CB], dut[TS][CB], yt[TS][CB], gt[TS][CB];
            memset(st, 0, sizeof(st));
            memset(ab, 0, sizeof(ab));
            for (int n = 0; n < NST; ++n)
                for (int l = 0; l < nc; ++l) ab[n][l] = A[(size_t)(c0 + l) * NST + n];
            for (int step0 = 0; step0 < L; step0 += TS) {
                const int nts = L - step0 < TS ? L - step0 : TS;
                for (int i = 0; i < nts; ++i) {
                    const size_t t = r0 + (d == 0 ? step0 + i : L - 1 - (step0 + i));
                    const float* dl = delta + t * E + c0;
                    const float* ur = xc + t * E + c0;
                    const float* zr = xz + t * 2 * E + E + c0;
                    for (int l = 0; l < nc; ++l) {
                        const float dv = softplus_f(dl[l] + dt_b[c0 + l]);
                        const float uv = ur[l];
                        dvt[i][l] = dv;
                        dut[i][l] = dv * uv;
                        yt[i][l] = Dskip[c0 + l] * uv;
                        gt[i][l] = silu_f(zr[l]);
                    }
                }
                for (int i = 0; i < nts; ++i) {
                    const size_t t = r0 + (d == 0 ? step0 + i : L - 1 - (step0 + i));
                    const float* Bt = dbl + t * XP + R;
                    const float* Ct = Bt + NST;
                    for (int n = 0; n < NST; ++n) {
                        const float bn = Bt[n], cn = Ct[n];
                        for (int l = 0; l < CB; ++l) {
                            const float a = vexpf(dvt[i][l] * ab[n][l]);
                            st[n][l] = a * st[n][l] + dut[i][l] * bn;
                            yt[i][l] += st[n][l] * cn;
                        }
                    }
                }
                for (int i = 0; i < nts; ++i) {
                    const size_t t = r0 + (d == 0 ? step0 + i : L - 1 - (step0 + i));
                    float* yr = y + t * E + c0;
                    for (int l = 0; l < nc; ++l) {
                        const float g = yt[i][l] * gt[i][l];                        /* each direction rounded, then summed */
                        if (acc) yr[l] = rb ? rbf(yr[l] + rbf(g)) : yr[l] + g;
                        else yr[l] = rb ? rbf(g) : g;
                    }
                }
            }
        }
}

/* ORACLE_TIMING=1: per-operator wall time of one forward on stderr (where the CPU baseline's time goes) */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
enum { T_NORM, T_INPROJ, T_ROUND, T_CONV, T_XPROJ, T_DTPROJ, T_SCAN, T_OUTPROJ, T_HEAD, T_MISC, T_N };
static const char* const t_names[T_N] = {"add_rmsnorm", "in_proj", "round", "conv_silu", "x_proj", "dt_proj", "scan", "out_proj", "head", "misc"};
#define TIMED(slot, stmt) do { const double _t0 = now_s(); stmt; tsum[slot] += now_s() - _t0; } while (0)

int oracle_forward(const oracle_model* m, const int32_t* ids, int B, int L, float* logits, float* hidden) {
    const int D = m->d_model, E = m->d_inner, R = m->dt_rank, XP = R + 2 * NST, S = 2 * B;
    const size_t rows = (size_t)S * L;
    /* the whole 2B-strand batch walks the stack layer by layer (as the HIP engine does); every operator is
     * OpenMP-parallel inside, so a small sample still uses all cores */
    const size_t nfl = rows * ((size_t)3 * D + 2 * E + E + XP + E + E + (m->ref_order ? D : 0)) + (size_t)E * NST;
    float* buf = (float*)malloc(sizeof(float) * nfl);
    int32_t* tok = (int32_t*)malloc(sizeof(int32_t) * rows);
    if (!buf || !tok) { free(buf); free(tok); return -1; }
    float* res = buf;
    float* u = res + rows * D;
    float* h = u + rows * D;
    float* xz = h + rows * D;
    float* xc = xz + rows * 2 * E;
    float* dbl = xc + rows * E;
    float* delta = dbl + rows * XP;
    float* y = delta + rows * E;
    float* A = y + rows * E;
    float* h2 = A + (size_t)E * NST;   /* ref_order only: out_proj of the reverse direction */

    for (int s = 0; s < S; ++s)
        for (int t = 0; t < L; ++t)
            tok[(size_t)s * L + t] = s < B ? (ids[(size_t)s * L + t] & 7)
                                           : m->complement[ids[(size_t)(s - B) * L + (L - 1 - t)] & 7];
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)rows; ++r)
        memcpy(h + (size_t)r * D, m->emb + (size_t)(tok[r] & 7) * D, sizeof(float) * D);

    const int rb = m->emulate_bf16;
    double tsum[T_N] = {0};
    const double t_begin = now_s();
    TIMED(T_ROUND, round_rows(h, rows * D, rb));
    for (int li = 0; li < m->n_layer; ++li) {
        const oracle_layer* ly = &m->layers[li];
        TIMED(T_NORM, add_rmsnorm(h, res, li > 0, ly->norm_w, u, (int64_t)rows, D, m->eps, rb));
        TIMED(T_INPROJ, linear_nt(u, D, ly->in_proj, D, xz, 2 * E, (int)rows, 2 * E));
        TIMED(T_ROUND, round_rows(xz, rows * 2 * E, rb));
        TIMED(T_MISC, memset(y, 0, sizeof(float) * rows * E));
        for (int d = 0; d < 2; ++d) {
            TIMED(T_CONV, conv_silu(xz, ly->conv_w[d], ly->conv_b[d], xc, S, L, E, d, rb));
            TIMED(T_XPROJ, linear_nt(xc, E, ly->x_proj[d], E, dbl, XP, (int)rows, XP));
            TIMED(T_ROUND, round_rows(dbl, rows * XP, rb));
            TIMED(T_DTPROJ, linear_nt(dbl, XP, ly->dt_w[d], R, delta, E, (int)rows, E));
            TIMED(T_ROUND, round_rows(delta, rows * E, rb));
#pragma omp parallel for schedule(static)
            for (int i = 0; i < E * NST; ++i) A[i] = -expf(ly->A_log[d][i]);
            TIMED(T_SCAN, scan_dir(xc, delta, dbl, xz, A, ly->dt_b[d], ly->Dskip[d], y, S, L, E, R, d, rb, !m->ref_order));
            if (m->ref_order) {      /* each Mamba call ends in its own (tied) out_proj, output stored in the model dtype */
                float* o = d == 0 ? h : h2;
                TIMED(T_OUTPROJ, linear_nt(y, E, ly->out_proj, E, o, D, (int)rows, D));
                TIMED(T_ROUND, round_rows(o, rows * D, rb));
            }
        }
        if (m->ref_order) {          /* BiMambaWrapper "add": out_fwd + out_rev */
            const double t0 = now_s();
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)(rows * D); ++i) h[i] += h2[i];
            tsum[T_MISC] += now_s() - t0;
        } else {
            TIMED(T_OUTPROJ, linear_nt(y, E, ly->out_proj, E, h, D, (int)rows, D));    /* tied out_proj folded, as the engine does */
        }
        TIMED(T_ROUND, round_rows(h, rows * D, rb));
    }
    float* Hall = u;   /* final normalised hidden [S, L, D] */
    TIMED(T_NORM, add_rmsnorm(h, res, 1, m->norm_f, Hall, (int64_t)rows, D, m->eps, rb));

    const double t_head = now_s();
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < L; ++l) {
            const float* hf = Hall + ((size_t)b * L + l) * D;
            const float* hr = Hall + ((size_t)(B + b) * L + (L - 1 - l)) * D;
            if (hidden) {
                float* o = hidden + ((size_t)b * L + l) * 2 * D;
                for (int c = 0; c < D; ++c) { o[c] = hf[c]; o[D + c] = hr[D - 1 - c]; }
            }
            if (logits) {
                for (int v = 0; v < 8; ++v) {
                    const float* e0 = m->emb + (size_t)v * D;
                    const float* e1 = m->emb + (size_t)m->complement[v] * D;
                    float a = 0.f, bsum = 0.f;
                    for (int c = 0; c < D; ++c) { a += hf[c] * e0[c]; bsum += hr[c] * e1[c]; }
                    logits[((size_t)b * L + l) * 8 + v] = rb ? rbf(rbf(a) + rbf(bsum)) : a + bsum;
                }
            }
        }
    tsum[T_HEAD] = now_s() - t_head;
    if (getenv("ORACLE_TIMING")) {
        fprintf(stderr, "oracle_forward B=%d L=%d: %.2f s |", B, L, now_s() - t_begin);
        for (int i = 0; i < T_N; ++i) fprintf(stderr, " %s %.2f", t_names[i], tsum[i]);
        fprintf(stderr, "\n");
    }
    free(buf);
    free(tok);
    return 0;
}
