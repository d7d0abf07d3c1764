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
 * Parallelisation: one strand per OpenMP thread (strands are independent); inside a strand plain
 * blocked loops the compiler vectorises.  No BLAS, no intrinsics: this is a scalar-source port.
 */
#include <math.h>
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
    int32_t complement[8];
    const float* emb;     /* [8, D] (tied LM head) */
    const float* norm_f;  /* [D] */
    const oracle_layer* layers;
} oracle_model;

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* expf with a Cephes-style polynomial so that the channel loop vectorises (rel. error ~1e-7) */
static inline float vexpf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float fx = floorf(x * 1.44269504088896341f + 0.5f);
    const float r = (x - fx * 0.693359375f) - fx * -2.12194440e-4f;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    p = p * r * r + r + 1.0f;
    union { float f; int32_t i; } u;
    u.i = ((int32_t)fx + 127) << 23;
    return p * u.f;
}

/* logf for normal positive x, Cephes polynomial (vectorisable; rel. error ~1e-7) */
static inline float vlogf(float x) {
    union { float f; int32_t i; } u;
    u.f = x;
    int32_t e = ((u.i >> 23) & 255) - 126;
    u.i = (u.i & 0x007fffff) | 0x3f000000;   /* mantissa in [0.5, 1) */
    float m = u.f;
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
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

/* C[M,N] = A[M,K] . W[N,K]^T  (F.linear without bias).  4x4 register block of VL-wide partial sums. */
static void linear_nt(const float* A, int lda, const float* W, int K, float* C, int ldc, int M, int N) {
    const int Kv = K - K % VL;
    for (int i0 = 0; i0 < M; i0 += 4) {
        const int mi = M - i0 < 4 ? M - i0 : 4;
        for (int j0 = 0; j0 < N; j0 += 4) {
            const int nj = N - j0 < 4 ? N - j0 : 4;
            if (mi == 4 && nj == 4) {
                float acc[4][4][VL] = {{{0.f}}};
                const float* a0 = A + (size_t)i0 * lda;
                const float* w0 = W + (size_t)j0 * K;
                for (int k = 0; k < Kv; k += VL)
                    for (int i = 0; i < 4; ++i)
                        for (int j = 0; j < 4; ++j)
#pragma omp simd
                            for (int l = 0; l < VL; ++l)
                                acc[i][j][l] += a0[(size_t)i * lda + k + l] * w0[(size_t)j * K + k + l];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        float s = 0.f;
                        for (int l = 0; l < VL; ++l) s += acc[i][j][l];
                        for (int kk = Kv; kk < K; ++kk) s += a0[(size_t)i * lda + kk] * w0[(size_t)j * K + kk];
                        C[(size_t)(i0 + i) * ldc + j0 + j] = s;
                    }
            } else {
                for (int i = 0; i < mi; ++i)
                    for (int j = 0; j < nj; ++j) {
                        const float* a = A + (size_t)(i0 + i) * lda;
                        const float* w = W + (size_t)(j0 + j) * K;
                        float part[VL] = {0.f};
                        for (int k = 0; k < Kv; k += VL)
                            for (int l = 0; l < VL; ++l) part[l] += a[k + l] * w[k + l];
                        float s = 0.f;
                        for (int l = 0; l < VL; ++l) s += part[l];
                        for (int kk = Kv; kk < K; ++kk) s += a[kk] * w[kk];
                        C[(size_t)(i0 + i) * ldc + j0 + j] = s;
                    }
            }
        }
    }
}

/* rms_norm_fn(prenorm=True): res = x (+ res); y = res * rsqrt(mean(res^2)+eps) * w */
static void add_rmsnorm(const float* x, float* res, int has_res, const float* w, float* y, int L, int D, float eps) {
    for (int t = 0; t < L; ++t) {
        float ss = 0.f;
        float* r = res + (size_t)t * D;
        const float* xr = x + (size_t)t * D;
        for (int c = 0; c < D; ++c) {
            const float v = has_res ? xr[c] + r[c] : xr[c];
            r[c] = v;
            ss += v * v;
        }
        const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
        for (int c = 0; c < D; ++c) y[(size_t)t * D + c] = r[c] * rstd * w[c];
    }
}

/* one strand through the whole stack; H out: final normalised hidden [L, D] */
static void strand_forward(const oracle_model* m, const int32_t* tok /*[L]*/, int L, float* H, float* scratch) {
    const int D = m->d_model, E = m->d_inner, R = m->dt_rank, XP = R + 2 * NST;
    float* res = scratch;                         /* [L, D] */
    float* u = res + (size_t)L * D;               /* [L, D] */
    float* h = u + (size_t)L * D;                 /* [L, D] */
    float* xz = h + (size_t)L * D;                /* [L, 2E] */
    float* xc = xz + (size_t)L * 2 * E;           /* [L, E] */
    float* dbl = xc + (size_t)L * E;              /* [L, XP] */
    float* delta = dbl + (size_t)L * XP;          /* [L, E] */
    float* y = delta + (size_t)L * E;             /* [L, E] */
    float* A = y + (size_t)L * E;                 /* [E, N] */

    for (int t = 0; t < L; ++t) memcpy(h + (size_t)t * D, m->emb + (size_t)(tok[t] & 7) * D, sizeof(float) * D);
    for (int li = 0; li < m->n_layer; ++li) {
        const oracle_layer* ly = &m->layers[li];
        add_rmsnorm(h, res, li > 0, ly->norm_w, u, L, D, m->eps);
        linear_nt(u, D, ly->in_proj, D, xz, 2 * E, L, 2 * E);
        memset(y, 0, sizeof(float) * (size_t)L * E);
        for (int d = 0; d < 2; ++d) {
            /* conv1d (width 4) + SiLU; direction 1 = anti-causal (the reverse Mamba on unflipped rows) */
            const float* cw = ly->conv_w[d];
            const float* cb = ly->conv_b[d];
            for (int t = 0; t < L; ++t)
                for (int c = 0; c < E; ++c) {
                    float a = cb[c];
                    for (int k = 0; k < 4; ++k) {
                        const int tt = d == 0 ? t - 3 + k : t + 3 - k;
                        if (tt >= 0 && tt < L) a += cw[c * 4 + k] * xz[(size_t)tt * 2 * E + c];
                    }
                    xc[(size_t)t * E + c] = silu_f(a);
                }
            linear_nt(xc, E, ly->x_proj[d], E, dbl, XP, L, XP);
            linear_nt(dbl, XP, ly->dt_w[d], R, delta, E, L, E);
            for (int i = 0; i < E * NST; ++i) A[i] = -expf(ly->A_log[d][i]);
            /* selective scan, channel blocks of VL lanes, sequential in t */
            for (int c0 = 0; c0 < E; c0 += VL) {
                float st[NST][VL], ab[NST][VL];
                memset(st, 0, sizeof(st));
                for (int n = 0; n < NST; ++n)
                    for (int l = 0; l < VL; ++l) ab[n][l] = A[(size_t)(c0 + l) * NST + n];
                for (int step = 0; step < L; ++step) {
                    const int t = d == 0 ? step : L - 1 - step;
                    const float* Bt = dbl + (size_t)t * XP + R;
                    const float* Ct = Bt + NST;
                    float dv[VL], du[VL], yv[VL];
                    for (int l = 0; l < VL; ++l) {
                        const int c = c0 + l;
                        dv[l] = softplus_f(delta[(size_t)t * E + c] + ly->dt_b[d][c]);
                        const float uv = xc[(size_t)t * E + c];
                        du[l] = dv[l] * uv;
                        yv[l] = ly->Dskip[d][c] * uv;
                    }
                    for (int n = 0; n < NST; ++n) {
                        const float bn = Bt[n], cn = Ct[n];
                        for (int l = 0; l < VL; ++l) {
                            const float a = vexpf(dv[l] * ab[n][l]);
                            st[n][l] = a * st[n][l] + du[l] * bn;
                            yv[l] += st[n][l] * cn;
                        }
                    }
                    for (int l = 0; l < VL; ++l) {
                        const int c = c0 + l;
                        y[(size_t)t * E + c] += yv[l] * silu_f(xz[(size_t)t * 2 * E + E + c]);
                    }
                }
            }
        }
        linear_nt(y, E, ly->out_proj, E, h, D, L, D);
    }
    add_rmsnorm(h, res, 1, m->norm_f, H, L, D, m->eps);
}

size_t oracle_scratch_floats(const oracle_model* m, int L) {
    const size_t D = m->d_model, E = m->d_inner, XP = m->dt_rank + 2 * NST;
    return (size_t)L * (3 * D + 2 * E + E + XP + E + E) + E * NST + 64;
}

/*
 * ids [B, L] int32 -> logits [B, L, 8] (may be NULL), hidden [B, L, 2D] (may be NULL).
 * hidden[b,l] = cat(Hf[b,l], reverse_channels(Hr[b,L-1-l])); logits = Hf.Emb^T + Hr[L-1-l].Emb[comp]^T.
 * returns 0, or -1 on allocation failure.
 */
int oracle_forward(const oracle_model* m, const int32_t* ids, int B, int L, float* logits, float* hidden) {
    const int D = m->d_model, S = 2 * B;
    float* Hall = (float*)malloc(sizeof(float) * (size_t)S * L * D);
    if (!Hall) return -1;
    int err = 0;
#pragma omp parallel
    {
        float* scratch = (float*)malloc(sizeof(float) * oracle_scratch_floats(m, L));
        int32_t* tok = (int32_t*)malloc(sizeof(int32_t) * (size_t)L);
        if (!scratch || !tok) {
#pragma omp atomic write
            err = -1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int s = 0; s < S; ++s) {
                if (s < B) {
                    for (int t = 0; t < L; ++t) tok[t] = ids[(size_t)s * L + t] & 7;
                } else {
                    for (int t = 0; t < L; ++t) tok[t] = m->complement[ids[(size_t)(s - B) * L + (L - 1 - t)] & 7];
                }
                strand_forward(m, tok, L, Hall + (size_t)s * L * D, scratch);
            }
        }
        free(scratch);
        free(tok);
    }
    if (err) { free(Hall); return err; }
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
                    logits[((size_t)b * L + l) * 8 + v] = a + bsum;
                }
            }
        }
    free(Hall);
    return 0;
}
