/*
 * Host-side helper library (libpcad_host.so): evaluation of an XGBoost gbtree ensemble on a block of embedding rows, all host
 * cores.  Replaces, for the reference's `predict_XGBoost.py` (src/predict_XGBoost.py:28-31,60: `xgb.XGBClassifier.load_model` +
 * `predict_proba`), the XGBoost C++ predictor that is not available offline; the numpy walk in xgb_predict.py is the same
 * algorithm one tree at a time (21 s per 100 000 x 1 024 chunk with the reference's 1 000 trees of depth 6 - a quarter of the
 * GPU time of that chunk; this form: well under a second).
 *
 * Model layout = XGBoost's JSON arrays, concatenated over trees: node k of tree t is entry tree_off[t] + k of left / right
 * (child indices inside the tree, -1 = leaf), feat (split_indices), cond (split_conditions: threshold, or the leaf value in a
 * leaf), dleft (default_left).  Rule: x < threshold -> left; missing (NaN) -> default_left.  Margins are accumulated in double
 * in tree order, starting from base_margin - the same sequence of additions as the numpy form, so the two agree bit for bit.
 */
#include <math.h>
#include <stdint.h>

#define ROW_BLOCK 64

int pcad_host_version(void) { return 1; }

/* X [rows, ldx] fp32 row-major; out [rows] double.  Returns 0, or -1 on a malformed argument. */
int pcad_xgb_margin(const float* X, int64_t rows, int64_t ldx, int32_t n_features, int32_t n_trees, const int64_t* tree_off,
                    const int32_t* left, const int32_t* right, const int32_t* feat, const float* cond, const uint8_t* dleft,
                    double base_margin, double* out) {
    if (rows < 0 || n_trees < 0 || ldx < n_features || (rows > 0 && (!X || !out)) ||
        (n_trees > 0 && (!tree_off || !left || !right || !feat || !cond || !dleft)))
        return -1;
    /* cheap pre-pass: a split feature outside [0, n_features) of an internal node would read outside the row (the Python loader
     * validates trees too; this entry point must not trust its caller) */
    if (n_trees > 0) {
        const int64_t n_nodes = tree_off[n_trees];
        for (int64_t k = 0; k < n_nodes; ++k)
            if (left[k] != -1 && (feat[k] < 0 || feat[k] >= n_features)) return -1;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t r0 = 0; r0 < rows; r0 += ROW_BLOCK) {
        const int64_t r1 = r0 + ROW_BLOCK < rows ? r0 + ROW_BLOCK : rows;
        double acc[ROW_BLOCK];
        for (int64_t r = r0; r < r1; ++r) acc[r - r0] = base_margin;
        /* trees outermost inside a row block: a tree's nodes stay in L1 while the block's rows walk it */
        for (int32_t t = 0; t < n_trees; ++t) {
            const int64_t o = tree_off[t];
            const int32_t *L = left + o, *R = right + o, *F = feat + o;
            const float* Cn = cond + o;
            const uint8_t* Dl = dleft + o;
            for (int64_t r = r0; r < r1; ++r) {
                const float* x = X + r * ldx;
                int32_t n = 0;
                while (L[n] != -1) {
                    const float v = x[F[n]];
                    const int go_left = isnan(v) ? Dl[n] != 0 : v < Cn[n];
                    n = go_left ? L[n] : R[n];
                }
                acc[r - r0] += (double)Cn[n];
            }
        }
        for (int64_t r = r0; r < r1; ++r) out[r] = acc[r - r0];
    }
    return 0;
}
