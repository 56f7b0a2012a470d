/* sbayes_oracle_c.c -- plain-C restatement of ONE eval of the sBayes mixture log-likelihood  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Same role and rules as oracle/sbayes_oracle.py (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may
 * build, load or call it; the product never does): a second, independently written checker of SURVEY.md 8(d)'s expression
 *
 *     LL = sum_{n, f: not NA} log sum_c w[n, f, c] * p_c[g_c(n), f, x(n, f)]
 *
 * and the COMPILED single-thread CPU baseline of bench.py (`cpu_baseline_compiled`).  The real reference JIT-compiles
 * compute_component_likelihood (sbayes/model/likelihood.py:104, @njit) and dirichlet_categorical_logpdf (sbayes/util.py:1373)
 * when numba is installed and leaves the rest of the eval to NumPy; numba is not in this image, so no figure for that path can
 * be measured.  This file compiles the WHOLE eval (gcc -O3): an upper bound on what the numba-accelerated reference can
 * reach on one core for this expression.
 *
 * Restated reference lines (float32 rounding points kept):
 *   tables    p_c = float32((counts_c + prior_c) / sum_s(...))                      sbayes/sampling/conditionals.py:171-188, sbayes/util.py:990-1007
 *   gather    lh[n, f, c] = p_c[g_c(n), f, x(n, f)]; later groups overwrite earlier ones; objects in no group of c: 0; NA: 1
 *                                                                                   sbayes/model/likelihood.py:104-133, conditionals.py:219-221
 *   weights   w = float32(pattern * weights) / float32 sum over c, per has_components pattern      likelihood.py:171-190
 *   combine   log(sum_c float64(w) * lh) summed over the non-NA observations         sbayes/sampling/loggers.py:355-357
 * Pinned by tests/test_oracle_c_cpu.py: == the NumPy oracle (itself pinned on the reference's recorded outputs) at 1e-12 relative on
 * the cfg1 / south_america / test_files fixtures and the headline workload, and == the reference's recorded mixture_ll.
 * Sums here run in index order (NumPy sums pairwise): the tables can differ from NumPy's in the last float32 bit of a few
 * entries, which moves the summed log-likelihood by ~1e-13 relative -- far inside the tolerance, and the reason this file is a
 * checker at a tolerance, not a bit-exact twin.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* returns 0 on success; *out = LL.  Arrays are C-contiguous:
 *   features [N][F][S] bool (one byte each; all-zero row = NA)      groups[c] -> [G_c][N] bool
 *   counts[c] -> [G_c][F][S] float32                                 conc[c] -> [G_c][F][S] float64 (per group)
 *   weights [F][C] float32 */
int sbo_mixture_loglik(const uint8_t* features, int N, int F, int S, int C, const int32_t* n_groups, const uint8_t* const* groups,
                       const float* const* counts, const double* const* conc, const float* weights, double* out) {
    if (N < 1 || F < 1 || S < 1 || C < 1 || C > 8 || !features || !n_groups || !groups || !counts || !conc || !weights || !out) return 1;
    const size_t FS = (size_t)F * S;
    int rc = 0;
    float** probs = (float**)calloc((size_t)C, sizeof(float*));
    int32_t* gid = (int32_t*)malloc((size_t)C * N * sizeof(int32_t));       /* [C][N] group of the object, -1 = none */
    int16_t* state = (int16_t*)malloc((size_t)N * F * sizeof(int16_t));      /* [N][F] state index, -1 = NA */
    if (!probs || !gid || !state) { rc = 2; goto done; }
    /* tables: normalize(counts + prior) rounded to float32 */
    for (int c = 0; c < C; ++c) {
        const int G = n_groups[c];
        probs[c] = (float*)malloc(((size_t)G * FS + 1) * sizeof(float));
        if (!probs[c]) { rc = 2; goto done; }
        for (size_t gf = 0; gf < (size_t)G * F; ++gf) {
            const float* cn = counts[c] + gf * S;
            const double* a = conc[c] + gf * S;
            double tot = 0.0;
            for (int s = 0; s < S; ++s) tot += (double)cn[s] + a[s];
            if (!(tot > 0.0)) { rc = 3; goto done; }                          /* util.py:1006 asserts positive sums */
            for (int s = 0; s < S; ++s) probs[c][gf * S + s] = (float)(((double)cn[s] + a[s]) / tot);
        }
    }
    /* one id per object and component: the LAST group containing it (likelihood.py:126-130) */
    for (int c = 0; c < C; ++c)
        for (int n = 0; n < N; ++n) {
            int32_t id = -1;
            for (int g = 0; g < n_groups[c]; ++g) if (groups[c][(size_t)g * N + n]) id = g;
            gid[(size_t)c * N + n] = id;
        }
    for (size_t nf = 0; nf < (size_t)N * F; ++nf) {
        int16_t x = -1;
        for (int s = 0; s < S; ++s) if (features[nf * S + s]) { x = (int16_t)s; break; }
        state[nf] = x;
    }
    {
        /* normalize_weights once per has_components PATTERN (the reference does the same: np.unique over the rows, likelihood.py:183) */
        const int n_pat = 1 << C;
        float* wn = (float*)malloc((size_t)n_pat * F * C * sizeof(float));
        uint8_t* wn_done = (uint8_t*)calloc((size_t)n_pat, 1);
        if (!wn || !wn_done) { free(wn); free(wn_done); rc = 2; goto done; }
        double total = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* tab[8];
            int pat = 0;
            for (int c = 0; c < C; ++c) {
                const int32_t g = gid[(size_t)c * N + n];
                tab[c] = g >= 0 ? probs[c] + (size_t)g * FS : NULL;
                if (g >= 0) pat |= 1 << c;
            }
            float* w = wn + (size_t)pat * F * C;
            if (!wn_done[pat]) {
                for (int f = 0; f < F; ++f) {
                    float wsum = 0.0f;
                    for (int c = 0; c < C; ++c) { w[f * C + c] = (pat >> c & 1) ? weights[(size_t)f * C + c] : 0.0f; wsum += w[f * C + c]; }
                    for (int c = 0; c < C; ++c) w[f * C + c] = w[f * C + c] / wsum;          /* float32 division (likelihood.py:186-187) */
                }
                wn_done[pat] = 1;
            }
            double row = 0.0;
            for (int f = 0; f < F; ++f) {
                const int x = state[(size_t)n * F + f];
                if (x < 0) continue;                                        /* NA: likelihood 1, masked out of the sum */
                double v = 0.0;
                for (int c = 0; c < C; ++c)
                    if (tab[c]) v += (double)w[f * C + c] * (double)tab[c][(size_t)f * S + x];
                row += log(v);
            }
            total += row;
        }
        *out = total;
        free(wn); free(wn_done);
    }
done:
    if (probs) for (int c = 0; c < C; ++c) free(probs[c]);
    free(probs); free(gid); free(state);
    return rc;
}
