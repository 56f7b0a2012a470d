/* abi_smoke.c -- plain-C caller of the engine's C ABI (include/sbe_engine.h): proves the boundary is
 * usable without Python / NumPy / torch.  Reads a small binary case written by the pytest wrapper
 * (tests/test_gpu_c_abi.py), evaluates it through the ABI and prints the results as text.
 *   gcc -O2 -I include tests/c/abi_smoke.c -o abi_smoke -ldl      (the .so is dlopen'ed)             */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sbe_engine.h"

#define LOAD(name) name##_t p_##name = (name##_t)dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }
typedef int (*sbe_create_t)(sbe_engine**, int, int, int, int, int, const int32_t*, int, const uint8_t*);
typedef int (*sbe_destroy_t)(sbe_engine*);
typedef const char* (*sbe_last_error_t)(const sbe_engine*);
typedef int (*sbe_set_groups_t)(sbe_engine*, int, int, const uint8_t*);
typedef int (*sbe_set_source_t)(sbe_engine*, int, const uint8_t*);
typedef int (*sbe_recount_t)(sbe_engine*, int, int);
typedef int (*sbe_set_concentration_t)(sbe_engine*, int, const double*, int);
typedef int (*sbe_update_probs_t)(sbe_engine*, int, int, double, double, const double*);
typedef int (*sbe_set_weights_t)(sbe_engine*, int, const float*);
typedef int (*sbe_mixture_loglik_t)(sbe_engine*, int, double*);
typedef int (*sbe_collapsed_loglik_t)(sbe_engine*, int, int, double*, float*);
typedef int (*sbe_get_info_t)(const sbe_engine*, sbe_info*);
typedef int (*sbe_host_group_ids_t)(const uint8_t*, int, int64_t, const int32_t*, int, int, int32_t*);
typedef int (*sbe_host_source_ids_t)(const uint8_t*, int64_t, int, int, const int32_t*, int, uint8_t*);
typedef int (*sbe_host_touched_groups_t)(const int32_t*, const int32_t*, int64_t, int, int32_t*, int32_t*);
typedef int (*sbe_gibbs_propose_supported_t)(sbe_engine*);
typedef int (*sbe_gibbs_propose_t)(sbe_engine*, int, int, const int32_t*, int, double, double, int, const double*, uint8_t*, float*, float*,
                                   int32_t*, int32_t*, float*);
typedef int (*sbe_test_roundtrip_t)(sbe_engine*, int, int);
typedef int (*sbe_collapsed_and_source_prior_t)(sbe_engine*, int, double*, double*);

static void* slurp(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s libsbe_engine.so case.bin\n", argv[0]); return 1; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(sbe_create) LOAD(sbe_destroy) LOAD(sbe_last_error) LOAD(sbe_set_groups) LOAD(sbe_set_source) LOAD(sbe_recount)
    LOAD(sbe_set_concentration) LOAD(sbe_update_probs) LOAD(sbe_set_weights) LOAD(sbe_mixture_loglik)
    LOAD(sbe_collapsed_loglik) LOAD(sbe_get_info)
    LOAD(sbe_host_group_ids) LOAD(sbe_host_source_ids) LOAD(sbe_host_touched_groups)
    LOAD(sbe_gibbs_propose_supported) LOAD(sbe_gibbs_propose) LOAD(sbe_test_roundtrip) LOAD(sbe_collapsed_and_source_prior)
    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror("case"); return 1; }
    int32_t hdr[4];                                   /* N, F, S, C */
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 3;
    const int N = hdr[0], F = hdr[1], S = hdr[2], C = hdr[3];
    int32_t* G = (int32_t*)slurp(f, (size_t)C * sizeof(int32_t));
    uint8_t* feats = (uint8_t*)slurp(f, (size_t)N * F * S);
    sbe_engine* e = NULL;
    if (p_sbe_create(&e, 0, N, F, S, C, G, 2, feats)) { fprintf(stderr, "create: %s\n", p_sbe_last_error(NULL)); return 4; }
    uint8_t* clusters = NULL;
    for (int c = 0; c < C; ++c) {
        uint8_t* groups = (uint8_t*)slurp(f, (size_t)G[c] * N);
        double* conc = (double*)slurp(f, (size_t)G[c] * F * S * sizeof(double));
        if (p_sbe_set_groups(e, 0, c, groups) || p_sbe_set_concentration(e, c, conc, 1)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 5; }
        if (c == 0) clusters = groups; else free(groups);
        free(conc);
    }
    uint8_t* source = (uint8_t*)slurp(f, (size_t)N * F * C);
    float* weights = (float*)slurp(f, (size_t)F * C * sizeof(float));
    if (p_sbe_set_source(e, 0, source) || p_sbe_recount(e, 0, -1)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 6; }
    for (int c = 0; c < C; ++c)
        if (p_sbe_update_probs(e, 0, c, 0.0, 0.0, NULL)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 7; }
    if (p_sbe_set_weights(e, 0, weights)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 8; }
    double ll = 0.0, collapsed = 0.0;
    if (p_sbe_mixture_loglik(e, 0, &ll)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 9; }
    for (int c = 0; c < C; ++c) {
        double* pg = (double*)malloc((size_t)G[c] * sizeof(double));
        if (p_sbe_collapsed_loglik(e, 0, c, pg, NULL)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 10; }
        for (int g = 0; g < G[c]; ++g) collapsed += pg[g];
        free(pg);
    }
    sbe_info info;
    p_sbe_get_info(e, &info);
    printf("mixture_ll %.17g\ncollapsed_ll %.17g\nn_na %lld\n", ll, collapsed, (long long)info.n_na);
    /* error behaviour: a bad slot is reported through the return code + message, not a crash */
    if (p_sbe_mixture_loglik(e, 7, &ll) == 0) { fprintf(stderr, "bad slot accepted\n"); return 11; }
    printf("error_text %s\n", p_sbe_last_error(e));
    /* ABI 6: a group matrix with an object in two rows is TAKEN (the last group is the object's id: what an uncached likelihood
       evaluation ends up with, likelihood.py:126-130) and the slot is marked; a call that would derive COUNTS from one id per object
       -- sbe_recount -- refuses the marked slot with SBE_ERR_DATA naming object, groups, component (needs two clusters).  The original
       matrix clears the mark and the slot evaluates as before. */
    if (G[0] >= 2) {
        uint8_t* bad = (uint8_t*)malloc((size_t)G[0] * N);
        memcpy(bad, clusters, (size_t)G[0] * N);
        int n_first = -1;
        for (int n = 0; n < N && n_first < 0; ++n) if (bad[n]) n_first = n;      /* a member of cluster 0 ... */
        if (n_first >= 0) bad[(size_t)N + n_first] = 1;                            /* ... also put into cluster 1 */
        const int rc_set = p_sbe_set_groups(e, 0, 0, bad);
        const int rc = p_sbe_recount(e, 0, -1);
        printf("overlap_set_rc %d\noverlap_rc %d\noverlap_text %s\n", rc_set, rc, p_sbe_last_error(e));
        double again = 0.0;
        if (p_sbe_set_groups(e, 0, 0, clusters) || p_sbe_recount(e, 0, -1)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 12; }
        for (int c = 0; c < C; ++c) if (p_sbe_update_probs(e, 0, c, 0.0, 0.0, NULL)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 12; }
        if (p_sbe_mixture_loglik(e, 0, &again) || again != ll) { fprintf(stderr, "state not restored by the original matrix\n"); return 12; }
        free(bad);
    }
    /* round 4: the host helpers of the marshalling (no device): ids of the first 5 objects */
    {
        int32_t objs[5] = {0, 1, 2, 3, 4}, gids[5], gids2[5], touched[64], n_touched = 0;
        uint8_t* sids = (uint8_t*)malloc((size_t)5 * F);
        const int n = N < 5 ? N : 5;
        if (p_sbe_host_group_ids(clusters, G[0], N, objs, n, 0, gids) != 0) return 13;
        if (p_sbe_host_source_ids(source, N, F, C, objs, n, sids) != 0) return 14;
        for (int i = 0; i < n; ++i) gids2[i] = gids[i] < 0 ? -1 : (gids[i] + 1) % G[0];
        if (p_sbe_host_touched_groups(gids, gids2, n, G[0], touched, &n_touched) != 0) return 15;
        printf("host_gids");
        for (int i = 0; i < n; ++i) printf(" %d", gids[i]);
        printf("\nhost_sids");
        for (int i = 0; i < n; ++i) printf(" %d", (int)sids[(size_t)i * F]);
        printf("\nhost_touched %d\n", n_touched);
        free(sids);
    }
    /* round 4, second session: GibbsSampleSource._propose in ONE call (sbe_gibbs_propose), every uniform 0.5: the first
       objects' source redrawn into slot 1; drawn ids, the groups touched and the sum of the count rows (a redraw moves
       counts between rows: they sum to zero) */
    if (p_sbe_test_roundtrip(e, 4, 3) != 0) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 16; }
    {   /* Model.__call__ = likelihood + prior in one call: sum of the per-group values, sum of the per-object source prior */
        int g_total = 0;
        for (int c = 0; c < C; ++c) g_total += G[c];
        double* per_group = (double*)malloc((size_t)g_total * sizeof(double));
        double* per_object = (double*)malloc((size_t)N * sizeof(double));
        if (p_sbe_collapsed_and_source_prior(e, 0, per_group, per_object)) { fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 18; }
        double sum_g = 0.0, sum_o = 0.0;
        for (int g = 0; g < g_total; ++g) sum_g += per_group[g];
        for (int n = 0; n < N; ++n) sum_o += per_object[n];
        printf("fused_collapsed_ll %.17g\nfused_source_prior %.17g\n", sum_g, sum_o);
        free(per_group); free(per_object);
    }
    if (p_sbe_gibbs_propose_supported(e) == 1) {
        const int n = N < 5 ? N : 5;
        int32_t objs[5] = {0, 1, 2, 3, 4}, touched[256], n_touched = 0;
        double* z = (double*)malloc((size_t)n * F * sizeof(double));
        for (int i = 0; i < n * F; ++i) z[i] = 0.5;
        uint8_t* ids = (uint8_t*)malloc((size_t)n * F);
        float* sel = (float*)malloc((size_t)n * F * sizeof(float));
        float* back = (float*)malloc((size_t)n * F * sizeof(float));
        float* rows = (float*)malloc((size_t)256 * F * S * sizeof(float));
        if (p_sbe_gibbs_propose(e, 0, 1, objs, n, 1.0, 1.0, 0, z, ids, sel, back, touched, &n_touched, rows)) {
            fprintf(stderr, "%s\n", p_sbe_last_error(e)); return 17;
        }
        double row_sum = 0.0;
        for (long i = 0; i < (long)n_touched * F * S; ++i) row_sum += rows[i];
        printf("propose_ids");
        for (int i = 0; i < n * F && i < 24; ++i) printf(" %d", (int)ids[i]);
        printf("\npropose_touched %d\npropose_row_sum %.1f\npropose_sel0 %.9g\n", n_touched, row_sum, (double)sel[0]);
        free(z); free(ids); free(sel); free(back); free(rows);
    }
    free(clusters);
    p_sbe_destroy(e);
    fclose(f);
    return 0;
}
