// sbe_engine_stateless.hip -- unit 2 of 4: the reference's free functions as stateless calls (normalize, Dirichlet-categorical
// log-pdf, effect counts, normalize_weights), the first forms of the operator kernels that take their tables from the caller,
// and the source-posterior family (posterior, draw, transition log-probability).
#include "sbe_engine_internal.hip.h"

extern "C" {

// =============================================================================================
// Stateless entry points: the reference's free functions, one call each, no slot involved.
// They run the same kernels on scratch buffers.
// =============================================================================================
int sbe_normalize_tables(sbe_engine* e, const float* counts, int n_groups, const double* conc, int conc_per_group,
                         double temperature, double prior_temperature, const double* unif_counts, float* out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;              // empty table set (a component without groups)
    CHECK_PTR(e, counts); CHECK_PTR(e, conc); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (prior_temperature > 0.0 && !unif_counts) return fail(e, SBE_ERR_ARG, "prior_temperature given without unif_counts (conditionals.py:114)");
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S, n = (int64_t)n_groups * fs;
    const size_t cb = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    const size_t ab = ((size_t)(conc_per_group ? n : fs) * sizeof(double) + 255) / 256 * 256;
    const size_t ub = ((size_t)fs * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * cb + ab + ub);
    if (rc) return rc;
    float* d_cnt = (float*)e->d_scratch;
    double* d_a = (double*)(e->d_scratch + cb);
    double* d_u = (double*)(e->d_scratch + cb + ab);
    float* d_out = (float*)(e->d_scratch + cb + ab + ub);
    { int _urc = upload(e, d_cnt, counts, (size_t)n * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_a, conc, (size_t)(conc_per_group ? n : fs) * sizeof(double)); if (_urc) return _urc; }
    const double* d_unif = nullptr;
    if (prior_temperature > 0.0) {
        { int _urc = upload(e, d_u, unif_counts, (size_t)fs * sizeof(double)); if (_urc) return _urc; }
        d_unif = d_u;
    }
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_probs<float><<<div_up((int64_t)n_groups * e->F, 256), 256, 0, e->stream>>>(
        d_cnt, d_a, d_unif, d_out, 0, n_groups, e->F, e->S, temperature, prior_temperature, conc_per_group ? 1 : 0, e->d_status);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, d_out, (size_t)n * sizeof(float));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE])
        return fail(e, SBE_ERR_DATA, "normalize: %d rows have a non-positive sum (sbayes/util.py:1006 assert)", e->h_status[ST_BAD_NORMALIZE]);
    return SBE_OK;
}

int sbe_dirichlet_logpdf(sbe_engine* e, const float* counts, int n_groups, const double* conc, int conc_per_group,
                         float* per_feature_out, double* per_group_out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;
    CHECK_PTR(e, counts); CHECK_PTR(e, conc);
    if (!per_feature_out && !per_group_out) return fail(e, SBE_ERR_ARG, "no output requested");
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S, n = (int64_t)n_groups * fs;
    const size_t cb = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    const size_t ab = ((size_t)(conc_per_group ? n : fs) * sizeof(double) + 255) / 256 * 256;
    const size_t pf = ((size_t)n_groups * e->F * sizeof(float) + 255) / 256 * 256;
    int rc = ensure_scratch(e, cb + ab + pf + (size_t)n_groups * sizeof(double));
    if (rc) return rc;
    float* d_cnt = (float*)e->d_scratch;
    double* d_a = (double*)(e->d_scratch + cb);
    float* d_pf = (float*)(e->d_scratch + cb + ab);
    double* d_pg = (double*)(e->d_scratch + cb + ab + pf);
    { int _urc = upload(e, d_cnt, counts, (size_t)n * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_a, conc, (size_t)(conc_per_group ? n : fs) * sizeof(double)); if (_urc) return _urc; }
    k_dcl<float><<<div_up((int64_t)n_groups * e->F, 256), 256, 0, e->stream>>>(d_cnt, d_a, d_pf, 0, n_groups, e->F, e->S, conc_per_group ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    if (per_group_out) {
        k_group_sum_f32<<<div_up((int64_t)n_groups * 8, 64), 64, 0, e->stream>>>(d_pf, d_pg, n_groups, e->F);
        HIPCHK(e, hipGetLastError());
        rc = d2h(e, per_group_out, d_pg, (size_t)n_groups * sizeof(double));
        if (rc) return rc;
    }
    if (per_feature_out) return d2h(e, per_feature_out, d_pf, (size_t)n_groups * e->F * sizeof(float));
    return SBE_OK;
}

int sbe_effect_counts(sbe_engine* e, const uint8_t* groups, int n_groups, const uint8_t* source_is_component,
                      const int32_t* objects, int n_subset, float* out) {
    CHECK_ENGINE(e);
    if (n_groups == 0) return SBE_OK;              // compute_effect_counts of an empty group matrix: empty counts
    CHECK_PTR(e, groups); CHECK_PTR(e, source_is_component); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (n_subset < -1 || (n_subset > 0 && !objects)) return fail(e, SBE_ERR_ARG, "bad object subset");
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, F = e->F, S = e->S;
    const int64_t n_out = (int64_t)n_groups * F * S;
    const int n_listed = n_subset < 0 ? N : n_subset;
    const size_t gb = ((size_t)n_groups * N + 255) / 256 * 256;
    const size_t mb = ((size_t)N * F + 255) / 256 * 256;
    const size_t ob = ((size_t)std::max(n_listed, 1) * sizeof(int32_t) + 255) / 256 * 256;
    const size_t cb = ((size_t)n_out * sizeof(int32_t) + 255) / 256 * 256;
    int rc = ensure_scratch(e, gb + mb + ob + 2 * cb);
    if (rc) return rc;
    uint8_t* d_groups = e->d_scratch;
    uint8_t* d_mask = e->d_scratch + gb;
    int32_t* d_obj = (int32_t*)(e->d_scratch + gb + mb);
    int32_t* d_cnt = (int32_t*)(e->d_scratch + gb + mb + ob);
    float* d_out = (float*)(e->d_scratch + gb + mb + ob + cb);
    { int _urc = upload(e, d_groups, groups, (size_t)n_groups * N); if (_urc) return _urc; }
    { int _urc = upload(e, d_mask, source_is_component, (size_t)N * F); if (_urc) return _urc; }
    if (n_subset > 0) { int _urc = upload(e, d_obj, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
    HIPCHK(e, hipMemsetAsync(d_cnt, 0, (size_t)n_out * sizeof(int32_t), e->stream));
    if (n_listed > 0) {
        int ft = 32;
        auto lds_for = [&](int t) { return (size_t)n_groups * t * S * sizeof(int32_t); };
        while (ft > 1 && lds_for(ft) > 64 * 1024) ft >>= 1;
        if (lds_for(ft) > 150 * 1024) return fail(e, SBE_ERR_ARG, "effect counts: %d groups x %d states exceed the LDS histogram", n_groups, S);
        const int n_ftiles = div_up(F, ft);
        int chunks = std::max(1, std::min(div_up(n_listed, 32), div_up(2 * e->compute_units, n_ftiles)));
        const int opc = div_up(n_listed, chunks);
        chunks = div_up(n_listed, opc);
        k_effect_counts<<<dim3(n_ftiles, chunks), kBlock, lds_for(ft), e->stream>>>(
            e->d_state, d_groups, d_mask, n_subset >= 0 ? d_obj : nullptr, n_listed, opc, N, F, S, e->Fp, n_groups, ft, d_cnt);
        HIPCHK(e, hipGetLastError());
    }
    k_i32_to_f32<<<div_up(n_out, 256), 256, 0, e->stream>>>(d_cnt, d_out, n_out);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n_out * sizeof(float));
}

int sbe_normalize_weights(sbe_engine* e, const float* weights, int n_comp, const uint8_t* has_components, int n_rows,
                          float* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, weights);
    if (n_comp < 1 || n_comp > kMaxComponents) return fail(e, SBE_ERR_ARG, "n_comp=%d unsupported (1..%d)", n_comp, kMaxComponents);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, has_components); CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = n_rows, F = e->F, C = n_comp;     // the rows are whatever the caller hands over (has_components[available], operators.py:1086)
    // row form (k_normalize_weight_rows): no pattern sort on either side, one launch; a few rows read the weights out of
    // the mapped staging ring, many rows out of device memory (every block reads them once)
    const size_t wb = al256((size_t)F * C * sizeof(float)), hb = al256((size_t)N * C);
    const int64_t n_out = (int64_t)N * F * C;
    int rc = ensure_scratch(e, wb + hb + (size_t)n_out * sizeof(float));
    if (rc) return rc;
    const void *v_w = e->d_scratch, *v_hc;
    const size_t w_lds = (size_t)F * C * sizeof(float);
    const bool in_lds = w_lds <= ((size_t)64 << 10);              // (beyond that the blocks read the device copy in place)
    if (N <= 256 && in_lds) rc = stage(e, weights, w_lds, e->d_scratch, &v_w);
    else rc = upload(e, e->d_scratch, weights, w_lds);
    if (rc) return rc;
    rc = stage(e, has_components, (size_t)N * C, e->d_scratch + wb, &v_hc);
    if (rc) return rc;
    void* d_out;
    rc = out_target(e, (size_t)n_out * sizeof(float), e->d_scratch + wb + hb, &d_out);
    if (rc) return rc;
    const DoneSig done = out_done(e, d_out, (unsigned)div_up(N, kNwRows));
    k_normalize_weight_rows<<<div_up(N, kNwRows), 256, in_lds ? w_lds : 0, e->stream>>>(
        (const float*)v_w, (const uint8_t*)v_hc, (float*)d_out, N, F, C, in_lds ? 1 : 0, done);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, (size_t)n_out * sizeof(float), done);
}

// ---- cluster-membership marginals ---------------------------------------------
int sbe_cluster_marginals(sbe_engine* e, int slot, const float* table, const int32_t* objects, int n_objects_av,
                          double prior_temperature, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, table); CHECK_PTR(e, out);
    if (n_objects_av < 0) return fail(e, SBE_ERR_ARG, "n_objects_av=%d", n_objects_av);
    if (n_objects_av == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "prior_temperature must be positive");
    for (int i = 0; i < n_objects_av; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    Slot& s = e->slots[slot];
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int F = e->F, S = e->S, C = e->C;
    const size_t tb = ((size_t)F * S * sizeof(float) + 255) / 256 * 256;
    const size_t ob = ((size_t)n_objects_av * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)2 * n_objects_av * sizeof(double);
    // table, object list and the result live in host-mapped memory (a few KB each): ONE kernel and one
    // synchronisation, no copy-engine operation in the chain
    rc = ensure_io(e, tb + ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, table, (size_t)F * S * sizeof(float));
    memcpy(e->h_io + tb, objects, (size_t)n_objects_av * sizeof(int32_t));
    const float* d_tab = (const float*)e->d_io;
    const int32_t* d_obj = (const int32_t*)(e->d_io + tb);
    double* d_out = (double*)(e->d_io + tb + ob);
    const double inv = 1.0 / prior_temperature;
    const DoneSig done = next_done(e, (unsigned)n_objects_av);
    k_cluster_marginals<<<n_objects_av, kBlock, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        e->d_probs + (int64_t)slot * e->table_elems(), d_tab, e->d_weights + (int64_t)slot * F * C,
        e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, d_obj, n_objects_av, d_out,
        reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + tb + ob, out_bytes);
    return synced(e);
}

// ---- ClusterJump.get_jump_lh / expected_confounder_features (operators.py:1679-1722, 1342-1379) -----------------
int sbe_jump_lh(sbe_engine* e, int slot, const float* pconf, const float* p_source, const float* p_target,
                const int32_t* objects, int n_members, double prior_temperature, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, p_source); CHECK_PTR(e, p_target); CHECK_PTR(e, out);
    if (n_members < 0) return fail(e, SBE_ERR_ARG, "n_members=%d", n_members);
    if (n_members == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "prior_temperature must be positive");
    for (int i = 0; i < n_members; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / weights not set", slot);
    const int n_conf_groups = e->Gtot - e->G[0];
    if (n_conf_groups > 0) CHECK_PTR(e, pconf);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int F = e->F, S = e->S, C = e->C;
    const size_t fs = (size_t)F * S * sizeof(float);
    const size_t cb = ((size_t)std::max(n_conf_groups, 1) * fs + 255) / 256 * 256;
    const size_t tb = (fs + 255) / 256 * 256;
    const size_t ob = ((size_t)n_members * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)2 * n_members * sizeof(double);
    // tables, member list and result in host-mapped memory: ONE kernel, one synchronisation (as sbe_cluster_marginals)
    rc = ensure_io(e, cb + 2 * tb + ob + out_bytes);
    if (rc) return rc;
    if (n_conf_groups > 0) memcpy(e->h_io, pconf, (size_t)n_conf_groups * fs);
    memcpy(e->h_io + cb, p_source, fs);
    memcpy(e->h_io + cb + tb, p_target, fs);
    memcpy(e->h_io + cb + 2 * tb, objects, (size_t)n_members * sizeof(int32_t));
    const double inv = 1.0 / prior_temperature;
    const DoneSig done = next_done(e, (unsigned)n_members);
    k_jump_lh<<<n_members, kBlock, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        (const float*)e->d_io, (const float*)(e->d_io + cb), (const float*)(e->d_io + cb + tb),
        e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
        (const int32_t*)(e->d_io + cb + 2 * tb), n_members, (double*)(e->d_io + cb + 2 * tb + ob),
        reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, e->G[0], done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + cb + 2 * tb + ob, out_bytes);
    return synced(e);
}

// ---- GibbsSampleWeights.source_lh_by_feature (operators.py:677-685) -----------------------------------------------
int sbe_source_lh_by_feature(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_io(e, (size_t)e->F * sizeof(float));
    if (rc) return rc;
    const DoneSig done = next_done(e, (unsigned)div_up(e->F, kSlfFT));
    k_source_lh_by_feature<<<div_up(e->F, kSlfFT), 1024, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
        e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (float*)e->d_io, e->N, e->F, e->C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io, (size_t)e->F * sizeof(float));
    return synced(e);
}

// ---- data-parallel cores of Gibbs source resampling ---------------------------
namespace {

// Shared front end of the source-posterior family: argument checks, object upload, kernel arguments.
// Scratch layout: [objects | extra bytes requested by the caller].
int source_posterior_setup(sbe_engine* e, int slot, const int32_t* objects, int n_sub, double temperature,
                           double prior_temperature, int from_prior, size_t extra_bytes, SrcPostArgs* a,
                           uint8_t** d_extra) {
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->slots[slot].patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    rc = ensure_scratch(e, ob + extra_bytes);
    if (rc) return rc;
    *d_extra = e->d_scratch + ob;
    // the object list is read once per thread: out of the host-mapped staging ring in place (no copy operation in front of
    // the kernel); lists beyond the ring's direct size go up with a copy
    const void* v_obj;
    rc = stage(e, objects, (size_t)n_sub * sizeof(int32_t), e->d_scratch, &v_obj);
    if (rc) return rc;
    const int32_t* d_obj = (const int32_t*)v_obj;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    *a = SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_pid + (int64_t)slot * e->Np,
                     e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
                     d_obj, n_sub, e->Np, e->F, e->S, e->C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                     from_prior != 0};
    return SBE_OK;
}

int source_posterior_status(sbe_engine* e, bool waited = false) {
    int rc = waited ? take_status(e) : read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE])
        return fail(e, SBE_ERR_DATA, "normalize: %d observations have a non-positive posterior sum (sbayes/util.py:1006 assert)", e->h_status[ST_BAD_NORMALIZE]);
    return SBE_OK;
}

// log_q = sum_i log(p_sel[i]) (fp64, fixed order) -> *out; optionally the selected probabilities themselves.
int finish_log_q(sbe_engine* e, const float* d_psel, int64_t n, double* d_partials, double* log_q_out, float* p_selected_out) {
    const int nb = (int)std::min<int64_t>(div_up(n, 4 * kBlock), 256);
    k_sum_log_f32<<<nb, kBlock, 0, e->stream>>>(d_psel, n, d_partials);
    HIPCHK(e, hipGetLastError());
    k_reduce_partials<<<1, kBlock, 0, e->stream>>>(d_partials, 0, nb, d_partials + 256, 0, 1, StepFinish{});
    HIPCHK(e, hipGetLastError());
    int rc = d2h(e, log_q_out, d_partials + 256, sizeof(double));
    if (rc) return rc;
    if (p_selected_out) { rc = d2h(e, p_selected_out, d_psel, (size_t)n * sizeof(float)); if (rc) return rc; }
    return source_posterior_status(e);
}

}  // namespace

int sbe_source_posterior(sbe_engine* e, int slot, const int32_t* objects, int n_sub, double temperature,
                         double prior_temperature, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    const int64_t n_out = (int64_t)std::max(n_sub, 0) * e->F * e->C;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, 0, (size_t)n_out * sizeof(float), &a, &d_extra);
    if (rc || n_sub == 0) return rc;
    void* d_out;
    rc = out_target(e, (size_t)n_out * sizeof(float), d_extra, &d_out);
    if (rc) return rc;
    const unsigned nb = (unsigned)div_up((int64_t)n_sub * e->F, 256);
    const DoneSig done = out_done(e, d_out, nb);
    k_source_posterior<<<nb, 256, 0, e->stream>>>(a, (float*)d_out, e->d_status, done);
    HIPCHK(e, hipGetLastError());
    rc = out_fetch(e, out, d_out, (size_t)n_out * sizeof(float), done);
    if (rc) return rc;
    return source_posterior_status(e, /*waited=*/true);
}

int sbe_sample_source(sbe_engine* e, int slot, int dst_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, double* log_q_out,
                      float* p_selected_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_SLOT(e, dst_slot); CHECK_PTR(e, log_q_out);
    if (!e->slots[dst_slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set (rows outside the subset would be undefined)", dst_slot);
    if (n_sub == 0) { *log_q_out = 0.0; return SBE_OK; }
    const int64_t n_obs = (int64_t)std::max(n_sub, 0) * e->F;
    const size_t zb = z ? ((size_t)n_obs * sizeof(double) + 255) / 256 * 256 : 0;    // z == NULL: device Philox stream
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, from_prior, zb + pb + 257 * sizeof(double), &a, &d_extra);
    if (rc) return rc;
    double* d_z = (double*)d_extra;
    float* d_psel = (float*)(d_extra + zb);
    double* d_partials = (double*)(d_extra + zb + pb);
    if (z) { int _urc = upload(e, d_z, z, (size_t)n_obs * sizeof(double)); if (_urc) return _urc; }
    k_sample_source<<<div_up(n_obs, 256), 256, 0, e->stream>>>(a, z ? d_z : nullptr, e->rng_seed, e->rng_draw,
                                                               e->d_src + (int64_t)dst_slot * e->N * e->Fp, d_psel, e->d_status, nullptr);
    bump_src(e, dst_slot);
    if (!z) ++e->rng_draw;
    HIPCHK(e, hipGetLastError());
    return finish_log_q(e, d_psel, n_obs, d_partials, log_q_out, p_selected_out);
}

int sbe_set_rng(sbe_engine* e, uint64_t seed, uint64_t draw) {
    CHECK_ENGINE(e);
    e->rng_seed = seed;
    e->rng_draw = draw;
    return SBE_OK;
}

int sbe_get_rng(sbe_engine* e, uint64_t* seed, uint64_t* draw) {
    CHECK_ENGINE(e); CHECK_PTR(e, seed); CHECK_PTR(e, draw);
    *seed = e->rng_seed;
    *draw = e->rng_draw;
    return SBE_OK;
}

int sbe_test_philox(sbe_engine* e, const uint32_t* ctr_key, int n, uint32_t* out) {
    CHECK_ENGINE(e);
    if (n <= 0) return SBE_OK;
    CHECK_PTR(e, ctr_key); CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t ib = ((size_t)n * 6 * sizeof(uint32_t) + 255) / 256 * 256;
    int rc = ensure_scratch(e, ib + (size_t)n * 4 * sizeof(uint32_t));
    if (rc) return rc;
    { int _urc = upload(e, e->d_scratch, ctr_key, (size_t)n * 6 * sizeof(uint32_t)); if (_urc) return _urc; }
    k_test_philox<<<div_up(n, 256), 256, 0, e->stream>>>((const uint32_t*)e->d_scratch, n, (uint32_t*)(e->d_scratch + ib));
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch + ib, (size_t)n * 4 * sizeof(uint32_t));
}

int sbe_source_logprob(sbe_engine* e, int slot, int src_slot, const int32_t* objects, int n_sub, double temperature,
                       double prior_temperature, int from_prior, double* log_q_out, float* p_selected_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_SLOT(e, src_slot); CHECK_PTR(e, log_q_out);
    if (!e->slots[src_slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", src_slot);
    if (n_sub == 0) { *log_q_out = 0.0; return SBE_OK; }
    const int64_t n_obs = (int64_t)std::max(n_sub, 0) * e->F;
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    SrcPostArgs a; uint8_t* d_extra = nullptr;
    int rc = source_posterior_setup(e, slot, objects, n_sub, temperature, prior_temperature, from_prior, pb + 257 * sizeof(double), &a, &d_extra);
    if (rc) return rc;
    float* d_psel = (float*)d_extra;
    double* d_partials = (double*)(d_extra + pb);
    k_source_logprob<<<div_up(n_obs, 256), 256, 0, e->stream>>>(a, e->d_src + (int64_t)src_slot * e->N * e->Fp, d_psel, e->d_status, nullptr);
    HIPCHK(e, hipGetLastError());
    return finish_log_q(e, d_psel, n_obs, d_partials, log_q_out, p_selected_out);
}

int sbe_subset_lh(sbe_engine* e, const int32_t* objects, int n_sub, int n_comp, const float* tables,
                  const int32_t* table_offsets, int n_tables_total, const int32_t* group_idx, double temperature,
                  float* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, out);
    if (n_sub < 0 || n_comp < 1 || n_comp > kMaxComponents || n_tables_total < 1) return fail(e, SBE_ERR_ARG, "bad sizes");
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, tables); CHECK_PTR(e, table_offsets); CHECK_PTR(e, group_idx);
    if (!(temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperature must be positive");
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    for (int c = 0; c < n_comp; ++c) {
        const int hi = (c + 1 < n_comp ? table_offsets[c + 1] : n_tables_total) - table_offsets[c];
        if (table_offsets[c] < 0 || hi < 0) return fail(e, SBE_ERR_ARG, "bad table offsets");
        for (int i = 0; i < n_sub; ++i)
            if (group_idx[(size_t)c * n_sub + i] >= hi) return fail(e, SBE_ERR_ARG, "group index out of range in component %d", c);
    }
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    const size_t tb = ((size_t)n_tables_total * fs * sizeof(float) + 255) / 256 * 256;
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const size_t gb = ((size_t)n_comp * n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const int64_t n_out = (int64_t)n_sub * e->F * n_comp;
    int rc = ensure_scratch(e, tb + ob + gb + 256 + (size_t)n_out * sizeof(float));
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    int32_t* d_obj = (int32_t*)(e->d_scratch + tb);
    int32_t* d_gi = (int32_t*)(e->d_scratch + tb + ob);
    int32_t* d_off = (int32_t*)(e->d_scratch + tb + ob + gb);
    float* d_out = (float*)(e->d_scratch + tb + ob + gb + 256);
    { int _urc = upload(e, d_tab, tables, (size_t)n_tables_total * fs * sizeof(float)); if (_urc) return _urc; }
    { int _urc = upload(e, d_obj, objects, (size_t)n_sub * sizeof(int32_t)); if (_urc) return _urc; }
    { int _urc = upload(e, d_gi, group_idx, (size_t)n_comp * n_sub * sizeof(int32_t)); if (_urc) return _urc; }
    { int _urc = upload(e, d_off, table_offsets, (size_t)n_comp * sizeof(int32_t)); if (_urc) return _urc; }
    const double inv_t = 1.0 / temperature;
    k_subset_lh<<<div_up((int64_t)n_sub * e->F, 256), 256, 0, e->stream>>>(
        e->d_state, d_tab, d_off, d_gi, d_obj, n_sub, d_out, e->F, e->S, n_comp, e->Fp, (float)inv_t, inv_t != 1.0);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n_out * sizeof(float));
}

}  // extern "C"
