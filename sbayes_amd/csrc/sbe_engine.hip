// sbe_engine.hip -- host side of the MI355X sBayes likelihood engine, unit 1 of 4: lifetime, options, the slots' state (setters,
// getters, counts, tables, weights), the resident evaluations (a1 / a3 / a6 surfaces, the fused mixture log-likelihood, the
// collapsed likelihood), slot copies, timing and the self-tests -- the C ABI of include/sbe_engine.h over the gfx950 kernels.
//
// Plain HIP runtime (own stream, own events, own device memory); no torch, no compatibility layer.  One engine = one
// process' view of one GPU.  The one-hot feature block and every slot's state stay resident in HBM; only small tables /
// id vectors cross PCIe per call.
#include "sbe_engine_internal.hip.h"

extern "C" {


int sbe_abi_version(void) { return SBE_ABI_VERSION; }

int sbe_device_count(int* out_count) {
    if (!out_count) return fail(nullptr, SBE_ERR_ARG, "null out_count");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { *out_count = 0; return fail(nullptr, SBE_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(err)); }
    *out_count = n;
    return SBE_OK;
}

const char* sbe_last_error(const sbe_engine* e) { return e ? e->last_error.c_str() : g_last_error.c_str(); }

// ---- host helpers of the drop-in layer's marshalling (no device, no engine) -----------------------------------------
// What update_feature_counts hands to sbe_counts_delta is derived from the samples' own arrays: the listed objects'
// group id per component (group_assignment[:, object_subset], counts.py:21-24) and source component per observation
// (source[object_subset], counts.py:25-27).  In NumPy that derivation is a dozen small array operations per MCMC step
// (tools/host_residual.py: the host layer, not the device, bounds the patched sampler); here it is one pass each.
int sbe_host_group_ids(const uint8_t* groups, int n_groups, int64_t n_objects, const int32_t* objects, int n, int offset,
                       int32_t* ids_out) {
    return sbeh_group_ids(groups, n_groups, n_objects, objects, n, offset, ids_out);
}

int sbe_host_source_ids(const uint8_t* source, int64_t n_objects, int n_features, int n_components, const int32_t* objects, int n,
                        uint8_t* ids_out) {
    return sbeh_source_ids(source, n_objects, n_features, n_components, objects, n, ids_out);
}

int sbe_host_touched_groups(const int32_t* gid_old, const int32_t* gid_new, int64_t count, int n_groups_total,
                            int32_t* touched_out, int32_t* n_touched_out) {
    return sbeh_touched_groups(gid_old, gid_new, count, n_groups_total, touched_out, n_touched_out);
}

int sbe_host_subset_ids(const int32_t* objects, int n, int64_t n_objects, int n_features, int n_components,
                        const int32_t* n_groups, const uint8_t* const* groups_new, const uint8_t* const* groups_old,
                        const uint8_t* source_new, const uint8_t* source_old,
                        int32_t* gid_new_out, int32_t* gid_old_out, uint8_t* sid_new_out, uint8_t* sid_old_out) {
    return sbeh_subset_ids(objects, n, n_objects, n_features, n_components, n_groups, groups_new, groups_old, source_new, source_old,
                           gid_new_out, gid_old_out, sid_new_out, sid_old_out);
}

int64_t sbe_host_diff_rows(const void* rows, void* mirror, int64_t n_rows, int64_t row_bytes, int32_t* changed_out) {
    return sbeh_diff_rows(rows, mirror, n_rows, row_bytes, changed_out);
}

int sbe_destroy(sbe_engine* e) {
    if (!e) return SBE_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    delete e->pool;
    for (auto& ln : e->lanes) {
        if (ln.h_payload) (void)hipHostFree(ln.h_payload);
        if (ln.h_step) (void)hipHostFree(ln.h_step);
        if (ln.d_pf) (void)hipFree(ln.d_pf);
        if (ln.d_stamp) (void)hipFree(ln.d_stamp);
        if (ln.d_status) (void)hipFree(ln.d_status);
    }
    if (e->d_batch_meta) (void)hipFree(e->d_batch_meta);
    if (e->h_batch_payload) (void)hipHostFree(e->h_batch_payload);
    if (e->d_batch_payload) (void)hipFree(e->d_batch_payload);
    if (e->h_step) (void)hipHostFree(e->h_step);
    if (e->h_step_payload) (void)hipHostFree(e->h_step_payload);
    if (e->h_io) (void)hipHostFree(e->h_io);
    void* dev_ptrs[] = {e->d_step_pf, e->d_step_pg, e->d_logtab, e->d_state_h, e->d_toff, e->d_tid, e->d_tuple_g, e->d_tuple_p, e->d_state_q, e->d_probs_t, e->d_wpat_t, e->d_onehot, e->d_state, e->d_gid, e->d_pid, e->d_src, e->d_counts, e->d_probs,
                        e->d_weights, e->d_wpat, e->d_patbits, e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a, e->d_unif, e->d_unif_res, e->d_comp_of_group, e->d_partials, e->d_rowoff,
                        e->d_status, e->d_changed, e->d_step_stamp, e->d_scratch, e->d_xt, e->d_arrive, e->d_logtab_fine, e->d_rowoff_s, e->d_rs_nq, e->d_state_s};
    for (void* p : dev_ptrs) if (p) (void)hipFree(p);
    if (e->h_results) (void)hipHostFree(e->h_results);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_flag) (void)hipHostFree(e->h_flag);
    if (e->h_done) (void)hipHostFree(e->h_done);
    if (e->h_chunk_flags) (void)hipHostFree(e->h_chunk_flags);
    if (e->d_chunk_tickets) (void)hipFree(e->d_chunk_tickets);
    if (e->h_stream) (void)hipHostFree(e->h_stream);
    if (e->d_ticket) (void)hipFree(e->d_ticket);
    if (e->h_pinned) (void)hipHostFree(e->h_pinned);
    if (e->h_arena) (void)hipHostFree(e->h_arena);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SBE_OK;
}

int sbe_create(sbe_engine** out, int device, int n_objects, int n_features, int n_states,
               int n_components, const int32_t* n_groups, int n_slots, const uint8_t* features_onehot) {
    if (!out) return fail(nullptr, SBE_ERR_ARG, "null out handle");
    *out = nullptr;
    if (!features_onehot || !n_groups) return fail(nullptr, SBE_ERR_ARG, "null features / n_groups");
    if (n_objects < 1 || n_features < 1) return fail(nullptr, SBE_ERR_ARG, "empty feature block (%d objects x %d features)", n_objects, n_features);
    if (n_states < 1 || n_states > 254) return fail(nullptr, SBE_ERR_ARG, "n_states=%d unsupported (1..254; state index is one byte, 0xFF = NA)", n_states);
    if (n_components < 1 || n_components > kMaxComponents) return fail(nullptr, SBE_ERR_ARG, "n_components=%d unsupported (1..%d)", n_components, kMaxComponents);
    if (n_slots < 1 || n_slots > 16384) return fail(nullptr, SBE_ERR_ARG, "n_slots=%d unsupported (1..16384)", n_slots);
    int64_t gtot = 0;
    for (int c = 0; c < n_components; ++c) {
        // a component may have no group at all (n_clusters == 0, the confounders-only baseline: the reference's
        // initializer returns an empty cluster matrix, sbayes/sampling/initializers.py:357): its tables are empty and
        // every object is in "no group" of it
        if (n_groups[c] < 0) return fail(nullptr, SBE_ERR_ARG, "component %d has %d groups", c, n_groups[c]);
        gtot += n_groups[c];
    }
    if (gtot < 1) return fail(nullptr, SBE_ERR_ARG, "no component has any group");
    if (gtot >= 0xFFFF) return fail(nullptr, SBE_ERR_ARG, "%lld groups in total exceed the 16-bit group index", (long long)gtot);

    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev == 0)
        return fail(nullptr, SBE_ERR_NODEVICE, "no HIP device available (%s); the engine has no CPU fallback",
                    err != hipSuccess ? hipGetErrorString(err) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, SBE_ERR_ARG, "device %d out of range [0,%d)", device, ndev);

    sbe_engine* e = new sbe_engine();
    e->device = device;
    e->N = n_objects; e->F = n_features; e->S = n_states; e->C = n_components; e->n_slots = n_slots;
    e->G.assign(n_groups, n_groups + n_components);
    e->goff.resize(n_components);
    for (int c = 0, o = 0; c < n_components; ++c) { e->goff[c] = o; o += n_groups[c]; }
    e->Gtot = (int)gtot;
    e->Fp = round_up(n_features, 64);
    e->rs_pitch = round_up(n_features * n_states, 16);
    e->Pmax = std::min(1 << n_components, 64);
    e->Np = round_up(n_objects, 4);
    e->NQ = e->Np / 4;
    {   // v2 feature-tile width: widest of 64/32/16 whose LDS image leaves two blocks per CU
        const char* env = getenv("SBE_FT");
        int ft = 64;
        auto lds_for = [&](int t) { return ((size_t)(gtot + 1) * t * n_states) * sizeof(float) + (size_t)e->Pmax * n_components * t * sizeof(double) + 8 * 1024; };
        while (ft > 16 && lds_for(ft) > 78 * 1024) ft >>= 1;
        if (env && (atoi(env) == 64 || atoi(env) == 32 || atoi(env) == 16)) ft = atoi(env);
        if (lds_for(ft) > 156 * 1024) {
            if (env) { delete e; return fail(nullptr, SBE_ERR_ARG, "probability tables too large for LDS staging at the forced tile width (G_total=%lld, S=%d)", (long long)gtot, n_states); }
            ft = 16;                 // very many groups x states: no LDS staging of tables (L2-served gathers)
            e->direct = true;
        }
        if (getenv("SBE_DIRECT") && atoi(getenv("SBE_DIRECT")) == 1) { ft = 16; e->direct = true; }   // experiments / tests
        e->ft = ft;
        e->n_ftiles = div_up(n_features, ft);
        e->Fq = round_up(n_features, 64);       // row pitch of the quad-interleaved state streams (>= any tiling of F)
    }
    e->conc_set.assign(n_components, 0);
    e->slots.resize(n_slots);
    e->src_sync.resize(n_slots);
    e->ids_sync.resize(n_slots);
    for (Slot& s : e->slots) {
        s.h_gid.assign((size_t)n_components * n_objects, kNoGroup);
        s.probs_set.assign(n_components, 0);
        s.counts_set.assign(n_components, 0);
    }

#define CREATE_CHK(call)                                                                        \
    do {                                                                                        \
        hipError_t _e2 = (call);                                                                \
        if (_e2 != hipSuccess) {                                                                \
            int _rc = fail(nullptr, SBE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e2)); \
            sbe_destroy(e);                                                                     \
            return _rc;                                                                         \
        }                                                                                       \
    } while (0)
#define CREATE_RC(expr)                                        \
    do {                                                       \
        int _rc = (expr);                                      \
        if (_rc) { g_last_error = e->last_error; sbe_destroy(e); return _rc; } \
    } while (0)

    CREATE_CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    e->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* env = getenv("SBE_MFMA_MIN_BATCH")) { if (atoi(env) > 0) e->mfma_min_batch = atoi(env); }      // (A/B runs, tests)
    if (const char* env = getenv("SBE_MFMA_WIDE_MIN_SHARE")) { if (atoi(env) >= 0) e->mfma_wide_min_share = atoi(env); }
    if (const char* env = getenv("SBE_MFMA_SMALL_SL4")) e->mfma_small_sl4 = atoi(env) != 0;
    if (const char* env = getenv("SBE_MFMA_MIN_OBS")) { if (atoll(env) > 0) e->mfma_min_obs = atoll(env); }
    if (const char* env = getenv("SBE_ROWS_SORTED")) e->opt_rows_sorted = atoi(env);
    snprintf(e->device_name, sizeof e->device_name, "%s%s%s", prop.name, prop.name[0] ? " " : "", prop.gcnArchName);
    CREATE_CHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreate(&e->ev0));
    CREATE_CHK(hipEventCreate(&e->ev1));
    for (int i = 0; i < 64; ++i) {                  // event pool of sbe_kernel_timing (grown on demand beyond this)
        hipEvent_t ev;
        CREATE_CHK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }

    const int64_t N = e->N, F = e->F, S = e->S, C = e->C, NS = n_slots;
    CREATE_RC(dmalloc(e, &e->d_onehot, N * e->rs_pitch));
    CREATE_RC(dmalloc(e, &e->d_state, N * e->Fp));
    CREATE_RC(dmalloc(e, &e->d_state_q, (int64_t)e->NQ * e->Fq * 4));
    CREATE_RC(dmalloc(e, &e->d_probs_t, NS * e->probs_t_elems()));
    CREATE_RC(dmalloc(e, &e->d_wpat_t, NS * e->wpat_t_elems()));
    CREATE_RC(dmalloc(e, &e->d_gid, NS * C * e->Np));
    CREATE_RC(dmalloc(e, &e->d_pid, NS * e->Np));
    CREATE_RC(dmalloc(e, &e->d_src, NS * N * e->Fp));
    CREATE_RC(dmalloc(e, &e->d_counts, NS * e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_probs, NS * e->table_elems() + F * S));      // + a row of ones [F][S] } what k_mixture_tuple_mfma reads for
    CREATE_RC(dmalloc(e, &e->d_weights, NS * F * C));
    CREATE_RC(dmalloc(e, &e->d_wpat, NS * e->Pmax * F * C + F * C));        // + a row of ones [F][C] } tuples / groups that are not there
    {
        const std::vector<float> ones((size_t)(F * std::max(S, C)), 1.0f);
        CREATE_CHK(hipMemcpy(e->d_probs + NS * e->table_elems(), ones.data(), F * S * sizeof(float), hipMemcpyHostToDevice));
        CREATE_CHK(hipMemcpy(e->d_wpat + NS * e->Pmax * F * C, ones.data(), F * C * sizeof(float), hipMemcpyHostToDevice));
    }
    CREATE_RC(dmalloc(e, &e->d_patbits, NS * e->Pmax));
    CREATE_RC(dmalloc(e, &e->d_tid, NS * e->Np));
    CREATE_RC(dmalloc(e, &e->d_toff, NS * e->Np + 64));       // + padding: the kernel prefetches 16 entries ahead
    CREATE_CHK(hipMemsetAsync(e->d_toff, 0, (NS * e->Np + 64) * sizeof(uint32_t), e->stream));
    if (C <= 4) {   // k_mixture_rows: widest tile whose LDS image (tables f32 [(Gtot+1)][S+1][ft] + f64 weight planes) fits
        // (sized for half the possible has_components patterns: a component every object has -- `universal` -- halves
        //  them; a launch whose slots really have more falls back to k_mixture_v2) + the waves' offset slots
        const int p_assumed = std::max(1, e->Pmax / 2);
        auto rows_lds = [&](int t) { return (size_t)(gtot + 1) * (S + 1) * t * 4 + (size_t)p_assumed * ((C + 1) / 2) * t * 16 + (size_t)kRowsWaves * (kWave / t) * (C + 1) * 16; };
        e->rows_ft = rows_lds(32) <= 160 * 1024 - 512 ? 32 : rows_lds(16) <= 160 * 1024 - 512 ? 16 : 0;
        if (const char* env = getenv("SBE_ROWS_FT")) { const int v = atoi(env); if (v == 0 || ((v == 16 || v == 32) && rows_lds(v) <= 160 * 1024 - 512)) e->rows_ft = v; }
        if (e->rows_ft) {
            CREATE_RC(dmalloc(e, &e->d_rowoff, NS * (C + 1) * e->Np));
            e->rowoff_epoch.assign(n_slots, ~0ull);
        }
    }
    {   // table of tab_log_pos: interval centres c_i = 1 + (i + 1/2)/128 (c_0 = 1), {RN(1/c), RN(-log(RN(1/c)))}
        std::vector<double> tab(2 * kLogTabEntries);
        for (int i = 0; i < kLogTabEntries; ++i) {
            const double c = i == 0 ? 1.0 : 1.0 + (i + 0.5) / kLogTabEntries;
            const double inv_c = 1.0 / c;
            tab[2 * i] = inv_c;
            tab[2 * i + 1] = i == 0 ? 0.0 : (double)(-logl((long double)inv_c));
        }
        CREATE_RC(dmalloc(e, &e->d_logtab, (int64_t)kLogTabEntries));
        CREATE_CHK(hipMemcpy(e->d_logtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (e->ft == 64 && S <= 127 && (int64_t)e->NQ * e->Fq * 8 < ((int64_t)1 << 31)) {
        CREATE_RC(dmalloc(e, &e->d_state_h, (int64_t)e->NQ * e->Fq * 4));
        k_init_state_h<<<div_up((int64_t)e->NQ * e->Fq, 256), 256, 0, e->stream>>>(e->d_state_h, (int64_t)e->NQ * e->Fq, e->Fq, e->S);
        CREATE_CHK(hipGetLastError());
    }
    CREATE_RC(dmalloc(e, &e->d_tuple_g, NS * kMaxTuples * kMaxComponents));
    CREATE_RC(dmalloc(e, &e->d_tuple_p, NS * (int64_t)kMaxTuples));
    CREATE_CHK(hipMemsetAsync(e->d_tid, 0, NS * e->Np, e->stream));
    CREATE_RC(dmalloc(e, &e->d_conc, e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_unif, F * S));
    CREATE_RC(dmalloc(e, &e->d_unif_res, F * S));
    CREATE_RC(dmalloc(e, &e->d_lg_conc, e->table_elems()));
    CREATE_RC(dmalloc(e, &e->d_sum_a, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_lg_sum_a, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_comp_of_group, e->Gtot));
    {
        std::vector<int32_t> cog(std::max(e->Gtot, 1), 0);
        for (int c = 0; c < e->C; ++c)
            for (int g = 0; g < e->G[c]; ++g) cog[e->goff[c] + g] = c;
        CREATE_CHK(hipMemcpy(e->d_comp_of_group, cog.data(), (size_t)std::max(e->Gtot, 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // partials: worst-case block count of the fused kernel (ft = 16, one packed step per thread)
    {
        const int64_t min_objs = kBlock / (16 / 4);
        e->partials_stride = std::max<int64_t>(div_up(F, 16) * std::max<int64_t>(div_up(N, min_objs), 4 * e->compute_units), 1024);
    }
    CREATE_RC(dmalloc(e, &e->d_partials, NS * e->partials_stride));
    CREATE_RC(dmalloc(e, &e->d_arrive, NS + 1));
    CREATE_CHK(hipMemset(e->d_arrive, 0, (size_t)(NS + 1) * sizeof(unsigned)));
    CREATE_RC(dmalloc(e, &e->d_status, (int64_t)ST_WORDS));
    CREATE_RC(dmalloc(e, &e->d_changed, (int64_t)e->Gtot));
    CREATE_RC(dmalloc(e, &e->d_step_stamp, (int64_t)e->Gtot));
    CREATE_CHK(hipMemsetAsync(e->d_step_stamp, 0, e->Gtot * sizeof(uint32_t), e->stream));
    CREATE_RC(dmalloc(e, &e->d_step_pf, (int64_t)e->Gtot * F));
    CREATE_RC(dmalloc(e, &e->d_step_pg, (int64_t)e->Gtot));
    {   // one-call step: payload layout (every section 16-byte aligned) and the mapped result block
        auto al = [](size_t v) { return (v + 15) / 16 * 16; };
        e->step_max_rows = (int)std::min<int64_t>(N, 256);
        size_t o = 0;
        e->sl.ids = o;      o = al(o + (size_t)e->Np * 2);
        e->sl.pid = o;      o = al(o + (size_t)e->Np);
        e->sl.tid = o;      o = al(o + (size_t)e->Np);
        e->sl.toff = o;     o = al(o + (size_t)e->Np * 4);
        e->sl.tuple_g = o;  o = al(o + (size_t)kMaxTuples * kMaxComponents * 2);
        e->sl.tuple_p = o;  o = al(o + (size_t)kMaxTuples);
        e->sl.patbits = o;  o = al(o + (size_t)e->Pmax * 4);
        e->sl.weights = o;  o = al(o + (size_t)(F * C) * 4);
        e->sl.row_of = o;   o = al(o + (size_t)e->Np * 2);
        e->sl.subset = o;   o = al(o + (size_t)e->Np * 4);
        e->sl.stale = o;    o = al(o + (size_t)e->step_max_rows * 4);
        e->sl.objects = o;  o = al(o + (size_t)e->step_max_rows * 4);
        e->sl.rows = o;     o = al(o + (size_t)e->step_max_rows * (size_t)(F * C));
        e->sl.total = o;
        CREATE_CHK(hipHostMalloc((void**)&e->h_step_payload, e->sl.total, hipHostMallocMapped));
        CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_step_payload, e->h_step_payload, 0));
        const size_t hb = ((size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int) + (size_t)e->Gtot + 7) / 8 * 8 + 2 * sizeof(double);   // (step_host_lq_offset + log_q, log_q_back)
        CREATE_CHK(hipHostMalloc((void**)&e->h_step, hb, hipHostMallocMapped));
        memset(e->h_step, 0, hb);
        CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_step_host, e->h_step, 0));
    }
    CREATE_CHK(hipHostMalloc((void**)&e->h_results, NS * sizeof(double), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_results, e->h_results, 0));
    CREATE_CHK(hipHostMalloc((void**)&e->h_status, ST_WORDS * sizeof(int), hipHostMallocDefault));
    memset(e->h_status, 0, ST_WORDS * sizeof(int));
    CREATE_CHK(hipHostMalloc((void**)&e->h_flag, ST_WORDS * sizeof(int), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_flag, e->h_flag, 0));
    memset(e->h_flag, 0, ST_WORDS * sizeof(int));
    CREATE_CHK(hipHostMalloc((void**)&e->h_done, 64, hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_done, e->h_done, 0));
    memset(e->h_done, 0, 64);
    CREATE_CHK(hipMalloc((void**)&e->d_ticket, 64));
    CREATE_CHK(hipMemsetAsync(e->d_ticket, 0, 64, e->stream));
    CREATE_CHK(hipHostMalloc((void**)&e->h_chunk_flags, sbe_engine::kMaxChunks * sizeof(unsigned long long), hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_chunk_flags, e->h_chunk_flags, 0));
    memset(e->h_chunk_flags, 0, sbe_engine::kMaxChunks * sizeof(unsigned long long));
    CREATE_CHK(hipMalloc((void**)&e->d_chunk_tickets, sbe_engine::kMaxChunks * sizeof(unsigned)));
    CREATE_CHK(hipMemsetAsync(e->d_chunk_tickets, 0, sbe_engine::kMaxChunks * sizeof(unsigned), e->stream));
    e->arena_bytes = (size_t)16 << 20;
    CREATE_CHK(hipHostMalloc((void**)&e->h_arena, e->arena_bytes, hipHostMallocMapped));
    CREATE_CHK(hipHostGetDevicePointer((void**)&e->d_arena, e->h_arena, 0));

    CREATE_CHK(hipMemsetAsync(e->d_status, 0, ST_WORDS * sizeof(int), e->stream));
    CREATE_CHK(hipMemcpyAsync(e->d_status + ST_FLAG_PTR, &e->d_flag, sizeof(int*), hipMemcpyHostToDevice, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_onehot, 0, N * e->rs_pitch, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_state, 0xFF, N * e->Fp, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_src, 0xFF, NS * N * e->Fp, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_gid, 0xFF, NS * C * e->Np * sizeof(uint16_t), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_pid, 0, NS * e->Np, e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_state_q, S, (int64_t)e->NQ * e->Fq * 4, e->stream));   // NA / padding byte = S
    CREATE_CHK(hipMemsetAsync(e->d_probs_t, 0, NS * e->probs_t_elems() * sizeof(float), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_wpat_t, 0, NS * e->wpat_t_elems() * sizeof(double), e->stream));
    CREATE_CHK(hipMemsetAsync(e->d_counts, 0, NS * e->table_elems() * sizeof(int32_t), e->stream));

    // ingest: raw one-hot -> normalised padded copy + packed state index + validation
    CREATE_RC(ensure_scratch(e, (size_t)(N * F * S)));
    CREATE_CHK(hipMemcpyAsync(e->d_scratch, features_onehot, (size_t)(N * F * S), hipMemcpyHostToDevice, e->stream));
    k_ingest_onehot<<<div_up(N * F, 256), 256, 0, e->stream>>>(e->d_scratch, e->d_onehot, e->d_state, e->d_state_q,
                                                              e->d_state_h, e->N, e->F, e->S, e->rs_pitch, e->Fp, e->Fq, e->d_status);
    CREATE_CHK(hipGetLastError());
    CREATE_CHK(hipMemcpyAsync(e->h_status, e->d_status, ST_FLAG_PTR * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    CREATE_CHK(hipStreamSynchronize(e->stream));
    if (e->h_status[ST_MULTI_STATE] != 0) {
        int rc = fail(nullptr, SBE_ERR_DATA, "features are not one-hot: %d (object, feature) rows have more than one state set",
                      e->h_status[ST_MULTI_STATE]);
        sbe_destroy(e);
        return rc;
    }
    e->n_na = e->h_status[ST_NA_COUNT];
#undef CREATE_CHK
#undef CREATE_RC
    *out = e;
    return SBE_OK;
}

int sbe_get_info(const sbe_engine* e, sbe_info* out) {
    CHECK_ENGINE(e);
    if (!out) return fail(nullptr, SBE_ERR_ARG, "null out");
    memset(out, 0, sizeof *out);
    out->abi_version = SBE_ABI_VERSION;
    out->device = e->device;
    out->n_objects = e->N; out->n_features = e->F; out->n_states = e->S;
    out->n_components = e->C; out->n_slots = e->n_slots; out->n_groups_total = e->Gtot;
    out->n_na = e->n_na; out->hbm_bytes = e->hbm_bytes; out->compute_units = e->compute_units;
    memcpy(out->device_name, e->device_name, sizeof out->device_name);
    return SBE_OK;
}

int sbe_get_na(const sbe_engine* ce, uint8_t* out_na) {
    sbe_engine* e = const_cast<sbe_engine*>(ce);
    CHECK_ENGINE(e); CHECK_PTR(e, out_na);
    HIPCHK(e, hipSetDevice(e->device));
    std::vector<uint8_t> st((size_t)e->N * e->Fp);
    int rc = d2h(e, st.data(), e->d_state, st.size());
    if (rc) return rc;
    for (int n = 0; n < e->N; ++n)
        for (int f = 0; f < e->F; ++f) out_na[(size_t)n * e->F + f] = st[(size_t)n * e->Fp + f] == kNA;
    return SBE_OK;
}

int sbe_set_option(sbe_engine* e, int option, int value) {
    CHECK_ENGINE(e);
    if (option == SBE_OPT_MIXTURE_KERNEL && (value == SBE_MIXTURE_PACKED || value == SBE_MIXTURE_ONEHOT || value == SBE_MIXTURE_PACKED_GENERAL || value == SBE_MIXTURE_PACKED_TUPLE || value == SBE_MIXTURE_PACKED_TUPLE_LDS || value == SBE_MIXTURE_ONEHOT_GENERAL || value == SBE_MIXTURE_PACKED_V2 || value == SBE_MIXTURE_PACKED_TUPLE_MFMA)) { e->opt_kernel = value; return SBE_OK; }
    if (option == SBE_OPT_LOG_MODE && (value == SBE_LOG_PER_OBS || value == SBE_LOG_PRODUCT)) { e->opt_log = value; return SBE_OK; }
    if (option == SBE_OPT_STEP_FORM && (value == 0 || value == 1)) { e->opt_step_form = value; return SBE_OK; }
    if (option == SBE_OPT_STEP_DERIVE && (value == 0 || value == 1)) { e->opt_step_derive = value; return SBE_OK; }
    if (option == SBE_OPT_FUSE_TABLES && (value == 0 || value == 1)) { e->opt_fuse_tables = value; return SBE_OK; }
    if (option == SBE_OPT_DEFERRED_CHECKS && (value == 0 || value == 1)) {
        if (!value && e->status_pending) { HIPCHK(e, hipStreamSynchronize(e->stream)); int rc = synced(e); e->opt_deferred = 0; return rc; }
        e->opt_deferred = value;
        return SBE_OK;
    }
    return fail(e, SBE_ERR_ARG, "unknown option %d / value %d", option, value);
}

int sbe_sync(sbe_engine* e) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return synced(e);
}

// ---- a1 -------------------------------------------------------------------------------------
int sbe_component_lh(sbe_engine* e, const void* probs, int probs_f64, int n_groups, const uint8_t* groups,
                     const int64_t* changed_groups, int n_changed, double* out, int64_t out_stride_n_bytes,
                     int64_t out_stride_f_bytes, double na_value) {
    CHECK_ENGINE(e); CHECK_PTR(e, probs); CHECK_PTR(e, groups); CHECK_PTR(e, out);
    if (n_groups < 1) return fail(e, SBE_ERR_ARG, "n_groups=%d", n_groups);
    if (n_changed < 0 || (n_changed > 0 && !changed_groups)) return fail(e, SBE_ERR_ARG, "bad changed_groups");
    for (int k = 0; k < n_changed; ++k)
        if (changed_groups[k] < 0 || changed_groups[k] >= n_groups)
            return fail(e, SBE_ERR_ARG, "changed_groups[%d]=%lld out of range [0,%d)", k, (long long)changed_groups[k], n_groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, F = e->F, S = e->S;
    // row selector: -2 untouched, -1 zero, >=0 table index (last changed group wins)
    std::vector<int32_t> sel(N, -2);
    for (int n = 0; n < N; ++n) {
        bool any = false;
        for (int g = 0; g < n_groups && !any; ++g) any = groups[(size_t)g * N + n] != 0;
        if (!any) sel[n] = -1;
    }
    for (int k = 0; k < n_changed; ++k) {
        const int64_t g = changed_groups[k];
        const uint8_t* row = groups + (size_t)g * N;
        for (int n = 0; n < N; ++n) if (row[n]) sel[n] = (int32_t)g;
    }
    const size_t esz = probs_f64 ? sizeof(double) : sizeof(float);
    const size_t tab_bytes = (size_t)n_groups * F * S * esz;
    const size_t tab_pad = (tab_bytes + 255) / 256 * 256;
    const size_t sel_bytes = ((size_t)N * sizeof(int32_t) + 255) / 256 * 256;
    const size_t out_bytes = (size_t)N * F * sizeof(double);
    int rc = ensure_scratch(e, tab_pad + sel_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* d_tab = e->d_scratch;
    int32_t* d_sel = (int32_t*)(e->d_scratch + tab_pad);
    double* d_out = (double*)(e->d_scratch + tab_pad + sel_bytes);
    {   // table + row selector: one enqueue when they fit the mapped ring (upload_segments), two copies otherwise
        const UploadSeg segs[2] = {{d_tab, probs, tab_bytes}, {d_sel, sel.data(), (size_t)N * sizeof(int32_t)}};
        int _urc = upload_segments(e, segs, 2);
        if (_urc) return _urc;
    }
    const int blocks = div_up((int64_t)N * F, 512);       // (k_component_lh: two output elements per thread)
    const int32_t* selp = sel.data();
    char* base = (char*)out;
    auto scatter_rows = [=](const double* dense, int n0, int n1) {   // rows the call writes: the caller's (strided) view <- staging rows
        for (int n = n0; n < n1; ++n) {
            if (selp[n] == -2) continue;
            char* row = base + (int64_t)n * out_stride_n_bytes;
            const double* srow = dense + (size_t)n * F;
            if (out_stride_f_bytes == (int64_t)sizeof(double)) memcpy(row, srow, (size_t)F * sizeof(double));
            else for (int f = 0; f < F; ++f) *(double*)(row + (int64_t)f * out_stride_f_bytes) = srow[f];
        }
    };
    static const bool no_stream = [] { const char* v = getenv("SBE_STREAM_RESULTS"); return v && atoi(v) == 0; }();   // (A/B)
    if (out_bytes >= ((size_t)1 << 19) && !no_stream) {
        // large result: the kernel stores it into host-mapped staging and reports chunk by chunk; the scatter into the
        // caller's view runs on the host pool while the later chunks cross PCIe (stream_result)
        rc = ensure_stream(e, out_bytes);
        if (rc) return rc;
        const StreamPlan plan = plan_stream(e, (unsigned)blocks, (size_t)512 * sizeof(double), out_bytes);
        double* s_out = (double*)e->d_stream;
        if (probs_f64) k_component_lh<double><<<plan.grid, 256, 0, e->stream>>>(e->d_state, (const double*)d_tab, d_sel, s_out, N, F, S, e->Fp, na_value, plan.sig);
        else k_component_lh<float><<<plan.grid, 256, 0, e->stream>>>(e->d_state, (const float*)d_tab, d_sel, s_out, N, F, S, e->Fp, na_value, plan.sig);
        HIPCHK(e, hipGetLastError());
        const double* dense = (const double*)e->h_stream;
        const int rows_per_job = std::max(1, (int)(((size_t)64 << 10) / ((size_t)F * sizeof(double))));
        const int n_jobs = div_up(N, rows_per_job);
        return stream_result(e, plan, n_jobs,
                             [=](int j) { return (size_t)j * rows_per_job * F * sizeof(double); },
                             [=](int j) { return (size_t)std::min(N, (j + 1) * rows_per_job) * F * sizeof(double); },
                             [=](int j) { scatter_rows(dense, j * rows_per_job, std::min(N, (j + 1) * rows_per_job)); });
    }
    if (probs_f64) k_component_lh<double><<<blocks, 256, 0, e->stream>>>(e->d_state, (const double*)d_tab, d_sel, d_out, N, F, S, e->Fp, na_value);
    else k_component_lh<float><<<blocks, 256, 0, e->stream>>>(e->d_state, (const float*)d_tab, d_sel, d_out, N, F, S, e->Fp, na_value);
    HIPCHK(e, hipGetLastError());
    rc = ensure_pinned(e, out_bytes);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(e->h_pinned, d_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    scatter_rows((const double*)e->h_pinned, 0, N);
    return SBE_OK;
}


int sbe_set_groups(sbe_engine* e, int slot, int component, const uint8_t* groups) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] > 0) CHECK_PTR(e, groups);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, G = e->G[component], off = e->goff[component];
    std::vector<uint16_t> ids(N, kNoGroup);
    char msg[320];
    OverlapNote note;
    (void)matrix_to_ids(groups, G, N, off, component, ids.data(), msg, sizeof msg, &note);      // (permissive: the last group wins)
    Slot& s = e->slots[slot];
    if (note.found) {
        if (!s.overlap_mask) { s.ov_obj = note.obj; s.ov_g1 = note.g1; s.ov_g2 = note.g2; s.ov_comp = component; }
        s.overlap_mask |= 1u << component;
    } else {
        s.overlap_mask &= ~(1u << component);
    }
    return set_gid_common(e, slot, component, ids);
}

int sbe_set_group_ids(sbe_engine* e, int slot, int component, const int32_t* ids_in) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component); CHECK_PTR(e, ids_in);
    HIPCHK(e, hipSetDevice(e->device));
    const int N = e->N, G = e->G[component], off = e->goff[component];
    std::vector<uint16_t> ids(N, kNoGroup);
    for (int n = 0; n < N; ++n) {
        if (ids_in[n] >= G) return fail(e, SBE_ERR_ARG, "group id %d of object %d out of range [0,%d)", ids_in[n], n, G);
        if (ids_in[n] >= 0) ids[n] = (uint16_t)(off + ids_in[n]);
    }
    e->slots[slot].overlap_mask &= ~(1u << component);       // (one id per object: no overlap by construction)
    return set_gid_common(e, slot, component, ids);
}

int sbe_get_group_ids(sbe_engine* e, int slot, int component, int32_t* ids_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component); CHECK_PTR(e, ids_out);
    HIPCHK(e, hipSetDevice(e->device));
    Slot& s = e->slots[slot];
    if (!s.groups_set) return fail(e, SBE_ERR_STATE, "slot %d: groups not set", slot);
    if (s.gid_pending) { int rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    // what the DEVICE holds (not the host mirror): the ids every resident kernel reads
    std::vector<uint16_t> ids((size_t)e->N);
    int rc = d2h(e, ids.data(), e->d_gid + ((int64_t)slot * e->C + component) * e->Np, ids.size() * sizeof(uint16_t));
    if (rc) return rc;
    const int off = e->goff[component];
    for (int n = 0; n < e->N; ++n) ids_out[n] = ids[n] == kNoGroup ? -1 : (int32_t)ids[n] - off;
    return SBE_OK;
}

// ---- source -----------------------------------------------------------------------------------
int sbe_set_source(sbe_engine* e, int slot, const uint8_t* source) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, source);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bytes = (size_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, bytes);
    if (rc) return rc;
    rc = clear_status_word(e, ST_MULTI_SOURCE);
    if (rc) return rc;
    { int _urc = upload(e, e->d_scratch, source, bytes); if (_urc) return _urc; }
    k_ingest_source<<<div_up((int64_t)e->N * e->F, 256), 256, 0, e->stream>>>(
        e->d_scratch, nullptr, e->d_src + (int64_t)slot * e->N * e->Fp, e->N, e->F, e->C, e->Fp, e->d_status);
    HIPCHK(e, hipGetLastError());
    bump_src(e, slot);
    e->slots[slot].source_set = true;
    return check_after(e, ST_MULTI_SOURCE);
}

int sbe_set_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows, const uint8_t* rows) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, rows);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    for (int i = 0; i < n_rows; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t row_bytes = (size_t)n_rows * e->F * e->C;
    const size_t row_pad = (row_bytes + 255) / 256 * 256;
    int rc = ensure_scratch(e, row_pad + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    rc = clear_status_word(e, ST_MULTI_SOURCE);
    if (rc) return rc;
    const void *v_rows, *v_obj;
    rc = stage(e, rows, row_bytes, e->d_scratch, &v_rows);
    if (rc) return rc;
    rc = stage(e, objects, (size_t)n_rows * sizeof(int32_t), e->d_scratch + row_pad, &v_obj);
    if (rc) return rc;
    if (e->batch && e->batch->n_src_blocks == 0 && v_rows != (const void*)e->d_scratch) {     // (staged in the ring: sbe_set_slot_delta launches it)
        SetterJobs& j = *e->batch;
        j.src_rows = (const uint8_t*)v_rows; j.src_objects = (const int32_t*)v_obj; j.src_id = e->d_src + (int64_t)slot * e->N * e->Fp;
        j.src_n = n_rows; j.src_F = e->F; j.src_C = e->C; j.src_Fp = e->Fp; j.src_status = e->d_status;
        j.n_src_blocks = (unsigned)div_up((int64_t)n_rows * e->F, 256);
    } else {
        k_ingest_source<<<div_up((int64_t)n_rows * e->F, 256), 256, 0, e->stream>>>(
            (const uint8_t*)v_rows, (const int32_t*)v_obj, e->d_src + (int64_t)slot * e->N * e->Fp, n_rows, e->F, e->C, e->Fp, e->d_status);
        HIPCHK(e, hipGetLastError());
    }
    bump_src(e, slot);
    return check_after(e, ST_MULTI_SOURCE);
}

int sbe_get_source_rows(sbe_engine* e, int slot, const int32_t* objects, int n_rows, uint8_t* rows_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, rows_out);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (!e->slots[slot].source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", slot);
    for (int i = 0; i < n_rows; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t row_bytes = (size_t)n_rows * e->F * e->C;
    const size_t row_pad = (row_bytes + 255) / 256 * 256;
    int rc = ensure_scratch(e, row_pad + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    int32_t* d_obj = (int32_t*)(e->d_scratch + row_pad);
    { int _urc = upload(e, d_obj, objects, (size_t)n_rows * sizeof(int32_t)); if (_urc) return _urc; }
    k_expand_source<<<div_up((int64_t)n_rows * e->F, 256), 256, 0, e->stream>>>(
        e->d_src + (int64_t)slot * e->N * e->Fp, d_obj, e->d_scratch, n_rows, e->F, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, rows_out, e->d_scratch, row_bytes);
}

// ---- counts -----------------------------------------------------------------------------------
int sbe_recount(sbe_engine* e, int slot, int component) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (component != -1) CHECK_COMP(e, component);
    Slot& s = e->slots[slot];
    if (!s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: source not set", slot);
    if (!s.groups_set) return fail(e, SBE_ERR_STATE, "slot %d: groups not set", slot);
    { int orc = reject_overlap(e, slot, "sbe_recount"); if (orc) return orc; }
    HIPCHK(e, hipSetDevice(e->device));
    // the kernel counts every component in one pass (each observation has one source
    // component); a single-component request recounts all and is still exact.
    HIPCHK(e, hipMemsetAsync(e->d_counts + (int64_t)slot * e->table_elems(), 0, e->table_elems() * sizeof(int32_t), e->stream));
    int rc = counts_launch(e, slot, +1, slot, 0, nullptr, e->N, slot, false, nullptr);
    if (rc) return rc;
    std::fill(s.counts_set.begin(), s.counts_set.end(), 1);
    return SBE_OK;
}

int sbe_update_counts(sbe_engine* e, int slot_new, int slot_old, const int32_t* objects, int n_subset,
                      uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot_new); CHECK_SLOT(e, slot_old);
    if (n_subset < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d", n_subset);
    if (n_subset > 0) CHECK_PTR(e, objects);
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    { int orc = reject_overlap(e, slot_new, "sbe_update_counts"); if (!orc) orc = reject_overlap(e, slot_old, "sbe_update_counts"); if (orc) return orc; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemsetAsync(e->d_changed, 0, e->Gtot, e->stream));
    if (n_subset > 0) {
        int rc = ensure_scratch(e, (size_t)n_subset * sizeof(int32_t));
        if (rc) return rc;
        { int _urc = upload(e, e->d_scratch, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
        rc = counts_launch(e, slot_new, +1, slot_old, -1, (const int32_t*)e->d_scratch, n_subset, slot_new, true, e->d_changed);
        if (rc) return rc;
    }
    if (changed_groups_out) {
        int rc = d2h(e, changed_groups_out, e->d_changed, e->Gtot);
        if (rc) return rc;
    }
    return SBE_OK;
}

int sbe_accumulate_counts(sbe_engine* e, int slot, const int32_t* objects, int n_subset, int sign,
                          uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (sign != 1 && sign != -1) return fail(e, SBE_ERR_ARG, "sign must be +1 or -1");
    if (n_subset < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d", n_subset);
    if (n_subset > 0) CHECK_PTR(e, objects);
    for (int i = 0; i < n_subset; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    { int orc = reject_overlap(e, slot, "sbe_accumulate_counts"); if (orc) return orc; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemsetAsync(e->d_changed, 0, e->Gtot, e->stream));
    if (n_subset > 0) {
        int rc = ensure_scratch(e, (size_t)n_subset * sizeof(int32_t));
        if (rc) return rc;
        { int _urc = upload(e, e->d_scratch, objects, (size_t)n_subset * sizeof(int32_t)); if (_urc) return _urc; }
        rc = counts_launch(e, slot, sign, slot, 0, (const int32_t*)e->d_scratch, n_subset, slot, true, e->d_changed);
        if (rc) return rc;
    }
    if (changed_groups_out) return d2h(e, changed_groups_out, e->d_changed, e->Gtot);
    return SBE_OK;
}

int sbe_set_counts(sbe_engine* e, int slot, int component, const float* counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->slots[slot].counts_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, counts);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const void* v_counts;
    rc = stage(e, counts, n * sizeof(float), e->d_scratch, &v_counts);
    if (rc) return rc;
    int32_t* dst = e->d_counts + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    k_f32_to_i32<<<div_up(n, 256), 256, 0, e->stream>>>((const float*)v_counts, dst, n);
    HIPCHK(e, hipGetLastError());
    e->slots[slot].counts_set[component] = 1;
    return SBE_OK;
}

int sbe_get_counts(sbe_engine* e, int slot, int component, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;
    CHECK_PTR(e, out);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const int32_t* src = e->d_counts + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    void* d_out;
    rc = out_target(e, n * sizeof(float), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, 1024), 64);
    const DoneSig done = out_done(e, d_out, blocks);
    if (done.flag) k_i32_to_f32_done<<<blocks, 1024, 0, e->stream>>>(src, (float*)d_out, n, done);
    else k_i32_to_f32<<<div_up(n, 256), 256, 0, e->stream>>>(src, (float*)d_out, n);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, n * sizeof(float), done);
}

// ---- concentration / probs ----------------------------------------------------------------------
int sbe_set_concentration(sbe_engine* e, int component, const double* conc, int per_group) {
    CHECK_ENGINE(e); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->conc_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, conc);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    double* dst = e->d_conc + (int64_t)e->goff[component] * fs;
    if (per_group) {
        { int _urc = upload(e, dst, conc, e->G[component] * fs * sizeof(double)); if (_urc) return _urc; }
    } else {
        for (int g = 0; g < e->G[component]; ++g)
            { int _urc = upload(e, dst + g * fs, conc, fs * sizeof(double)); if (_urc) return _urc; }
    }
    // the count-independent lgamma terms of this component's tables, for the one-call steps
    const int g_lo = e->goff[component], g_hi = g_lo + e->G[component];
    k_conc_lgamma<<<div_up((int64_t)(g_hi - g_lo) * e->F, 256), 256, 0, e->stream>>>(e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a,
                                                                              g_lo, g_hi, e->F, e->S);
    HIPCHK(e, hipGetLastError());
    e->conc_set[component] = 1;
    return SBE_OK;
}

int sbe_update_probs_mask(sbe_engine* e, int slot, unsigned component_mask, double temperature, double prior_temperature,
                          const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (e->C < 32 && (component_mask >> e->C) != 0) return fail(e, SBE_ERR_ARG, "component mask 0x%x names components >= %d", component_mask, e->C);
    Slot& s = e->slots[slot];
    bool any = false;
    for (int c = 0; c < e->C; ++c) {
        if (!((component_mask >> c) & 1u)) continue;
        if (e->G[c] == 0) { s.probs_set[c] = 1; continue; }                        // component without groups: empty tables
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        any = true;
    }
    if (!any) return SBE_OK;
    if (prior_temperature > 0.0 && !unif_counts) return fail(e, SBE_ERR_ARG, "prior_temperature given without unif_counts (conditionals.py:114)");
    HIPCHK(e, hipSetDevice(e->device));
    const double* d_unif = nullptr;
    if (prior_temperature > 0.0) {
        { int _urc = upload(e, e->d_unif, unif_counts, (size_t)e->F * e->S * sizeof(double)); if (_urc) return _urc; }
        d_unif = e->d_unif;
    }
    int rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    for (int c = 0; c < e->C; ++c) {                       // runs of selected components: adjacent group ranges, one launch
        if (!((component_mask >> c) & 1u) || e->G[c] == 0) continue;
        const int g_lo = e->goff[c];
        int g_hi = g_lo + e->G[c];
        s.probs_set[c] = 1;
        while (c + 1 < e->C && ((component_mask >> (c + 1)) & 1u)) { ++c; g_hi = e->goff[c] + e->G[c]; s.probs_set[c] = 1; }
        k_probs<int32_t><<<div_up((int64_t)(g_hi - g_lo) * e->F, 256), 256, 0, e->stream>>>(
            e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, d_unif,
            e->d_probs + (int64_t)slot * e->table_elems(), g_lo, g_hi, e->F, e->S, temperature, prior_temperature, 1, e->d_status, 0,
            e->d_probs_t + (int64_t)slot * e->probs_t_elems(), e->Gtot, e->ft);   // (+ the tile-transposed copy: one launch)
        HIPCHK(e, hipGetLastError());
    }
    return check_after(e, ST_BAD_NORMALIZE);
}

int sbe_update_probs(sbe_engine* e, int slot, int component, double temperature, double prior_temperature,
                     const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    return sbe_update_probs_mask(e, slot, 1u << component, temperature, prior_temperature, unif_counts);
}

int sbe_set_probs(sbe_engine* e, int slot, int component, const float* probs) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) { e->slots[slot].probs_set[component] = 1; return SBE_OK; }
    CHECK_PTR(e, probs);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    float* dst = e->d_probs + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    int rc = h2d(e, dst, probs, n * sizeof(float));
    if (rc) return rc;
    rc = retile_probs(e, slot, component);
    if (rc) return rc;
    e->slots[slot].probs_set[component] = 1;
    return SBE_OK;
}

// every component's table in one call (recalculate_feature_counts reads them all back: counts.py:35-52)
int sbe_get_counts_all(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (e->Gtot == 0) return SBE_OK;
    CHECK_PTR(e, out);
    for (int c = 0; c < e->C; ++c)
        if (e->G[c] > 0 && !e->slots[slot].counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->Gtot * e->F * e->S;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    const int32_t* src = e->d_counts + (int64_t)slot * e->table_elems();
    void* d_out;
    rc = out_target(e, n * sizeof(float), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, 1024), 64);
    const DoneSig done = out_done(e, d_out, blocks);
    if (done.flag) k_i32_to_f32_done<<<blocks, 1024, 0, e->stream>>>(src, (float*)d_out, n, done);
    else k_i32_to_f32<<<div_up(n, 256), 256, 0, e->stream>>>(src, (float*)d_out, n);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, out, d_out, n * sizeof(float), done);
}

int sbe_get_probs(sbe_engine* e, int slot, int component, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;
    CHECK_PTR(e, out);
    if (!e->slots[slot].probs_set[component]) return fail(e, SBE_ERR_STATE, "slot %d: probs of component %d not set", slot, component);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->G[component] * e->F * e->S;
    const float* src = e->d_probs + (int64_t)slot * e->table_elems() + (int64_t)e->goff[component] * e->F * e->S;
    return d2h(e, out, src, n * sizeof(float));
}

// ---- weights ------------------------------------------------------------------------------------
int sbe_set_weights(sbe_engine* e, int slot, const float* weights) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, weights);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_patterns_and_weights(e, slot, weights);
    if (rc) return rc;
    e->slots[slot].weights_set = true;
    return SBE_OK;
}

int sbe_get_weights(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    if (!e->slots[slot].weights_set) return fail(e, SBE_ERR_STATE, "slot %d: weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    return d2h(e, out, e->d_weights + (int64_t)slot * e->F * e->C, (size_t)e->F * e->C * sizeof(float));
}

int sbe_get_weights_normalized(sbe_engine* e, int slot, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { int rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, n * sizeof(float));
    if (rc) return rc;
    k_expand_weights<<<div_up(n, 256), 256, 0, e->stream>>>(e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
                                                           e->d_pid + (int64_t)slot * e->Np, (float*)e->d_scratch, e->N, e->F, e->C);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(float));
}

// ---- a3 / a2 / a6 dense outputs ------------------------------------------------------------------
int sbe_likelihood_per_component(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    int rc = check_slot_ready(e, slot, false);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->N * e->F * e->C;
    const int blocks = div_up(n, 512);                    // (k_lh_dense: two output elements per thread)
    static const bool no_stream = [] { const char* v = getenv("SBE_STREAM_RESULTS"); return v && atoi(v) == 0; }();   // (A/B)
    if ((size_t)n * sizeof(double) >= ((size_t)1 << 19) && !no_stream) {
        // large result: stored by the kernel into host-mapped staging, copied to the caller chunk by chunk on the host
        // pool while the rest crosses PCIe (stream_result)
        const size_t bytes = (size_t)n * sizeof(double);
        rc = ensure_stream(e, bytes);
        if (rc) return rc;
        const StreamPlan plan = plan_stream(e, (unsigned)blocks, (size_t)512 * sizeof(double), bytes);
        k_lh_dense<<<plan.grid, 256, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_probs + (int64_t)slot * e->table_elems(),
            (double*)e->d_stream, e->N, e->Np, e->F, e->S, e->C, e->Fp, plan.sig);
        HIPCHK(e, hipGetLastError());
        constexpr size_t kJob = (size_t)64 << 10;
        const int n_jobs = (int)((bytes + kJob - 1) / kJob);
        uint8_t* dst = (uint8_t*)out;
        const uint8_t* stage = e->h_stream;
        rc = stream_result(e, plan, n_jobs,
                           [=](int j) { return (size_t)j * kJob; },
                           [=](int j) { return std::min(bytes, (size_t)(j + 1) * kJob); },
                           [=](int j) { const size_t o = (size_t)j * kJob; memcpy(dst + o, stage + o, std::min(kJob, bytes - o)); });
        return rc ? rc : synced(e);               // (the call waited for the device: a deferred data check is delivered here)
    }
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    k_lh_dense<<<blocks, 256, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_probs + (int64_t)slot * e->table_elems(),
        (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(double));
}

int sbe_likelihood_per_component_exact(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source not set", slot);
    for (int c = 0; c < e->C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t n = (int64_t)e->N * e->F * e->C;
    int rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_lh_exact<<<div_up((int64_t)e->N * e->F, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_gid + (int64_t)slot * e->C * e->Np,
        e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp, e->d_status);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, e->d_scratch, n * sizeof(double));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE]) return fail(e, SBE_ERR_DATA, "normalize: non-positive row sum in leave-one-out tables (sbayes/util.py:1006 assert)");
    return SBE_OK;
}

int sbe_observation_lh(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    int rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->slots[slot].patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F;
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    k_observation_lh<<<div_up(n, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * e->C * e->Np, e->d_pid + (int64_t)slot * e->Np,
        e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C,
        (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C, e->Fp);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, e->d_scratch, n * sizeof(double));
}

// ---- north-star scalar ----------------------------------------------------------------------------
int sbe_mixture_loglik_batch_async(sbe_engine* e, int first_slot, int n) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range [%d,%d) out of range", first_slot, first_slot + n);
    HIPCHK(e, hipSetDevice(e->device));
    return enqueue_mixture(e, first_slot, n, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS);
}

int sbe_fetch_results(sbe_engine* e, int first_slot, int n, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, out);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipStreamSynchronize(e->stream));     // results were written straight into mapped host memory
    memcpy(out, e->h_results + first_slot, (size_t)n * sizeof(double));
    return synced(e);
}

int sbe_mixture_loglik_batch(sbe_engine* e, int first_slot, int n, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, out);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipSetDevice(e->device));
    DoneSig done;
    int rc = enqueue_mixture(e, first_slot, n, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, &done);
    if (rc) return rc;
    rc = wait_done(e, done);                        // results were written straight into mapped host memory
    if (rc) return rc;
    memcpy(out, e->h_results + first_slot, (size_t)n * sizeof(double));
    return synced(e);
}

int sbe_mixture_loglik(sbe_engine* e, int slot, double* out) { return sbe_mixture_loglik_batch(e, slot, 1, out); }

// ---- collapsed likelihood -------------------------------------------------------------------------
// a7/a8 of the groups [g_lo, g_lo + G) of a slot: per-group float64 into host-mapped memory (and, optionally, the
// float32 per-feature rows into d_pf).  One launch (k_collapsed_groups) when a group's F*S terms fit LDS, the k_dcl /
// k_group_sum_f32 pair otherwise.  Ends with the stream synchronised; results at e->h_io.
static int collapsed_groups(sbe_engine* e, int slot, int g_lo, int G, float* d_pf) {
    int rc = ensure_io(e, (size_t)G * sizeof(double));
    if (rc) return rc;
    const int32_t* counts = e->d_counts + (int64_t)slot * e->table_elems();
    const size_t lds = (size_t)e->F * e->S * sizeof(double) + (size_t)e->F * sizeof(float);
    if (lds <= ((size_t)96 << 10)) {
        const DoneSig done = next_done(e, (unsigned)G);
        k_collapsed_groups<int32_t><<<G, 1024, lds, e->stream>>>(counts, e->d_conc, e->d_lg_conc, e->d_sum_a, e->d_lg_sum_a, d_pf,
                                                                  (double*)e->d_io, g_lo, e->F, e->S, done);
        HIPCHK(e, hipGetLastError());
        rc = wait_done(e, done);
        if (rc) return rc;
        return synced(e);
    } else {
        float* pf = d_pf;
        if (!pf) {
            rc = ensure_scratch(e, (size_t)G * e->F * sizeof(float));
            if (rc) return rc;
            pf = (float*)e->d_scratch;
        }
        k_dcl<int32_t><<<div_up((int64_t)G * e->F, 256), 256, 0, e->stream>>>(counts, e->d_conc, pf, g_lo, g_lo + G, e->F, e->S, 1);
        HIPCHK(e, hipGetLastError());
        k_group_sum_f32<<<div_up((int64_t)G * 8, 64), 64, 0, e->stream>>>(pf, (double*)e->d_io, G, e->F);
        HIPCHK(e, hipGetLastError());
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return synced(e);
}

int sbe_collapsed_loglik_all(sbe_engine* e, int slot, double* per_group_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_group_out);
    Slot& s = e->slots[slot];
    for (int c = 0; c < e->C; ++c) {
        if (e->G[c] == 0) continue;
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = collapsed_groups(e, slot, 0, e->Gtot, nullptr);
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)e->Gtot * sizeof(double));
    return SBE_OK;
}

int sbe_collapsed_loglik(sbe_engine* e, int slot, int component, double* per_group_out, float* per_feature_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_COMP(e, component);
    if (e->G[component] == 0) return SBE_OK;       // no group: Likelihood.compute_lh_clusters sums an empty cache
    CHECK_PTR(e, per_group_out);
    Slot& s = e->slots[slot];
    if (!s.counts_set[component]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, component);
    if (!e->conc_set[component]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", component);
    HIPCHK(e, hipSetDevice(e->device));
    const int G = e->G[component], g_lo = e->goff[component];
    float* d_pf = nullptr;
    int rc = SBE_OK;
    if (per_feature_out) {
        rc = ensure_scratch(e, (size_t)G * e->F * sizeof(float));
        if (rc) return rc;
        d_pf = (float*)e->d_scratch;
    }
    rc = collapsed_groups(e, slot, g_lo, G, d_pf);  // (the G doubles land in host-mapped memory: no copy-engine hop in the chain)
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)G * sizeof(double));
    if (per_feature_out) return d2h(e, per_feature_out, d_pf, (size_t)G * e->F * sizeof(float));
    return SBE_OK;
}


int sbe_test_fast_log(sbe_engine* e, const double* in, int n, double* out_fast, double* out_lib) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out_fast); CHECK_PTR(e, out_lib);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 3 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_f = (double*)(e->d_scratch + b);
    double* d_l = (double*)(e->d_scratch + 2 * b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_fast_log<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, d_f, d_l, n);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out_fast, d_f, (size_t)n * sizeof(double));
    if (rc) return rc;
    return d2h(e, out_lib, d_l, (size_t)n * sizeof(double));
}

int sbe_test_lgamma(sbe_engine* e, const double* in, int n, double* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_o = (double*)(e->d_scratch + b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_lgamma<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, d_o, n);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_o, (size_t)n * sizeof(double));
}


int sbe_test_roundtrip(sbe_engine* e, int n_blocks, int mode) {
    CHECK_ENGINE(e);
    if (n_blocks < 1 || n_blocks > 65535) return fail(e, SBE_ERR_ARG, "n_blocks=%d", n_blocks);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = ensure_io(e, 256 + (size_t)n_blocks * sizeof(double));
    if (rc) return rc;
    const DoneSig done = next_done(e, (unsigned)n_blocks);
    k_test_roundtrip<<<n_blocks, 64, 0, e->stream>>>((const int32_t*)e->d_io, (double*)(e->d_io + 256), mode, done);
    HIPCHK(e, hipGetLastError());
    return sync_and_report(e, done);
}

int sbe_test_tab_log(sbe_engine* e, const double* in, int n, double* out) {
    CHECK_ENGINE(e); CHECK_PTR(e, in); CHECK_PTR(e, out);
    if (n < 1) return fail(e, SBE_ERR_ARG, "n=%d", n);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t b = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    int rc = ensure_scratch(e, 2 * b);
    if (rc) return rc;
    double* d_in = (double*)e->d_scratch;
    double* d_out = (double*)(e->d_scratch + b);
    { int _urc = upload(e, d_in, in, (size_t)n * sizeof(double)); if (_urc) return _urc; }
    k_test_tab_log<<<div_up(n, 256), 256, 0, e->stream>>>(d_in, e->d_logtab, d_out, n);
    HIPCHK(e, hipGetLastError());
    return d2h(e, out, d_out, (size_t)n * sizeof(double));
}

// ---- slots ------------------------------------------------------------------------------------------
int sbe_copy_slot(sbe_engine* e, int dst, int src) {
    CHECK_ENGINE(e); CHECK_SLOT(e, dst); CHECK_SLOT(e, src);
    if (dst == src) return SBE_OK;
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t N = e->N, F = e->F, C = e->C, T = e->table_elems();
    // every per-slot array in ONE launch (twelve hipMemcpyAsync calls cost ~40 us of host time per step)
    CopySegs cs{};
    uint32_t run = 0;
    auto seg = [&](auto* ptr, int64_t elems) {
        const int64_t bytes = elems * (int64_t)sizeof(*ptr);
        cs.src[cs.n] = reinterpret_cast<const uint32_t*>(ptr + (int64_t)src * elems);
        cs.dst[cs.n] = reinterpret_cast<uint32_t*>(ptr + (int64_t)dst * elems);
        run += (uint32_t)(bytes / 4);
        cs.end[cs.n++] = run;
    };
    seg(e->d_gid, C * e->Np); seg(e->d_pid, (int64_t)e->Np); seg(e->d_src, N * e->Fp); seg(e->d_counts, T); seg(e->d_probs, T);
    seg(e->d_probs_t, e->probs_t_elems()); seg(e->d_wpat_t, e->wpat_t_elems());
    seg(e->d_tid, (int64_t)e->Np); seg(e->d_toff, (int64_t)e->Np); seg(e->d_tuple_g, (int64_t)kMaxTuples * kMaxComponents); seg(e->d_tuple_p, (int64_t)kMaxTuples);
    seg(e->d_weights, F * C); seg(e->d_wpat, (int64_t)e->Pmax * F * C); seg(e->d_patbits, (int64_t)e->Pmax);
    k_multi_copy<<<std::min<int64_t>(div_up(run, 256), 4 * e->compute_units), 256, 0, e->stream>>>(cs);
    HIPCHK(e, hipGetLastError());
    bump_src(e, dst);
    bump_ids(e, dst);
    e->slots[dst] = e->slots[src];
    return SBE_OK;
}

// ---- measurement -------------------------------------------------------------------------------------
int sbe_timer_start(sbe_engine* e) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    return SBE_OK;
}

int sbe_timer_stop(sbe_engine* e, float* elapsed_ms) {
    CHECK_ENGINE(e); CHECK_PTR(e, elapsed_ms);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return SBE_OK;
}

int sbe_timer_mark(sbe_engine* e) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    return SBE_OK;
}

int sbe_timer_elapsed(sbe_engine* e, float* elapsed_ms) {
    CHECK_ENGINE(e); CHECK_PTR(e, elapsed_ms);
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return SBE_OK;
}

int sbe_kernel_timing(sbe_engine* e, int enable, int* n_launches, float* main_kernel_avg_ms) {
    CHECK_ENGINE(e);
    HIPCHK(e, hipSetDevice(e->device));
    if (enable == 1) { e->ev_timing = true; e->ev_used = 0; return SBE_OK; }     // start: forget earlier pairs
    if (enable == 2) { e->ev_timing = false; return SBE_OK; }                    // pause: keep the recorded pairs
    if (enable == 3) { e->ev_timing = true; return SBE_OK; }                     // resume
    if (enable != 0) return fail(e, SBE_ERR_ARG, "sbe_kernel_timing: enable=%d", enable);
    CHECK_PTR(e, n_launches); CHECK_PTR(e, main_kernel_avg_ms);
    e->ev_timing = false;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    double acc = 0.0;
    for (int it = 0; it < e->ev_used; ++it) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]));
        acc += ms;
    }
    *n_launches = e->ev_used;
    *main_kernel_avg_ms = e->ev_used ? (float)(acc / e->ev_used) : 0.f;
    e->ev_used = 0;
    return synced(e);
}

const char* sbe_last_mixture_kernel(const sbe_engine* e) { return e ? e->last_kernel : "none"; }

int sbe_profile_mixture(sbe_engine* e, int first_slot, int n, int iters, float* total_ms, float* main_kernel_avg_ms) {
    CHECK_ENGINE(e); CHECK_SLOT(e, first_slot); CHECK_PTR(e, total_ms); CHECK_PTR(e, main_kernel_avg_ms);
    if (iters < 1 || iters > 100000) return fail(e, SBE_ERR_ARG, "iters=%d", iters);
    if (n < 1 || first_slot + n > e->n_slots) return fail(e, SBE_ERR_ARG, "slot range out of range");
    HIPCHK(e, hipSetDevice(e->device));
    // one event pair per launch around the dominant kernel only (the fused gather/log/reduce
    // kernel); the small fixed-order partial reduction that follows is outside the pair.
    while ((int)e->ev_pool.size() < 2 * iters) {
        hipEvent_t ev;
        HIPCHK(e, hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    const int mode = e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS;
    int rc = enqueue_mixture(e, first_slot, n, mode);     // resolves lazily-built state
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    for (int it = 0; it < iters; ++it) {
        rc = launch_mixture(e, first_slot, n, mode, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]);
        if (rc) return rc;
    }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(total_ms, e->ev0, e->ev1));
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[2 * it], e->ev_pool[2 * it + 1]));
        acc += ms;
    }
    *main_kernel_avg_ms = (float)(acc / iters);
    return SBE_OK;
}

}  // extern "C"
