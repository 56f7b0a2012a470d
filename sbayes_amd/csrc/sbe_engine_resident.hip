// sbe_engine_resident.hip -- unit 3 of 4: the delta / resident forms the drop-in host layer calls once per operator step (count
// deltas, row setters, the fused kept-observations likelihood and Gibbs resampling, cluster-membership marginals, jump scores,
// source prior, the exact per-observation form).
#include "sbe_engine_internal.hip.h"

extern "C" {

// ---- round 3: delta / resident forms for the drop-in host layer ----------------------------------------------------
// What the unchanged reference sampler asks per MCMC step goes over PCIe as object lists and a few changed rows
// (last row): no [N][F] mask, no whole [G][F][S] table.

int sbe_set_uniform_counts(sbe_engine* e, const double* unif_counts) {
    CHECK_ENGINE(e); CHECK_PTR(e, unif_counts);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload(e, e->d_unif_res, unif_counts, (size_t)e->F * e->S * sizeof(double));
    if (rc) return rc;
    e->unif_set = true;
    return SBE_OK;
}

// `follow_slot` >= 0 (sbe_counts_delta_apply): the slot's resident counts -- the OLD state's -- take the difference, and with
// `follow_probs` the probability rows of the touched groups are rebuilt: inside the tile kernel (the usual case), by one
// more kernel behind the general one.
static int counts_delta_impl(sbe_engine* e, int follow_slot, int follow_probs, int follow_source, const int32_t* objects, int n_subset, const int32_t* gid_old,
                             const int32_t* gid_new, const uint8_t* src_old, const uint8_t* src_new, const int32_t* touched, int n_touched,
                             float* out_diff) {
    CHECK_ENGINE(e);
    if (n_subset < 0 || n_touched < 0) return fail(e, SBE_ERR_ARG, "n_subset=%d n_touched=%d", n_subset, n_touched);
    if (n_touched == 0) return SBE_OK;
    CHECK_PTR(e, touched); CHECK_PTR(e, out_diff);
    const int F = e->F, S = e->S, C = e->C;
    const size_t out_bytes = (size_t)n_touched * F * S * sizeof(float);
    if (n_subset == 0) { memset(out_diff, 0, out_bytes); return SBE_OK; }
    CHECK_PTR(e, objects); CHECK_PTR(e, gid_old); CHECK_PTR(e, gid_new); CHECK_PTR(e, src_old); CHECK_PTR(e, src_new);
    int rc = check_objects(e, objects, n_subset);
    if (rc) return rc;
    std::vector<int32_t> comp(n_touched);
    for (int t = 0; t < n_touched; ++t) {
        if (touched[t] < 0 || touched[t] >= e->Gtot) return fail(e, SBE_ERR_ARG, "touched group %d out of range [0,%d)", touched[t], e->Gtot);
        int c = 0;
        while (c + 1 < C && touched[t] >= e->goff[c + 1]) ++c;
        comp[t] = c;
    }
    for (int64_t i = 0; i < (int64_t)C * n_subset; ++i)
        if (gid_old[i] < -1 || gid_old[i] >= e->Gtot || gid_new[i] < -1 || gid_new[i] >= e->Gtot)
            return fail(e, SBE_ERR_ARG, "group index out of range in the subset's ids");
    DeltaFollow follow{};
    if (follow_slot >= 0) {
        const Slot& sl = e->slots[follow_slot];
        for (int t = 0; t < n_touched; ++t) {
            if (!sl.counts_set[comp[t]])
                return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set (sbe_counts_delta_apply adds to resident tables)", follow_slot, comp[t]);
            if (follow_probs && (!sl.probs_set[comp[t]] || !e->conc_set[comp[t]]))
                return fail(e, SBE_ERR_STATE, "slot %d: probability tables / concentration of component %d not set (update_probs = 1 rebuilds "
                            "the rows of tables that exist: sbe_update_probs first)", follow_slot, comp[t]);
        }
        if (follow_source && !sl.source_set)
            return fail(e, SBE_ERR_STATE, "slot %d: source not set (update_source = 1 patches resident rows)", follow_slot);
        if (follow_source) follow.src = e->d_src + (int64_t)follow_slot * e->N * e->Fp;
        follow.counts = e->d_counts + (int64_t)follow_slot * e->table_elems();
        if (follow_probs) {
            follow.conc = e->d_conc; follow.probs = e->d_probs + (int64_t)follow_slot * e->table_elems();
            follow.probs_t = e->d_probs_t + (int64_t)follow_slot * e->probs_t_elems(); follow.status = e->d_status; follow.ft = e->ft;
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (follow.probs) { rc = clear_status_word(e, ST_BAD_NORMALIZE); if (rc) return rc; }
    // inputs (a few KB) in host-mapped memory, read by the kernel in place; the diff rows come back the same way when
    // they are small, through the staging copy otherwise
    const size_t ob = al256((size_t)n_subset * 4), gb = al256((size_t)C * n_subset * 4), sb = al256((size_t)n_subset * F);
    const size_t tb = al256((size_t)n_touched * 4);
    const bool mapped_out = out_bytes <= ((size_t)1 << 18);
    rc = ensure_io(e, ob + 2 * gb + 2 * sb + 2 * tb + (mapped_out ? out_bytes : 0));
    if (rc) return rc;
    uint8_t* h = e->h_io;
    size_t o = 0;
    const size_t o_obj = o;  memcpy(h + o, objects, (size_t)n_subset * 4); o += ob;
    const size_t o_go = o;   memcpy(h + o, gid_old, (size_t)C * n_subset * 4); o += gb;
    const size_t o_gn = o;   memcpy(h + o, gid_new, (size_t)C * n_subset * 4); o += gb;
    const size_t o_so = o;   memcpy(h + o, src_old, (size_t)n_subset * F); o += sb;
    const size_t o_sn = o;   memcpy(h + o, src_new, (size_t)n_subset * F); o += sb;
    const size_t o_t = o;    memcpy(h + o, touched, (size_t)n_touched * 4); o += tb;
    const size_t o_tc = o;   memcpy(h + o, comp.data(), (size_t)n_touched * 4); o += tb;
    // small subsets (the usual update_feature_counts: a few dozen objects): one launch, a block per 16-feature tile stages
    // what it needs of the mapped block into LDS in one PCIe round trip and serves every touched group (k_counts_delta_tile)
    const size_t tile_lds = ((size_t)n_touched * kDeltaFT * S + (size_t)e->Gtot + (size_t)n_touched + (size_t)n_subset * (1 + 2 * C)) * sizeof(int32_t) +
                            (size_t)2 * n_subset * kDeltaFT;
    if (mapped_out && n_subset <= kDeltaTileMaxN && tile_lds <= ((size_t)64 << 10) && e->opt_fuse_tables) {
        const uint8_t* din = e->d_io;
        float* d_out = (float*)(e->d_io + o);
        const unsigned blocks = (unsigned)div_up(F, kDeltaFT);
        const DoneSig done = next_done(e, blocks);
        k_counts_delta_tile<<<blocks, kBlock, tile_lds, e->stream>>>(
            e->d_state, (const int32_t*)(din + o_obj), n_subset, (const int32_t*)(din + o_go), (const int32_t*)(din + o_gn),
            din + o_so, din + o_sn, (const int32_t*)(din + o_t), n_touched, d_out, F, S, e->Fp, C, e->Gtot, done, follow);
        HIPCHK(e, hipGetLastError());
        rc = wait_done(e, done);                             // (the difference is complete; the slot follows behind the flag)
        if (!rc) rc = synced(e);
        if (rc) return rc;
        memcpy(out_diff, h + o, out_bytes);
        // the rebuilt rows may raise normalize's data check after the flag: reported like a setter's (deferred mode: by the
        // next call that waits for the device)
        return follow.probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_counts_delta_apply") : SBE_OK;
    }
    // larger subsets: the kernel walks the listed objects one after another (ids, then the object's rows): out of
    // host-mapped memory every step of that walk would be a PCIe round trip, so the packed inputs go to device memory with
    // ONE copy from the pinned block; the diff rows come back through the mapped block (posted writes)
    const size_t in_bytes = o;
    const bool out_in_block = mapped_out && !follow.counts;     // (a following slot reads the rows after the call has returned: device memory)
    rc = ensure_scratch(e, in_bytes + (out_in_block ? 0 : out_bytes));
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(e->d_scratch, h, in_bytes, hipMemcpyHostToDevice, e->stream));
    const uint8_t* din = e->d_scratch;
    float* d_out = out_in_block ? (float*)(e->d_io + o) : (float*)(e->d_scratch + in_bytes);
    const DoneSig done = out_in_block ? next_done(e, (unsigned)(n_touched * div_up(F, kDeltaFT))) : DoneSig{};
    k_counts_delta<<<dim3(n_touched, div_up(F, kDeltaFT)), kBlock, (size_t)kDeltaFT * S * sizeof(int32_t), e->stream>>>(
        e->d_state, (const int32_t*)(din + o_obj), n_subset, (const int32_t*)(din + o_go), (const int32_t*)(din + o_gn),
        din + o_so, din + o_sn, (const int32_t*)(din + o_t), (const int32_t*)(din + o_tc), d_out, F, S, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    if (follow.counts) {
        k_add_count_rows<<<div_up((int64_t)n_touched * F, 256), 256, 0, e->stream>>>(d_out, (const int32_t*)(din + o_t), n_touched, F, S, e->Gtot, follow);
        HIPCHK(e, hipGetLastError());
    }
    if (follow.src) {
        k_set_source_ids<<<div_up((int64_t)n_subset * F, 256), 256, 0, e->stream>>>(din + o_sn, (const int32_t*)(din + o_obj), n_subset, F, e->Fp, follow.src);
        HIPCHK(e, hipGetLastError());
    }
    if (!out_in_block) {
        rc = d2h(e, out_diff, d_out, out_bytes);
        if (rc || !follow.probs) return rc;
        return sync_and_report(e);
    }
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(out_diff, h + o, out_bytes);
    return synced(e);
}

int sbe_counts_delta(sbe_engine* e, const int32_t* objects, int n_subset, const int32_t* gid_old, const int32_t* gid_new,
                     const uint8_t* src_old, const uint8_t* src_new, const int32_t* touched, int n_touched, float* out_diff) {
    return counts_delta_impl(e, -1, 0, 0, objects, n_subset, gid_old, gid_new, src_old, src_new, touched, n_touched, out_diff);
}

int sbe_counts_delta_apply(sbe_engine* e, int slot, int update_probs, int update_source, const int32_t* objects, int n_subset,
                           const int32_t* gid_old, const int32_t* gid_new, const uint8_t* src_old, const uint8_t* src_new,
                           const int32_t* touched, int n_touched, float* out_diff) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    return counts_delta_impl(e, slot, update_probs, update_source, objects, n_subset, gid_old, gid_new, src_old, src_new, touched, n_touched, out_diff);
}

static int set_counts_rows_impl(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows, bool with_probs) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_rows < 0) return fail(e, SBE_ERR_ARG, "n_rows=%d", n_rows);
    if (n_rows == 0) return SBE_OK;
    CHECK_PTR(e, group_idx); CHECK_PTR(e, rows);
    for (int i = 0; i < n_rows; ++i)
        if (group_idx[i] < 0 || group_idx[i] >= e->Gtot) return fail(e, SBE_ERR_ARG, "group index %d out of range [0,%d)", group_idx[i], e->Gtot);
    // rows PATCH a table: the component's counts must be resident already (sbe_set_counts / sbe_recount / a step),
    // else the patched rows would sit among uninitialised ones and counts_set would stay false
    for (int i = 0; i < n_rows; ++i) {
        int c = 0;
        while (c + 1 < e->C && group_idx[i] >= e->goff[c + 1]) ++c;
        if (!e->slots[slot].counts_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set (sbe_set_counts_rows patches resident tables: "
                        "send the component whole with sbe_set_counts first); row for group %d refused", slot, c, group_idx[i]);
        if (with_probs && (!e->slots[slot].probs_set[c] || !e->conc_set[c]))
            return fail(e, SBE_ERR_STATE, "slot %d: probability tables / concentration of component %d not set (sbe_set_counts_rows_probs "
                        "rebuilds the rows of tables that exist: sbe_update_probs first); row for group %d refused", slot, c, group_idx[i]);
    }
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t fs = (int64_t)e->F * e->S;
    const size_t rb = al256((size_t)n_rows * fs * sizeof(float));
    int rc = ensure_scratch(e, rb + (size_t)n_rows * sizeof(int32_t));
    if (rc) return rc;
    const void *v_rows, *v_idx;
    rc = stage(e, rows, (size_t)n_rows * fs * sizeof(float), e->d_scratch, &v_rows);
    if (rc) return rc;
    rc = stage(e, group_idx, (size_t)n_rows * sizeof(int32_t), e->d_scratch + rb, &v_idx);
    if (rc) return rc;
    if (with_probs) { rc = clear_status_word(e, ST_BAD_NORMALIZE); if (rc) return rc; }
    const int kind = !with_probs ? 0 : e->S <= 8 ? 8 : e->S <= 16 ? 16 : 1;
    const int lanes_per_row = kind == 8 ? 8 : kind == 16 ? 16 : 1;
    const int64_t n_threads = kind == 0 ? (int64_t)n_rows * fs : (int64_t)n_rows * e->F * lanes_per_row;
    if (e->batch && e->batch->n_rows_blocks == 0 && v_rows != (const void*)e->d_scratch) {     // (staged in the ring: sbe_set_slot_delta launches it)
        SetterJobs& j = *e->batch;
        j.rows = CountRowsArgs{(const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc,
                               e->d_probs + (int64_t)slot * e->table_elems(), e->d_probs_t + (int64_t)slot * e->probs_t_elems(), n_rows,
                               e->F, e->S, e->Gtot, e->ft, e->d_status};
        j.rows_kind = kind; j.n_rows_blocks = (unsigned)div_up(n_threads, 256);
        return with_probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_set_counts_rows_probs") : SBE_OK;
    }
    if (!with_probs) {
        k_set_count_rows<<<div_up((int64_t)n_rows * fs, 256), 256, 0, e->stream>>>(
            (const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), n_rows, fs);
        HIPCHK(e, hipGetLastError());
        return SBE_OK;
    }
    auto launch = [&](auto kernel) {
        kernel<<<div_up(n_threads, 256), 256, 0, e->stream>>>(
            (const float*)v_rows, (const int32_t*)v_idx, e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc,
            e->d_probs + (int64_t)slot * e->table_elems(), e->d_probs_t + (int64_t)slot * e->probs_t_elems(), n_rows, e->F, e->S, e->Gtot,
            e->ft, e->d_status);
    };
    if (kind == 8) launch(k_set_count_rows_probs_x<8>);
    else if (kind == 16) launch(k_set_count_rows_probs_x<16>);
    else launch(k_set_count_rows_probs);
    HIPCHK(e, hipGetLastError());
    return check_after(e, ST_BAD_NORMALIZE, "sbe_set_counts_rows_probs");
}

int sbe_set_counts_rows(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows) {
    return set_counts_rows_impl(e, slot, group_idx, n_rows, rows, false);
}

int sbe_set_counts_rows_probs(sbe_engine* e, int slot, const int32_t* group_idx, int n_rows, const float* rows) {
    return set_counts_rows_impl(e, slot, group_idx, n_rows, rows, true);
}

// Several state-setting calls of one bind as ONE launch: sbe_set_groups (groups != NULL), sbe_set_counts_rows /
// sbe_set_counts_rows_probs (n_count_rows > 0), sbe_set_source_rows (n_src_rows > 0) -- the same checks, the same host-side
// work and the same results as those calls in that order; their kernels (the three jobs touch disjoint resident arrays) are
// collected and issued together (k_apply_setters) when their inputs went through the mapped ring, one by one otherwise.
int sbe_set_slot_delta(sbe_engine* e, int slot, int groups_component, const uint8_t* groups, const int32_t* count_idx, int n_count_rows,
                       const float* count_rows, int update_probs, const int32_t* src_objects, int n_src_rows, const uint8_t* src_rows) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    SetterJobs jobs{};
    const bool batching = e->opt_deferred && e->batch == nullptr;      // (immediate data checks synchronise inside every setter)
    if (batching) e->batch = &jobs;
    int rc = SBE_OK;
    if (groups) rc = sbe_set_groups(e, slot, groups_component, groups);
    if (!rc && n_count_rows) rc = set_counts_rows_impl(e, slot, count_idx, n_count_rows, count_rows, update_probs != 0);
    if (!rc && n_src_rows) rc = sbe_set_source_rows(e, slot, src_objects, n_src_rows, src_rows);
    if (batching) {
        e->batch = nullptr;
        const unsigned n_blocks = jobs.n_group_blocks + jobs.n_rows_blocks + jobs.n_src_blocks;
        if (n_blocks) {                         // (also after a later setter's error: the earlier ones' host state counts on their launch)
            k_apply_setters<<<n_blocks, 256, 0, e->stream>>>(jobs);
            if (hipGetLastError() != hipSuccess && !rc) rc = fail(e, SBE_ERR_HIP, "k_apply_setters launch failed");
        }
    }
    return rc;
}

static size_t gu_fused_lds_bytes(int R, int S, size_t in_bytes, int N) {
    return ((size_t)R * 16 * S + (size_t)(N + 31) / 32) * sizeof(int32_t) + in_bytes;
}
// `in_bytes` of the call's host-mapped block (a multiple of 256) are staged by the kernel; the arrays sit at these byte offsets
static GuFusedArgs gu_fused_args(sbe_engine* e, int slot, int i_cluster, int n_sub, int R, double temperature, double prior_temperature,
                                 const int32_t* table_offsets_host, size_t in_bytes, size_t group_idx_at, size_t hc_new_at, size_t hc_old_at) {
    GuFusedArgs fa{};
    const int C = e->C;
    fa.state = e->d_state; fa.gid = e->d_gid + (int64_t)slot * C * e->Np; fa.src = e->d_src + (int64_t)slot * e->N * e->Fp;
    fa.counts = e->d_counts + (int64_t)slot * e->table_elems();
    fa.mapped_in = reinterpret_cast<const uint32_t*>(e->d_io); fa.in_words = (int)(in_bytes / 4);
    fa.objects_word = 0; fa.group_idx_word = (int)(group_idx_at / 4); fa.hc_new_word = (int)(hc_new_at / 4); fa.hc_old_word = (int)(hc_old_at / 4);
    for (int c = 0; c < C; ++c) fa.table_offsets[c] = table_offsets_host[c];
    fa.conc = e->d_conc; fa.unif = e->d_unif_res; fa.temperature = temperature; fa.prior_temperature = prior_temperature;
    fa.status = e->d_status;
    fa.n_sub = n_sub; fa.i_cluster = i_cluster; fa.K = e->G[0]; fa.N = e->N; fa.Np = e->Np; fa.F = e->F; fa.S = e->S; fa.C = C;
    fa.Fp = e->Fp; fa.R = R;
    return fa;
}

int sbe_given_unchanged_lh(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                           double prior_temperature, float* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, out);
    const int N = e->N, F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source not set", slot);
    for (int c = 0; c < C; ++c) {
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (c > 0 && !s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    }
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    const int R = 1 + e->Gtot - K;                              // table rows: the cluster + every confounder group
    const int64_t fs = (int64_t)F * S;
    const size_t out_bytes = (size_t)n_sub * F * C * sizeof(float);
    const bool mapped_out = out_bytes <= ((size_t)1 << 18);
    // host-mapped inputs: object list | table row of each (component, subset object) | table offsets
    const size_t ob = al256((size_t)n_sub * 4), mb = 0, gb = al256((size_t)C * n_sub * 4), fb = al256((size_t)C * 4);
    rc = ensure_io(e, ob + mb + gb + fb + (mapped_out ? out_bytes : 0));
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * 4);
    int32_t* gi = (int32_t*)(h + ob + mb);
    int32_t* off = (int32_t*)(h + ob + mb + gb);
    for (int c = 0; c < C; ++c) {
        off[c] = c == 0 ? 0 : 1 + e->goff[c] - K;
        for (int i = 0; i < n_sub; ++i) {
            if (c == 0) { gi[i] = 0; continue; }                // every subset object sees the cluster's table (operators.py:884)
            const uint16_t gg = s.h_gid[(size_t)c * N + objects[i]];
            gi[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int)gg - e->goff[c];
        }
    }
    const size_t cb = al256((size_t)R * fs * sizeof(float));
    rc = ensure_scratch(e, cb + (mapped_out ? 0 : out_bytes));
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    float* d_out = mapped_out ? (float*)(e->d_io + ob + mb + gb + fb) : (float*)(e->d_scratch + cb);
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    // one launch (k_given_unchanged_fused: tables built per 16-feature tile in LDS and consumed there) when the tile's
    // image fits; otherwise -- or with SBE_OPT_FUSE_TABLES off -- the table kernel and the gather, two launches
    const size_t fused_lds = gu_fused_lds_bytes(R, S, ob + mb + gb, N);
    if (e->opt_fuse_tables && mapped_out && fused_lds <= kGuFusedLdsMax) {
        GuFusedArgs fa = gu_fused_args(e, slot, i_cluster, n_sub, R, temperature, prior_temperature, off, ob + mb + gb, ob + mb, 0, 0);
        const double inv_t = 1.0 / temperature;
        fa.out = d_out; fa.inv_t = (float)inv_t; fa.use_pow = inv_t != 1.0;
        const unsigned blocks = (unsigned)div_up(F, 16);
        const DoneSig done = next_done(e, blocks);
        k_given_unchanged_fused<false><<<blocks, kUnchangedBlock, fused_lds, e->stream>>>(fa, GuGibbsArgs{}, nullptr, nullptr, nullptr, done);
        HIPCHK(e, hipGetLastError());
        rc = sync_and_report(e, done);
        if (rc) return rc;
        memcpy(out, h + ob + mb + gb + fb, out_bytes);
        return SBE_OK;
    }
    // kept counts and their conditional_effect_mean (conditionals.py:105-122) in one launch: the cluster's row with the
    // cluster prior, the confounder rows with theirs
    const bool list_in_lds = (size_t)n_sub * sizeof(int32_t) <= ((size_t)32 << 10);
    if (((size_t)16 * S + (N + 31) / 32) * sizeof(int32_t) > ((size_t)96 << 10))
        return fail(e, SBE_ERR_ARG, "component_likelihood_given_unchanged: %d objects x %d states exceed the kernel's LDS image", N, S);
    k_unchanged_counts<<<dim3(R, div_up(F, 16)), kUnchangedBlock,
                         ((size_t)16 * S + (list_in_lds ? n_sub : 0) + (N + 31) / 32) * sizeof(int32_t), e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_src + (int64_t)slot * N * e->Fp,
        e->d_counts + (int64_t)slot * e->table_elems(), (const int32_t*)e->d_io, n_sub, e->d_comp_of_group,
        i_cluster, K, N, e->Np, F, S, e->Fp, e->d_conc, e->d_unif_res, temperature, prior_temperature, e->d_status, d_tab,
        list_in_lds ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    const double inv_t = 1.0 / temperature;
    const DoneSig done = mapped_out ? next_done(e, (unsigned)div_up((int64_t)n_sub * F, 256)) : DoneSig{};
    k_subset_lh<<<div_up((int64_t)n_sub * F, 256), 256, 0, e->stream>>>(
        e->d_state, d_tab, (const int32_t*)(e->d_io + ob + mb + gb), (const int32_t*)(e->d_io + ob + mb), (const int32_t*)e->d_io,
        n_sub, d_out, F, S, C, e->Fp, (float)inv_t, inv_t != 1.0, done);
    HIPCHK(e, hipGetLastError());
    if (!mapped_out) {
        rc = d2h(e, out, d_out, out_bytes);
        if (rc) return rc;
        return report_status(e);                    // (d2h synchronised)
    }
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, h + ob + mb + gb + fb, out_bytes);
    return SBE_OK;
}

// `gid_old` non-null: the count delta of the proposal as well (sbe_given_unchanged_gibbs_counts)
static int given_unchanged_gibbs_impl(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                                      double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                      const uint8_t* src_old, const double* z, uint8_t* src_new_out, float* sel_new_out,
                                      float* sel_back_out, const int32_t* gid_old, const int32_t* gid_new, int32_t* touched_out,
                                      int32_t* n_touched_out, float* diff_rows_out, int follow = 0, int follow_probs = 0) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot);
    const bool with_counts = gid_old != nullptr;
    if (with_counts) { CHECK_PTR(e, gid_new); CHECK_PTR(e, touched_out); CHECK_PTR(e, n_touched_out); CHECK_PTR(e, diff_rows_out); *n_touched_out = 0; }
    if (n_sub < 0) return fail(e, SBE_ERR_ARG, "n_sub=%d", n_sub);
    if (n_sub == 0) return SBE_OK;
    CHECK_PTR(e, objects); CHECK_PTR(e, hc_new); CHECK_PTR(e, hc_old); CHECK_PTR(e, src_old); CHECK_PTR(e, z);
    CHECK_PTR(e, src_new_out); CHECK_PTR(e, sel_new_out); CHECK_PTR(e, sel_back_out);
    const int N = e->N, F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    { int orc = reject_overlap(e, slot, "sbe_given_unchanged_gibbs"); if (orc) return orc; }
    for (int c = 0; c < C; ++c) {
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
        if (c > 0 && !s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
    }
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    for (int64_t i = 0; i < (int64_t)n_sub * F; ++i)
        if (src_old[i] != 0xFF && src_old[i] >= C) return fail(e, SBE_ERR_ARG, "old source component %d out of range [0,%d)", src_old[i], C);
    HIPCHK(e, hipSetDevice(e->device));
    const int R = 1 + e->Gtot - K;                              // table rows: the cluster + every confounder group
    const int64_t fs = (int64_t)F * S;
    const size_t nf = (size_t)n_sub * F;
    // host-mapped block: object list | table row per (component, subset object) | table offsets | has_components rows of
    // both samples | -- outputs: drawn component ids | selected probabilities, forward and backward.  The old source ids and
    // the uniforms (n F (1 + 8) bytes, read once, element-parallel) are staged through the ring / an upload.
    const size_t ob = al256((size_t)n_sub * 4), gb = al256((size_t)C * n_sub * 4), fb = al256((size_t)C * 4);
    const size_t hb = al256((size_t)n_sub * C);
    const size_t idb = al256(nf), selb = al256(nf * sizeof(float));
    // with the count delta: + the subset's global group ids in both samples and the touched groups (in), their rows (out)
    int n_touched = 0;
    if (with_counts) {
        if (sbeh_touched_groups(gid_old, gid_new, (int64_t)C * n_sub, e->Gtot, touched_out, &n_touched) != 0)
            return fail(e, SBE_ERR_ARG, "group index out of range in the subset's ids");
        *n_touched_out = n_touched;
    }
    const size_t gidb = with_counts ? al256((size_t)C * n_sub * 4) : 0, tchb = with_counts ? al256((size_t)std::max(n_touched, 1) * 4) : 0;
    const size_t rowb = with_counts ? al256((size_t)std::max(n_touched, 1) * fs * sizeof(float)) : 0;
    const size_t in_bytes = ob + gb + fb + 2 * hb + 2 * gidb + tchb, out_bytes = idb + 2 * selb + rowb;
    if (out_bytes > ((size_t)8 << 20)) return fail(e, SBE_ERR_ARG, "sbe_given_unchanged_gibbs: %d objects x %d features exceed the mapped result block", n_sub, F);
    rc = ensure_io(e, in_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * 4);
    int32_t* gi = (int32_t*)(h + ob);
    int32_t* off = (int32_t*)(h + ob + gb);
    for (int c = 0; c < C; ++c) {
        off[c] = c == 0 ? 0 : 1 + e->goff[c] - K;
        for (int i = 0; i < n_sub; ++i) {
            if (c == 0) { gi[i] = 0; continue; }                // every subset object sees the cluster's table (operators.py:884)
            const uint16_t gg = s.h_gid[(size_t)c * N + objects[i]];
            gi[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int)gg - e->goff[c];
        }
    }
    memcpy(h + ob + gb + fb, hc_new, (size_t)n_sub * C);
    memcpy(h + ob + gb + fb + hb, hc_old, (size_t)n_sub * C);
    const size_t o_gold = ob + gb + fb + 2 * hb, o_gnew = o_gold + gidb, o_tch = o_gnew + gidb;
    if (with_counts) {
        memcpy(h + o_gold, gid_old, (size_t)C * n_sub * 4);
        memcpy(h + o_gnew, gid_new, (size_t)C * n_sub * 4);
        memcpy(h + o_tch, touched_out, (size_t)n_touched * 4);
    }
    const size_t cb = al256((size_t)R * fs * sizeof(float)), zb = al256(nf * sizeof(double)), sob = al256(nf);
    rc = ensure_scratch(e, cb + zb + sob);
    if (rc) return rc;
    float* d_tab = (float*)e->d_scratch;
    const void *v_z, *v_so;
    rc = stage(e, z, nf * sizeof(double), e->d_scratch + cb, &v_z);
    if (rc) return rc;
    rc = stage(e, src_old, nf, e->d_scratch + cb + zb, &v_so);
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    GuGibbsArgs a{};
    a.state = e->d_state; a.tables = d_tab; a.table_offsets = (const int32_t*)(e->d_io + ob + gb);
    a.group_idx = (const int32_t*)(e->d_io + ob); a.objects = (const int32_t*)e->d_io;
    a.weights = e->d_weights + (int64_t)slot * F * C;
    a.hc_new = e->d_io + ob + gb + fb; a.hc_old = e->d_io + ob + gb + fb + hb;
    a.src_old = (const uint8_t*)v_so; a.z = (const double*)v_z;
    a.n_sub = n_sub; a.F = F; a.S = S; a.C = C; a.Fp = e->Fp;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    a.inv_t = (float)inv_t; a.inv_tp = (float)inv_tp; a.pow_lh = inv_t != 1.0; a.pow_w = inv_tp != 1.0; a.from_prior = from_prior ? 1 : 0;
    uint8_t* d_ids = e->d_io + in_bytes;
    // the slot follows the proposal (sbe_given_unchanged_gibbs_apply): its counts take the delta, the touched groups' probability
    // rows are rebuilt, the subset's source rows become the drawn ids -- behind the completion flag of the same launch
    follow = follow && with_counts && n_touched > 0;
    DeltaFollow dfollow{};
    if (follow) {
        const Slot& sl = e->slots[slot];
        for (int t = 0; t < n_touched; ++t) {
            int c = 0;
            while (c + 1 < C && touched_out[t] >= e->goff[c + 1]) ++c;
            if (follow_probs && !sl.probs_set[c])
                return fail(e, SBE_ERR_STATE, "slot %d: probability tables of component %d not set (update_probs = 1 rebuilds the rows of "
                            "tables that exist: sbe_update_probs first)", slot, c);
        }
        dfollow.counts = e->d_counts + (int64_t)slot * e->table_elems();
        dfollow.src = e->d_src + (int64_t)slot * N * e->Fp;
        if (follow_probs) {
            dfollow.conc = e->d_conc; dfollow.probs = e->d_probs + (int64_t)slot * e->table_elems();
            dfollow.probs_t = e->d_probs_t + (int64_t)slot * e->probs_t_elems(); dfollow.status = e->d_status; dfollow.ft = e->ft;
        }
    }
    const size_t fused_lds = gu_fused_lds_bytes(R, S, in_bytes, N) +
                             (with_counts ? ((size_t)n_touched * 16 * S + (size_t)e->Gtot) * sizeof(int32_t) : 0) +
                             (follow ? (size_t)n_touched * sizeof(int32_t) + (size_t)n_sub * 16 : 0);
    if (e->opt_fuse_tables && fused_lds <= kGuFusedLdsMax) {        // one launch (see sbe_given_unchanged_lh)
        GuFusedArgs fa = gu_fused_args(e, slot, i_cluster, n_sub, R, temperature, prior_temperature, off, in_bytes, ob,
                                       ob + gb + fb, ob + gb + fb + hb);
        if (with_counts) {
            fa.gid_old_word = (int)(o_gold / 4); fa.gid_new_word = (int)(o_gnew / 4); fa.n_touched = n_touched; fa.Gtot = e->Gtot;
            fa.touched = reinterpret_cast<const int32_t*>(e->d_io + o_tch);
            fa.rows_out = reinterpret_cast<float*>(d_ids + idb + 2 * selb);
            fa.follow = dfollow;
        }
        const unsigned blocks = (unsigned)div_up(F, 16);
        const DoneSig done = next_done(e, blocks);
        k_given_unchanged_fused<true><<<blocks, kUnchangedBlock, fused_lds, e->stream>>>(
            fa, a, d_ids, (float*)(d_ids + idb), (float*)(d_ids + idb + selb), done);
        HIPCHK(e, hipGetLastError());
        rc = sync_and_report(e, done);
        if (rc) return rc;
        memcpy(src_new_out, h + in_bytes, nf);
        memcpy(sel_new_out, h + in_bytes + idb, nf * sizeof(float));
        memcpy(sel_back_out, h + in_bytes + idb + selb, nf * sizeof(float));
        if (with_counts) memcpy(diff_rows_out, h + in_bytes + idb + 2 * selb, (size_t)n_touched * fs * sizeof(float));
        // (rows rebuilt behind the flag may raise normalize's data check: reported like a setter's)
        return dfollow.probs ? check_after(e, ST_BAD_NORMALIZE, "sbe_given_unchanged_gibbs_apply") : SBE_OK;
    }
    const bool list_in_lds = (size_t)n_sub * sizeof(int32_t) <= ((size_t)32 << 10);
    if (((size_t)16 * S + (N + 31) / 32) * sizeof(int32_t) > ((size_t)96 << 10))
        return fail(e, SBE_ERR_ARG, "gibbs_sample_source: %d objects x %d states exceed the kernel's LDS image", N, S);
    k_unchanged_counts<<<dim3(R, div_up(F, 16)), kUnchangedBlock,
                         ((size_t)16 * S + (list_in_lds ? n_sub : 0) + (N + 31) / 32) * sizeof(int32_t), e->stream>>>(
        e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_src + (int64_t)slot * N * e->Fp,
        e->d_counts + (int64_t)slot * e->table_elems(), (const int32_t*)e->d_io, n_sub, e->d_comp_of_group,
        i_cluster, K, N, e->Np, F, S, e->Fp, e->d_conc, e->d_unif_res, temperature, prior_temperature, e->d_status, d_tab,
        list_in_lds ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    const unsigned blocks = (unsigned)div_up((int64_t)nf, 256);
    const DoneSig done = next_done(e, blocks);
    k_given_unchanged_gibbs<<<blocks, kBlock, 0, e->stream>>>(a, d_ids, (float*)(d_ids + idb), (float*)(d_ids + idb + selb), e->d_status, done);
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(src_new_out, h + in_bytes, nf);
    memcpy(sel_new_out, h + in_bytes + idb, nf * sizeof(float));
    memcpy(sel_back_out, h + in_bytes + idb + selb, nf * sizeof(float));
    if (with_counts)       // (this shape has no one-launch form: the count delta by its own call, from the ids just drawn)
        return counts_delta_impl(e, follow ? slot : -1, follow_probs, follow ? 1 : 0, objects, n_sub, gid_old, gid_new, src_old, src_new_out,
                                 touched_out, n_touched, diff_rows_out);
    return SBE_OK;
}

int sbe_given_unchanged_gibbs(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                              double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                              const uint8_t* src_old, const double* z, uint8_t* src_new_out, float* sel_new_out,
                              float* sel_back_out) {
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int sbe_given_unchanged_gibbs_counts(sbe_engine* e, int slot, int i_cluster, const int32_t* objects, int n_sub, double temperature,
                                     double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                     const uint8_t* src_old, const double* z, const int32_t* gid_old, const int32_t* gid_new,
                                     uint8_t* src_new_out, float* sel_new_out, float* sel_back_out, int32_t* touched_out,
                                     int32_t* n_touched_out, float* diff_rows_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, gid_old);
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, gid_old, gid_new, touched_out, n_touched_out,
                                      diff_rows_out);
}

int sbe_given_unchanged_gibbs_apply(sbe_engine* e, int slot, int update_probs, int i_cluster, const int32_t* objects, int n_sub,
                                    double temperature, double prior_temperature, int from_prior, const uint8_t* hc_new, const uint8_t* hc_old,
                                    const uint8_t* src_old, const double* z, const int32_t* gid_old, const int32_t* gid_new,
                                    uint8_t* src_new_out, float* sel_new_out, float* sel_back_out, int32_t* touched_out,
                                    int32_t* n_touched_out, float* diff_rows_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, gid_old);
    return given_unchanged_gibbs_impl(e, slot, i_cluster, objects, n_sub, temperature, prior_temperature, from_prior, hc_new, hc_old,
                                      src_old, z, src_new_out, sel_new_out, sel_back_out, gid_old, gid_new, touched_out, n_touched_out,
                                      diff_rows_out, 1, update_probs);
}

int sbe_cluster_posterior_marginals(sbe_engine* e, int slot, int i_cluster, double temperature, double prior_temperature,
                                    const int32_t* objects, int n_objects_av, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    if (n_objects_av < 0) return fail(e, SBE_ERR_ARG, "n_objects=%d", n_objects_av);
    if (n_objects_av == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    const int F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_cluster < 0 || i_cluster >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [0,%d)", i_cluster, K);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_objects_av);
    if (rc) return rc;
    rc = check_slot_ready(e, slot, true);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.counts_set[0] || !e->conc_set[0]) return fail(e, SBE_ERR_STATE, "slot %d: cluster counts / concentration not set", slot);
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t fs = (int64_t)F * S;
    // the candidate table: conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior, T) (operators.py:1046-1052)
    // from the slot's resident counts -- nothing table-sized crosses PCIe.  Fused form (k_cluster_marginals_ws): builder
    // waves of every block put it into LDS while the block's object waves run their load chains; otherwise -- table beyond
    // 64 KB, more than kWsC components, SBE_OPT_FUSE_TABLES off -- a table kernel in front of the block-per-object kernel.
    const size_t cand_bytes = (size_t)fs * sizeof(float);
    const bool fused = e->opt_fuse_tables && C <= kWsC && cand_bytes <= ((size_t)64 << 10);
    const size_t ob = al256((size_t)n_objects_av * sizeof(int32_t));
    const size_t out_bytes = (size_t)2 * n_objects_av * sizeof(double);
    rc = ensure_io(e, ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_objects_av * sizeof(int32_t));
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const int32_t* cnt = e->d_counts + (int64_t)slot * e->table_elems();
    const double inv = 1.0 / prior_temperature;
    DoneSig done;
    if (fused) {
        InlineTables tin{};
        tin.row[0] = RowSource{cnt + (int64_t)i_cluster * fs, e->d_conc + (int64_t)i_cluster * fs};
        tin.unif = e->d_unif_res; tin.temperature = temperature; tin.prior_temperature = prior_temperature; tin.status = e->d_status;
        tin.n_rows = F;
        const unsigned blocks = (unsigned)div_up(n_objects_av, kWsObjWaves);
        done = next_done(e, blocks);
        k_cluster_marginals_ws<<<blocks, kWsBlock, cand_bytes, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_probs + (int64_t)slot * e->table_elems(), e->d_weights + (int64_t)slot * F * C,
            e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, (const int32_t*)e->d_io, n_objects_av,
            (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done, tin);
    } else {
        rc = ensure_scratch(e, cand_bytes);
        if (rc) return rc;
        float* d_tab = (float*)e->d_scratch;
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(
            cnt, e->d_conc, e->d_unif_res, d_tab, i_cluster, i_cluster + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_cluster * fs);
        HIPCHK(e, hipGetLastError());
        done = next_done(e, (unsigned)n_objects_av);
        k_cluster_marginals<<<n_objects_av, kBlock, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_probs + (int64_t)slot * e->table_elems(), d_tab, e->d_weights + (int64_t)slot * F * C,
            e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0, (const int32_t*)e->d_io, n_objects_av,
            (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C, e->Fp, done);
    }
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + ob, out_bytes);
    return SBE_OK;
}

int sbe_jump_lh_resident(sbe_engine* e, int slot, int i_source, int i_target, double temperature, double prior_temperature,
                         const int32_t* objects, int n_members, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    if (n_members < 0) return fail(e, SBE_ERR_ARG, "n_members=%d", n_members);
    if (n_members == 0) return SBE_OK;
    CHECK_PTR(e, objects);
    const int F = e->F, S = e->S, C = e->C, K = e->G[0];
    if (i_source < 0 || i_source >= K || i_target < 0 || i_target >= K) return fail(e, SBE_ERR_ARG, "cluster index out of range");
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    int rc = check_objects(e, objects, n_members);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / weights not set", slot);
    for (int c = 0; c < C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    if (!e->unif_set) return fail(e, SBE_ERR_STATE, "uniform concentration not set (sbe_set_uniform_counts)");
    HIPCHK(e, hipSetDevice(e->device));
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t fs = (int64_t)F * S;
    const int n_conf = e->Gtot - K;
    // tempered tables of the two clusters and of every confounder group (ClusterEffectProposals.posterior_counts +
    // normalize, operators.py:1254-1259, 1364-1371) from the slot's resident counts; the reference uses the CLUSTER
    // prior's uniform concentration for every component (operators.py:1352).  Fused form (k_jump_lh_ws): builder waves
    // put them into LDS, one launch; otherwise (tables beyond 64 KB, C > kWsC, option off) three table kernels in front.
    const size_t built_bytes = (size_t)(2 + n_conf) * fs * sizeof(float);
    const bool fused = e->opt_fuse_tables && C <= kWsC && built_bytes <= ((size_t)64 << 10);
    const size_t ob = al256((size_t)n_members * sizeof(int32_t));
    const size_t out_bytes = (size_t)2 * n_members * sizeof(double);
    rc = ensure_io(e, ob + out_bytes);
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_members * sizeof(int32_t));
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const int32_t* cnt = e->d_counts + (int64_t)slot * e->table_elems();
    const double inv = 1.0 / prior_temperature;
    DoneSig done;
    if (fused) {
        InlineTables tin{};
        tin.row[0] = RowSource{cnt + (int64_t)i_source * fs, e->d_conc + (int64_t)i_source * fs};
        tin.row[1] = RowSource{cnt + (int64_t)i_target * fs, e->d_conc + (int64_t)i_target * fs};
        tin.counts = cnt; tin.conc = e->d_conc; tin.first_conf_group = K;
        tin.unif = e->d_unif_res; tin.temperature = temperature; tin.prior_temperature = prior_temperature; tin.status = e->d_status;
        tin.n_rows = (2 + n_conf) * F;
        const unsigned blocks = (unsigned)div_up(n_members, kWsObjWaves);
        done = next_done(e, blocks);
        k_jump_lh_ws<<<blocks, kWsBlock, built_bytes, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np,
            e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
            (const int32_t*)e->d_io, n_members, (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C,
            e->Fp, K, done, tin);
    } else {
        rc = ensure_scratch(e, (size_t)(2 + std::max(n_conf, 1)) * fs * sizeof(float));
        if (rc) return rc;
        float* d_ps = (float*)e->d_scratch;
        float* d_pt = d_ps + fs;
        float* d_pc = d_pt + fs;
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_ps, i_source, i_source + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_source * fs);
        HIPCHK(e, hipGetLastError());
        k_probs<int32_t><<<div_up((int64_t)F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_pt, i_target, i_target + 1, F, S,
            temperature, prior_temperature, 1, e->d_status, -(int64_t)i_target * fs);
        HIPCHK(e, hipGetLastError());
        if (n_conf > 0) {
            k_probs<int32_t><<<div_up((int64_t)n_conf * F, 256), 256, 0, e->stream>>>(cnt, e->d_conc, e->d_unif_res, d_pc, K, e->Gtot, F, S,
                temperature, prior_temperature, 1, e->d_status, -(int64_t)K * fs);
            HIPCHK(e, hipGetLastError());
        }
        done = next_done(e, (unsigned)n_members);
        k_jump_lh<<<n_members, kBlock, 0, e->stream>>>(
            e->d_state, e->d_gid + (int64_t)slot * C * e->Np, e->d_pid + (int64_t)slot * e->Np, d_pc, d_ps, d_pt,
            e->d_weights + (int64_t)slot * F * C, e->d_patbits + (int64_t)slot * e->Pmax, (float)inv, inv != 1.0 ? 1 : 0,
            (const int32_t*)e->d_io, n_members, (double*)(e->d_io + ob), reinterpret_cast<const f64x2_t*>(e->d_logtab), e->Np, F, S, C,
            e->Fp, K, done);
    }
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(out, e->h_io + ob, out_bytes);
    return SBE_OK;
}

// ---- source prior and the LikelihoodLogger row --------------------------------
int sbe_source_prior(sbe_engine* e, int slot, double* per_object_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_object_out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_scratch(e, (size_t)e->N * sizeof(double));
    if (rc) return rc;
    void* d_out;
    rc = out_target(e, (size_t)e->N * sizeof(double), e->d_scratch, &d_out);
    if (rc) return rc;
    const unsigned nb = (unsigned)div_up(e->N, 1024 / kWave);
    const DoneSig done = out_done(e, d_out, nb);
    k_source_prior<<<nb, 1024, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
        e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (double*)d_out, e->N, e->F, e->C, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    return out_fetch(e, per_object_out, d_out, (size_t)e->N * sizeof(double), done);
}

// Model.__call__ = likelihood + prior (sbayes/model/model.py:47-51): what sbe_collapsed_loglik_all and sbe_source_prior
// return, for the same slot state, in ONE launch and one synchronisation (k_collapsed_source_prior).  Shapes whose
// group terms exceed the LDS budget take the two calls one after the other.
int sbe_collapsed_and_source_prior(sbe_engine* e, int slot, double* per_group_out, double* per_object_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, per_group_out); CHECK_PTR(e, per_object_out);
    Slot& s = e->slots[slot];
    for (int c = 0; c < e->C; ++c) {
        if (e->G[c] == 0) continue;
        if (!s.counts_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts of component %d not set", slot, c);
        if (!e->conc_set[c]) return fail(e, SBE_ERR_STATE, "concentration of component %d not set", c);
    }
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    const size_t lds = (size_t)e->F * e->S * sizeof(double) + (size_t)e->F * sizeof(float);
    const size_t gb = al256((size_t)e->Gtot * sizeof(double)), out_bytes = gb + (size_t)e->N * sizeof(double);
    if (e->Gtot == 0 || lds > ((size_t)96 << 10) || out_bytes > kMappedOutMax || !poll_done_enabled()) {
        int rc = sbe_collapsed_loglik_all(e, slot, per_group_out);
        if (rc) return rc;
        return sbe_source_prior(e, slot, per_object_out);
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    rc = ensure_io(e, out_bytes);
    if (rc) return rc;
    const unsigned nb = (unsigned)e->Gtot + (unsigned)div_up(e->N, 1024 / kWave);
    const DoneSig done = next_done(e, nb);
    const SourcePriorArgs sp{e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_pid + (int64_t)slot * e->Np,
                             e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, (double*)(e->d_io + gb), e->N, e->F, e->C, e->Fp};
    k_collapsed_source_prior<<<nb, 1024, lds, e->stream>>>(e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, e->d_lg_conc,
                                                          e->d_sum_a, e->d_lg_sum_a, (double*)e->d_io, e->Gtot, e->F, e->S, sp, done);
    HIPCHK(e, hipGetLastError());
    rc = wait_done(e, done);
    if (rc) return rc;
    memcpy(per_group_out, e->h_io, (size_t)e->Gtot * sizeof(double));
    memcpy(per_object_out, e->h_io + gb, (size_t)e->N * sizeof(double));
    return synced(e);
}

int sbe_observation_lh_exact(sbe_engine* e, int slot, double* out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, slot); CHECK_PTR(e, out);
    Slot& s = e->slots[slot];
    if (!s.groups_set || !s.source_set || !s.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", slot);
    { int orc = reject_overlap(e, slot, "sbe_observation_lh_exact"); if (orc) return orc; }
    for (int c = 0; c < e->C; ++c)
        if (!s.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = SBE_OK;
    if (s.patterns_dirty) { rc = upload_patterns_and_weights(e, slot); if (rc) return rc; }
    const int64_t n = (int64_t)e->N * e->F;
    rc = ensure_scratch(e, n * sizeof(double));
    if (rc) return rc;
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    k_lh_exact<<<div_up(n, 256), 256, 0, e->stream>>>(
        e->d_state, e->d_src + (int64_t)slot * e->N * e->Fp, e->d_gid + (int64_t)slot * e->C * e->Np,
        e->d_counts + (int64_t)slot * e->table_elems(), e->d_conc, (double*)e->d_scratch, e->N, e->Np, e->F, e->S, e->C,
        e->Fp, e->d_status, e->d_wpat + (int64_t)slot * e->Pmax * e->F * e->C, e->d_pid + (int64_t)slot * e->Np);
    HIPCHK(e, hipGetLastError());
    rc = d2h(e, out, e->d_scratch, n * sizeof(double));
    if (rc) return rc;
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status[ST_BAD_NORMALIZE]) return fail(e, SBE_ERR_DATA, "normalize: non-positive row sum in leave-one-out tables (sbayes/util.py:1006 assert)");
    return SBE_OK;
}

}  // extern "C"
