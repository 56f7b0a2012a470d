// sbe_engine_steps.hip -- unit 4 of 4: one MCMC step per call (sbe_step, sbe_step_delta, sbe_gibbs_step), their batched forms over
// many chains (host worker pool, packed payloads) and the one-call Gibbs proposal.
#include "sbe_engine_internal.hip.h"

extern "C" {

// ---- one MCMC step in one call: delta in, likelihoods out (north_star: "only the proposed cluster-
// assignment delta crosses PCIe") ---------------------------------------------------------------------
namespace {
// Inputs of k_step_core that differ between the one-call MCMC step (payload from the host) and the one-call Gibbs
// step (new source sampled on the device).
struct CoreInputs {
    const void* ids_new = nullptr;       // component-0 group ids of the candidate [Np] u16, or nullptr (unchanged)
    const void* pid = nullptr; const void* tid = nullptr; const void* toff = nullptr;      // with ids_new: the tables
    const void* tuple_g = nullptr; const void* tuple_p = nullptr; const void* patbits = nullptr;   // derived from them
    const void* weights = nullptr;       // new weights [F][C] f32, or nullptr (unchanged)
    const int16_t* row_of = nullptr;     // [Np]: >= 0 marks an object whose source changes
    const uint8_t* rows = nullptr; const int32_t* objects = nullptr; int n_changed = 0;   // payload source rows
    const uint8_t* src_new = nullptr;    // device-sampled source (the candidate's array) instead of payload rows
    const int32_t* subset = nullptr; int n_subset = 0;      // objects whose counts may change
    int P = 1;                           // has_components patterns of the candidate
    // source array of the candidate slot: the whole array is copied from the current slot (full_src_copy), or only the
    // rows in which the candidate slot is known to differ from it (sbe_engine::SrcSync)
    bool full_src_copy = true; const int32_t* stale = nullptr; int n_stale = 0;
    // delta layout (sbe_step_batch_delta): patched id arrays, per-subset-entry row / cluster id (StepCore)
    const int32_t* patch_n = nullptr; const uint16_t* patch_gid = nullptr; const uint8_t* patch_pid = nullptr;
    const uint8_t* patch_tid = nullptr; int n_patch = -1;
    const int16_t* sub_row = nullptr; const uint16_t* sub_gid0 = nullptr;
};

// the single-step calls' lane: the engine's own payload / result blocks
sbe_engine::Lane lane0(sbe_engine* e) {
    return sbe_engine::Lane{e->h_step_payload, e->d_step_payload, e->h_step, e->d_step_host, e->d_step_pf, e->d_step_stamp, e->step_id,
                            e->d_status};
}

// kernel 1 of the one-call steps: candidate slot = current slot + inputs, count delta, every table.
// build_step_core fills the kernel's argument block for one chain (lane); the caller launches it.
int build_step_core(sbe_engine* e, sbe_engine::Lane& lane, int cur_slot, int cand_slot, const CoreInputs& in,
                    StepCore& a, size_t& lds_out, int& n_blocks_out, int n_chains = 1) {
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const bool regroup = in.ids_new != nullptr;
    a = StepCore{};
    uint32_t run = 0;
    auto seg = [&](auto* base, int64_t elems, const void* other_src) {       // per-slot array `base`, elems per slot
        const int64_t bytes = elems * (int64_t)sizeof(*base);
        a.cs.src[a.cs.n] = other_src ? reinterpret_cast<const uint32_t*>(other_src)
                                     : reinterpret_cast<const uint32_t*>(base + (int64_t)cur_slot * elems);
        a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(base + (int64_t)cand_slot * elems);
        run += (uint32_t)(bytes / 4);
        a.cs.end[a.cs.n++] = run;
    };
    uint16_t* g_cur = e->d_gid + (int64_t)cur_slot * C * Np;
    const bool delta = in.n_patch >= 0;          // sbe_step_batch_delta: the per-object id arrays are patched, not copied
    if (!delta) {
        // gid: component 0 from the inputs when the clusters changed; the other components from the current slot
        uint16_t* g_cand = e->d_gid + (int64_t)cand_slot * C * Np;
        a.cs.src[a.cs.n] = reinterpret_cast<const uint32_t*>(regroup ? in.ids_new : (const void*)g_cur);
        a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(g_cand);
        run += (uint32_t)(Np * 2 / 4); a.cs.end[a.cs.n++] = run;
        if (C > 1) {
            a.cs.src[a.cs.n] = reinterpret_cast<const uint32_t*>(g_cur + Np);
            a.cs.dst[a.cs.n] = reinterpret_cast<uint32_t*>(g_cand + Np);
            run += (uint32_t)((int64_t)(C - 1) * Np * 2 / 4); a.cs.end[a.cs.n++] = run;
        }
        seg(e->d_pid, (int64_t)Np, regroup ? in.pid : nullptr);
        seg(e->d_tid, (int64_t)Np, regroup ? in.tid : nullptr);
        seg(e->d_toff, (int64_t)Np, regroup ? in.toff : nullptr);
    }
    a.n_patch = in.n_patch; a.patch_n = in.patch_n; a.patch_gid = in.patch_gid; a.patch_pid = in.patch_pid; a.patch_tid = in.patch_tid;
    a.gid_dst = e->d_gid + (int64_t)cand_slot * C * Np; a.pid_dst = e->d_pid + (int64_t)cand_slot * Np;
    a.tid_dst = e->d_tid + (int64_t)cand_slot * Np; a.toff_dst = e->d_toff + (int64_t)cand_slot * Np;
    a.toff_mul = (uint32_t)(e->S + 1) * 512u;
    a.sub_row = in.sub_row; a.sub_gid0 = in.sub_gid0;
    seg(e->d_tuple_g, (int64_t)kMaxTuples * kMaxComponents, in.tuple_g);
    seg(e->d_tuple_p, (int64_t)kMaxTuples, in.tuple_p);
    seg(e->d_patbits, (int64_t)e->Pmax, in.patbits);
    seg(e->d_weights, (int64_t)F * C, in.weights);
    a.src_seg = a.cs.n;
    if (in.full_src_copy) seg(e->d_src, (int64_t)N * e->Fp, nullptr);
    a.stale = in.stale; a.n_stale = in.full_src_copy ? 0 : in.n_stale;
    a.src_cur_rows = e->d_src + (int64_t)cur_slot * N * e->Fp;
    a.row_of = in.row_of;
    a.rows = in.rows;
    a.objects = in.objects;
    a.src_dst = e->d_src + (int64_t)cand_slot * N * e->Fp;
    a.n_changed = in.n_changed; a.F = F; a.C = C; a.Fp = e->Fp; a.status = lane.d_status;
    // tile blocks
    a.state = e->d_state; a.gid_cur = g_cur;
    a.ids_new = reinterpret_cast<const uint16_t*>(in.ids_new);
    a.src_cur = e->d_src + (int64_t)cur_slot * N * e->Fp;
    a.src_new = in.src_new;
    a.subset = in.subset; a.n_subset = in.n_subset;
    a.counts_cur = e->d_counts + (int64_t)cur_slot * e->table_elems();
    a.counts_new = e->d_counts + (int64_t)cand_slot * e->table_elems();
    a.conc = e->d_conc; a.lg_conc = e->d_lg_conc; a.sum_a = e->d_sum_a; a.lg_sum_a = e->d_lg_sum_a;
    a.probs = e->d_probs + (int64_t)cand_slot * e->table_elems();
    a.probs_t = e->d_probs_t + (int64_t)cand_slot * e->probs_t_elems();
    a.per_feature = lane.d_pf;
    if (++lane.step_id == 0) {                // stamp wrap-around (2^32 steps): start over from clean stamps
        HIPCHK(e, hipMemsetAsync(lane.d_stamp, 0, e->Gtot * sizeof(uint32_t), e->stream));
        lane.step_id = 1;
    }
    a.stamp = lane.d_stamp; a.step_id = lane.step_id;
    a.Np = Np; a.S = e->S; a.Gtot = e->Gtot; a.ft = e->ft;
    a.ftc = (int)std::max<int64_t>(1, std::min<int64_t>(8, 2048 / ((int64_t)e->Gtot * e->S)));
    a.n_tile_blocks = div_up(F, a.ftc);
    // weight blocks
    a.weights = in.weights ? reinterpret_cast<const float*>(in.weights) : e->d_weights + (int64_t)cur_slot * F * C;
    a.pattern_bits = in.patbits ? reinterpret_cast<const uint32_t*>(in.patbits) : e->d_patbits + (int64_t)cur_slot * e->Pmax;
    a.wpat = e->d_wpat + (int64_t)cand_slot * e->Pmax * F * C;
    a.wpat_t = e->d_wpat_t + (int64_t)cand_slot * e->wpat_t_elems();
    a.P = in.P; a.Pmax = e->Pmax; a.n_weight_blocks = div_up((int64_t)in.P * F, kBlock);
    const int64_t E = (int64_t)e->Gtot * a.ftc * e->S, R = (int64_t)e->Gtot * a.ftc;
    const size_t lds = (size_t)((E * 4 + 15) / 16 * 16) + (size_t)(2 * E + R) * sizeof(double);
    // copy blocks: enough to fill the chip for ONE chain; a batch of chains shares it (64 chains x 68 four-KB copy
    // blocks made the batched launch workgroup-dispatch bound: 6 000 blocks, 99 us)
    a.n_copy_blocks = std::max(1, (int)std::min<int64_t>(div_up(run, 1024), std::max(4, 2 * e->compute_units / std::max(1, n_chains))));
    lds_out = lds;
    n_blocks_out = a.n_tile_blocks + a.n_weight_blocks + a.n_copy_blocks;
    return SBE_OK;
}

int launch_step_core(sbe_engine* e, int cur_slot, int cand_slot, const CoreInputs& in) {
    sbe_engine::Lane lane = lane0(e);
    StepCore a; size_t lds = 0; int n_blocks = 0;
    int rc = build_step_core(e, lane, cur_slot, cand_slot, in, a, lds, n_blocks);
    e->step_id = lane.step_id;
    if (rc) return rc;
    k_step_core<<<n_blocks, kBlock, lds, e->stream>>>(a);
    HIPCHK(e, hipGetLastError());
    return SBE_OK;
}

// the step epilogue's mapped-memory block: [Gtot] f64 | [ST_WORDS] i32 | [Gtot] u8 (padded to 8) | [2] f64 (log_q, log_q_back)
inline size_t step_host_lq_offset(const sbe_engine* e) {
    return ((size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int) + (size_t)e->Gtot + 7) / 8 * 8;
}

StepFinish make_step_finish_lane(sbe_engine* e, const sbe_engine::Lane& lane) {
    StepFinish fin{};
    fin.per_feature = lane.d_pf;
    fin.group_out = reinterpret_cast<double*>(lane.d_step_host);
    fin.status = lane.d_status;
    fin.status_out = reinterpret_cast<int*>(lane.d_step_host + (size_t)e->Gtot * sizeof(double));
    fin.changed = nullptr; fin.stamp = lane.d_stamp; fin.step_id = lane.step_id;
    fin.changed_out = lane.d_step_host + (size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int);
    fin.Gtot = e->Gtot; fin.F = e->F;
    return fin;
}
StepFinish make_step_finish(sbe_engine* e) { return make_step_finish_lane(e, lane0(e)); }

// after the synchronisation that ends a one-call step: data checks, then the results out of the mapped block
int read_step_results_lane(sbe_engine* e, const uint8_t* h_step, int cand_slot, double* group_logliks_out, double* mixture_out,
                      uint8_t* changed_groups_out, const char* bad_norm_what, int* d_status = nullptr, int chain = -1) {
    const int* hst = reinterpret_cast<const int*>(h_step + (size_t)e->Gtot * sizeof(double));
    if (hst[ST_BAD_NORMALIZE] || hst[ST_MULTI_SOURCE]) {
        const int bad_norm = hst[ST_BAD_NORMALIZE], multi_src = hst[ST_MULTI_SOURCE];
        (void)hipMemsetAsync((d_status ? d_status : e->d_status) + ST_BAD_NORMALIZE, 0, 2 * sizeof(int), e->stream);
        if (!d_status || d_status == e->d_status) e->h_flag[ST_BAD_NORMALIZE] = e->h_flag[ST_MULTI_SOURCE] = 0;
        char who[32] = "";
        if (chain >= 0) snprintf(who, sizeof who, "chain %d: ", chain);
        if (bad_norm) return fail(e, SBE_ERR_DATA, "%snormalize: %d %s have a non-positive sum (sbayes/util.py:1006 assert)", who, bad_norm, bad_norm_what);
        return fail(e, SBE_ERR_DATA, "%ssource is not one-hot over components in %d observations", who, multi_src);
    }
    memcpy(group_logliks_out, h_step, (size_t)e->Gtot * sizeof(double));
    if (changed_groups_out) memcpy(changed_groups_out, h_step + (size_t)e->Gtot * sizeof(double) + ST_WORDS * sizeof(int), (size_t)e->Gtot);
    *mixture_out = e->h_results[cand_slot];
    return SBE_OK;
}
int read_step_results(sbe_engine* e, int cand_slot, double* group_logliks_out, double* mixture_out,
                      uint8_t* changed_groups_out, const char* bad_norm_what) {
    return read_step_results_lane(e, e->h_step, cand_slot, group_logliks_out, mixture_out, changed_groups_out, bad_norm_what);
}

}  // namespace

static int step_lean(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                     int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                     double* mixture_out, uint8_t* changed_groups_out);
static int step_general(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                        int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                        double* mixture_out, uint8_t* changed_groups_out);

int sbe_step(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
             int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
             double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_changed < 0 || (n_changed > 0 && (!changed_objects || !source_rows)))
        return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing for n_changed=%d", n_changed);
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    { int orc = reject_overlap(e, cur_slot, "one-call step"); if (orc) return orc; }
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", cur_slot, c);
    for (int i = 0; i < n_changed; ++i)
        if (changed_objects[i] < 0 || changed_objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", changed_objects[i]);
    HIPCHK(e, hipSetDevice(e->device));
    // few-launch form (one H2D payload, four kernels, results through mapped memory) whenever the step fits its
    // payload; the call-by-call form otherwise (many changed rows, never-uploaded patterns) or on request
    static const bool force_general = getenv("SBE_STEP_GENERAL") && atoi(getenv("SBE_STEP_GENERAL")) == 1;
    if (!force_general && e->opt_step_form == 0 && n_changed <= e->step_max_rows && !cur.patterns_dirty &&
        (int64_t)e->Gtot * e->S * 28 <= 60 * 1024)        // (k_step_core's LDS image of one feature column)
        return step_lean(e, cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights,
                         group_logliks_out, mixture_out, changed_groups_out);
    return step_general(e, cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights,
                        group_logliks_out, mixture_out, changed_groups_out);
}

// Host half of a lean step for one chain: the candidate's host state `cd` (= current + delta) and the step's payload
// packed into the lane's host-mapped block; `in` receives the device-side views of that payload.  Touches only the lane,
// `cd` and read-only engine state, so the chains of a batch can be prepared by several host threads at once.
static int prepare_step(sbe_engine* e, const sbe_engine::Lane& lane, int cur_slot, int cand_slot, const uint8_t* clusters,
                        const int32_t* changed_objects, int n_changed, const uint8_t* source_rows, const float* weights,
                        Slot& cd, CoreInputs& in, std::string* err, std::vector<int32_t>* moved_out = nullptr) {
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const Slot& cur = e->slots[cur_slot];
    cd = cur;                                 // host state of the candidate (committed by the caller)
    // objects whose counts may change: listed source rows + objects whose cluster membership changed
    static thread_local std::vector<int32_t> mv;                      // (no allocation per step: the chains of a batch are
    static thread_local std::vector<uint16_t> mv_old;                  //  prepared by pool threads)
    mv.clear(); mv_old.clear();
    bool full_derive = false;
    const bool regroup = clusters != nullptr;
    if (regroup) {
        const int K = e->G[0];
        uint16_t* ids = cd.h_gid.data();      // component 0: group offset 0
        {
            char msg[320];
            if (!matrix_to_ids(clusters, K, N, 0, 0, ids, msg, sizeof msg)) { *err = msg; return SBE_ERR_DATA; }
        }
        const uint16_t* old_ids = cur.h_gid.data();
        for (int n = 0; n < N; ++n) if (ids[n] != old_ids[n]) { mv.push_back(n); mv_old.push_back(old_ids[n]); }
        // pattern ids and group tuples of the candidate: the moved objects' entries updated in place (cd holds the
        // current slot's tables), the full derivation when the set of patterns changes, the counts are not there
        // (slot never derived in full) or on request
        if (e->opt_step_derive == 1 || (int)mv.size() > N / 8 ||
            !update_patterns_and_tuples(e, cd, mv.data(), mv_old.data(), (int)mv.size())) {
            full_derive = true;                        // pattern ranks / tuple numbers of ANY object may change
            derive_patterns(e, cd);
            if ((int)cd.patterns.size() > e->Pmax) {
                char buf[160];
                snprintf(buf, sizeof buf, "%zu distinct has_components patterns exceed capacity %d", cd.patterns.size(), e->Pmax);
                *err = buf;
                return SBE_ERR_ARG;
            }
            derive_tuples(e, cd);
        }
        cd.patterns_dirty = false;
        cd.group_epoch = ++e->epoch_counter;
    }
    if (moved_out) {
        *moved_out = mv;
        if (full_derive) moved_out->assign(1, -1);     // "every entry may differ": commit_ids_sync leaves no usable record
    }
    // ---- payload: packed in host-mapped pinned memory; the kernels read it in place (a few tens of KB over
    // PCIe, no copy engine in the chain).  Every step ends with a stream synchronisation, so the lane's buffer is
    // free again when the next step starts.
    const auto& L = e->sl;
    uint8_t* st = lane.h_payload;
    int n_subset = 0;
    {   // sorted union of the objects that changed cluster and the objects with new source rows
        int32_t* sub = reinterpret_cast<int32_t*>(st + L.subset);
        if (n_changed == 0) { memcpy(sub, mv.data(), mv.size() * sizeof(int32_t)); n_subset = (int)mv.size(); }
        else {
            static thread_local std::vector<int32_t> ch;
            ch.assign(changed_objects, changed_objects + n_changed);
            std::sort(ch.begin(), ch.end());
            ch.erase(std::unique(ch.begin(), ch.end()), ch.end());
            n_subset = (int)(std::set_union(mv.begin(), mv.end(), ch.begin(), ch.end(), sub) - sub);
        }
    }
    if (regroup) {
        memcpy(st + L.ids, cd.h_gid.data(), (size_t)N * 2);
        if (Np > N) memset(st + L.ids + (size_t)N * 2, 0xFF, (size_t)(Np - N) * 2);
        memset(st + L.pid, 0, Np); memcpy(st + L.pid, cd.h_pid.data(), N);
        memcpy(st + L.tid, cd.h_tid.data(), Np);
        memcpy(st + L.toff, cd.h_toff.data(), (size_t)Np * 4);
        memcpy(st + L.tuple_g, cd.h_tuple_g.data(), (size_t)kMaxTuples * kMaxComponents * 2);
        memcpy(st + L.tuple_p, cd.h_tuple_p.data(), kMaxTuples);
        memset(st + L.patbits, 0, (size_t)e->Pmax * 4);
        memcpy(st + L.patbits, cd.patterns.data(), cd.patterns.size() * 4);
    }
    if (weights) memcpy(st + L.weights, weights, (size_t)F * C * 4);
    if (n_changed > 0) {
        int16_t* row_of = reinterpret_cast<int16_t*>(st + L.row_of);
        std::fill(row_of, row_of + Np, (int16_t)-1);
        for (int i = 0; i < n_changed; ++i) row_of[changed_objects[i]] = (int16_t)i;    // (a repeated object: last row wins)
        memcpy(st + L.objects, changed_objects, (size_t)n_changed * 4);
        memcpy(st + L.rows, source_rows, (size_t)n_changed * F * C);
    }
    const uint8_t* pl = lane.d_payload;
    in = CoreInputs{};
    if (regroup) {
        in.ids_new = pl + L.ids; in.pid = pl + L.pid; in.tid = pl + L.tid; in.toff = pl + L.toff;
        in.tuple_g = pl + L.tuple_g; in.tuple_p = pl + L.tuple_p; in.patbits = pl + L.patbits;
    }
    if (weights) in.weights = pl + L.weights;
    if (n_changed > 0) {
        in.row_of = reinterpret_cast<const int16_t*>(pl + L.row_of);
        in.rows = pl + L.rows;
        in.objects = reinterpret_cast<const int32_t*>(pl + L.objects);
        in.n_changed = n_changed;
    }
    in.subset = reinterpret_cast<const int32_t*>(pl + L.subset); in.n_subset = n_subset;
    in.P = (int)cd.patterns.size();
    {   // source array: rows to bring over from the current slot (sbe_engine::SrcSync), or the whole array
        const sbe_engine::SrcSync& rec = e->src_sync[cand_slot];
        const sbe_engine::SrcSync& cs = e->src_sync[cur_slot];
        if (rec.peer == cur_slot && rec.peer_version == cs.version && rec.own_version == rec.version &&
            (int)rec.diff.size() <= e->step_max_rows) {
            int32_t* stale = reinterpret_cast<int32_t*>(st + L.stale);
            const int16_t* row_of = n_changed > 0 ? reinterpret_cast<const int16_t*>(st + L.row_of) : nullptr;
            int ns = 0;
            for (int32_t n : rec.diff)                       // (rows this step rewrites anyway are left to it)
                if (!row_of || row_of[n] < 0) stale[ns++] = n;
            in.full_src_copy = false;
            in.stale = reinterpret_cast<const int32_t*>(pl + L.stale);
            in.n_stale = ns;
        }
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    cd.weights_set = true;
    return SBE_OK;
}

// ... and its id arrays = the current slot's except the moved objects' entries
static void commit_ids_sync(sbe_engine* e, int cur_slot, int cand_slot, const std::vector<int32_t>& moved) {
    sbe_engine::SrcSync& rc = e->ids_sync[cand_slot];
    sbe_engine::SrcSync& cu = e->ids_sync[cur_slot];
    ++rc.version;
    if (moved.size() == 1 && moved[0] < 0) {           // the candidate's tables were derived afresh (prepare_step): the two
        rc.peer = cu.peer = -1;                        // slots' pattern / tuple numbering is unrelated from here on
        rc.diff.clear(); cu.diff.clear();
        return;
    }
    rc.peer = cur_slot; rc.peer_version = cu.version; rc.own_version = rc.version; rc.diff = moved;
    cu.peer = cand_slot; cu.peer_version = rc.version; cu.own_version = cu.version; cu.diff = moved;
}

// after a one-call step was enqueued: the candidate's source = the current slot's except the rows the step wrote
static void commit_src_sync(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* changed_objects, int n_changed) {
    sbe_engine::SrcSync& rc = e->src_sync[cand_slot];
    sbe_engine::SrcSync& cu = e->src_sync[cur_slot];
    ++rc.version;
    rc.peer = cur_slot; rc.peer_version = cu.version; rc.own_version = rc.version;
    rc.diff.assign(changed_objects, changed_objects + n_changed);
    cu.peer = cand_slot; cu.peer_version = rc.version; cu.own_version = cu.version;
    cu.diff = rc.diff;
}

static int step_lean(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                     int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                     double* mixture_out, uint8_t* changed_groups_out) {
    if (e->status_pending) {                  // deliver a deferred data check before this step reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    // SBE_STEP_TIMING=1: host-side phase times (prepare / enqueue / wait), printed every 2000 steps (diagnostic)
    static const bool timing = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) == 1;
    static double t_acc[3] = {0, 0, 0};
    static int t_n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    Slot cd;
    CoreInputs in;
    {
        std::string err;
        int rc = prepare_step(e, lane0(e), cur_slot, cand_slot, clusters, changed_objects, n_changed, source_rows, weights, cd, in, &err,
                              &e->step_moved);
        if (rc) return fail(e, rc, "%s", err.c_str());
    }
    const auto t1 = std::chrono::steady_clock::now();
    // ---- kernel 1: candidate slot = current slot + payload, count delta, every table ---------------------------
    int rc = launch_step_core(e, cur_slot, cand_slot, in);
    if (rc) return rc;
    commit_src_sync(e, cur_slot, cand_slot, changed_objects, n_changed);
    commit_ids_sync(e, cur_slot, cand_slot, e->step_moved);
    e->slots[cand_slot] = cd;
    // ---- kernels 2 + 3: fused mixture eval, reduction + step epilogue (mapped-memory results) -------------------
    StepFinish fin = make_step_finish(e);
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    rc = wait_done(e, done);
    if (rc) return rc;
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        t_acc[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t_acc[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
        t_acc[2] += std::chrono::duration<double, std::micro>(t3 - t2).count();
        if (++t_n == 2000) {
            fprintf(stderr, "[sbe_step] host prepare %.1f us, enqueue %.1f us, wait %.1f us per step\n", t_acc[0] / t_n, t_acc[1] / t_n, t_acc[2] / t_n);
            t_acc[0] = t_acc[1] = t_acc[2] = 0; t_n = 0;
        }
    }
    return read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "rows");
}

// ---- batched multi-chain step (missing #4): B chains' deltas in ONE call ------------------------------
// The reference steps its chains one after the other in one Python loop (MCMC.generate_samples,
// sbayes/sampling/mcmc.py:237-241).  Here the chains' candidate slots are built by ONE launch of k_step_core_batch
// (chain <-> blockIdx.y), evaluated by ONE launch of the fused mixture kernel over the candidate slot list and
// finished by ONE launch of k_reduce_partials (a reduction block and an epilogue block per chain); the host halves
// (candidate host state, payload packing) run on a small pool of worker threads.  One synchronisation per batch.
namespace {
int ensure_lanes(sbe_engine* e, int n) {
    const size_t hb = step_host_lq_offset(e) + 2 * sizeof(double);
    while ((int)e->lanes.size() < n) {
        sbe_engine::Lane ln{};
        // (no payload block of its own: a batch's payloads live back to back in h_batch_payload / d_batch_payload)
        HIPCHK(e, hipHostMalloc((void**)&ln.h_step, hb, hipHostMallocMapped));
        memset(ln.h_step, 0, hb);
        HIPCHK(e, hipHostGetDevicePointer((void**)&ln.d_step_host, ln.h_step, 0));
        HIPCHK(e, hipMalloc((void**)&ln.d_pf, (size_t)e->Gtot * e->F * sizeof(float)));
        HIPCHK(e, hipMalloc((void**)&ln.d_stamp, (size_t)e->Gtot * sizeof(uint32_t)));
        HIPCHK(e, hipMemsetAsync(ln.d_stamp, 0, (size_t)e->Gtot * sizeof(uint32_t), e->stream));
        HIPCHK(e, hipMalloc((void**)&ln.d_status, ST_WORDS * sizeof(int)));
        HIPCHK(e, hipMemsetAsync(ln.d_status, 0, ST_WORDS * sizeof(int), e->stream));
        ln.step_id = 0;
        e->lanes.push_back(ln);
    }
    return SBE_OK;
}
}  // namespace


int sbe_step_batch(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                   const uint8_t* clusters, const uint8_t* clusters_mask, const int32_t* rows_ptr,
                   const int32_t* changed_objects, const uint8_t* source_rows, const float* weights,
                   const uint8_t* weights_mask, double* group_logliks_out, double* mixture_out,
                   uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, cur_slots); CHECK_PTR(e, cand_slots); CHECK_PTR(e, rows_ptr);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (n_chains < 1 || n_chains > e->n_slots / 2) return fail(e, SBE_ERR_ARG, "n_chains=%d (1..%d: two slots per chain)", n_chains, e->n_slots / 2);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024) return fail(e, SBE_ERR_ARG, "sbe_step_batch: tables too large for the one-launch step (G_total=%d, S=%d)", e->Gtot, e->S);
    const int N = e->N, F = e->F, C = e->C, K = e->G[0];
    if (rows_ptr[0] != 0) return fail(e, SBE_ERR_ARG, "rows_ptr[0] must be 0");
    // SBE_STEP_TIMING=1: per-phase wall clock of this call on stderr (tools/prof_step_batch.py)
    static const bool timing = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) != 0;
    using clk = std::chrono::steady_clock;
    clk::time_point tp[12]; int ntp = 0;
    auto mark = [&] { if (timing && ntp < 12) tp[ntp++] = clk::now(); };
    mark();
    {   // argument checks before anything is touched
        std::vector<uint8_t> used(e->n_slots, 0);
        for (int i = 0; i < n_chains; ++i) {
            const int a = cur_slots[i], b = cand_slots[i];
            if (a < 0 || a >= e->n_slots || b < 0 || b >= e->n_slots || a == b) return fail(e, SBE_ERR_ARG, "chain %d: bad slots (%d, %d)", i, a, b);
            if (used[a] || used[b]) return fail(e, SBE_ERR_ARG, "chain %d: slot used by another chain of the batch", i);
            used[a] = used[b] = 1;
            const int nr = rows_ptr[i + 1] - rows_ptr[i];
            if (nr < 0 || nr > e->step_max_rows) return fail(e, SBE_ERR_ARG, "chain %d: %d changed source rows (0..%d per chain in a batched step)", i, nr, e->step_max_rows);
            if (nr > 0 && (!changed_objects || !source_rows)) return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing");
            for (int k = rows_ptr[i]; k < rows_ptr[i + 1]; ++k)
                if (changed_objects[k] < 0 || changed_objects[k] >= N) return fail(e, SBE_ERR_ARG, "chain %d: object index %d out of range", i, changed_objects[k]);
            const Slot& cur = e->slots[a];
            if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", a);
            { int orc = reject_overlap(e, a, "one-call step"); if (orc) return orc; }
            for (int c = 0; c < C; ++c)
                if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", a, c);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    for (int i = 0; i < n_chains; ++i)       // never-uploaded patterns of a current slot (first step after set_groups)
        if (e->slots[cur_slots[i]].patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slots[i]); if (rc) return rc; }
    int rc = ensure_lanes(e, n_chains);
    if (rc) return rc;
    rc = ensure_step_pool(e);
    if (rc) return rc;
    mark();
    // ---- host halves, in parallel over the chains ---------------------------------------------------------------
    if ((int)e->batch_cands.size() < n_chains) e->batch_cands.resize(n_chains);
    if ((int)e->batch_moved.size() < n_chains) e->batch_moved.resize(n_chains);
    std::vector<Slot>& cds = e->batch_cands;
    std::vector<CoreInputs> ins(n_chains);
    // payload blocks: chain i's used prefix (fixed sections + its changed rows) at a running offset of one pinned block
    std::vector<size_t> pay_off(n_chains + 1, 0);
    for (int i = 0; i < n_chains; ++i)
        pay_off[i + 1] = pay_off[i] + (e->sl.rows + (size_t)(rows_ptr[i + 1] - rows_ptr[i]) * F * C + 255) / 256 * 256;
    if (pay_off[n_chains] > e->batch_payload_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->h_batch_payload) { HIPCHK(e, hipHostFree(e->h_batch_payload)); e->h_batch_payload = nullptr; }
        if (e->d_batch_payload) { HIPCHK(e, hipFree(e->d_batch_payload)); e->d_batch_payload = nullptr; }
        e->batch_payload_bytes = pay_off[n_chains] + pay_off[n_chains] / 2;
        HIPCHK(e, hipHostMalloc((void**)&e->h_batch_payload, e->batch_payload_bytes, hipHostMallocDefault));
        HIPCHK(e, hipMalloc((void**)&e->d_batch_payload, e->batch_payload_bytes));
    }
    // ---- device argument blocks, per part: StepCore per chain | StepFinish per chain | candidate slot list --------
    // A large batch is cut into two parts: the host halves of the second run while the device works on the first
    // (measured, headline shape: 256 chains 655 us in two parts against ~740 in one; at 64 chains the per-part
    // fixed costs of the three launches outweigh the overlap -- 245 us in one part, 271 in two).
    int n_parts = n_chains >= 128 ? 2 : 1;
    if (const char* env = getenv("SBE_STEP_PARTS")) n_parts = std::max(1, std::min(atoi(env), n_chains));
    const int per_part = div_up(n_chains, n_parts);
    const size_t part_cores = ((size_t)per_part * sizeof(StepCore) + 255) / 256 * 256;
    const size_t part_fins = ((size_t)per_part * sizeof(StepFinish) + 255) / 256 * 256;
    const size_t part_bytes = part_cores + part_fins + ((size_t)per_part * sizeof(int32_t) + 255) / 256 * 256;
    const size_t meta_bytes = part_bytes * n_parts;
    if (meta_bytes > e->batch_meta_bytes) {
        if (e->d_batch_meta) { HIPCHK(e, hipStreamSynchronize(e->stream)); HIPCHK(e, hipFree(e->d_batch_meta)); }
        e->batch_meta_bytes = meta_bytes + meta_bytes / 2;
        HIPCHK(e, hipMalloc((void**)&e->d_batch_meta, e->batch_meta_bytes));
    }
    std::vector<uint8_t> meta(meta_bytes);
    std::vector<int> rcs(n_chains, SBE_OK);
    std::vector<std::string> errs(n_chains);
    DoneSig batch_done{};
    for (int part = 0; part < n_parts; ++part) {
        const int i0 = part * per_part, i1 = std::min(n_chains, i0 + per_part), np = i1 - i0;
        if (np <= 0) break;
        // the payload goes up in chunks of kCopyChunk chains, each sent as soon as its chains are prepared: the copy
        // (1.6 MB for 64 headline chains, ~40 us) runs under the preparation of the chains behind it
        constexpr int kCopyChunk = 16;
        const int n_copy_chunks = div_up(np, kCopyChunk);
        std::vector<std::atomic<int>> chunk_done(n_copy_chunks);
        for (auto& c : chunk_done) c.store(0, std::memory_order_relaxed);
        int next_copy = 0;
        hipError_t copy_err = hipSuccess;
        auto send_ready = [&]() {
            while (next_copy < n_copy_chunks) {
                const int c0 = i0 + next_copy * kCopyChunk, c1 = std::min(i1, c0 + kCopyChunk);
                if (chunk_done[next_copy].load(std::memory_order_acquire) < c1 - c0) break;
                const hipError_t he = hipMemcpyAsync(e->d_batch_payload + pay_off[c0], e->h_batch_payload + pay_off[c0],
                                                     pay_off[c1] - pay_off[c0], hipMemcpyHostToDevice, e->stream);
                if (he != hipSuccess) copy_err = he;
                ++next_copy;
            }
        };
        const std::function<void()> poll = send_ready;
        e->pool->run(np, [&](int j) {
            const int i = i0 + j;
            const bool regroup = clusters && (!clusters_mask || clusters_mask[i]);
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0;
            sbe_engine::Lane lane = e->lanes[i];                      // this chain's lane with its slice of the packed payload
            lane.h_payload = e->h_batch_payload + pay_off[i];
            lane.d_payload = e->d_batch_payload + pay_off[i];
            rcs[i] = prepare_step(e, lane, cur_slots[i], cand_slots[i], regroup ? clusters + (size_t)i * K * N : nullptr,
                                  nr ? changed_objects + r0 : nullptr, nr, nr ? source_rows + (size_t)r0 * F * C : nullptr,
                                  reweight ? weights + (size_t)i * F * C : nullptr, cds[i], ins[i], &errs[i], &e->batch_moved[i]);
            chunk_done[j / kCopyChunk].fetch_add(1, std::memory_order_release);
        }, &poll);
        send_ready();
        HIPCHK(e, copy_err);
        for (int i = i0; i < i1; ++i)
            if (rcs[i]) { (void)hipStreamSynchronize(e->stream); return fail(e, rcs[i], "chain %d: %s", i, errs[i].c_str()); }
        if (part == 0) mark();
        uint8_t* pm = meta.data() + (size_t)part * part_bytes;
        StepCore* cores = reinterpret_cast<StepCore*>(pm);
        StepFinish* fins = reinterpret_cast<StepFinish*>(pm + part_cores);
        int32_t* slot_list = reinterpret_cast<int32_t*>(pm + part_cores + part_fins);
        size_t lds = 0; int max_blocks = 0;
        for (int i = i0; i < i1; ++i) {
            size_t l = 0; int nb = 0;
            rc = build_step_core(e, e->lanes[i], cur_slots[i], cand_slots[i], ins[i], cores[i - i0], l, nb, n_chains);
            if (rc) return rc;
            lds = std::max(lds, l); max_blocks = std::max(max_blocks, nb);
            fins[i - i0] = make_step_finish_lane(e, e->lanes[i]);
            slot_list[i - i0] = cand_slots[i];
        }
        if (part == 0) mark();
        uint8_t* dm = e->d_batch_meta + (size_t)part * part_bytes;
        rc = upload(e, dm, pm, part_bytes);
        if (rc) return rc;
        if (part == 0) mark();
        k_step_core_batch<<<dim3(max_blocks, np), kBlock, lds, e->stream>>>(reinterpret_cast<const StepCore*>(dm));
        HIPCHK(e, hipGetLastError());
        if (part == 0) mark();
        for (int i = i0; i < i1; ++i) {
            commit_src_sync(e, cur_slots[i], cand_slots[i], changed_objects ? changed_objects + rows_ptr[i] : nullptr,
                            rows_ptr[i + 1] - rows_ptr[i]);
            commit_ids_sync(e, cur_slots[i], cand_slots[i], e->batch_moved[i]);
            std::swap(e->slots[cand_slots[i]], cds[i]);                               // (swap: both keep their storage)
        }
        if (part == 0) mark();
        rc = launch_mixture(e, 0, np, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, nullptr,
                            cand_slots + i0, reinterpret_cast<const int32_t*>(dm + part_cores + part_fins),
                            reinterpret_cast<const StepFinish*>(dm + part_cores), part == n_parts - 1 ? &batch_done : nullptr);
        if (rc) return rc;
    }
    mark();
    { int wrc = wait_done(e, batch_done); if (wrc) return wrc; }       // (the last part's reduction carries the flag)
    mark();
    // every chain's results are delivered; a chain whose proposal was malformed (its own data-check words) is reported
    // by index after that -- the other chains' outputs stay usable
    int first_bad = SBE_OK;
    std::string first_msg;
    for (int i = 0; i < n_chains; ++i) {
        rc = read_step_results_lane(e, e->lanes[i].h_step, cand_slots[i], group_logliks_out + (size_t)i * e->Gtot, mixture_out + i,
                               changed_groups_out ? changed_groups_out + (size_t)i * e->Gtot : nullptr, "rows", e->lanes[i].d_status, i);
        if (rc && !first_bad) { first_bad = rc; first_msg = e->last_error; }
    }
    if (first_bad) return fail(e, first_bad, "%s", first_msg.c_str());
    mark();
    if (timing && ntp == 10) {
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(tp[b] - tp[a]).count(); };
        fprintf(stderr, "[sbe_step_batch] %d chains (first part): checks %.1f | prepare (pool) %.1f | step cores %.1f | upload %.1f | launch core %.1f | "
                        "slot moves %.1f | mixture + reduce launches and the other parts %.1f | wait for the device %.1f | read results %.1f us\n", n_chains,
                us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9));
    }
    return SBE_OK;
}

// ---- batched step, DELTA form (round 3) -------------------------------------------------------
// The same step as sbe_step_batch with the proposal handed over as what it is -- a few moved objects:
//     moved_objects / moved_cluster (CSR by moved_ptr): objects that change cluster and their new cluster (-1: none)
//     changed_objects / source_rows (CSR by rows_ptr):  objects whose source rows change, each listed once
// A chain's two slots differ only in what its LAST step changed (SrcSync records for the source rows and for the id
// arrays), so the candidate is built by PATCHING: host mirror, device id arrays and source rows in O(delta); no
// [K][N] matrix is scanned, no slot state copied, no [N]-sized array packed or sent.  A chain whose records do not hold
// (first sweep, a slot touched by another call) or whose step changes the SET of has_components patterns / overflows
// the tuple table goes through sbe_step_batch itself (cluster matrix rebuilt from the ids); results are identical.
namespace {

struct DeltaPlan {           // per chain: payload section offsets (bytes from the chain's base) and capacities
    size_t subset, sub_row, sub_gid0, patch_n, patch_gid, patch_pid, patch_tid, stale, objects, tuple_g, tuple_p, patbits, weights, rows, total;
};

inline size_t al16(size_t v) { return (v + 15) / 16 * 16; }

DeltaPlan plan_delta(const sbe_engine* e, int n_mv, int n_changed, int n_last_ids, int n_last_src, bool reweight) {
    DeltaPlan p{};
    const size_t n_sub = (size_t)n_mv + n_changed, n_patch = (size_t)n_mv + n_last_ids;
    size_t o = 0;
    p.subset = o;    o = al16(o + n_sub * 4);
    p.sub_row = o;   o = al16(o + n_sub * 2);
    p.sub_gid0 = o;  o = al16(o + n_sub * 2);
    p.patch_n = o;   o = al16(o + n_patch * 4);
    p.patch_gid = o; o = al16(o + n_patch * 2);
    p.patch_pid = o; o = al16(o + n_patch);
    p.patch_tid = o; o = al16(o + n_patch);
    p.stale = o;     o = al16(o + (size_t)n_last_src * 4);
    p.objects = o;   o = al16(o + (size_t)n_changed * 4);
    p.tuple_g = o;   o = al16(o + (size_t)kMaxTuples * kMaxComponents * 2);
    p.tuple_p = o;   o = al16(o + (size_t)kMaxTuples);
    p.patbits = o;   o = al16(o + (size_t)e->Pmax * 4);
    p.weights = o;   o = al16(o + (reweight ? (size_t)e->F * e->C * 4 : 0));
    p.rows = o;      o = al16(o + (size_t)n_changed * e->F * e->C);
    p.total = (o + 255) / 256 * 256;
    return p;
}

// Host half of one chain in the delta form.  Patches e->slots[cand_slot] in place (it holds the current slot's state
// except the entries of the last step's moved objects).  Returns 1 when the chain must take the classic path instead
// (pattern set changes, tuple table full); the candidate's host state is then unspecified (the classic path rewrites it).
int prepare_step_delta(sbe_engine* e, uint8_t* h_base, const uint8_t* d_base, const DeltaPlan& L, int cur_slot, int cand_slot,
                       const int32_t* mv_objects, const int32_t* mv_cluster, int n_mv, const int32_t* changed_objects, int n_changed,
                       const uint8_t* source_rows, const float* weights, CoreInputs& in, std::vector<int32_t>& moved_out) {
    const int F = e->F, C = e->C;
    const Slot& cur = e->slots[cur_slot];
    Slot& cd = e->slots[cand_slot];
    const sbe_engine::SrcSync& irec = e->ids_sync[cand_slot];
    // 1. the candidate's host mirror back to the current slot's state: entries of the last step's moved objects
    for (int32_t n : irec.diff) {
        cd.h_gid[n] = cur.h_gid[n]; cd.h_pid[n] = cur.h_pid[n]; cd.h_tid[n] = cur.h_tid[n]; cd.h_toff[n] = cur.h_toff[n];
    }
    cd.patterns = cur.patterns; cd.n_tuples = cur.n_tuples;
    cd.h_tuple_g = cur.h_tuple_g; cd.h_tuple_p = cur.h_tuple_p; cd.pat_cnt = cur.pat_cnt; cd.tup_cnt = cur.tup_cnt;
    cd.inc_ok = cur.inc_ok; cd.patterns_dirty = false;
    cd.groups_set = cur.groups_set; cd.weights_set = true; cd.source_set = cur.source_set;
    cd.counts_set = cur.counts_set;
    // 2. this step's moves
    static thread_local std::vector<int32_t> mv;
    static thread_local std::vector<uint16_t> mv_old;
    mv.clear(); mv_old.clear();
    for (int i = 0; i < n_mv; ++i) {
        const int n = mv_objects[i];
        const uint16_t g_new = mv_cluster[i] < 0 ? kNoGroup : (uint16_t)mv_cluster[i];
        if (cd.h_gid[n] == g_new) continue;                              // (not a move)
        mv.push_back(n); mv_old.push_back(cd.h_gid[n]);
        cd.h_gid[n] = g_new;
    }
    if (!mv.empty()) {
        if (e->opt_step_derive == 1 || !update_patterns_and_tuples(e, cd, mv.data(), mv_old.data(), (int)mv.size())) return 1;
        cd.group_epoch = ++e->epoch_counter;
    } else cd.group_epoch = cur.group_epoch;
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    moved_out = mv;
    // 3. payload
    uint8_t* st = h_base;
    in = CoreInputs{};
    int32_t* sub = reinterpret_cast<int32_t*>(st + L.subset);
    int16_t* sub_row = reinterpret_cast<int16_t*>(st + L.sub_row);
    uint16_t* sub_gid0 = reinterpret_cast<uint16_t*>(st + L.sub_gid0);
    int n_subset = 0;
    {   // sorted union of moved and changed objects; per entry its row in `rows` (-1: none) and its candidate cluster id
        static thread_local std::vector<std::pair<int32_t, int32_t>> ch;     // (object, row)
        ch.clear();
        for (int i = 0; i < n_changed; ++i) ch.emplace_back(changed_objects[i], i);
        std::sort(ch.begin(), ch.end());
        std::sort(mv.begin(), mv.end());
        size_t a = 0, b = 0;
        while (a < mv.size() || b < ch.size()) {
            int32_t n; int r = -1;
            if (b == ch.size() || (a < mv.size() && mv[a] < ch[b].first)) n = mv[a++];
            else { n = ch[b].first; r = ch[b].second; if (a < mv.size() && mv[a] == n) ++a; ++b; }
            sub[n_subset] = n; sub_row[n_subset] = (int16_t)r; sub_gid0[n_subset] = cd.h_gid[n];
            ++n_subset;
        }
    }
    int n_patch = 0;
    {   // id entries to (re)write in the candidate's device arrays: last step's leftovers and this step's moves
        int32_t* pn = reinterpret_cast<int32_t*>(st + L.patch_n);
        uint16_t* pg = reinterpret_cast<uint16_t*>(st + L.patch_gid);
        uint8_t* pp = st + L.patch_pid; uint8_t* pt = st + L.patch_tid;
        auto put = [&](int32_t n) { pn[n_patch] = n; pg[n_patch] = cd.h_gid[n]; pp[n_patch] = cd.h_pid[n]; pt[n_patch] = cd.h_tid[n]; ++n_patch; };
        for (int32_t n : irec.diff) put(n);
        for (int32_t n : mv) put(n);                                      // (an object in both lists: same value twice)
    }
    const bool tables_changed = cd.patterns != cur.patterns || cd.h_tuple_p != cur.h_tuple_p || cd.h_tuple_g != cur.h_tuple_g;
    if (tables_changed) {
        memcpy(st + L.tuple_g, cd.h_tuple_g.data(), (size_t)kMaxTuples * kMaxComponents * 2);
        memcpy(st + L.tuple_p, cd.h_tuple_p.data(), kMaxTuples);
        memset(st + L.patbits, 0, (size_t)e->Pmax * 4);
        memcpy(st + L.patbits, cd.patterns.data(), cd.patterns.size() * 4);
        in.tuple_g = d_base + L.tuple_g; in.tuple_p = d_base + L.tuple_p; in.patbits = d_base + L.patbits;
    }
    if (weights) { memcpy(st + L.weights, weights, (size_t)F * C * 4); in.weights = d_base + L.weights; }
    if (n_changed > 0) {
        memcpy(st + L.objects, changed_objects, (size_t)n_changed * 4);
        memcpy(st + L.rows, source_rows, (size_t)n_changed * F * C);
        in.rows = d_base + L.rows;
        in.objects = reinterpret_cast<const int32_t*>(d_base + L.objects);
        in.n_changed = n_changed;
    }
    in.subset = reinterpret_cast<const int32_t*>(d_base + L.subset); in.n_subset = n_subset;
    in.sub_row = reinterpret_cast<const int16_t*>(d_base + L.sub_row);
    in.sub_gid0 = reinterpret_cast<const uint16_t*>(d_base + L.sub_gid0);
    in.patch_n = reinterpret_cast<const int32_t*>(d_base + L.patch_n);
    in.patch_gid = reinterpret_cast<const uint16_t*>(d_base + L.patch_gid);
    in.patch_pid = d_base + L.patch_pid; in.patch_tid = d_base + L.patch_tid; in.n_patch = n_patch;
    in.P = (int)cd.patterns.size();
    {   // source rows to bring over from the current slot (the last step's rows that this step does not rewrite)
        const sbe_engine::SrcSync& rec = e->src_sync[cand_slot];
        int32_t* stale = reinterpret_cast<int32_t*>(st + L.stale);
        int ns = 0;
        for (int32_t n : rec.diff) {
            bool rewritten = false;
            for (int i = 0; i < n_changed && !rewritten; ++i) rewritten = changed_objects[i] == n;
            if (!rewritten) stale[ns++] = n;
        }
        in.full_src_copy = false;
        in.stale = reinterpret_cast<const int32_t*>(d_base + L.stale);
        in.n_stale = ns;
    }
    return 0;
}

bool sync_valid(const sbe_engine::SrcSync& rec, const sbe_engine::SrcSync& peer, int peer_slot, int cap) {
    return rec.peer == peer_slot && rec.peer_version == peer.version && rec.own_version == rec.version && (int)rec.diff.size() <= cap;
}

}  // namespace

int sbe_step_batch_delta(sbe_engine* e, int n_chains, const int32_t* cur_slots, const int32_t* cand_slots,
                         const int32_t* moved_ptr, const int32_t* moved_objects, const int32_t* moved_cluster,
                         const int32_t* rows_ptr, const int32_t* changed_objects, const uint8_t* source_rows,
                         const float* weights, const uint8_t* weights_mask, double* group_logliks_out, double* mixture_out,
                         uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_PTR(e, cur_slots); CHECK_PTR(e, cand_slots); CHECK_PTR(e, moved_ptr); CHECK_PTR(e, rows_ptr);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    const auto t_start = std::chrono::steady_clock::now();
    std::chrono::steady_clock::time_point tq[8]; int nq = 0;
    auto markd = [&] { if (nq < 8) tq[nq++] = std::chrono::steady_clock::now(); };
    if (n_chains < 1 || n_chains > e->n_slots / 2) return fail(e, SBE_ERR_ARG, "n_chains=%d (1..%d: two slots per chain)", n_chains, e->n_slots / 2);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024) return fail(e, SBE_ERR_ARG, "sbe_step_batch_delta: tables too large for the one-launch step (G_total=%d, S=%d)", e->Gtot, e->S);
    const int N = e->N, F = e->F, C = e->C, K = e->G[0];
    if (rows_ptr[0] != 0 || moved_ptr[0] != 0) return fail(e, SBE_ERR_ARG, "rows_ptr[0] / moved_ptr[0] must be 0");
    {
        std::vector<uint8_t> used(e->n_slots, 0);
        std::vector<uint32_t> seen(N, 0), seen_mv(N, 0);
        for (int i = 0; i < n_chains; ++i) {
            const int a = cur_slots[i], b = cand_slots[i];
            if (a < 0 || a >= e->n_slots || b < 0 || b >= e->n_slots || a == b) return fail(e, SBE_ERR_ARG, "chain %d: bad slots (%d, %d)", i, a, b);
            if (used[a] || used[b]) return fail(e, SBE_ERR_ARG, "chain %d: slot used by another chain of the batch", i);
            used[a] = used[b] = 1;
            const int nr = rows_ptr[i + 1] - rows_ptr[i], nm = moved_ptr[i + 1] - moved_ptr[i];
            if (nr < 0 || nr > e->step_max_rows) return fail(e, SBE_ERR_ARG, "chain %d: %d changed source rows (0..%d per chain in a batched step)", i, nr, e->step_max_rows);
            if (nm < 0 || nm > N) return fail(e, SBE_ERR_ARG, "chain %d: %d moved objects", i, nm);
            if ((nr > 0 && (!changed_objects || !source_rows)) || (nm > 0 && (!moved_objects || !moved_cluster))) return fail(e, SBE_ERR_ARG, "chain %d: delta arrays missing", i);
            for (int k = rows_ptr[i]; k < rows_ptr[i + 1]; ++k) {
                const int n = changed_objects[k];
                if (n < 0 || n >= N) return fail(e, SBE_ERR_ARG, "chain %d: object index %d out of range", i, n);
                if (seen[n] == (uint32_t)(2 * i + 1)) return fail(e, SBE_ERR_ARG, "chain %d: object %d listed twice in changed_objects", i, n);
                seen[n] = (uint32_t)(2 * i + 1);
            }
            for (int k = moved_ptr[i]; k < moved_ptr[i + 1]; ++k) {
                const int n = moved_objects[k];
                if (n < 0 || n >= N) return fail(e, SBE_ERR_ARG, "chain %d: moved object index %d out of range", i, n);
                if (moved_cluster[k] < -1 || moved_cluster[k] >= K) return fail(e, SBE_ERR_ARG, "chain %d: cluster %d out of range [-1,%d)", i, moved_cluster[k], K);
                // a repeated moved object would be patched twice (pattern counts decremented for a pattern the object
                // was never in, its count delta added twice) while the matrix form resolves it last-wins: rejected
                if (seen_mv[n] == (uint32_t)(i + 1)) return fail(e, SBE_ERR_ARG, "chain %d: object %d listed twice in moved_objects", i, n);
                seen_mv[n] = (uint32_t)(i + 1);
            }
            const Slot& cur = e->slots[a];
            if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", a);
            { int orc = reject_overlap(e, a, "one-call step"); if (orc) return orc; }
            for (int c = 0; c < C; ++c)
                if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", a, c);
        }
    }
    HIPCHK(e, hipSetDevice(e->device));
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    for (int i = 0; i < n_chains; ++i)
        if (e->slots[cur_slots[i]].patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slots[i]); if (rc) return rc; }
    int rc = ensure_lanes(e, n_chains);
    if (rc) return rc;
    rc = ensure_step_pool(e);
    if (rc) return rc;
    if ((int)e->batch_moved.size() < n_chains) e->batch_moved.resize(n_chains);
    markd();                                     // 0: checks done
    // ---- which chains can be patched; their payload plans -------------------------------------------------------------
    std::vector<int> fast, slow;
    std::vector<DeltaPlan> plans(n_chains);
    std::vector<size_t> pay_off(n_chains + 1, 0);
    for (int i = 0; i < n_chains; ++i) {
        const int a = cur_slots[i], b = cand_slots[i];
        const bool ok = sync_valid(e->ids_sync[b], e->ids_sync[a], a, e->step_max_rows) &&
                        sync_valid(e->src_sync[b], e->src_sync[a], a, e->step_max_rows) &&
                        e->slots[a].inc_ok && e->slots[a].n_tuples > 0 && e->slots[b].h_gid.size() == e->slots[a].h_gid.size();
        if (ok) {
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            plans[i] = plan_delta(e, moved_ptr[i + 1] - moved_ptr[i], rows_ptr[i + 1] - rows_ptr[i], (int)e->ids_sync[b].diff.size(),
                                  (int)e->src_sync[b].diff.size(), reweight);
            pay_off[i + 1] = pay_off[i] + plans[i].total;
            fast.push_back(i);
        } else { pay_off[i + 1] = pay_off[i]; slow.push_back(i); }
    }
    // Chains that cannot be patched run through the classic entry point (cluster matrices rebuilt from the ids).  That call
    // uses the lanes, the payload block and the per-chain scratch of ITS chain numbering and synchronises: it runs either
    // before the patched chains are prepared or after their results have been read, never in between.
    auto run_classic = [&](const std::vector<int>& which, int& rc_out, std::string& msg_out) {
        rc_out = SBE_OK;
        if (which.empty()) return;
        const int ns = (int)which.size();
        std::vector<int32_t> s_cur(ns), s_cand(ns), s_ptr(ns + 1, 0), s_objs;
        std::vector<uint8_t> s_cl((size_t)ns * K * N, 0), s_rows, s_wm(ns, 0);
        std::vector<float> s_w(weights ? (size_t)ns * F * C : 0);
        for (int j = 0; j < ns; ++j) {
            const int i = which[j];
            s_cur[j] = cur_slots[i]; s_cand[j] = cand_slots[i];
            const Slot& cur = e->slots[cur_slots[i]];
            std::vector<uint16_t> ids(cur.h_gid.begin(), cur.h_gid.begin() + N);
            for (int k = moved_ptr[i]; k < moved_ptr[i + 1]; ++k) ids[moved_objects[k]] = moved_cluster[k] < 0 ? kNoGroup : (uint16_t)moved_cluster[k];
            uint8_t* cl = s_cl.data() + (size_t)j * K * N;
            for (int n = 0; n < N; ++n) if (ids[n] != kNoGroup) cl[(size_t)ids[n] * N + n] = 1;
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0;
            s_ptr[j + 1] = s_ptr[j] + nr;
            if (nr) {
                s_objs.insert(s_objs.end(), changed_objects + r0, changed_objects + r0 + nr);
                s_rows.insert(s_rows.end(), source_rows + (size_t)r0 * F * C, source_rows + (size_t)(r0 + nr) * F * C);
            }
            if (weights && (!weights_mask || weights_mask[i])) { s_wm[j] = 1; memcpy(&s_w[(size_t)j * F * C], weights + (size_t)i * F * C, (size_t)F * C * 4); }
        }
        std::vector<double> s_glh((size_t)ns * e->Gtot), s_mix(ns);
        std::vector<uint8_t> s_chg((size_t)ns * e->Gtot);
        rc_out = sbe_step_batch(e, ns, s_cur.data(), s_cand.data(), s_cl.data(), nullptr, s_ptr.data(), s_objs.empty() ? nullptr : s_objs.data(),
                                s_rows.empty() ? nullptr : s_rows.data(), weights ? s_w.data() : nullptr, weights ? s_wm.data() : nullptr,
                                s_glh.data(), s_mix.data(), s_chg.data());
        if (rc_out) { msg_out = e->last_error; return; }
        for (int j = 0; j < ns; ++j) {
            const int i = which[j];
            memcpy(group_logliks_out + (size_t)i * e->Gtot, &s_glh[(size_t)j * e->Gtot], (size_t)e->Gtot * sizeof(double));
            mixture_out[i] = s_mix[j];
            if (changed_groups_out) memcpy(changed_groups_out + (size_t)i * e->Gtot, &s_chg[(size_t)j * e->Gtot], (size_t)e->Gtot);
        }
    };
    int slow_rc = SBE_OK; std::string slow_msg;
    run_classic(slow, slow_rc, slow_msg);        // (chains without usable records: before the patched chains touch anything)
    markd();                                     // 1: plans (+ the unpatched chains)
    // ---- host halves of the patched chains (pool) ----------------------------------------------------------------------
    // The delta payload is small (~10 KB per chain): it goes up in ONE copy together with the launch arguments (every
    // copy-engine operation costs ~10 us of latency in the stream; the chunked upload of the matrix form paid five).
    std::vector<CoreInputs> ins(n_chains);
    std::vector<int> fb(n_chains, 0);
    const int nf = (int)fast.size();
    const size_t part_cores = ((size_t)std::max(nf, 1) * sizeof(StepCore) + 255) / 256 * 256;
    const size_t part_fins = ((size_t)std::max(nf, 1) * sizeof(StepFinish) + 255) / 256 * 256;
    const size_t meta_bytes = part_cores + part_fins + ((size_t)std::max(nf, 1) * sizeof(int32_t) + 255) / 256 * 256;
    const size_t meta_off = pay_off[n_chains];
    if (meta_off + meta_bytes > e->batch_payload_bytes) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->h_batch_payload) { HIPCHK(e, hipHostFree(e->h_batch_payload)); e->h_batch_payload = nullptr; }
        if (e->d_batch_payload) { HIPCHK(e, hipFree(e->d_batch_payload)); e->d_batch_payload = nullptr; }
        e->batch_payload_bytes = (meta_off + meta_bytes) * 3 / 2;
        HIPCHK(e, hipHostMalloc((void**)&e->h_batch_payload, e->batch_payload_bytes, hipHostMallocDefault));
        HIPCHK(e, hipMalloc((void**)&e->d_batch_payload, e->batch_payload_bytes));
    }
    if (nf > 0) {
        e->pool->run(nf, [&](int j) {
            const int i = fast[j];
            const bool reweight = weights && (!weights_mask || weights_mask[i]);
            const int r0 = rows_ptr[i], nr = rows_ptr[i + 1] - r0, m0 = moved_ptr[i], nm = moved_ptr[i + 1] - m0;
            fb[i] = prepare_step_delta(e, e->h_batch_payload + pay_off[i], e->d_batch_payload + pay_off[i], plans[i], cur_slots[i], cand_slots[i],
                                       nm ? moved_objects + m0 : nullptr, nm ? moved_cluster + m0 : nullptr, nm,
                                       nr ? changed_objects + r0 : nullptr, nr, nr ? source_rows + (size_t)r0 * F * C : nullptr,
                                       reweight ? weights + (size_t)i * F * C : nullptr, ins[i], e->batch_moved[i]);
        });
    }
    markd();                                     // 2: host halves
    std::vector<int> go, late;                             // patched chains that stay on the fast path / that turned out not to
    for (int i : fast) { if (fb[i]) { late.push_back(i); bump_ids(e, cand_slots[i]); } else go.push_back(i); }
    const int ng = (int)go.size();
    DoneSig fast_done{};
    if (ng > 0) {
        uint8_t* pm = e->h_batch_payload + meta_off;
        uint8_t* dm = e->d_batch_payload + meta_off;
        StepCore* cores = reinterpret_cast<StepCore*>(pm);
        StepFinish* fins = reinterpret_cast<StepFinish*>(pm + part_cores);
        int32_t* slot_list = reinterpret_cast<int32_t*>(pm + part_cores + part_fins);
        std::vector<int32_t> cand_go(ng);
        size_t lds = 0; int max_blocks = 0;
        for (int j = 0; j < ng; ++j) {
            const int i = go[j];
            size_t l = 0; int nb = 0;
            rc = build_step_core(e, e->lanes[i], cur_slots[i], cand_slots[i], ins[i], cores[j], l, nb, ng);
            if (rc) return rc;
            lds = std::max(lds, l); max_blocks = std::max(max_blocks, nb);
            fins[j] = make_step_finish_lane(e, e->lanes[i]);
            slot_list[j] = cand_go[j] = cand_slots[i];
        }
        HIPCHK(e, hipMemcpyAsync(e->d_batch_payload, e->h_batch_payload, meta_off + meta_bytes, hipMemcpyHostToDevice, e->stream));
        k_step_core_batch<<<dim3(max_blocks, ng), kBlock, lds, e->stream>>>(reinterpret_cast<const StepCore*>(dm));
        HIPCHK(e, hipGetLastError());
        markd();                                 // 3: step cores built, uploaded, launched
        hipEvent_t tev_a = nullptr, tev_b = nullptr;       // (sbe_kernel_timing brackets the fused kernel of a batched step too)
        rc = next_timing_events(e, &tev_a, &tev_b);
        if (rc) return rc;
        rc = launch_mixture(e, 0, ng, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, tev_a, tev_b, nullptr,
                            cand_go.data(), reinterpret_cast<const int32_t*>(dm + part_cores + part_fins),
                            reinterpret_cast<const StepFinish*>(dm + part_cores), &fast_done);
        for (int j = 0; j < ng; ++j) {           // (bookkeeping under the device work: everything is enqueued)
            const int i = go[j];
            commit_src_sync(e, cur_slots[i], cand_slots[i], changed_objects ? changed_objects + rows_ptr[i] : nullptr, rows_ptr[i + 1] - rows_ptr[i]);
            commit_ids_sync(e, cur_slots[i], cand_slots[i], e->batch_moved[i]);
        }
        if (rc) return rc;
    }
    static const bool timing_d = getenv("SBE_STEP_TIMING") && atoi(getenv("SBE_STEP_TIMING")) != 0;
    const auto t_enq = std::chrono::steady_clock::now();
    { int wrc = wait_done(e, fast_done); if (wrc) return wrc; }        // (no patched chain: nothing was launched, plain wait)
    if (timing_d) {
        const auto t_done = std::chrono::steady_clock::now();
        auto us = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[sbe_step_batch_delta] %d chains: %d patched, %d + %d through sbe_step_batch | checks %.1f | plans %.1f | host halves (pool) %.1f | cores + launch %.1f | "
                        "mixture launches + records %.1f | wait %.1f us\n", n_chains, ng, (int)slow.size(), (int)late.size(), nq > 0 ? us(t_start, tq[0]) : 0.0,
                nq > 1 ? us(tq[0], tq[1]) : 0.0, nq > 2 ? us(tq[1], tq[2]) : 0.0, nq > 3 ? us(tq[2], tq[3]) : 0.0, nq > 3 ? us(tq[3], t_enq) : 0.0, us(t_enq, t_done));
    }
    int first_bad = slow_rc; std::string first_msg = slow_msg;
    for (int i : go) {
        rc = read_step_results_lane(e, e->lanes[i].h_step, cand_slots[i], group_logliks_out + (size_t)i * e->Gtot, mixture_out + i,
                                    changed_groups_out ? changed_groups_out + (size_t)i * e->Gtot : nullptr, "rows", e->lanes[i].d_status, i);
        if (rc && !first_bad) { first_bad = rc; first_msg = e->last_error; }
    }
    // ---- chains whose step turned out to need the full derivation: now that the patched chains' results are out ---------
    if (!late.empty()) {
        int late_rc = SBE_OK; std::string late_msg;
        run_classic(late, late_rc, late_msg);
        if (late_rc && !first_bad) { first_bad = late_rc; first_msg = late_msg; }
    }
    if (first_bad) return fail(e, first_bad, "%s", first_msg.c_str());
    return SBE_OK;
}

// The single-chain step in delta form: sbe_step with the proposal as moved objects + changed rows (see
// sbe_step_batch_delta); the payload sits in the lane's host-mapped block, so no copy-engine operation is in the chain.
int sbe_step_delta(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* moved_objects, const int32_t* moved_cluster, int n_moved,
                   const int32_t* changed_objects, int n_changed, const uint8_t* source_rows, const float* weights,
                   double* group_logliks_out, double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    const int N = e->N, C = e->C, K = e->G[0];
    if (n_moved < 0 || n_moved > N || (n_moved > 0 && (!moved_objects || !moved_cluster))) return fail(e, SBE_ERR_ARG, "moved_objects / moved_cluster missing for n_moved=%d", n_moved);
    if (n_changed < 0 || (n_changed > 0 && (!changed_objects || !source_rows))) return fail(e, SBE_ERR_ARG, "changed_objects / source_rows missing for n_changed=%d", n_changed);
    bool dup = false;                  // a repeated object in either list: the matrix form resolves it (last entry wins)
    {
        static thread_local std::vector<uint32_t> stamp;
        static thread_local uint32_t epoch = 0;
        if ((int)stamp.size() < N || ++epoch == 0) { stamp.assign(N, 0); epoch = 1; }
        for (int i = 0; i < n_moved; ++i) {
            if (moved_objects[i] < 0 || moved_objects[i] >= N) return fail(e, SBE_ERR_ARG, "moved object index %d out of range", moved_objects[i]);
            if (moved_cluster[i] < -1 || moved_cluster[i] >= K) return fail(e, SBE_ERR_ARG, "cluster %d out of range [-1,%d)", moved_cluster[i], K);
            dup = dup || stamp[moved_objects[i]] == epoch;
            stamp[moved_objects[i]] = epoch;
        }
    }
    for (int i = 0; i < n_changed; ++i) {
        if (changed_objects[i] < 0 || changed_objects[i] >= N) return fail(e, SBE_ERR_ARG, "object index %d out of range", changed_objects[i]);
        for (int j = 0; j < i && !dup && n_changed <= 64; ++j) dup = changed_objects[j] == changed_objects[i];
    }
    const Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    { int orc = reject_overlap(e, cur_slot, "one-call step"); if (orc) return orc; }
    for (int c = 0; c < C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c]) return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration of component %d not set", cur_slot, c);
    HIPCHK(e, hipSetDevice(e->device));
    // the matrix form serves whatever the patching form cannot: no usable records, a large or repeated row list, a
    // step that needs the full derivation, tables too large for the one-launch step
    auto classic = [&]() {
        std::vector<uint16_t> ids(cur.h_gid.begin(), cur.h_gid.begin() + N);
        for (int k = 0; k < n_moved; ++k) ids[moved_objects[k]] = moved_cluster[k] < 0 ? kNoGroup : (uint16_t)moved_cluster[k];
        std::vector<uint8_t> cl((size_t)std::max(K, 1) * N, 0);
        for (int n = 0; n < N; ++n) if (ids[n] != kNoGroup) cl[(size_t)ids[n] * N + n] = 1;
        return sbe_step(e, cur_slot, cand_slot, n_moved > 0 ? cl.data() : nullptr, changed_objects, n_changed, source_rows, weights,
                        group_logliks_out, mixture_out, changed_groups_out);
    };
    const bool ok = !dup && n_changed <= std::min(e->step_max_rows, 64) && e->opt_step_form == 0 && !cur.patterns_dirty &&
                    (int64_t)e->Gtot * e->S * 28 <= 60 * 1024 &&
                    sync_valid(e->ids_sync[cand_slot], e->ids_sync[cur_slot], cur_slot, e->step_max_rows) &&
                    sync_valid(e->src_sync[cand_slot], e->src_sync[cur_slot], cur_slot, e->step_max_rows) &&
                    cur.inc_ok && cur.n_tuples > 0 && e->slots[cand_slot].h_gid.size() == cur.h_gid.size();
    if (!ok) return classic();
    const DeltaPlan plan = plan_delta(e, n_moved, n_changed, (int)e->ids_sync[cand_slot].diff.size(), (int)e->src_sync[cand_slot].diff.size(), weights != nullptr);
    if (plan.total > e->sl.total) return classic();
    if (e->status_pending) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    CoreInputs in;
    if (prepare_step_delta(e, e->h_step_payload, e->d_step_payload, plan, cur_slot, cand_slot, moved_objects, moved_cluster, n_moved,
                           changed_objects, n_changed, source_rows, weights, in, e->step_moved)) {
        bump_ids(e, cand_slot);
        return classic();
    }
    int rc = launch_step_core(e, cur_slot, cand_slot, in);
    if (rc) return rc;
    commit_src_sync(e, cur_slot, cand_slot, changed_objects, n_changed);
    commit_ids_sync(e, cur_slot, cand_slot, e->step_moved);
    StepFinish fin = make_step_finish(e);
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    rc = wait_done(e, done);
    if (rc) return rc;
    return read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "rows");
}

// ---- one-call Gibbs source step (GibbsSampleSource._propose, operators.py:495-552, on the resident state) ------
// candidate = current with the source of the listed objects redrawn from its posterior ON THE DEVICE; count delta,
// every table, both transition log-probabilities, collapsed per-group and mixture log-likelihood of the candidate:
// five launches, one synchronisation; objects (and the caller's uniforms) are read from host-mapped memory, all
// results arrive through host-mapped memory.
int sbe_gibbs_step(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                   double prior_temperature, int from_prior, const double* z, double* log_q_out, double* log_q_back_out,
                   double* group_logliks_out, double* mixture_out, uint8_t* changed_groups_out) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, log_q_out); CHECK_PTR(e, log_q_back_out); CHECK_PTR(e, group_logliks_out); CHECK_PTR(e, mixture_out);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_sub < 1) return fail(e, SBE_ERR_ARG, "n_sub=%d (nothing to resample)", n_sub);
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    { int orc = reject_overlap(e, cur_slot, "one-call step"); if (orc) return orc; }
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c] || !cur.probs_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration / probability tables of component %d not set", cur_slot, c);
    for (int i = 0; i < n_sub; ++i)
        if (objects[i] < 0 || objects[i] >= e->N) return fail(e, SBE_ERR_ARG, "object index %d out of range", objects[i]);
    if ((int64_t)e->Gtot * e->S * 28 > 60 * 1024)
        return fail(e, SBE_ERR_ARG, "one-call Gibbs step: tables too large for the fused table kernel (G_total=%d, S=%d)", e->Gtot, e->S);
    HIPCHK(e, hipSetDevice(e->device));
    if (cur.patterns_dirty) { int rc = upload_patterns_and_weights(e, cur_slot); if (rc) return rc; }
    if (e->status_pending) {                  // deliver a deferred data check before this step reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        int rc = synced(e);
        if (rc) return rc;
    }
    const int N = e->N, Np = e->Np, F = e->F, C = e->C;
    const int64_t n_obs = (int64_t)n_sub * F;
    // host-mapped inputs: objects | row_of marks | uniforms (when they are few; a large block is copied instead)
    const size_t ob = ((size_t)n_sub * sizeof(int32_t) + 255) / 256 * 256;
    const size_t rb = ((size_t)Np * sizeof(int16_t) + 255) / 256 * 256;
    const size_t zbytes = z ? (size_t)n_obs * sizeof(double) : 0;
    const bool z_mapped = zbytes <= ((size_t)1 << 19);
    int rc = ensure_io(e, ob + rb + (z_mapped ? zbytes : 0));
    if (rc) return rc;
    memcpy(e->h_io, objects, (size_t)n_sub * sizeof(int32_t));
    int16_t* row_of = reinterpret_cast<int16_t*>(e->h_io + ob);
    std::fill(row_of, row_of + Np, (int16_t)-1);
    for (int i = 0; i < n_sub; ++i) row_of[objects[i]] = 0;
    const int32_t* d_obj = reinterpret_cast<const int32_t*>(e->d_io);
    // device scratch: selected probabilities (forward / back), their partial log sums, uniforms if copied
    const size_t pb = ((size_t)n_obs * sizeof(float) + 255) / 256 * 256;
    const size_t zb = (z && !z_mapped) ? (zbytes + 255) / 256 * 256 : 0;
    const int nblk = div_up(n_obs, kBlock);                   // one log-sum partial per block of the two posterior kernels
    const size_t qb_bytes = ((size_t)nblk * sizeof(double) + 255) / 256 * 256;
    rc = ensure_scratch(e, 2 * pb + 2 * qb_bytes + zb);
    if (rc) return rc;
    float* d_psel_f = (float*)e->d_scratch;
    float* d_psel_b = (float*)(e->d_scratch + pb);
    double* d_part_f = (double*)(e->d_scratch + 2 * pb);
    double* d_part_b = (double*)(e->d_scratch + 2 * pb + qb_bytes);
    const double* d_z = nullptr;
    if (z && z_mapped) { memcpy(e->h_io + ob + rb, z, zbytes); d_z = reinterpret_cast<const double*>(e->d_io + ob + rb); }
    else if (z) {
        double* dz = (double*)(e->d_scratch + 2 * pb + 2 * qb_bytes);
        int _urc = upload(e, dz, z, zbytes); if (_urc) return _urc;
        d_z = dz;
    }
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    auto post_args = [&](int slot) {
        return SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * C * Np, e->d_pid + (int64_t)slot * Np,
                           e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * F * C,
                           d_obj, n_sub, Np, F, e->S, C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                           from_prior != 0};
    };
    uint8_t* src_cand = e->d_src + (int64_t)cand_slot * N * e->Fp;
    bump_src(e, cand_slot);                  // (the Gibbs step draws into the candidate's array and copies the rest in full)
    bump_ids(e, cand_slot);
    // 1: the draw (posterior from the current tables) -> the candidate's source rows of the listed objects, and the
    //    per-block partial sums of log_q
    k_sample_source<<<nblk, kBlock, 0, e->stream>>>(post_args(cur_slot), d_z, e->rng_seed, e->rng_draw, src_cand, d_psel_f, e->d_status, d_part_f);
    if (!z) ++e->rng_draw;
    HIPCHK(e, hipGetLastError());
    // 2: the rest of the candidate slot, its count delta and every one of its tables
    Slot cd = cur;
    {
        CoreInputs in;
        in.row_of = reinterpret_cast<const int16_t*>(e->d_io + ob);
        in.src_new = src_cand;
        in.subset = d_obj; in.n_subset = n_sub;
        in.P = (int)cd.patterns.size();
        rc = launch_step_core(e, cur_slot, cand_slot, in);
        if (rc) return rc;
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    e->slots[cand_slot] = cd;
    // 3: log_q_back -- the candidate's posterior evaluated at the CURRENT source assignment (+ its partial sums)
    k_source_logprob<<<nblk, kBlock, 0, e->stream>>>(post_args(cand_slot), e->d_src + (int64_t)cur_slot * N * e->Fp, d_psel_b, e->d_status, d_part_b);
    HIPCHK(e, hipGetLastError());
    // 4 + 5: fused mixture eval, reduction + epilogue (per-group collapsed values, flags, checks, log_q / log_q_back)
    StepFinish fin = make_step_finish(e);
    fin.lq_partials[0] = d_part_f; fin.lq_partials[1] = d_part_b; fin.lq_n[0] = fin.lq_n[1] = nblk;
    fin.lq_out = reinterpret_cast<double*>(e->d_step_host + step_host_lq_offset(e));
    DoneSig done;
    rc = launch_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS, nullptr, nullptr, &fin, nullptr, nullptr,
                        nullptr, &done);
    if (rc) return rc;
    rc = wait_done(e, done);
    if (rc) return rc;
    rc = read_step_results(e, cand_slot, group_logliks_out, mixture_out, changed_groups_out, "posterior rows / table rows");
    if (rc) return rc;
    const double* lq = reinterpret_cast<const double*>(e->h_step + step_host_lq_offset(e));
    *log_q_out = lq[0];
    *log_q_back_out = lq[1];
    return SBE_OK;
}

static int step_general(sbe_engine* e, int cur_slot, int cand_slot, const uint8_t* clusters, const int32_t* changed_objects,
                        int n_changed, const uint8_t* source_rows, const float* weights, double* group_logliks_out,
                        double* mixture_out, uint8_t* changed_groups_out) {
    Slot& cur = e->slots[cur_slot];
    const int saved_deferred = e->opt_deferred;
    e->opt_deferred = 1;                       // no intermediate synchronisation inside the step
    auto done = [&](int rc) { e->opt_deferred = saved_deferred; return rc; };
    int rc = sbe_copy_slot(e, cand_slot, cur_slot);
    if (rc) return done(rc);
    const int N = e->N;
    // objects whose counts may change: listed source rows + objects whose cluster membership changed
    std::vector<uint8_t> moved(N, 0);
    for (int i = 0; i < n_changed; ++i) moved[changed_objects[i]] = 1;
    if (clusters) {
        const int K = e->G[0];
        std::vector<uint16_t> ids(N, kNoGroup);
        char msg[320];
        if (!matrix_to_ids(clusters, K, N, 0, 0, ids.data(), msg, sizeof msg)) return done(fail(e, SBE_ERR_DATA, "%s", msg));   // component 0: offset 0
        for (int n = 0; n < N; ++n) if (ids[n] != cur.h_gid[n]) moved[n] = 1;
        rc = set_gid_common(e, cand_slot, 0, ids);
        if (rc) return done(rc);
    }
    if (n_changed > 0) {
        rc = sbe_set_source_rows(e, cand_slot, changed_objects, n_changed, source_rows);
        if (rc) return done(rc);
    }
    if (weights) rc = sbe_set_weights(e, cand_slot, weights);
    else if (e->slots[cand_slot].patterns_dirty) rc = upload_patterns_and_weights(e, cand_slot);
    if (rc) return done(rc);
    std::vector<int32_t> subset;
    for (int n = 0; n < N; ++n) if (moved[n]) subset.push_back(n);
    rc = sbe_update_counts(e, cand_slot, cur_slot, subset.data(), (int)subset.size(), nullptr);
    if (rc) return done(rc);
    // probability tables of every component in one launch (+ their tile-transposed copy)
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return done(rc);
    k_probs<int32_t><<<div_up((int64_t)e->Gtot * e->F, 256), 256, 0, e->stream>>>(
        e->d_counts + (int64_t)cand_slot * e->table_elems(), e->d_conc, nullptr,
        e->d_probs + (int64_t)cand_slot * e->table_elems(), 0, e->Gtot, e->F, e->S, 0.0, 0.0, 1, e->d_status);
    k_tile_probs<<<div_up((int64_t)e->Gtot * e->S * e->ft * e->n_ftiles, 256), 256, 0, e->stream>>>(
        e->d_probs + (int64_t)cand_slot * e->table_elems(), e->d_probs_t + (int64_t)cand_slot * e->probs_t_elems(),
        0, e->Gtot, e->Gtot, e->F, e->S, e->ft, e->n_ftiles);
    HIPCHK(e, hipGetLastError());
    std::fill(e->slots[cand_slot].probs_set.begin(), e->slots[cand_slot].probs_set.end(), 1);
    rc = check_after(e, ST_BAD_NORMALIZE);
    if (rc) return done(rc);
    // collapsed likelihood of every group (a7/a8) into a device buffer
    k_dcl<int32_t><<<div_up((int64_t)e->Gtot * e->F, 256), 256, 0, e->stream>>>(
        e->d_counts + (int64_t)cand_slot * e->table_elems(), e->d_conc, e->d_step_pf, 0, e->Gtot, e->F, e->S, 1);
    k_group_sum_f32<<<div_up((int64_t)e->Gtot * 8, 64), 64, 0, e->stream>>>(e->d_step_pf, e->d_step_pg, e->Gtot, e->F);
    HIPCHK(e, hipGetLastError());
    rc = enqueue_mixture(e, cand_slot, 1, e->opt_log == SBE_LOG_PRODUCT ? LOG_PRODUCT : LOG_PER_OBS);
    if (rc) return done(rc);
    // one read-back, one synchronisation
    const size_t pg_bytes = (size_t)e->Gtot * sizeof(double);
    rc = ensure_pinned(e, pg_bytes + (size_t)e->Gtot);
    if (rc) return done(rc);
    HIPCHK(e, hipMemcpyAsync(e->h_pinned, e->d_step_pg, pg_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h_pinned + pg_bytes, e->d_changed, (size_t)e->Gtot, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    memcpy(group_logliks_out, e->h_pinned, pg_bytes);
    if (changed_groups_out) memcpy(changed_groups_out, e->h_pinned + pg_bytes, (size_t)e->Gtot);
    *mixture_out = e->h_results[cand_slot];
    return done(synced(e));
}

// ---- self-test hook: the table-build log against the device library's log -------------------------------
// GibbsSampleSource._propose (operators.py:495-552) for the drop-in layer in ONE call: sbe_gibbs_step's device chain -- the
// draw into the candidate slot, the rest of the slot with its count delta and tables (k_step_core), the backward
// probabilities -- and then, instead of the likelihoods a resident chain wants, what the reference's sample bookkeeping
// wants: the drawn ids, both selected-probability arrays and the count rows that changed, all in the host-mapped block,
// one completion flag.  (As eight engine calls -- copy_slot, sample_source, update_counts, update_probs, source_logprob,
// get_source_rows, counts_delta -- the same work cost 190 us per proposal in the sampler replay, seven stream
// synchronisations among them.)
int sbe_gibbs_propose_supported(sbe_engine* e) {                       // 1: the CHAIN form fits (the tile form is tried first, per call)
    CHECK_ENGINE(e);
    return ((int64_t)e->Gtot * e->S * 28 <= 60 * 1024) ? 1 : 0;          // (the fused table kernel of the step core)
}

// `follow` (sbe_gibbs_propose_apply): when the proposal touches any group, the CURRENT slot takes it -- counts, the touched
// groups' tables, the drawn source rows -- inside the tile kernel (tables and ids behind its completion flag), or as a copy of
// the candidate slot behind the chain form.
static int gibbs_propose_impl(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                              double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                              float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out, int follow) {
    CHECK_ENGINE(e); CHECK_SLOT(e, cur_slot); CHECK_SLOT(e, cand_slot);
    CHECK_PTR(e, src_new_out); CHECK_PTR(e, sel_out); CHECK_PTR(e, sel_back_out); CHECK_PTR(e, touched_out); CHECK_PTR(e, n_touched_out);
    CHECK_PTR(e, diff_rows_out); CHECK_PTR(e, z);
    if (cur_slot == cand_slot) return fail(e, SBE_ERR_ARG, "current and candidate slot must differ");
    if (n_sub < 1) return fail(e, SBE_ERR_ARG, "n_sub=%d (nothing to resample)", n_sub);
    CHECK_PTR(e, objects);
    if (!(temperature > 0.0) || !(prior_temperature > 0.0)) return fail(e, SBE_ERR_ARG, "temperatures must be positive");
    Slot& cur = e->slots[cur_slot];
    if (!cur.groups_set || !cur.source_set || !cur.weights_set) return fail(e, SBE_ERR_STATE, "slot %d: groups / source / weights not set", cur_slot);
    { int orc = reject_overlap(e, cur_slot, "one-call step"); if (orc) return orc; }
    for (int c = 0; c < e->C; ++c)
        if (!cur.counts_set[c] || !e->conc_set[c] || !cur.probs_set[c])
            return fail(e, SBE_ERR_STATE, "slot %d: counts / concentration / probability tables of component %d not set", cur_slot, c);
    int rc = check_objects(e, objects, n_sub);
    if (rc) return rc;
    const int N = e->N, Np = e->Np, F = e->F, C = e->C, S = e->S;
    const int64_t n_obs = (int64_t)n_sub * F, fs = (int64_t)F * S;
    // the groups the subset's objects are in (their count rows are the only ones the redraw can change), ascending
    std::vector<uint8_t> seen((size_t)e->Gtot, 0);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < n_sub; ++i) {
            const uint16_t gg = cur.h_gid[(size_t)c * N + objects[i]];
            if (gg != kNoGroup) seen[gg] = 1;
        }
    int n_touched = 0;
    for (int g = 0; g < e->Gtot; ++g) if (seen[g]) touched_out[n_touched++] = g;
    *n_touched_out = n_touched;
    HIPCHK(e, hipSetDevice(e->device));
    if (cur.patterns_dirty) { rc = upload_patterns_and_weights(e, cur_slot); if (rc) return rc; }
    if (e->status_pending) {                  // deliver a deferred data check before this call reuses the words
        HIPCHK(e, hipStreamSynchronize(e->stream));
        rc = synced(e);
        if (rc) return rc;
    }
    // ---- tile form: the whole proposal in ONE kernel (k_gibbs_propose_tile), no candidate slot built ----
    {
        const size_t t_ob = al256((size_t)n_sub * sizeof(int32_t)), t_gb = al256((size_t)C * n_sub * sizeof(int32_t));
        const size_t t_tb = al256((size_t)std::max(n_touched, 1) * sizeof(int32_t));
        const size_t t_in = t_ob + t_gb + t_tb;
        const size_t t_lds = t_in + ((size_t)e->Gtot + (size_t)n_touched * 16 * S) * sizeof(int32_t) + (size_t)n_sub * 16;
        const size_t zbytes_t = (size_t)n_obs * sizeof(double);
        const bool z_map = zbytes_t <= ((size_t)1 << 19);
        const size_t t_zb = z_map ? al256(zbytes_t) : 0;
        const size_t t_idb = al256((size_t)n_obs), t_selb = al256((size_t)n_obs * sizeof(float));
        const size_t t_rowb = al256((size_t)std::max(n_touched, 1) * fs * sizeof(float));
        const size_t t_out = t_idb + 2 * t_selb + t_rowb;
        // (a block serves ALL listed objects for its 16 features: beyond ~128 objects the chain form's grid over every
        //  observation is the faster one -- 154 us against 64 us at 1000 objects, 13 us against 64 us at 30)
        const bool chain_possible = sbe_gibbs_propose_supported(e) == 1;
        if (e->opt_fuse_tables && t_lds <= kGuFusedLdsMax && t_out <= ((size_t)8 << 20) && (n_sub <= 128 || !chain_possible)) {
            rc = ensure_io(e, t_in + t_zb + t_out);
            if (rc) return rc;
            uint8_t* h = e->h_io;
            memcpy(h, objects, (size_t)n_sub * sizeof(int32_t));
            int32_t* gl = reinterpret_cast<int32_t*>(h + t_ob);
            for (int c = 0; c < C; ++c)
                for (int i = 0; i < n_sub; ++i) {
                    const uint16_t gg = cur.h_gid[(size_t)c * N + objects[i]];
                    gl[(size_t)c * n_sub + i] = gg == kNoGroup ? -1 : (int32_t)gg;
                }
            memcpy(h + t_ob + t_gb, touched_out, (size_t)n_touched * sizeof(int32_t));
            const double* d_zt;
            if (z_map) { memcpy(h + t_in, z, zbytes_t); d_zt = reinterpret_cast<const double*>(e->d_io + t_in); }
            else {
                rc = ensure_scratch(e, al256(zbytes_t));
                if (rc) return rc;
                int urc = upload(e, e->d_scratch, z, zbytes_t); if (urc) return urc;
                d_zt = reinterpret_cast<const double*>(e->d_scratch);
            }
            rc = clear_status_word(e, ST_BAD_NORMALIZE);
            if (rc) return rc;
            uint8_t* d_o = e->d_io + t_in + t_zb;
            GibbsTileArgs ta{};
            ta.state = e->d_state; ta.gid = e->d_gid + (int64_t)cur_slot * C * Np; ta.pid = e->d_pid + (int64_t)cur_slot * Np;
            ta.src = e->d_src + (int64_t)cur_slot * N * e->Fp; ta.probs = e->d_probs + (int64_t)cur_slot * e->table_elems();
            ta.wpat = e->d_wpat + (int64_t)cur_slot * e->Pmax * F * C; ta.counts = e->d_counts + (int64_t)cur_slot * e->table_elems();
            ta.conc = e->d_conc;
            ta.mapped_in = reinterpret_cast<const uint32_t*>(e->d_io); ta.in_words = (int)(t_in / 4);
            ta.objects_word = 0; ta.gid_word = (int)(t_ob / 4); ta.touched_word = (int)((t_ob + t_gb) / 4);
            ta.z = d_zt;
            ta.ids_out = d_o; ta.sel_out = (float*)(d_o + t_idb); ta.back_out = (float*)(d_o + t_idb + t_selb);
            ta.rows_out = (float*)(d_o + t_idb + 2 * t_selb);
            ta.n_sub = n_sub; ta.n_touched = n_touched; ta.Gtot = e->Gtot; ta.Np = Np; ta.F = F; ta.S = S; ta.C = C; ta.Fp = e->Fp;
            const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
            ta.inv_t = inv_t; ta.inv_tp = (float)inv_tp; ta.pow_lh = inv_t != 1.0; ta.pow_w = inv_tp != 1.0; ta.from_prior = from_prior != 0;
            ta.status = e->d_status;
            if (follow && n_touched > 0) {
                ta.follow.counts = e->d_counts + (int64_t)cur_slot * e->table_elems();
                ta.follow.probs = e->d_probs + (int64_t)cur_slot * e->table_elems();
                ta.follow.probs_t = e->d_probs_t + (int64_t)cur_slot * e->probs_t_elems();
                ta.follow.ft = e->ft;
                ta.follow.src = e->d_src + (int64_t)cur_slot * N * e->Fp;
            }
            const unsigned blocks = (unsigned)div_up(F, 16);
            const DoneSig done = next_done(e, blocks);
            k_gibbs_propose_tile<<<blocks, kTileBlock, t_lds, e->stream>>>(ta, done);
            HIPCHK(e, hipGetLastError());
            rc = sync_and_report(e, done);
            if (rc) return rc;
            const uint8_t* ho = h + t_in + t_zb;
            memcpy(src_new_out, ho, (size_t)n_obs);
            memcpy(sel_out, ho + t_idb, (size_t)n_obs * sizeof(float));
            memcpy(sel_back_out, ho + t_idb + t_selb, (size_t)n_obs * sizeof(float));
            memcpy(diff_rows_out, ho + t_idb + 2 * t_selb, (size_t)n_touched * fs * sizeof(float));
            return SBE_OK;
        }
    }
    // ---- chain form (tables beyond the tile kernel's LDS image, or SBE_OPT_FUSE_TABLES off): the candidate slot is built ----
    if (!sbe_gibbs_propose_supported(e))
        return fail(e, SBE_ERR_ARG, "sbe_gibbs_propose: tables too large for the fused table kernel (G_total=%d, S=%d)", e->Gtot, e->S);
    // host-mapped block: objects | row_of marks | touched | uniforms (when few) || ids | sel | sel_back | count rows
    const size_t ob = al256((size_t)n_sub * sizeof(int32_t)), rb = al256((size_t)Np * sizeof(int16_t));
    const size_t tb = al256((size_t)std::max(n_touched, 1) * sizeof(int32_t));
    const size_t zbytes = (size_t)n_obs * sizeof(double);
    const bool z_mapped = zbytes <= ((size_t)1 << 19);
    const size_t zb = z_mapped ? al256(zbytes) : 0;
    const size_t idb = al256((size_t)n_obs), selb = al256((size_t)n_obs * sizeof(float));
    const size_t rowb = al256((size_t)std::max(n_touched, 1) * fs * sizeof(float));
    const size_t in_bytes = ob + rb + tb + zb, out_bytes = idb + 2 * selb + rowb;
    if (out_bytes > ((size_t)8 << 20)) return fail(e, SBE_ERR_ARG, "sbe_gibbs_propose: %d objects x %d features exceed the mapped result block", n_sub, F);
    rc = ensure_io(e, in_bytes + out_bytes);
    if (rc) return rc;
    uint8_t* h = e->h_io;
    memcpy(h, objects, (size_t)n_sub * sizeof(int32_t));
    int16_t* row_of = reinterpret_cast<int16_t*>(h + ob);
    std::fill(row_of, row_of + Np, (int16_t)-1);
    for (int i = 0; i < n_sub; ++i) row_of[objects[i]] = 0;
    memcpy(h + ob + rb, touched_out, (size_t)n_touched * sizeof(int32_t));
    const int32_t* d_obj = reinterpret_cast<const int32_t*>(e->d_io);
    const int nblk = div_up(n_obs, kBlock);
    const size_t qb_bytes = al256((size_t)nblk * sizeof(double));
    rc = ensure_scratch(e, 2 * qb_bytes + (z_mapped ? 0 : al256(zbytes)));
    if (rc) return rc;
    double* d_part_f = (double*)e->d_scratch;
    double* d_part_b = (double*)(e->d_scratch + qb_bytes);
    const double* d_z;
    if (z_mapped) { memcpy(h + ob + rb + tb, z, zbytes); d_z = reinterpret_cast<const double*>(e->d_io + ob + rb + tb); }
    else {
        double* dz = (double*)(e->d_scratch + 2 * qb_bytes);
        int urc = upload(e, dz, z, zbytes); if (urc) return urc;
        d_z = dz;
    }
    uint8_t* d_ids = e->d_io + in_bytes;
    float* d_psel_f = (float*)(d_ids + idb);
    float* d_psel_b = (float*)(d_ids + idb + selb);
    float* d_rows = (float*)(d_ids + idb + 2 * selb);
    rc = clear_status_word(e, ST_BAD_NORMALIZE);
    if (rc) return rc;
    const double inv_t = 1.0 / temperature, inv_tp = 1.0 / prior_temperature;
    auto post_args = [&](int slot) {
        return SrcPostArgs{e->d_state, e->d_gid + (int64_t)slot * C * Np, e->d_pid + (int64_t)slot * Np,
                           e->d_probs + (int64_t)slot * e->table_elems(), e->d_wpat + (int64_t)slot * e->Pmax * F * C,
                           d_obj, n_sub, Np, F, S, C, e->Fp, inv_t, (float)inv_tp, inv_t != 1.0, inv_tp != 1.0,
                           from_prior != 0};
    };
    uint8_t* src_cand = e->d_src + (int64_t)cand_slot * N * e->Fp;
    bump_src(e, cand_slot);
    bump_ids(e, cand_slot);
    // 1: the draw (posterior from the current tables) -> the candidate's source rows of the listed objects; p[drawn] out
    k_sample_source<<<nblk, kBlock, 0, e->stream>>>(post_args(cur_slot), d_z, e->rng_seed, e->rng_draw, src_cand, d_psel_f, e->d_status, d_part_f);
    HIPCHK(e, hipGetLastError());
    // 2: the rest of the candidate slot, its count delta and every one of its tables
    Slot cd = cur;
    {
        CoreInputs in;
        in.row_of = reinterpret_cast<const int16_t*>(e->d_io + ob);
        in.src_new = src_cand;
        in.subset = d_obj; in.n_subset = n_sub;
        in.P = (int)cd.patterns.size();
        rc = launch_step_core(e, cur_slot, cand_slot, in);
        if (rc) return rc;
    }
    std::fill(cd.probs_set.begin(), cd.probs_set.end(), 1);
    e->slots[cand_slot] = cd;
    // 3: the candidate's posterior evaluated at the CURRENT source assignment; p_back[old source] out
    k_source_logprob<<<nblk, kBlock, 0, e->stream>>>(post_args(cand_slot), e->d_src + (int64_t)cur_slot * N * e->Fp, d_psel_b, e->d_status, d_part_b);
    HIPCHK(e, hipGetLastError());
    // 4: drawn ids and changed count rows, completion
    const int64_t n_el = n_obs + (int64_t)n_touched * fs;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n_el, 256), 256);
    const DoneSig done = next_done(e, blocks);
    k_gibbs_fetch<<<blocks, 256, 0, e->stream>>>(src_cand, d_obj, n_sub, d_ids, e->d_counts + (int64_t)cur_slot * e->table_elems(),
                                                e->d_counts + (int64_t)cand_slot * e->table_elems(),
                                                reinterpret_cast<const int32_t*>(e->d_io + ob + rb), n_touched, d_rows, F, S, e->Fp, done);
    HIPCHK(e, hipGetLastError());
    rc = sync_and_report(e, done);
    if (rc) return rc;
    memcpy(src_new_out, h + in_bytes, (size_t)n_obs);
    memcpy(sel_out, h + in_bytes + idb, (size_t)n_obs * sizeof(float));
    memcpy(sel_back_out, h + in_bytes + idb + selb, (size_t)n_obs * sizeof(float));
    memcpy(diff_rows_out, h + in_bytes + idb + 2 * selb, (size_t)n_touched * fs * sizeof(float));
    if (follow && n_touched > 0) return sbe_copy_slot(e, cur_slot, cand_slot);       // (the candidate IS the proposal: one copy launch)
    return SBE_OK;
}

int sbe_gibbs_propose(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                      double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                      float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out) {
    return gibbs_propose_impl(e, cur_slot, cand_slot, objects, n_sub, temperature, prior_temperature, from_prior, z, src_new_out, sel_out,
                              sel_back_out, touched_out, n_touched_out, diff_rows_out, 0);
}

int sbe_gibbs_propose_apply(sbe_engine* e, int cur_slot, int cand_slot, const int32_t* objects, int n_sub, double temperature,
                            double prior_temperature, int from_prior, const double* z, uint8_t* src_new_out, float* sel_out,
                            float* sel_back_out, int32_t* touched_out, int32_t* n_touched_out, float* diff_rows_out) {
    return gibbs_propose_impl(e, cur_slot, cand_slot, objects, n_sub, temperature, prior_temperature, from_prior, z, src_new_out, sel_out,
                              sel_back_out, touched_out, n_touched_out, diff_rows_out, 1);
}

}  // extern "C"
