/* Host-side marshalling helpers of the drop-in layer (no device, no engine, plain C): shared by the engine library, which
 * exports them through the C ABI (include/sbe_engine.h: sbe_host_*), and by the CPython extension sbayes_amd/_sbe_pyhost
 * (sbe_pyhost.c), which calls them on NumPy buffers without the ctypes round trip.  Return 0 = ok, 1 = a listed object has
 * no single id (in several groups of one component, or listed twice), -1 = bad argument. */
#ifndef SBE_HOST_HELPERS_H
#define SBE_HOST_HELPERS_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ids_out[i] = offset + g for the one row g of `groups` ([n_groups][n_objects] bool) that has objects[i] set, -1 if none
 * (group_assignment[:, object_subset], sbayes/sampling/counts.py:21-24) */
static inline int sbeh_group_ids(const uint8_t* groups, int n_groups, int64_t n_objects, const int32_t* objects, int n, int offset,
                                 int32_t* ids_out) {
    if ((n_groups > 0 && !groups) || (n > 0 && (!objects || !ids_out)) || n_groups < 0 || n < 0) return -1;
    for (int i = 0; i < n; ++i) {
        const int64_t o = objects[i];
        if (o < 0 || o >= n_objects) return -1;
        int32_t id = -1;
        for (int g = 0; g < n_groups; ++g)
            if (groups[(int64_t)g * n_objects + o]) {
                if (id >= 0) return 1;                 /* in several groups: no single id (the caller counts per group) */
                id = offset + g;
            }
        ids_out[i] = id;
    }
    return 0;
}

/* ids_out[i][f] = the component c with source[objects[i]][f][c] set, 0xFF if none (source[object_subset, :, c],
 * counts.py:25-27); `source` is [n_objects][F][C] bool */
static inline int sbeh_source_ids(const uint8_t* source, int64_t n_objects, int n_features, int n_components, const int32_t* objects,
                                  int n, uint8_t* ids_out) {
    if ((n > 0 && (!source || !objects || !ids_out)) || n < 0 || n_features < 0 || n_components < 1 || n_components > 254) return -1;
    const int64_t row = (int64_t)n_features * n_components;
    for (int i = 0; i < n; ++i) {
        const int64_t o = objects[i];
        if (o < 0 || o >= n_objects) return -1;
        const uint8_t* src = source + o * row;
        uint8_t* out = ids_out + (int64_t)i * n_features;
        /* first set component (argmax), BIT-SELECTED -- no data-dependent branch: the source of an observation is as good as
           random, and the first-hit scan with its mispredicted branches cost ~9 cycles per observation: a third of the host layer's
           update_feature_counts at 1000 x 200 x 2 (round 6).  These loops vectorise. */
        if (n_components == 2) {
            for (int f = 0; f < n_features; ++f) {
                const uint8_t m0 = (uint8_t)-(src[2 * f] != 0), m1 = (uint8_t)-(src[2 * f + 1] != 0);
                out[f] = (uint8_t)(~m0 & ((uint8_t)(m1 & 1) | (uint8_t)~m1));          /* m0: 0; else m1: 1; else 0xFF */
            }
        } else {
            for (int f = 0; f < n_features; ++f) {
                uint8_t id = 0xFF;
                for (int c = n_components - 1; c >= 0; --c) {
                    const uint8_t m = (uint8_t)-(src[(int64_t)f * n_components + c] != 0);
                    id = (uint8_t)((id & ~m) | ((uint8_t)c & m));
                }
                out[f] = id;
            }
        }
    }
    return 0;
}

/* the sorted distinct group indices >= 0 among gid_old / gid_new (np.union1d without the -1s) */
static inline int sbeh_touched_groups(const int32_t* gid_old, const int32_t* gid_new, int64_t count, int n_groups_total,
                                      int32_t* touched_out, int32_t* n_touched_out) {
    if (count < 0 || n_groups_total < 0 || !touched_out || !n_touched_out || (count > 0 && (!gid_old || !gid_new))) return -1;
    static __thread uint8_t* seen = NULL;
    static __thread int seen_cap = 0;
    if (seen_cap < n_groups_total) {
        uint8_t* p = (uint8_t*)realloc(seen, (size_t)n_groups_total);
        if (!p) return -1;
        seen = p; seen_cap = n_groups_total;
    }
    if (n_groups_total) memset(seen, 0, (size_t)n_groups_total);
    for (int64_t i = 0; i < count; ++i) {
        const int32_t a = gid_old[i], b = gid_new[i];
        if (a < -1 || a >= n_groups_total || b < -1 || b >= n_groups_total) return -1;
        if (a >= 0) seen[a] = 1;
        if (b >= 0) seen[b] = 1;
    }
    int32_t n = 0;
    for (int32_t g = 0; g < n_groups_total; ++g) if (seen[g]) touched_out[n++] = g;       /* ascending, like np.union1d */
    *n_touched_out = n;
    return 0;
}

/* everything sbe_counts_delta needs about the listed objects in one pass (drop-in update_feature_counts, counts.py:55-95) */
static inline int sbeh_subset_ids(const int32_t* objects, int n, int64_t n_objects, int n_features, int n_components,
                                  const int32_t* n_groups, const uint8_t* const* groups_new, const uint8_t* const* groups_old,
                                  const uint8_t* source_new, const uint8_t* source_old,
                                  int32_t* gid_new_out, int32_t* gid_old_out, uint8_t* sid_new_out, uint8_t* sid_old_out) {
    if (n < 0 || n_objects < 0 || n_features < 0 || n_components < 1 || n_components > 254 || !n_groups || !groups_new || !groups_old) return -1;
    if (n == 0) return 0;
    if (!objects || !source_new || !source_old || !gid_new_out || !gid_old_out || !sid_new_out || !sid_old_out) return -1;
    /* a listed object counts once (the reference's fancy index would count a repeated one twice): stamps, not a clear per call */
    static __thread uint32_t* seen = NULL;
    static __thread int64_t seen_cap = 0;
    static __thread uint32_t stamp = 0;
    if (seen_cap < n_objects) {
        uint32_t* p = (uint32_t*)realloc(seen, (size_t)n_objects * sizeof(uint32_t));
        if (!p) return -1;
        seen = p; seen_cap = n_objects;
        memset(seen, 0, (size_t)n_objects * sizeof(uint32_t));
        stamp = 0;
    }
    if (++stamp == 0) { memset(seen, 0, (size_t)seen_cap * sizeof(uint32_t)); stamp = 1; }
    for (int i = 0; i < n; ++i) {
        const int64_t o = objects[i];
        if (o < 0 || o >= n_objects) return -1;
        if (seen[o] == stamp) return 1;
        seen[o] = stamp;
    }
    int offset = 0;
    for (int c = 0; c < n_components; ++c) {
        if (n_groups[c] < 0 || (n_groups[c] > 0 && (!groups_new[c] || !groups_old[c]))) return -1;
        int rc = sbeh_group_ids(groups_new[c], n_groups[c], n_objects, objects, n, offset, gid_new_out + (int64_t)c * n);
        if (rc) return rc;
        if (groups_old[c] == groups_new[c]) memcpy(gid_old_out + (int64_t)c * n, gid_new_out + (int64_t)c * n, (size_t)n * sizeof(int32_t));
        else if ((rc = sbeh_group_ids(groups_old[c], n_groups[c], n_objects, objects, n, offset, gid_old_out + (int64_t)c * n)) != 0) return rc;
        offset += n_groups[c];
    }
    if (sbeh_source_ids(source_new, n_objects, n_features, n_components, objects, n, sid_new_out) != 0) return -1;
    if (source_old == source_new) { if (sid_old_out != sid_new_out) memcpy(sid_old_out, sid_new_out, (size_t)n * n_features); }
    else if (sbeh_source_ids(source_old, n_objects, n_features, n_components, objects, n, sid_old_out) != 0) return -1;
    return 0;
}

/* the bind cache's content compare: rows of `rows` that differ bytewise from `mirror` are copied into `mirror`, their
 * indices written to changed_out (ascending); returns how many, -1 on a bad argument */
static inline int64_t sbeh_diff_rows(const void* rows, void* mirror, int64_t n_rows, int64_t row_bytes, int32_t* changed_out) {
    if (n_rows < 0 || row_bytes < 0 || n_rows > INT32_MAX) return -1;
    if (n_rows == 0 || row_bytes == 0) return 0;
    if (!rows || !mirror || !changed_out) return -1;
    const uint8_t* a = (const uint8_t*)rows;
    uint8_t* b = (uint8_t*)mirror;
    int64_t n = 0;
    for (int64_t r = 0; r < n_rows; ++r, a += row_bytes, b += row_bytes)
        if (memcmp(a, b, (size_t)row_bytes) != 0) { memcpy(b, a, (size_t)row_bytes); changed_out[n++] = (int32_t)r; }
    return n;
}

/* The same for the rows listed in cand_a and cand_b only (row indices in any order, repeats allowed; cand_b may be NULL): the
 * caller KNOWS every other row equals the mirror's (binding.py: source lineage).  Differing rows are copied into the mirror and
 * returned in ascending order (a repeated index compares equal the second time); -1 on an index out of range. */
static inline int64_t sbeh_diff_rows_among(const void* rows, void* mirror, int64_t n_rows, int64_t row_bytes, const int32_t* cand_a, int64_t na,
                                           const int32_t* cand_b, int64_t nb, int32_t* changed_out) {
    if (n_rows < 0 || row_bytes < 0 || n_rows > INT32_MAX || na < 0 || nb < 0) return -1;
    if (n_rows == 0 || row_bytes == 0) return 0;
    if (!rows || !mirror || !changed_out || (na && !cand_a) || (nb && !cand_b)) return -1;
    int64_t n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t* cand = pass ? cand_b : cand_a;
        const int64_t nc = pass ? nb : na;
        for (int64_t i = 0; i < nc; ++i) {
            const int64_t r = cand[i];
            if (r < 0 || r >= n_rows) return -1;
            const uint8_t* a = (const uint8_t*)rows + r * row_bytes;
            uint8_t* b = (uint8_t*)mirror + r * row_bytes;
            if (memcmp(a, b, (size_t)row_bytes) != 0) { memcpy(b, a, (size_t)row_bytes); changed_out[n++] = (int32_t)r; }
        }
    }
    for (int64_t i = 1; i < n; ++i) {                     /* (a handful of rows: insertion sort) */
        const int32_t v = changed_out[i];
        int64_t j = i;
        while (j > 0 && changed_out[j - 1] > v) { changed_out[j] = changed_out[j - 1]; --j; }
        changed_out[j] = v;
    }
    return n;
}

#endif
