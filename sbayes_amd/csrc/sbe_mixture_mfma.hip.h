// sbe_mixture_mfma.hip.h -- what the two matrix-pipe kernels of the batched group-tuple form share (sbe_mixture_mfma.hip:
// k_mixture_tuple_mfma, every wave counts and evaluates; sbe_mixture_mfma_ws.hip: k_mixture_tuple_mfma_ws, producer waves count,
// consumer waves evaluate): fragment types, the table-driven log, the tuple metadata and phase 0 of a block (metadata, log
// table and the A fragments -- the indicator image [tid == t] of the block's 16 slots -- into LDS).
#pragma once
#include <type_traits>

#include "sbe_mixture.hip.h"

namespace sbe {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const v4i_t lds_cv4i_t;

constexpr int kMfmaSlots = 16;          // slots per block in the narrow form (<= 8 tuples); wide forms: 4 (<= 32 tuples), 2 (<= 64)

// Table-driven log for this kernel's epilogue, G chains interleaved: log v = k ln2 + log c_i + log1p(r), r = m / c_i - 1, with
// its own FINER table (kFineLogEntries = 1024 intervals of the mantissa m in [1, 2): {RN(1/c_i) / 2, log c_i}, built by ensure_xt) so
// that |r| <= 2^-11 (2^-10 in interval 0) and the series stops after r^4: first dropped term r^5 / 5 <= 6e-18 per log (2e-16 in
// interval 0).  v_fma_f64 is the dearest instruction of the epilogue (tools/probe/valu_rates.hip: 6.5 cycles against 4.0 for
// 32-bit integer work at two waves per SIMD): against the 128-entry / r^5 form this is one FMA fewer per entry.
// The binary exponent is NOT folded in here: it is returned as an integer and summed exactly, count-weighted, by the caller (one
// v_mad_i32_i24 instead of a conversion and an FMA per entry; k ln2 is applied once per accumulator at the end).  So that the
// two sums never cancel, mantissas from sqrt(2) on (table index >= kFineLogSplit) count as m / 2 with k + 1 -- the exponent comes
// from hi + kFineLogCarry, which carries exactly there, and their table rows hold log(c_i / 2) -- so a probability in [0.707, 1)
// has k = 0 and a small negative mantissa part; the centres of the first and the last interval are 1 and 2 (log exactly 0):
// log 1 = 0 exactly, and for v = 1 - eps the result is log1p(-eps) to 1e-13 relative (a state that every table entry gives
// probability ~1 has a log-likelihood of ~0, which the reference gets to its own rounding: tools/fuzz_gpu.py checks to
// 1e-10 relative + 1e-16 per observation, and caught a first form that summed k ln2 and log m apart without the split:
// 2e-12 off at a log-likelihood of 1e-5).
constexpr int kFineLogEntries = 1024;
constexpr int kFineLogSplit = 424;                                        // 1 + 424/1024 = 1.4140625 ~ sqrt(2)
constexpr uint32_t kFineLogCarry = 0x00100000u - ((uint32_t)kFineLogSplit << 10);
// (round 6: the exponent comes back BIASED -- (hi + carry) >> 20, no "- 1023" per entry: the caller takes 1023 x the column's
//  object count off the integer sum once per lane -- and the mantissa's high word is one v_bfi_b32 with the mask in an SGPR and the
//  exponent pattern in a VGPR instead of v_and + v_or with two literals: two vector instructions fewer per table entry.)
constexpr int kFineLogBias = 1023;
template <int G>
__device__ __forceinline__ void tab_log4_n(const double (&v)[G], double (&out)[G], int (&kexp)[G], uint32_t tab, uint32_t one_hi /* 0x3FF00000 in a VGPR */) {
    f64x2_t e[G];
    double m[G], r[G], q[G];
    uint32_t mant_mask = 0x000FFFFFu;
    asm volatile("" : "+s"(mant_mask));                                          // (kept in an SGPR: v_bfi_b32 takes no literal)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t hi = (uint32_t)__double2hiint(v[g]);
        e[g] = *(lds_cf64x2_t*)(uintptr_t)(tab + ((hi >> 6) & 0x3FF0u));        // entry (hi >> 10) & 1023
        uint32_t mh;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(mh) : "s"(mant_mask), "v"(hi), "v"(one_hi));   // (hi & 0xFFFFF) | 0x3FF00000
        m[g] = __hiloint2double((int)mh, __double2loint(v[g]));
        kexp[g] = (int)((hi + kFineLogCarry) >> 20);                              // biased by kFineLogBias
    }
    // log1p(r) to r^4 in s = r / 2 (the table holds 1 / (2 c), so s comes out of the first FMA):
    //     r - r^2/2 + r^3/3 - r^4/4 = s (2 + s (-2 + s (8/3 - 4 s)))
    // Every constant but 8/3 is an inline operand, so the chain is five v_fma_f64 and nothing else (the Horner form in r needs
    // two constants in one instruction: a register move per entry, and a separate r^2 and final add).
#pragma unroll
    for (int g = 0; g < G; ++g) r[g] = fma(m[g], e[g].x, -0.5);               // s
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(r[g], -4.0, 8.0 / 3.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(r[g], q[g], -2.0);
#pragma unroll
    for (int g = 0; g < G; ++g) q[g] = fma(r[g], q[g], 2.0);
#pragma unroll
    for (int g = 0; g < G; ++g) out[g] = fma(r[g], q[g], e[g].y);
}

// per (slot of the block, tuple): byte offsets of the tuple's probability rows inside the probs array and of its
// pattern's weight rows inside the wpat array
template <int CT>
struct __attribute__((aligned(CT <= 1 ? 8 : (CT <= 3 ? 16 : 32)))) TupleMeta {
    uint32_t woff;
    uint32_t goff[CT];
};

// Phase 0 of a block of NTHR threads (all of them call it; the caller's barrier follows): tuple metadata, log table, A fragments.
// `slot_of(sl)`: absolute slot of the block's sl-th slot, or -1.  LDS: log table at tab_off (= 0), A fragments from a_off,
// metadata at `meta`.
// SL = slots per block (16 / 4 / 2): an M tile's 32 rows are 32 / SL tuples x SL slots, row = (tuple in tile) * SL + slot; the
// block holds TPB = MT * 32 / SL tuples.
template <int MT, int CT, bool FP4, int NTHR, typename SlotOf, int SL = kMfmaSlots>
__device__ __forceinline__ void mfma_phase0(unsigned char* lds_raw, const MfmaMixParams& p, const SlotOf& slot_of, uint32_t tab_off,
                                            uint32_t a_off, TupleMeta<CT>* meta, int KBp) {
    typedef TupleMeta<CT> Meta;
    constexpr int kMfmaThreads = NTHR;
    constexpr int TPT = 32 / SL, TPB = MT * TPT;                  // tuples per M tile / per block
    static_assert(SL == 16 || SL == 4 || SL == 2, "slots per block");
    static_assert(!(SL != 16 && !FP4), "the wide forms exist with FP4 operands only");
    // ---- phase 0: tuple metadata, log table, A fragments ------------------------------------------------------------
    // Offsets of a tuple that is not there (another slot's tuple, the padding tuple of an odd KT, a slot beyond the batch)
    // and of a component the tuple has no group in point at the rows of ONES behind the two arrays: no observation is
    // counted on the former (and log C is an ordinary number), the normalised weight of the latter is exactly 0.
    // Every global load of the phase is ASKED FOR FIRST -- the tuple ids of this thread's first units, its share of the log table,
    // the tuple's pattern and groups (unconditionally: the rows exist for every tuple index) -- and consumed afterwards: one round
    // trip to L2 / HBM instead of three or four dependent ones at the start of every block (round 6).
    auto meta_of = [&](int slot, int t, uint32_t pat, const uint32_t (&g)[CT]) -> Meta {
        Meta md;
        md.woff = p.wpat_ones_off;
#pragma unroll
        for (int c = 0; c < CT; ++c) md.goff[c] = p.probs_ones_off;
        if (slot >= 0 && t < p.KT && pat != 0xFFu) {
            md.woff = (uint32_t)(((int64_t)slot * p.wpat_stride + (int64_t)pat * p.F * CT) * 4);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if ((int)g[c] < p.Gtot) md.goff[c] = (uint32_t)(((int64_t)slot * p.probs_stride + (int64_t)g[c] * p.FS) * 4);
        }
        return md;
    };
    if constexpr (FP4) {
        // one unit = the 32 tuple ids of (slot sl, lane half h of k-block kb: 32 objects) -> the 2 MT indicator pieces (16 bytes =
        // 32 nibbles each).  UB units' ids are asked for together.
        const int n_units = SL * KBp * 2;
        constexpr int UB = 2;
        uint32_t d[UB][8];
        auto load_ids = [&](int u0) {
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * kMfmaThreads;
                const int sl = u % SL, n0 = (u / SL) * 32;     // (u / SL) = kb * 2 + h
                const int slot = u < n_units ? slot_of(sl) : -1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    d[k][i] = 0xFFFFFFFFu;                                        // matches no tuple
                    if (slot >= 0 && n0 + 4 * i + 4 <= p.Np)
                        d[k][i] = *reinterpret_cast<const uint32_t*>(p.tid + (int64_t)slot * p.tid_stride + n0 + 4 * i);
                }
            }
        };
        auto emit = [&](int u0) {
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * kMfmaThreads;
                if (u >= n_units) break;
                const int sl = u % SL, hk = u / SL;
                const int kb = hk >> 1, h = hk & 1;
#pragma unroll 8
                for (int t = 0; t < TPB; ++t) {
                    uint4 o;
                    uint32_t* ov = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) {
                        uint32_t eq[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const uint32_t x = d[k][2 * wv + q] ^ ((uint32_t)t * 0x01010101u);
                            const uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;      // bit 7 of a byte: byte != 0
                            eq[q] = (~nz >> 7) & 0x01010101u;
                        }
                        ov[wv] = (eq[0] << 1) | (eq[1] << 5);       // 1.0 = 0x2: low nibbles = objects 8 wv + b, high = 8 wv + 4 + b
                    }
                    // fragment (m = t / TPT, kb): lane = h * 32 + row, row = (t % TPT) * SL + sl
                    const uint32_t fl = (uint32_t)(h * 32 + (t % TPT) * SL + sl);
                    *reinterpret_cast<uint4*>(lds_raw + a_off + (((uint32_t)(t / TPT) * (uint32_t)KBp + (uint32_t)kb) * 64u + fl) * 16u) = o;
                }
            }
        };
        const int u_first = (int)threadIdx.x;
        load_ids(u_first);
        constexpr int LT = (2 * kFineLogEntries + kMfmaThreads - 1) / kMfmaThreads;
        double lt[LT];
#pragma unroll
        for (int k = 0; k < LT; ++k) {
            const int i = (int)threadIdx.x + k * kMfmaThreads;
            lt[k] = i < 2 * kFineLogEntries ? reinterpret_cast<const double*>(p.logtab)[i] : 0.0;
        }
        const bool has_meta = (int)threadIdx.x < SL * TPB;
        const int m_sl = (int)threadIdx.x / TPB, m_t = (int)threadIdx.x % TPB;
        const int m_slot = has_meta ? slot_of(m_sl) : -1;
        uint32_t m_pat = 0xFFu, m_g[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) m_g[c] = 0xFFFFFFFFu;
        if (m_slot >= 0) {                                   // (m_t < TPB <= kMaxTuples: the rows exist whatever KT is)
            m_pat = p.tuple_p[(int64_t)m_slot * p.tuple_p_stride + m_t];
#pragma unroll
            for (int c = 0; c < CT; ++c) m_g[c] = p.tuple_g[(int64_t)m_slot * p.tuple_g_stride + m_t * kMaxComponents + c];
        }
        if (has_meta) meta[m_sl * TPB + m_t] = meta_of(m_slot, m_t, m_pat, m_g);
#pragma unroll
        for (int k = 0; k < LT; ++k) {
            const int i = (int)threadIdx.x + k * kMfmaThreads;
            if (i < 2 * kFineLogEntries) reinterpret_cast<double*>(lds_raw + tab_off)[i] = lt[k];
        }
        for (int u0 = u_first; u0 < n_units; u0 += UB * kMfmaThreads) {
            if (u0 != u_first) load_ids(u0);
            emit(u0);
        }
    } else {
        if ((int)threadIdx.x < kMfmaSlots * 2 * MT) {
            const int sl = (int)threadIdx.x / (2 * MT), t = (int)threadIdx.x % (2 * MT);
            const int slot = slot_of(sl);
            uint32_t pat = 0xFFu, g[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) g[c] = 0xFFFFFFFFu;
            if (slot >= 0) {
                pat = p.tuple_p[(int64_t)slot * p.tuple_p_stride + t];
#pragma unroll
                for (int c = 0; c < CT; ++c) g[c] = p.tuple_g[(int64_t)slot * p.tuple_g_stride + t * kMaxComponents + c];
            }
            meta[sl * 2 * MT + t] = meta_of(slot, t, pat, g);
        }
        for (int i = (int)threadIdx.x; i < 2 * kFineLogEntries; i += kMfmaThreads)
            reinterpret_cast<double*>(lds_raw + tab_off)[i] = reinterpret_cast<const double*>(p.logtab)[i];
        // one unit = the 16 tuple ids of (slot sl, 16 objects) -> the 2 MT indicator pieces of those objects.  The ids of
        // UB units are asked for together (a unit at a time the block's start is four dependent trips to L2 / HBM)
        const int n_units = kMfmaSlots * KBp * 2;
        constexpr int UB = 4;
        for (int u0 = (int)threadIdx.x; u0 < n_units; u0 += UB * kMfmaThreads) {
            uint32_t d[UB][4];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * kMfmaThreads;
                const int sl = u & 15, n0 = (u >> 4) * 16;     // (u >> 4) = kb * 2 + h
                const int slot = u < n_units ? slot_of(sl) : -1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[k][i] = 0xFFFFFFFFu;                                        // matches no tuple
                    if (slot >= 0 && n0 + 4 * i + 4 <= p.Np)
                        d[k][i] = *reinterpret_cast<const uint32_t*>(p.tid + (int64_t)slot * p.tid_stride + n0 + 4 * i);
                }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * kMfmaThreads;
                if (u >= n_units) break;
                const int sl = u & 15, hk = u >> 4;
                const int kb = hk >> 1, h = hk & 1;
#pragma unroll
                for (int t = 0; t < 2 * MT; ++t) {
                    uint4 o;
                    uint32_t* ov = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t x = d[k][i] ^ ((uint32_t)t * 0x01010101u);
                        const uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;      // bit 7 of a byte: byte != 0
                        ov[i] = (~nz >> 7) & 0x01010101u;
                    }
                    // fragment (m = t >> 1, kb): lane = h * 32 + (t & 1) * 16 + sl
                    const uint32_t fl = (uint32_t)(h * 32 + (t & 1) * 16 + sl);
                    *reinterpret_cast<uint4*>(lds_raw + a_off + (((uint32_t)(t >> 1) * (uint32_t)KBp + (uint32_t)kb) * 64u + fl) * 16u) = o;
                }
            }
        }
    }
}

}  // namespace sbe
