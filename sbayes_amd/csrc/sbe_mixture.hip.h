// sbe_mixture.hip.h -- what the engine's host units and the units of the fused mixture kernels share: the launch parameters of
// the north-star kernels and their launchers.  The kernels themselves (sbe_kernels_mixture.hip.h: k_mixture_v2 / _onehot_v2 /
// _rows / _combo / _tuple64; sbe_mixture_mfma.hip: k_mixture_tuple_mfma) are templates and most of the library's compile time;
// they are compiled in units of their own (sbe_mixture.hip, sbe_mixture_tuple.hip, sbe_mixture_rows.hip, sbe_mixture_mfma.hip),
// so a change to the engine's host code or to any other kernel does not rebuild them.
#pragma once
#include "sbe_device_common.hip.h"

namespace sbe {

// Log-accumulation modes of the fused kernels:
//   LOG_PER_OBS : fp64 log per observation, fp64 sum
//   LOG_PRODUCT : the observation likelihoods of a step are multiplied into a running mantissa whose
//                 binary exponent is stripped with integer ops; one fp64 log per thread (see ProdAcc)
enum MixMode : int { LOG_PER_OBS = 0, LOG_PRODUCT = 1, WRITE_OBS = 2 };

constexpr int kRowsBlock = 1024;
constexpr int kRowsWaves = kRowsBlock / kWave;

struct Mix2Params {
    int N, NQ, Np, F, Fq, S, C, Gtot, P;
    int n_ftiles, quads_per_chunk;
    int n_work, n_batch;                           // (tile, chunk) work items per slot; slots in this launch
    int slot_groups, slots_per_group;              // XCD-aware block order (see k_mixture_v2)
    // group-tuple form (k_mixture_combo): per slot the distinct (g_0..g_{C-1}) tuples of its objects
    const uint8_t* tid;      int64_t tid_stride;       // [Np] tuple index per object
    const uint16_t* tuple_g; int64_t tuple_g_stride;   // [kMaxTuples][kMaxComponents] global group index (Gtot = none)
    const uint8_t* tuple_p;  int64_t tuple_p_stride;   // [kMaxTuples] pattern id of the tuple
    int KT;                                            // tuples used by the slots of this launch (max)
    int combo_w_off;                                   // byte offset of the weight tile in the combo kernel's LDS
    int combo_tab_off;                                 // byte offset of the one-hot byte -> (state, feature) table
    const uint32_t* state_q;                       // [NQ][Fq]
    const uint2* state_h;                          // [NQ][Fq] 4 x u16 prepared LDS offsets (k_mixture_tuple64), or null
    const uint32_t* toff;  int64_t toff_stride;    // per slot [Np] byte offset of the object's tuple block
    const double2* logtab;                         // [128] {1/c, log c} of tab_log_pos
    int gen_slots;                                 // k_mixture_tuple64 block order: slots per XCD and generation
    int rows_cum[5];                               // k_mixture_rows: cumulative per-mille shares of a block's steps by wave age class
    int ragged_w;                                  // valid features of the last tile if it runs in sub-row mode (<= 32), else 0
    const uint8_t* onehot; int rs_pitch;           // [N][rs_pitch] (one-hot variant)
    const uint16_t* gid;   int64_t gid_stride;     // per slot [C][Np]
    const uint8_t* pid;    int64_t pid_stride;     // per slot [Np]
    const float* probs_t;  int64_t probs_t_stride; // per slot [n_ftiles][(Gtot+1)*S*FT]
    const double* wpat_t;  int64_t wpat_t_stride;  // per slot [n_ftiles][Pmax*C*FT]
    int wpat_tile_stride;                          // Pmax*C*FT
    double* partials;      int64_t partials_stride;
    // final reduction inside the kernel (finish_partial): results != nullptr, tickets per slot in arrive[], optional signal
    double* results;
    unsigned* arrive;
    DoneSig done;
    int first_slot;
    const int32_t* slot_list;                      // slots of this launch (n_batch entries), or null: first_slot + i
    // rows kernel (k_mixture_rows): engine tile width of probs_t, canonical per-pattern weights, per-object row offsets
    int eft;                                       // tile width of probs_t (64 / 32 / 16)
    const float* wpat;     int64_t wpat_stride;    // per slot [Pmax][F][C] float32 normalised weights (a5)
    const uint32_t* rowoff; int64_t rowoff_stride; // per slot [C+1][Np]: LDS byte offsets (see k_rowoff)
    // pattern-sorted form of the rows kernel (k_rowsort): per slot the objects in pattern order, runs padded to whole wave steps
    const uint32_t* rowoff_s; int64_t rowoff_s_stride;   // per slot [NQs][C+1][4]: C table-row offsets, then pattern << 24 | state-row byte offset
    const int32_t* rs_nq;                          // per slot: quads of its padded order
    const uint8_t* state_s; int state_s_pitch;     // [N + 1][pitch] state index per observation, NA = S; row N: the null object (all NA)
};

// batched group-tuple form on the matrix pipe (sbe_mixture_mfma.hip: k_mixture_tuple_mfma)
struct MfmaMixParams {
    int F, S, FS, Gtot, Np;
    int NT, KBp;                                   // 32-column tiles of the (feature, state) axis; k-blocks of 64 (FP4 operands) / 32 (i8) objects, padded to a multiple of 4
    int KT;                                        // tuples used by the slots of this launch (max; <= 64)
    int SL;                                        // slots per block: 16 (KT <= 8), 4 (<= 32) or 2 (<= 64) -- the widest whose A image fits LDS
    int n_batch, n_split, nt_per_split;            // slots of the launch; column splits (blocks per group of 16 slots); column tiles per split
    int first_slot;
    const int32_t* slot_list;                      // slots of this launch (n_batch entries), or null: first_slot + i
    const uint8_t* xt;  uint32_t xt_bytes;         // [NT + 1][KBp][64][16] (+ PF fragments) one-hot block in fragment order (k_xt_frags); tile NT is zero
    const uint8_t* tid;      int64_t tid_stride;       // per slot [Np] tuple index per object
    const uint16_t* tuple_g; int64_t tuple_g_stride;   // per slot [kMaxTuples][kMaxComponents] global group index (Gtot = none)
    const uint8_t* tuple_p;  int64_t tuple_p_stride;   // per slot [kMaxTuples] pattern id of the tuple (0xFF: tuple not there)
    const float* probs;      int64_t probs_stride;  uint32_t probs_bytes;   // per slot [Gtot][F][S] float32 tables (a4), whole array < 4 GiB
    const float* wpat;       int64_t wpat_stride;   uint32_t wpat_bytes;    // per slot [Pmax][F][C] float32 normalised weights (a5)
    uint32_t probs_ones_off, wpat_ones_off;        // byte offsets of the rows of ones behind the two arrays (F*S / F*C floats)
    const double2* logtab;                         // [1024] {1/c, log c}: this kernel's own finer table (tab_log4_n)
    const int32_t* colcount;                       // [(NT + 1) * 32] objects per (feature, state) column over all tuples (k_colcount); zero padding
    const int32_t* colfeat;                        // [(NT + 1) * 32] feature of the column
    const int32_t* tile_prefix;                    // [NT + 2] observations counted in the column tiles before t
    double* partials;        int64_t partials_stride;
    // final reduction inside the kernel (results != nullptr): the LAST of a slot group's n_split blocks to finish -- tickets in
    // arrive[group], which it leaves at 0 -- adds the group's partial sums in split order and writes the 16 results; `done`
    // (optional) is signalled by those blocks, one per slot group
    double* results;
    unsigned* arrive;
    DoneSig done;
};

// waves per block of k_mixture_tuple64.  (8-wave blocks -- twice the waves per SIMD at the same LDS footprint -- were
// measured twice: 72.7 us at 80 VGPRs / 3 blocks per CU, 107 us at 64 VGPRs / 4 blocks per CU, against 61-63 us: the
// kernel does not fit those register budgets without spilling in its table build.)
constexpr int tuple64_waves() { return 4; }

// Launchers (sbe_mixture.hip): pick the template instance for (mode, tile width, component count) and enqueue it.
// mode: LOG_PER_OBS / LOG_PRODUCT; ft: 64 / 32 / 16; C: components (1..4 compile-time instances, else the runtime form).
void launch_v2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct);
void launch_oh2(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool direct);
void launch_combo(bool onehot, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st);
void launch_tuple64(int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st);
void launch_rows(int mode, int ft, int C, const Mix2Params& p, dim3 grid, size_t lds, hipStream_t st, bool sorted = false);
// the sorted form's inputs: the slot's objects by has_components pattern (one wave per slot), the NA = S state block
void launch_rowsort(const uint16_t* gid, const uint8_t* pid, uint32_t* out, int32_t* nq_out, int64_t gid_stride, int64_t pid_stride,
                    int64_t out_stride, int first_slot, const int32_t* slot_list, int n_slots, int N, int Np, int C, int Gtot, int Pmax,
                    uint32_t row_bytes, uint32_t state_pitch, int step_objects, hipStream_t st);
void launch_state_s(const uint8_t* state, uint8_t* state_s, int N, int F, int Fp, int pitch, int S, hipStream_t st);
// sbe_mixture_mfma.hip
bool tuple_mfma_fp4();                            // operand format of the count contraction: FP4 (default; a k-block = 64 objects) or i8 (32)
inline int tuple_mfma_kblock_objects() { return tuple_mfma_fp4() ? 64 : 32; }
int tuple_mfma_slots_per_block(int KT);           // 16 (KT <= 8) / 4 (<= 32) / 2 (<= 64): the MOST slots per block KT tuples allow; 0: the form does not apply
void launch_xt_frags(const uint8_t* state, uint8_t* xt, int N, int F, int S, int Fp, int NT, int KBp, bool fp4, hipStream_t st);
size_t column_tables_bytes(int NT);               // colcount | colfeat | tile_prefix (MfmaMixParams), one allocation
void launch_column_tables(const uint8_t* state, int32_t* out, int N, int F, int S, int Fp, int NT, hipStream_t st);
// sbe_mixture_mfma_ws.hip: the wave-specialised form (producer waves count, consumer waves evaluate); FP4 operands only
size_t tuple_mfma_ws_lds_bytes(int MT, int C, int KBp);
bool launch_tuple_mfma_ws(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st);
size_t tuple_mfma_lds_bytes(int MT, int C, int KBp);
void fine_log_table(double* tab);                 // [2 * 1024] {1/c, log c} of k_mixture_tuple_mfma's log (sbe_mixture_mfma.hip)
// false (nothing launched): an instance of the kernel carries static LDS, so its dynamic block does not start at address 0
bool launch_tuple_mfma(int C, const MfmaMixParams& p, dim3 grid, size_t lds, hipStream_t st);
constexpr int kTupleMfmaColsPerPass = 2;          // column tiles a wave of k_mixture_tuple_mfma owns per pass (= columns per lane)

}  // namespace sbe
