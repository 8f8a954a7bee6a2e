// tsr_internal.h -- shared internals of libtsr_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tsr.h"

#define TSR_API extern "C" __attribute__((visibility("default")))

void tsr_set_error(const char *fmt, ...);

#define TSR_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            tsr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                      \
            return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? TSR_ERR_NO_DEVICE \
                                                                          : TSR_ERR_HIP;  \
        }                                                                                 \
    } while (0)

#define TSR_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            tsr_set_error(__VA_ARGS__); \
            return TSR_ERR_INVALID;   \
        }                             \
    } while (0)

#define TSR_TRY(expr)                 \
    do {                              \
        int _s = (expr);              \
        if (_s != TSR_OK) return _s;  \
    } while (0)

// A growable device scratch buffer owned by the ctx (never shrinks).
struct tsr_devbuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct tsr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cus = 0;
    int64_t hbm_bytes = 0;
    char name[128] = {0};
    tsr_devbuf scratch[8];  // per-purpose scratch slots (see tsr_scratch)
    tsr_devbuf pinned;      // page-locked host staging for large downloads (see tsr_download)
    void *comm = nullptr;   // ncclComm_t when tsr_dist_init has run
    void *d_coll = nullptr; // 64 device bytes for the small collectives of tsr_bank_broadcast (allocated with the communicator)
    int nranks = 1, rank = 0;
    // kernel-choice options (include/tsr.h: tsr_ctx_set_option); defaults from the environment at creation only
    int opt_fast_kernel = 0, opt_exact_kernel = 0, opt_part_layout = 0, opt_xcd_map = 1;
    // TSR_EXACT on the matrix cores rests on v_mfma_f32_16x16x4_f32 evaluating its four products as an ascending fmaf
    // chain: 0 = not yet checked on this device, 1 = the probe matched the vector kernel bit for bit, -1 = it did not
    // (the vector kernel serves every TSR_EXACT call of this context, said once on stderr)
    int exact_mfma_state = 0;
    int last_bank_kernel = 0;               // TSR_KERNEL_* of the latest scoring pass (TSR_INFO_LAST_BANK_KERNEL)
    bool exact_mfma_could_not_run = false;  // the last probe failed to run (state stays 0: probed again)
};

int tsr_scratch(tsr_ctx *ctx, int slot, size_t bytes, void **out);
// Synchronous device -> pageable host copy on ctx->stream; large ones go through the page-locked staging
// buffer (a D2H straight into fresh pageable memory re-pins the pages on every call: 0.4 vs 5 ms for 3 MB).
int tsr_download(tsr_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int tsr_corpus_create_empty(tsr_ctx *ctx, const int64_t *frame_offsets, int64_t N, int64_t D,
                            struct tsr_corpus **out, bool alloc_bytes = true, const int64_t *d_off_ready = nullptr, bool tables = true);

// ---- bank ---------------------------------------------------------------------
// Exact path: templates in groups of TSR_MG; group g holds, for every (tau, d),
// TSR_MG floats  Wx[g][tau][d][m] = W_{g*MG+m}[tau][d]  (zero padded in tau beyond
// L_m, in d up to a multiple of 32 and for m beyond M), so that one wave-uniform
// 32-byte line feeds 8 templates for one feature.
#define TSR_MG 8

// TSR_FAST uses three limbs when the SHORTEST filter of the bank has at most this many taps (Lmin * D).
#define TSR_FAST_W24_BELOW_TAPS 4096
// ... and when ANY filter has more than this many taps: the two-limb path keeps its window sums in an
// int32 ring, which holds 16-bit weights (|q| <= 127*256) only up to floor((2^31-1) / (127*256)) set taps.
#define TSR_FAST_W24_ABOVE_TAPS 66052

// Fast (MFMA) path: see detect_mfma_kernel.inc for the limb layout.
struct tsr_bank {
    tsr_ctx *ctx = nullptr;
    int device = 0;
    int32_t M = 0;
    int64_t D = 0;
    int32_t Lmax = 0;
    int64_t sumL = 0;
    std::vector<int32_t> L;   // host copies
    std::vector<double> c;
    std::vector<float> W_cat; // host copy (needed to rebuild device forms after a broadcast)
    // device, exact path
    int32_t n_groups = 0;
    int32_t Dq = 0;                  // ceil(D/32): dwords of feature bits per frame
    float *d_Wx = nullptr;           // all groups back to back
    std::vector<int64_t> gx_off;     // float offset of group g in d_Wx
    std::vector<int32_t> gx_L;       // max L in group g
    int64_t *d_gx_off = nullptr;     // device copies of the two tables above
    int32_t *d_gx_L = nullptr;
    int32_t *d_L = nullptr;          // [M]
    float *d_cf = nullptr;           // [M]  f32(c_m)
    // exact path on the matrix cores (detect_exact_mfma.hip): B fragments [g][tau' = 0..Lg][Dp/16][lane] float4;
    // absent (null) when a weight is not finite or small enough for subnormal partial sums
    float *d_Wm = nullptr;
    int64_t *d_gm_off = nullptr;     // float4 offset of group g in d_Wm
    int32_t *d_gm_rows = nullptr;    // tap rows of group g in d_Wm: max(L_g) + 1 over the group's pair (2p, 2p + 1)
    int wm_Dp = 0;                   // resident row stride the fragments were laid out for
    int64_t wm_total4 = 0;           // float4 elements of d_Wm (without its two fragments of read-ahead slack)
    std::vector<int64_t> gm_off_h;   // host copies of d_gm_off / d_gm_rows
    std::vector<int32_t> gm_rows_h;
    // device, fast path (built lazily by detect_mfma.hip): [0] two int8 limbs, [1] three, [2] bf16, [3] bf16 hi + lo,
    // [4] two int8 limbs with two taps folded into K (tap2_bank_kernel, narrow feature rows)
    void *fast[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool tap2_unsupported = false;   // the kind-4 build was tried and the geometry does not fit
    // TSR_MX (detect_mx.hip): fp6 limb fragments of the Toeplitz form, built lazily
    void *mx = nullptr;
    bool mx_unsupported = false;
};

struct tsr_corpus {
    tsr_ctx *ctx = nullptr;
    int device = 0;
    int64_t N = 0, D = 0, P = 0;     // P = total frames
    int Dp = 0;                      // resident row stride: round_up(D, 16)
    uint8_t *d_feats = nullptr;      // [P, Dp] bytes in {0,1}, rows zero padded (+ zeroed slack
                                     // behind the last row so that tiles may over-read)
    int32_t *d_utt_end = nullptr;    // [P] end frame (exclusive) of the utterance holding p
    int32_t *d_utt_beg = nullptr;    // [P] first frame of the utterance holding p
    float *d_scores = nullptr;       // [P]
    int32_t *d_ids = nullptr;        // [P]
    bool scored = false;             // d_scores / d_ids hold the result of a max-mode scoring pass (they come from the pool uninitialised)
    std::vector<int64_t> offsets;    // host [N+1]
    // fast path: cached partition of the frame stream into per-workgroup segments
    int plan_key = 0;                // (segments, Lmax) the plan was built for
    int plan_nseg = 0;
    int32_t *d_plan = nullptr;       // [3][nseg]: first output, end of outputs, end of rows
    void *d_planes = nullptr;        // bit planes of the frames (detect_exact_mfma.hip), built on first use by TSR_EXACT
    int64_t planes_nb = 0;           // blocks of 32 frames per parity the planes cover
    uint16_t *d_feats_bf16 = nullptr; // bf16 copy of the frames (rows of bf16_stride bytes), built on first use by TSR_BF16
    int bf16_stride = 0;
    size_t feats_alloc = 0, bf16_alloc = 0;  // bytes allocated behind d_feats / d_feats_bf16 (frames + zeroed slack)
    // Bit form of the features, the resident form tsr_corpus_part_features leaves behind (an eighth of the byte form):
    // bit (d mod 32) of dword d / 32 of a frame = its feature d, and the dwords of a frame are laid out PHASE-MAJOR --
    // [bit_nph phases][bit_rows frames][bit_pw dwords], dword w of frame p at ((w / bit_pw) * bit_rows + p) * bit_pw +
    // w % bit_pw -- where a phase is the feature range one pass of the two-limb bank kernel holds in registers
    // (tsr_bit_geometry: a pure function of D).  A K-phase of the bank pass then reads one contiguous plane, 32 frames
    // of a tile = 32 * bit_pw consecutive dwords, and every byte of the form leaves HBM once per bank pass (row-major
    // bit rows made every phase pull every 128-byte line of the whole form).  bit_rows = P + 48 (read-ahead of the last
    // tile).  A corpus that has this form may lack d_feats until something needs byte rows (tsr_corpus_need_bytes).
    uint8_t *d_bits = nullptr;
    int bit_pw = 0, bit_nph = 0;
    int64_t bit_rows = 0;
    size_t bits_alloc = 0;
    // Bernoulli mixture: the posterior component probabilities the latest E-step left on the device
    float *d_aff = nullptr;          // [P][aff_K]
    int64_t aff_K = 0, aff_cap = 0;  // columns in use; capacity in floats
    std::vector<double> aff_colsum;  // [aff_K] column sums of d_aff (the M-step's weights x P)
    void *mx = nullptr;              // TSR_MX (detect_mx.hip): bit rows at even-aligned positions + the tile plan, built on first use
};

struct tsr_parts {
    tsr_ctx *ctx = nullptr;
    int device = 0;
    int64_t K = 0, P0 = 0, P1 = 0, E = 0;
    double *d_tab = nullptr;  // [q][k][2] = {log_invparts, log_parts}, q = (i*P1+j)*E+z
    // MFMA coder of the corpus front-end (parts_mfma.hip): 16-bit fixed-point weight limbs
    int mode = 0;             // TSR_PARTS_AUTO / _VALU / _MFMA (tsr_parts_set_mode)
    bool mfma_ok = false;     // tables and geometry fit the MFMA coder
    char mfma_why[96] = {0};  // ... or why not
    int nkb = 0, pm_NG = 0, pm_n_inv = 0;
    float pm_o0_const = 0, pm_o0_slack = 0;
    void *d_wq = nullptr;     // int8 [mt 2][kb][limb 2][lane 64][16]
    int32_t *d_kc = nullptr;  // [half 2][mt 2][16]: (base sum << 6) | part
    int64_t last_locations = 0, last_rechecked = -1;  // of the latest tsr_corpus_part_features (-1: VALU kernel ran)
};

// Read-ahead bound of the bank kernels' frame staging.  A workgroup stages whole 32-row tiles by LDS-DMA, AH tiles
// ahead of the one it computes, but only tiles t < nT = ceil((r1 - o0) / 32) of its segment, and r1 <= P
// (build_plan): the first row of the last staged tile is a frame of the corpus, at most row P - 1.  From that row a
// tile copy reads `tile_bytes` contiguous bytes (flat rows: LDS row = resident row) or, row by row,
// ceil(tile_bytes / SB) rows of src_slots 16-byte slots at a stride of Dp bytes.  So the staging never reads at or
// beyond the offset returned here (relative to the first frame of the array it reads, `phase_off` bytes into each
// row for K-phases) -- which has to lie inside the allocation: frames + zeroed slack.  Checked before every launch
// (TSR_REQUIRE), for the resident bytes and for the bf16 copy alike.
static inline size_t tsr_staging_read_end(int64_t P, int Dp, int SB, int src_slots, int tile_bytes, bool flat, size_t phase_off)
{
    if (P <= 0) return 0;
    const size_t first = (size_t)(P - 1) * (size_t)Dp + phase_off;
    if (flat) return first + (size_t)tile_bytes;
    const size_t rows = ((size_t)tile_bytes / 16 + (size_t)(SB >> 4) - 1) / (size_t)(SB >> 4);
    return first + (rows - 1) * (size_t)Dp + (size_t)src_slots * 16;
}

// K-blocks (of 32 features) a wave of the interleaved bank kernel keeps in registers (detect_mfma_kernel.inc: kMaxNKB)
#define TSR_FAST_MAX_NKB 12
// Phases and dwords per phase of the bit form of D features: what tsr_fast_bank_build derives from D for two limbs.
static inline void tsr_bit_geometry(int64_t D, int *pw, int *nph)
{
    const int nkb_total = (int)((D + 31) / 32);
    *nph = (nkb_total + TSR_FAST_MAX_NKB - 1) / TSR_FAST_MAX_NKB;
    *pw = (nkb_total + *nph - 1) / *nph;
}

// Byte rows of a corpus that so far holds only bit rows (unpacked on the device, kept); no-op otherwise.
int tsr_corpus_need_bytes(tsr_ctx *ctx, const struct tsr_corpus *cp);
// Bit rows for a corpus (allocation + zeroed slack only; the caller fills them).
int tsr_corpus_alloc_bits(tsr_ctx *ctx, struct tsr_corpus *cp, bool zero_tails = true);

// ---- kernels' host entry points (device pointers, enqueue on ctx->stream) -----
// detect_exact.hip
int tsr_launch_exact_bank_mfma(tsr_ctx *ctx, const tsr_bank *bank, struct tsr_corpus *cp, float *d_raw);
int tsr_launch_exact_bank(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_feats, int64_t P,
                          int Dp, float *d_raw /*[M][P]*/);
int tsr_launch_corpus_pack(tsr_ctx *ctx, const uint8_t *d_src, int64_t P, int D, int Dp,
                           uint8_t *d_dst);
int tsr_launch_bank_epilogue(tsr_ctx *ctx, const tsr_bank *bank, const float *d_raw, int64_t P,
                             const int32_t *d_utt_end, int flags, float *d_scores, int32_t *d_ids,
                             float *d_all /*[M][P] or null*/);
int tsr_launch_detect_f32(tsr_ctx *ctx, const float *d_F, int64_t T, int64_t D, const float *d_LF,
                          int64_t M, int64_t L, const float *d_cs, float *d_out, uint16_t *d_ids);
int tsr_launch_detect_u8_single(tsr_ctx *ctx, const uint8_t *d_F, int64_t T, int64_t D, const float *d_LF, int64_t L,
                                float *d_out);
int tsr_launch_max_detect_examples(tsr_ctx *ctx, const uint8_t *d_F, int64_t N, int64_t T,
                                   int64_t D, const double *d_LF, int64_t L,
                                   const int64_t *d_lengths, double *d_tmp, double *d_out);
int tsr_launch_part_chain_setup(tsr_ctx *ctx, const int64_t *d_ooff, const int32_t *d_beg, const int32_t *d_n0, int64_t N, int g0,
                                int64_t P_out, int32_t *d_src_row, int32_t *d_src_lo, int32_t *d_src_hi, int32_t *d_utt_beg,
                                int32_t *d_utt_end, uint32_t *d_bits, int64_t plane_dwords, int64_t tail_first, int tail_count,
                                int n_planes, int *d_work);
int tsr_launch_out_geometry(tsr_ctx *ctx, const int64_t *d_ooff, const int32_t *d_beg, const int32_t *d_n0, int64_t N,
                            int g0, int64_t P_out, int32_t *d_src_row, int32_t *d_src_lo, int32_t *d_src_hi);
int tsr_launch_code_parts_mag_aux(tsr_ctx *ctx, const uint8_t *d_Xe, const uint8_t *d_Xm, const uint8_t *d_Xa, int64_t X0,
                                  int64_t X1, int64_t E, int64_t MD, int64_t AD, int64_t P0, int64_t P1, const uint8_t *d_mask,
                                  const double *d_tab, int64_t K, double *d_out);
// classify.hip
int tsr_launch_window_argmax(tsr_ctx *ctx, const float *d_all, int64_t stride, int M, const int64_t *d_starts,
                             const int64_t *d_ends, int64_t W, float *d_score, int64_t *d_loc, int32_t *d_tid);
int tsr_launch_float_windows(tsr_ctx *ctx, const tsr_bank *bank, const float *d_X, const int32_t *d_item_row,
                             const int32_t *d_item_left, int64_t NW, float *d_out);
// Isolated-window classification on the matrix cores (detect_exact_mfma.hip): pairs of consecutive positions of a
// window, 16 pairs of one padding pattern per tile.  TSR_ERR_UNSUPPORTED when the bank has no matrix-core form.
struct tsr_iso_pair {   // one row of a tile: the positions item and item + 1 of one window
    int64_t f0;         // stream frame of the first position's first row (may lie in front of the corpus: padding)
    int32_t item;       // index of the first position in the item list (-1: filler row of a tile)
    int32_t left;       // rows the window has left at the first position; bit 30: the second position exists
};
int tsr_launch_isolated_mfma(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_X, int Dp, const float *d_bgd,
                             const tsr_iso_pair *d_pairs, const int32_t *d_tile_kr, int64_t n_tiles, int64_t NW, void *d_wb_scratch,
                             float *d_out, const int *d_n_tiles = nullptr);
// the tile tables made on the device from the windows' geometry (detect_exact_mfma.hip); buckets = (longest template group + 2)^2
#define TSR_ISO_MAX_BUCKETS 4096
int tsr_launch_isolated_tables(tsr_ctx *ctx, const int64_t *d_geo, int64_t W, int64_t NP, int RMAX, int *d_work, tsr_iso_pair *d_tiles,
                               int64_t tiles_bound, const int32_t **d_tile_kr, const int **d_n_tiles);
int tsr_launch_isolated_windows(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_X, int Dp, const float *d_bgd,
                                const int64_t *d_first, const int64_t *d_lo, const int64_t *d_hi,
                                const int32_t *d_left, int64_t NW, float *d_out);
int tsr_launch_peaks(tsr_ctx *ctx, const float *d_v, const int64_t *d_offsets, int64_t N, float threshold, int C0,
                     int64_t *d_counts, int32_t *d_times);
int tsr_launch_window_peak_max(tsr_ctx *ctx, const float *d_v, const int64_t *d_offsets, const int64_t *d_utt,
                               const int64_t *d_starts, const int64_t *d_ends, int64_t W, float *d_out);
int tsr_launch_mstep(tsr_ctx *ctx, const uint8_t *d_X, int64_t P, int D, int Dp, const float *d_aff, int K,
                     float *d_part, int S, float *d_out);
int tsr_launch_mixture_scores(tsr_ctx *ctx, const uint8_t *d_X, int64_t P, int D, int Dp, const float *d_odds, int Dq,
                              const double *d_c, int K, float *d_scores);
int tsr_launch_mixture_estep(tsr_ctx *ctx, const float *d_scores, int64_t P, int K, const double *d_log_w, int hard,
                             float *d_aff, double *d_blk, int *n_blocks);
int tsr_launch_link_partition(tsr_ctx *ctx, const int32_t *d_times, const int32_t *d_lengths, int64_t n, uint8_t *d_part);
int tsr_launch_cluster_times(tsr_ctx *ctx, const int32_t *d_times, const int64_t *d_offsets, const int64_t *d_counts, int64_t N,
                             float C0, float C1, const int32_t *d_lens, uint8_t *d_part, int64_t *d_n_clusters,
                             int32_t *d_starts, int32_t *d_ends);
// edges.hip
int tsr_launch_edge_blocks(tsr_ctx *ctx, const double *d_S, const uint8_t *d_mask, const int64_t *d_utt_off, const int32_t *d_items,
                           int64_t n_items, int H, int block_length, int spread, double q, const double *abst,
                           int Dp, uint8_t *d_out);
// edges2.hip
int tsr_launch_threshold_edgemap(tsr_ctx *ctx, double *d_E, const int64_t *d_breaks, int nb, int64_t T, double q,
                                 const double *d_abst, int binary, double *d_taus);
int tsr_launch_spread_edgemap(tsr_ctx *ctx, const double *d_in, double *d_out, const int64_t *d_breaks, int nb, int64_t rows,
                              int64_t T, const int32_t *d_orient, int spread);
int tsr_launch_magnitude(tsr_ctx *ctx, const double *d_S, int F0, int64_t T, int smooth, int down, int F, double *d_S2,
                         const void *d_wins, int n_wins, int whole, int m, int r, int min_count, uint8_t *d_E, uint8_t *d_out);
// parts_mfma.hip
int tsr_parts_mfma_prepare(tsr_ctx *ctx, tsr_parts *p, const double *lp, const double *lip);
void tsr_parts_mfma_free(tsr_parts *p);
int tsr_launch_code_parts_mfma(tsr_ctx *ctx, const tsr_parts *p, const uint8_t *d_X, int64_t P, int64_t X1,
                               int Dp, const int32_t *d_utt_end, int threshold, int fast, int32_t *d_argmax,
                               uint32_t *d_list, int *d_work, int ids8 = 0);
// detect_mfma.hip
int tsr_fast_bank_build(tsr_ctx *ctx, tsr_bank *bank, int kind);  // kind: index into tsr_bank::fast
int tsr_corpus_bf16_copy(tsr_ctx *ctx, struct tsr_corpus *cp);
void tsr_fast_bank_free(tsr_bank *bank);
int tsr_launch_tap2_bank(tsr_ctx *ctx, const tsr_bank *bank, struct tsr_corpus *cp, float *d_scores, int32_t *d_ids, float *d_all);
int tsr_launch_fast_bank(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *corpus, int flags,
                         float *d_scores, int32_t *d_ids, float *d_all);
// detect_mx.hip
int tsr_launch_mx_bank(tsr_ctx *ctx, const tsr_bank *bank, struct tsr_corpus *cp, float *d_scores, int32_t *d_ids, float *d_all);
void tsr_mx_bank_free(tsr_bank *bank);
void tsr_mx_corpus_free(struct tsr_corpus *cp);
// parts.hip
int tsr_launch_code_parts(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *d_X, int64_t X0,
                          int64_t X1, int threshold, int fast, double *d_out /*or null*/,
                          int32_t *d_argmax /*or null*/);
int tsr_launch_spread(tsr_ctx *ctx, const int32_t *d_A, int64_t X0, int64_t X1, int r0, int r1,
                      int64_t K, int g0, int g1, const uint8_t *d_graph, uint8_t *d_out);
int tsr_launch_spread_graph(tsr_ctx *ctx, uint8_t *d_X, int64_t n_loc, int64_t K,
                            const uint8_t *d_graph);
int tsr_launch_code_parts_stream(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *d_X, int64_t P,
                                 int64_t X1, int Dp, const int32_t *d_utt_end, int threshold, int fast,
                                 int32_t *d_argmax);
bool tsr_spread_packs_bits(int64_t N1, int r0, int64_t K, int Dp_out, int bit_words /* dwords of a frame's bit form; 0: bytes */);
int tsr_launch_spread_stream(tsr_ctx *ctx, const int32_t *d_A, int64_t N1, int r0, int r1, int64_t K,
                             int g0, int g1, int64_t O1, int64_t P_out, const int32_t *d_src_row,
                             const int32_t *d_src_lo, const int32_t *d_src_hi, const uint8_t *d_graph,
                             const uint64_t *d_graph_masks, int Dp_out, uint8_t *d_out, uint8_t *d_out_bits = nullptr, int bit_pw = 0,
                             int bit_nph = 0, int64_t bit_rows = 0, int ids8 = 0);
