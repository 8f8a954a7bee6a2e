/*
 * tsr.h -- C ABI of libtsr_hip.so, the MI355X (gfx950) implementation of the
 * sliding-window template-scoring path of markstoehr/Template-Speech-Recognition.
 *
 * The reference exposes this path as three Cython extension modules
 * (template_speech_rec/setup.py:35-41); there is no C interface to match, so
 * this header declares one entry point per reference function, with plain
 * pointers + sizes (no Python, NumPy or torch types).  The Python modules in
 * template-speech-recognition_amd/template_speech_rec/ bind these with ctypes
 * under the reference's own module and function names.
 *
 * Conventions
 *   - every function returns an int status (TSR_OK == 0); tsr_strerror() names
 *     it, tsr_last_error() gives the detailed message of the calling thread;
 *   - all host arrays are C-contiguous; the reference's trailing (frequency,
 *     part) axes are passed flattened as one feature axis D (the reference
 *     iterates frequency-outer, part-inner == flattened order);
 *   - inputs are borrowed and never modified (except tsr_spread_graph_patches,
 *     which mutates in place like code_parts.pyx:459-503); outputs are written
 *     to caller-allocated host buffers unless the name says `_device`;
 *   - a tsr_ctx owns one device, one HIP stream and its scratch buffers; it is
 *     not thread-safe: use one ctx per host thread / per GPU;
 *   - nothing here falls back to a CPU path: without a GPU every compute call
 *     fails with TSR_ERR_NO_DEVICE.
 */
#ifndef TSR_H
#define TSR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSR_VERSION 100 /* 0.1.0 */

enum {
    TSR_OK = 0,
    TSR_ERR_INVALID = 1,     /* bad shape / null pointer / size out of range      */
    TSR_ERR_HIP = 2,         /* a HIP runtime call failed (see tsr_last_error)    */
    TSR_ERR_NO_DEVICE = 3,   /* no usable gfx950 device                           */
    TSR_ERR_UNSUPPORTED = 4, /* valid request this build cannot serve             */
    TSR_ERR_NONFINITE = 5,   /* filter bank contains inf/nan                      */
    TSR_ERR_RCCL = 6         /* an RCCL call failed or librccl is unavailable     */
};

/* Scoring precision/algorithm selector (`flags` of the bank scorers).
 *   TSR_EXACT: f32 accumulation in the reference's order tau -> f -> k
 *              (compute_likelihood_linear_filter.pyx:162-175); raw scores are
 *              bit-identical to the Cython path.  Runs as v_mfma_f32_16x16x4_f32 chains (an
 *              ascending-k fmaf chain per output) over bit planes of the corpus; banks with
 *              a non-zero weight below 2^-100 take the one-lane-per-window vector kernel.
 *   TSR_FAST : MFMA path.  The bank is held as int8 limbs of a fixed-point encoding
 *              (per-template scale) and accumulated exactly in integers, so results
 *              are independent of summation order and bit-reproducible; the only
 *              error is the weight quantisation.  At least two limbs (16-bit weights):
 *              |error| <= 2^-16 * max|W_m| per SET tap of the window in the worst case,
 *              i.e. <= 2^-16 * (max|W_m| / mean|W|) relative to the sum of the magnitudes
 *              of the added terms (the norm in which a floating-point sum is accurate at
 *              all); the errors are unbiased, so the observed figure grows with the
 *              square root of the tap count -- about 5e-3 absolute on BASELINE config 2
 *              (11 520 taps, scores of order 10^3: 3e-6 relative), the size of the rounding
 *              noise of the reference's own sequential f32 sum.  A window whose score is of
 *              order one next to thousands of cancelling taps is NOT within 1e-4 of its own
 *              value in this mode -- nor is the reference's f32 result within that of the
 *              exact sum; ask for TSR_W24 or TSR_EXACT there.  A bank whose shortest filter has at
 *              most 4096 taps (L * D) always gets the third limb, and so does a bank with a
 *              filter of more than 66 052 taps (16-bit weights times that many set taps would
 *              overflow the two-limb path's int32 window sums): the weights never get fewer
 *              than 16 bits.
 *   TSR_FAST | TSR_W24: three limbs (23-bit weights, power-of-two scale),
 *              <= 2^-22 * max|W_m| per set tap, at 1.5x the matrix-core work. */
enum { TSR_EXACT = 0, TSR_FAST = 1, TSR_W24 = 2, TSR_BF16 = 4, TSR_MX = 8 };
/*   TSR_FAST | TSR_BF16 [| TSR_W24]: the bank as a bf16 MFMA GEMM (BASELINE configs[4]): weights rounded to bf16
 *              (relative error up to 2^-9 per weight: NOT within the 1e-4 bar) or, with TSR_W24, split into
 *              bf16 hi + lo (2^-17 per weight); f32 accumulation inside a 32-frame tile, tile sums rounded to
 *              fixed point and added as integers (order-independent, hence bit-reproducible, but not exact).
 *              Comparison variants only; D <= 384, otherwise the exact kernel serves the call.
 *   TSR_FAST | TSR_MX: 13-bit weights (|q| <= 4095, per-template scale) as three radix-16 digits, each an exact fp6
 *              value, on the block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4: fp4 frames x fp6
 *              weights at four times the bf16 rate, the digits' weights 1 / 16 / 256 applied by its block scales, one
 *              f32 accumulator holding the exact integer sum).  The bank pass is a skew-free Toeplitz product
 *              (csrc/detect_mx.hip) and the fastest path for a bank of up to eight templates (BASELINE config 2;
 *              rows wider than 512 features -- config 3's part corpora -- go through the kernel in feature phases,
 *              banks of more than eight templates in slices of eight, one launch each, merged in the result arrays);
 *              |error| <= 2^-13 * max|W_m| per SET tap in the worst case, unbiased -- observed 2e-5 relative on
 *              config 2 (scores of order 4 * 10^3), i.e. eight times TSR_FAST's figure and a fifth of the 1e-4 bar.
 *              Sums below 2^24 (config 2) are exact integers; beyond (wide rows) the f32 accumulator rounds at 2^-24
 *              relative per addition, in a fixed order: the same bits from launch to launch.  Max mode (scores + ids)
 *              and all-scores passes; filters of at most 4096 taps are served as plain TSR_FAST.  Stands alone
 *              beside TSR_FAST (not with TSR_W24 / TSR_BF16).  A request for this kernel, not a choice of the faster
 *              one: it pays M x (Lmax + 1) tap rows and whole rounds of workgroups (one per CU, 672 window starts
 *              each), so it is the faster for banks of near-equal filter lengths (config 2: 24 vs 33 us per 1000
 *              utterances; 64 templates of 32 taps: 183 vs 194 us; config 4's 64 templates of 24 .. 48 taps, dealt to
 *              the slices by length: 0.36 vs 0.38 ms per 1250 utterances) and the slower for a ragged bank of eight
 *              templates or fewer (one slice pays its longest filter: L_m = 28 .. 36: 39 vs 33 us). */

typedef struct tsr_ctx tsr_ctx;       /* device + stream + scratch                 */
typedef struct tsr_bank tsr_bank;     /* device-resident template bank             */
typedef struct tsr_corpus tsr_corpus; /* device-resident utterances + their scores */
typedef struct tsr_parts tsr_parts;   /* device-resident part models (front-end)   */

/* ---- library / context ---------------------------------------------------- */
int tsr_version(void);
const char *tsr_strerror(int status);
const char *tsr_last_error(void);
int tsr_device_count(int *count);
int tsr_ctx_create(int device, tsr_ctx **out);
int tsr_ctx_destroy(tsr_ctx *ctx);
int tsr_ctx_sync(tsr_ctx *ctx);
/* hipEvent pair on the ctx stream: kernel time of everything enqueued between. */
int tsr_timer_start(tsr_ctx *ctx);
int tsr_timer_stop(tsr_ctx *ctx, float *elapsed_ms);
/* Kernel-choice options of a context.  Which kernel produces a result never depends on the
 * process environment after the context exists: the defaults are read from the environment ONCE,
 * inside tsr_ctx_create (TSR_FASTK = tap2|old, TSR_EXACTK = valu|mfma, TSR_PART_LAYOUT = bytes|bits,
 * TSR_XCD_MAP = 0|1 -- diagnostics), and from then on only tsr_ctx_set_option
 * changes them.  Every choice yields the same results (bit for bit within TSR_EXACT, within
 * TSR_FAST's exact integer sums, and for the part features); the options exist for A/B timing.
 *   TSR_OPT_FAST_KERNEL   two-limb TSR_FAST kernel: AUTO (tap-pair kernel on narrow rows, else the
 *                         interleaved one), TSR_FAST_KERNEL_TAP2 (wherever it fits), TSR_FAST_KERNEL_INTERLEAVED
 *   TSR_OPT_EXACT_KERNEL  TSR_EXACT: AUTO (matrix cores where the bank has that form), TSR_EXACT_KERNEL_VALU
 *   TSR_OPT_PART_LAYOUT   what tsr_corpus_part_features leaves resident: AUTO (bit rows when the
 *                         dilation kernel can write them, K <= 64), TSR_PART_LAYOUT_BYTES, TSR_PART_LAYOUT_BITS
 *   TSR_OPT_XCD_MAP       XCD-aware placement of multi-group bank passes: 1 (default) / 0 */
enum { TSR_OPT_FAST_KERNEL = 1, TSR_OPT_EXACT_KERNEL = 2, TSR_OPT_PART_LAYOUT = 3, TSR_OPT_XCD_MAP = 4,
       /* read-only (tsr_ctx_get_option): 0 = TSR_EXACT has not met the matrix cores on this context yet, 1 = the one-off
        * probe (a small fixed bank scored by the matrix-core kernel and by the vector kernel) matched bit for bit,
        * -1 = it did not and every TSR_EXACT pass of this context stays on the vector kernel (said once on stderr) */
       TSR_INFO_EXACT_MFMA = 5,
       /* read-only: the bank kernel that served the latest scoring pass of this context (TSR_KERNEL_* below; 0 = none yet) */
       TSR_INFO_LAST_BANK_KERNEL = 6 };
enum { TSR_KERNEL_EXACT_VALU = 1, TSR_KERNEL_EXACT_MFMA = 2, TSR_KERNEL_INTERLEAVED = 3, TSR_KERNEL_TAP2 = 4, TSR_KERNEL_MX = 5 };
enum { TSR_OPT_AUTO = 0,
       TSR_FAST_KERNEL_TAP2 = 1, TSR_FAST_KERNEL_INTERLEAVED = 2,
       TSR_EXACT_KERNEL_VALU = 1, TSR_EXACT_KERNEL_MFMA = 2,
       TSR_PART_LAYOUT_BYTES = 1, TSR_PART_LAYOUT_BITS = 2 };
int tsr_ctx_set_option(tsr_ctx *ctx, int option, int value);
int tsr_ctx_get_option(const tsr_ctx *ctx, int option, int *value);
/* The large device arrays of destroyed corpora are kept in a process-wide pool (at most 2 GB) and handed to the next
 * corpus of about the same size -- a front-end call that makes a new corpus would otherwise pay a device allocation
 * and, when the old corpus goes, a device-wide synchronisation per array.  tsr_pool_trim returns the idle blocks to the
 * driver. */
int tsr_pool_trim(void);
/* device name, CU count, HBM bytes (for bench/DESIGN reporting) */
int tsr_device_info(tsr_ctx *ctx, char *name, int name_len, int *cus, int64_t *hbm_bytes);

/* ---- compute_likelihood_linear_filter.pyx, one utterance, one filter -------- */
/* detect(F, LF)            compute_likelihood_linear_filter.pyx:135-176
 * F u8 [T,D], LF f32 [L,D] -> out f32 [T]; out[t] = 0 for t > T-L.  T < L is
 * TSR_ERR_INVALID (the reference wraps a uint16 at :157 and reads out of bounds). */
int tsr_detect_u8(tsr_ctx *ctx, const uint8_t *F, int64_t T, int64_t D, const float *LF, int64_t L,
                  float *out);
/* detect_float / detect_float2d / detect_spectral   :178-219, :281-318, :322-359 */
int tsr_detect_f32(tsr_ctx *ctx, const float *F, int64_t T, int64_t D, const float *LF, int64_t L,
                   float *out);
/* detect_float_max(F, LF, cs)   :221-278.  LF f32 [M,L,D], cs f32 [M] ->
 * out f32 [T], ids u16 [T] (first maximum; zeros for t > T-L). */
int tsr_detect_f32_max(tsr_ctx *ctx, const float *F, int64_t T, int64_t D, const float *LF,
                       int64_t M, int64_t L, const float *cs, float *out, uint16_t *ids);
/* get_max_detect_examples(F, LF, lengths)   :21-76.  F u8 [N,T,D], LF f64 [L,D],
 * lengths i64 [N] -> out f64 [N] (filter truncated at the example end). */
int tsr_max_detect_examples_u8(tsr_ctx *ctx, const uint8_t *F, int64_t N, int64_t T, int64_t D,
                               const double *LF, int64_t L, const int64_t *lengths, double *out);

/* ---- template bank: get_train_data.py:962-992 (_compute_detection_E) --------- */
/* W_cat: template m is L[m]*D floats, templates concatenated in order; c f64 [M]
 * (estimate_template.py:230-251 produces exactly (W_m f32, c_m f64)). */
int tsr_bank_create(tsr_ctx *ctx, const float *W_cat, const int32_t *L, const double *c, int32_t M,
                    int64_t D, tsr_bank **out);
int tsr_bank_destroy(tsr_bank *bank);
int tsr_bank_info(const tsr_bank *bank, int32_t *M, int64_t *D, int32_t *Lmax, int64_t *sumL);

/* Score utterances against the bank: scores[p] = max_m (detect(E_n, W_m)[t] +
 * f32(c_m)), ids[p] = lowest m attaining it, p = frame_offsets[n] + t.
 * feats u8 [sum T_n, D] (utterance n = frames [frame_offsets[n],
 * frame_offsets[n+1])), frame_offsets i64 [N+1].  Positions t > T_n - L_m score
 * 0 + c_m exactly as the reference's zero tail does (quirk: they take part in
 * the max).  T_n < L_m is allowed here: every frame is then tail for m. */
int tsr_score_bank_u8(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *feats,
                      const int64_t *frame_offsets, int64_t N, int flags, float *scores,
                      int32_t *ids);
/* Same, but every template's score: per_template f32 [M, sum T_n]. */
int tsr_score_bank_all_u8(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *feats,
                          const int64_t *frame_offsets, int64_t N, int flags, float *per_template);

/* Device-resident steady state (what bench.py times): upload once, score many
 * times with inputs and outputs staying in HBM, download on demand. */
int tsr_corpus_create(tsr_ctx *ctx, const uint8_t *feats, const int64_t *frame_offsets, int64_t N,
                      int64_t D, tsr_corpus **out);
int tsr_corpus_destroy(tsr_corpus *corpus);
int tsr_corpus_score(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *corpus, int flags); /* async */
/* detect(E_n, W_m) + c_m of EVERY template at every frame of a resident corpus -> per_template
 * f32 [M, sum T_n] on the host (what tsr_score_bank_all_u8 returns, without re-uploading the features). */
int tsr_corpus_score_all(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *corpus, int flags, float *per_template);
int tsr_corpus_download(tsr_ctx *ctx, const tsr_corpus *corpus, float *scores, int32_t *ids);
int tsr_corpus_frames(const tsr_corpus *corpus, int64_t *total_frames);

/* ---- code_parts.pyx / spread_waliji_patches.pyx ------------------------------- */
/* code_parts (fast=0, code_parts.pyx:70-135) / code_parts_fast (fast=1, :247-295).
 * X u8 [X0,X1,E]; log_parts, log_invparts f64 [K,P0,P1,E]; out f64
 * [X0-P0+1, X1-P1+1, K+1].  `threshold` is compared as unsigned like the
 * generated C does, so a negative threshold makes every location background. */
int tsr_code_parts(tsr_ctx *ctx, const uint8_t *X, int64_t X0, int64_t X1, int64_t E,
                   const double *log_parts, const double *log_invparts, int64_t K, int64_t P0,
                   int64_t P1, int threshold, int fast, double *out);
/* code_parts_mag_aux (code_parts.pyx:138-242): code_parts on edges plus magnitude and auxiliary channels,
 * a location being coded iff X_thresh_mask at the patch centre is set.  X_edges u8 [X0,X1,E], X_mags u8
 * [X0,MD], X_aux u8 [X0,AD]; log tables f64: parts [K,P0,P1,E], mag [K,P0+MD], aux [K,P0*AD] (each with its
 * "inv" twin); mask u8 [X0,X1] -> out f64 [X0-P0+1, X1-P1+1, K+1], the terms added in the reference's order
 * (bit-identical map).  Integer divisions are the Python-2 / generated-C unsigned ones. */
int tsr_code_parts_mag_aux(tsr_ctx *ctx, const uint8_t *X_edges, const uint8_t *X_mags, const uint8_t *X_aux,
                           int64_t X0, int64_t X1, int64_t E, int64_t MD, int64_t AD, const double *log_parts,
                           const double *log_invparts, const double *log_mag_parts, const double *log_mag_invparts,
                           const double *log_aux_parts, const double *log_aux_invparts, int64_t K, int64_t P0,
                           int64_t P1, const uint8_t *X_thresh_mask, double *out);
/* spread_patches (code_parts.pyx:406-457) == spread_waliji_patches
 * (spread_waliji_patches.pyx:15-59).  X i64 [X0,X1] -> out u8 [X0,X1,K]. */
int tsr_spread_patches(tsr_ctx *ctx, const int64_t *X, int64_t X0, int64_t X1, int r0, int r1,
                       int64_t K, uint8_t *out);
/* spread_graph_patches (code_parts.pyx:459-503): X u8 [X0,X1,K] updated in
 * place, graph u8 [K,K]; sequential over the source part like the reference. */
int tsr_spread_graph_patches(tsr_ctx *ctx, uint8_t *X, int64_t X0, int64_t X1, int64_t K,
                             const uint8_t *graph);

/* Fused front-end, get_train_data.py:2619-2628 (get_part_features) and :183-191
 * (get_waliji_feature_map): code_parts[_fast] -> argmax (first max) ->
 * spread(r0,r1) -> [graph closure] -> [::g0, ::g1] without the f64 map ever
 * leaving the chip.  out u8 [ceil(N0/g0), ceil(N1/g1), K], N0 = X0-P0+1,
 * N1 = X1-P1+1; argmax_out (optional, may be NULL) i64 [N0,N1]. */
int tsr_parts_create(tsr_ctx *ctx, const double *log_parts, const double *log_invparts, int64_t K,
                     int64_t P0, int64_t P1, int64_t E, tsr_parts **out);
int tsr_parts_destroy(tsr_parts *parts);
int tsr_part_features(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *X, int64_t X0,
                      int64_t X1, int threshold, int fast, int r0, int r1, const uint8_t *graph,
                      int g0, int g1, uint8_t *out, int64_t *argmax_out);

/* The same chain over a whole device-resident corpus of edge maps (D = X1*E per frame): returns a
 * NEW corpus of part maps (D = O1*K per frame, utterance n has ceil((T_n-P0+1)/g0) frames) that
 * tsr_corpus_score consumes directly -- BASELINE config 3, no host round trip. */
int tsr_corpus_part_features(tsr_ctx *ctx, const tsr_parts *parts, const tsr_corpus *edges,
                             int threshold, int fast, int r0, int r1, const uint8_t *graph, int g0,
                             int g1, tsr_corpus **out);
/* The corpus chain has two coders with IDENTICAL results (the part ids of code_parts.pyx:70-135
 * followed by np.argmax): the VALU kernel that forms the reference's f64 sums, and the MFMA coder
 * that finds the index with exact 16-bit integer sums and re-evaluates in f64 only the locations
 * whose two best parts lie within the rounding window.  AUTO takes the MFMA coder whenever the
 * tables fit it (E == 8, K <= 63, P0*ceil(P1/2) <= 16, finite tables); MFMA makes the call fail
 * with TSR_ERR_UNSUPPORTED instead of falling back.
 * tsr_parts_info: whether the MFMA coder is available and, for the latest
 * tsr_corpus_part_features, the number of live-or-not patch locations scanned and how many of
 * them went through the f64 re-evaluation (-1: the VALU kernel ran).  Any pointer may be NULL. */
enum { TSR_PARTS_AUTO = 0, TSR_PARTS_VALU = 1, TSR_PARTS_MFMA = 2 };
int tsr_parts_set_mode(tsr_parts *parts, int mode);
int tsr_parts_info(const tsr_parts *parts, int *mfma_available, int64_t *locations, int64_t *rechecked);
/* Read back a corpus: feature bytes u8 [sum T_n, D]; its shape (frame_offsets i64 [N+1], may be NULL). */
int tsr_corpus_download_feats(tsr_ctx *ctx, const tsr_corpus *corpus, uint8_t *feats);
int tsr_corpus_shape(const tsr_corpus *corpus, int64_t *N, int64_t *D, int64_t *frame_offsets);
/* What a corpus keeps resident: *layout bit 0 = byte rows [sum T, round_up(D,16)] in {0,1}, bit 1 = bit rows
 * [sum T, round_up(ceil(D/8),16)] (bit d of a row = feature d; what tsr_corpus_part_features writes when it can:
 * an eighth of the bytes, read as they are by the two-limb TSR_FAST pass on wide rows; byte rows are derived from
 * them on first use by anything else and then kept); *feature_bytes = device bytes of those forms.  Pointers may be NULL. */
int tsr_corpus_layout(const tsr_corpus *corpus, int *layout, int64_t *feature_bytes);

/* ---- phone-window classification on top of the scores ----------- */
/* _compute_classification_E (get_train_data.py:718-773) over a resident corpus.  Window w is the
 * frame range [starts[w], ends[w]) of the stream (get_classify_windows :710-716 shifted by the
 * utterance's first frame); it must be non-empty and lie inside one utterance.  Per template the
 * FIRST maximum of detect(E, W_m) + c_m over the window (np.argmax), across templates the highest
 * value, the earlier template on ties (strict '<' at :768).  score f32 [W], loc i64 [W] (stream frame
 * of the maximum), template_id i32 [W].  Locations equal the reference's with TSR_EXACT; with
 * TSR_FAST the scores carry the 16-bit weight quantisation and a near-tie may resolve differently. */
int tsr_corpus_classify_windows(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *corpus,
                                const int64_t *starts, const int64_t *ends, int64_t W, int flags,
                                float *score, int64_t *loc, int32_t *template_id);
/* _compute_isolated_classification_E (get_train_data.py:838-925): W background-padded f32 windows
 * (built by get_isolated_classify_windows :781-835), rows concatenated in E_cat f32 [R, D], window w =
 * rows [row_offsets[w], row_offsets[w+1]); every template is run with detect_float's arithmetic
 * (compute_likelihood_linear_filter.pyx:178-219: f32, product rounded before the add) + f32(c_m), the
 * arg-max covers the first n_valid[w] positions of the window.  loc i32 [W] is the position inside
 * the window.  Bit-identical scores, hence identical locations and template ids. */
int tsr_classify_float_windows(tsr_ctx *ctx, const tsr_bank *bank, const float *E_cat,
                               const int64_t *row_offsets, const int32_t *n_valid, int64_t W,
                               float *score, int32_t *loc, int32_t *template_id);

/* The same on a resident BINARY corpus without materialising the windows (which are ~40 f32 rows per
 * phone, 4 KB per row at D=360): window w starts at stream frame first[w] -- possibly before its
 * utterance (front padding) -- has rows[w] rows, of which those whose frame lies in
 * [real_lo[w], real_hi[w]) are the corpus frames (as 0.0f/1.0f) and all others the background vector
 * bgd f32 [D] (the padding of :791-803 and the rows behind the phone's segment end, :826-832).
 * Identical scores to tsr_classify_float_windows on the materialised windows. */
int tsr_corpus_classify_isolated(tsr_ctx *ctx, const tsr_bank *bank, const tsr_corpus *corpus,
                                 const float *bgd, const int64_t *first, const int64_t *real_lo,
                                 const int64_t *real_hi, const int32_t *rows, const int32_t *n_valid,
                                 int64_t W, float *score, int32_t *loc, int32_t *template_id);

/* ---- peak picking on the score rows (roc_functions.py) ----------- */
/* Score source of the two functions below: `corpus` != NULL -> the scores tsr_corpus_score left on
 * the device (vals/offsets/N ignored); else host rows vals f32 [offsets[N]], utterance n =
 * [offsets[n], offsets[n+1]).
 * tsr_peaks = get_max_above_threshold (roc_functions.py:177-187) for every utterance: with
 * l = T_n - C0 + 1 the indices t in [1, l-1) where v[t] >= v[t-1], v[t] >= v[t+1] and
 * v[t] >= threshold (compared in f32), ascending.  counts i64 [N]; utterance n's indices are stored
 * from times[offsets[n]] on (times i32 [offsets[N]]). */
int tsr_peaks(tsr_ctx *ctx, const tsr_corpus *corpus, const float *vals, const int64_t *offsets, int64_t N,
              float threshold, int C0, int64_t *counts, int32_t *times);
/* _get_detect_clusters_single_threshold (roc_functions.py:356-383) for every utterance in one call:
 * get_max_above_threshold (tsr_peaks, with l = T_n - C0 + 1) -> cluster_times (cluster_times.pyx:133-222,
 * uint16 times, float32 C0/C1) -> get_cluster_starts_ends (roc_functions.py:189-210).  n_clusters i64 [N];
 * utterance n's (start, end) pairs are stored from starts/ends[offsets[n]] on (i32 [offsets[N]] each).
 * peak_counts i64 [N] / peak_times i32 [offsets[N]] optionally receive the peak lists (NULL to skip).
 * Utterances must be shorter than 65535 scores (the reference's uint16 peak times); C0 a whole number. */
int tsr_detect_clusters(tsr_ctx *ctx, const tsr_corpus *corpus, const float *vals, const int64_t *offsets, int64_t N,
                        float threshold, float C0, float C1, int64_t *n_clusters, int32_t *starts, int32_t *ends,
                        int64_t *peak_counts, int32_t *peak_times);
/* cluster_times (cluster_times.pyx:133-222) for one utterance: times u16 [n] strictly increasing ->
 * partition u8 [n + 1] (2 where a cluster starts and at the closing sentinel, 0 inside a cluster). */
int tsr_cluster_times(tsr_ctx *ctx, const uint16_t *times, int64_t n, float C0, float C1, uint8_t *partition);
/* cluster_times_template_lengths (cluster_times.pyx:221-310): the same state machine with the thresholds of
 * the detected templates -- a link opens a cluster when it is >= template_lengths[template_components[i-1]],
 * a cluster is split when it is longer than 1.5 x the length of its first detection's template.
 * template_components u16 [n] must index template_lengths u16 [n_lengths] (the reference does not check). */
int tsr_cluster_times_template_lengths(tsr_ctx *ctx, const uint16_t *times, const uint16_t *template_lengths,
                                       int64_t n_lengths, const uint16_t *template_components, int64_t n,
                                       uint8_t *partition);
/* cluster_times_variable_length (cluster_times.pyx:73-131).  The reference's split loop never changes the
 * partition (split_cluster :47-70 re-marks the end of the cluster it is asked to split), so the result is the
 * link partition: partition[i] = 1 where i == 0 or times[i] - times[i-1] >= detection_lengths[i-1], else 0.
 * times, detection_lengths i32 [n] -> partition u8 [n]. */
int tsr_cluster_times_variable_length(tsr_ctx *ctx, const int32_t *times, const int32_t *detection_lengths, int64_t n,
                                      uint8_t *partition);
/* get_max_detection_in_syllable_windows (roc_functions.py:88-139), the per-window part: window w is
 * the slice [starts[w], ends[w]) of row utt[w] of the zero-padded detection array (positions at or
 * beyond the utterance's T_n read 0, as in the reference's np.zeros rows; the caller clips ends to the
 * row width); out[w] = the largest inner value of the slice that is >= both its neighbours, -inf if
 * there is none. */
int tsr_window_peak_max(tsr_ctx *ctx, const tsr_corpus *corpus, const float *vals, const int64_t *offsets,
                        int64_t N, const int64_t *utt, const int64_t *starts, const int64_t *ends, int64_t W,
                        float *out);

/* ---- edge front-end (edge_signal_proc.py) -------------------------- */
/* get_edge_features, default branch (get_train_data.py:2547-2557): _edge_map_no_threshold
 * (edge_signal_proc.py:1697-1836) -> _edge_map_threshold_segments (:1537-1675: per block of
 * block_length frames and per orientation the quantile-th sorted positive value, floored at
 * abst_threshold[e], then the binary map spread +-spread_length steps along the orientation inside
 * the block; the trailing partial block is processed on the last block_length frames and only its own
 * frames are kept) -> reorg_part_for_fast_filtering.  S_cat f64 [sum T_n, H] is the spectrogram stream
 * TIME-major (the reference's S before the `.T` of get_edge_features(S.T, ...)); the result is a new
 * resident corpus with D = 8 H, byte h*8 + e of a frame = orientation e at frequency h, ready for
 * tsr_corpus_part_features / tsr_corpus_score.  abst_threshold f64 [8]; 0 <= quantile < 1. */
int tsr_edge_corpus_create(tsr_ctx *ctx, const double *S_cat, const int64_t *frame_offsets, int64_t N,
                           int64_t H, int block_length, int spread_length, double quantile,
                           const double *abst_threshold, tsr_corpus **out);
/* The magnitude_thresholded_edges branch (get_train_data.py:2528-2546): as above, with the edge strengths
 * zeroed wherever mask_cat u8 [sum T_n, H] (time-major like S_cat) is 0, before the thresholding. */
int tsr_edge_corpus_create_masked(tsr_ctx *ctx, const double *S_cat, const uint8_t *mask_cat, const int64_t *frame_offsets,
                                  int64_t N, int64_t H, int block_length, int spread_length, double quantile,
                                  const double *abst_threshold, tsr_corpus **out);
/* threshold_edgemap (edge_signal_proc.py:1184-1252, critical_bands=None) over threshold_edge_block
 * (:1257-1307): every row block [row_breaks[b], row_breaks[b+1]) of E f64 [rows, T], in place: the
 * int(quantile * n)-th of its n sorted positive values is the quantile; entries that are positive and
 * >= max(quantile value, abst_threshold[b]) become 1 (or keep their value when convert_to_binary == 0),
 * everything else 0; a block without positive values becomes 0.  thresholds f64 [n_blocks] (may be NULL)
 * receives the quantile values (NaN for empty blocks) -- what report_level=True returns.
 * spread_edgemap (:1397-1404) over _spread_edges (:1406-1417): every row block, in place, becomes the maximum
 * (floored at 0) over the shifts k * orientations[b] = (rows, columns), k in [-spread_length, spread_length],
 * zeros outside the block.  orientations == NULL is spread_type='all' of _edge_map_threshold_segments (:1623-1636,
 * :1664-1675): SciPy's maximum_filter over a (2 spread_length + 1)^2 box with zeros outside the row block. */
int tsr_threshold_edgemap(tsr_ctx *ctx, double *E, const int64_t *row_breaks, int32_t n_blocks, int64_t T,
                          double quantile, const double *abst_threshold, int32_t convert_to_binary, double *thresholds);
int tsr_spread_edgemap(tsr_ctx *ctx, double *E, const int64_t *row_breaks, int32_t n_blocks, int64_t T,
                       const int32_t *orientations, int32_t spread_length);
/* magnitude_features (edge_signal_proc.py:600-670; whole_block == 0) and magnitude_features_whole_block
 * (:673-747; whole_block != 0): S f64 [F0, T] (frequency x time) -> out u8 [T, F], F = ceil(F0 / down) (F0
 * when down == 0).  Optional maximum over mag_smooth_freq bins (SciPy's maximum_filter, reflected ends), then
 * every mag_downsample_freq-th bin; overlapping estimation windows of block_length frames; a cell is set when
 * its value is >= the window's quantile -- per bin, or one for all bins (whole_block); then an OR over
 * +-mag_spread_length frames, or (whole_block) the square spread of radius spread_radius with SciPy's
 * correlate-and-truncate arithmetic. */
int tsr_magnitude_features(tsr_ctx *ctx, const double *S, int64_t F0, int64_t T, int32_t block_length,
                           int32_t spread_radius, double quantile, int32_t mag_smooth_freq, int32_t mag_downsample_freq,
                           int32_t mag_spread_length, int32_t whole_block, uint8_t *out);

/* M-step sums of the Bernoulli mixture (bernoulli_mixture.py:179, np.dot(affinities.T, data_mat)) on a
 * resident binary corpus: out f32 [K, D], out[k][d] = sum_p weights[p][k] * x[p][d], weights f32 [P, K]
 * (one row per frame).  f32 accumulation in a fixed order (frames ascending inside <= 64 slices, slices
 * ascending): reproducible, but -- like the reference's BLAS product -- not a defined reference order. */
int tsr_corpus_weighted_sums(tsr_ctx *ctx, const tsr_corpus *corpus, const float *weights, int64_t K, float *out);

/* ---- Bernoulli-mixture EM on a resident data matrix (next: SURVEY 8(f) N4) ------
 * bernoulli_mixture.py:263-291 (_compute_loglikelihoods, get_template_loglikelihoods) and :175-182 (M_step).
 * `data` is a corpus of one-frame utterances (the 0/1 data matrix [N, D]); component k is given by its log-odds row
 * log_odds[k] = log T_k - log(1 - T_k) (f32 [K, D]) and consts[k] = sum_d log(1 - T_k[d]) (f64), so that
 *     log P(x_n | k) = consts[k] + sum_d x[n,d] log_odds[k,d]
 * (f32 products and partial sums, f64 across the lanes and for the constant, one rounding to f32: like the reference's
 * float32 BLAS product not a defined summation order, agreement to f32 rounding).
 * tsr_mixture_scores: those scores, f32 [K, N] (get_template_loglikelihoods transposed).
 * tsr_mixture_e_step: scores -> per row the max-shifted log-sum-exp over score + log_weights[k] -> *loglikelihood
 *   (sum over rows, f64, reproducible order) and the posterior component probabilities f32 [N, K] (one-hot on the
 *   first best score when hard_assignment, :278-281), which STAY ON THE DEVICE for the M-step; `affinities`
 *   (host, [N*K]) may be NULL.
 * tsr_mixture_m_step: weights[k] = mean_n a[n,k] (f64 [K]) and sums[k,d] = sum_n a[n,k] x[n,d]
 *   (f32 [K*D], fixed summation order) from the posteriors of the latest e-step. */
int tsr_mixture_scores(tsr_ctx *ctx, const tsr_corpus *data, const float *log_odds, const double *consts, int32_t K,
                       float *scores);
int tsr_mixture_e_step(tsr_ctx *ctx, tsr_corpus *data, const float *log_odds, const double *consts, int32_t K,
                       const double *log_weights, int32_t hard_assignment, double *loglikelihood, float *affinities);
int tsr_mixture_m_step(tsr_ctx *ctx, tsr_corpus *data, double *weights, float *sums);
/* the posteriors of the latest e-step, f32 [N, K] */
int tsr_corpus_download_affinities(tsr_ctx *ctx, const tsr_corpus *data, float *affinities);

/* ---- multi-GPU: utterances shard across ranks, the bank is broadcast once ----- */
/* One process per GPU.  Rank 0 calls tsr_dist_unique_id and hands the 128 bytes
 * to the other ranks over any host channel; every rank then calls
 * tsr_dist_init.  tsr_bank_broadcast replaces the contents of `bank` on every
 * rank with rank `root`'s (an ncclBroadcast over xGMI of the device-resident
 * bank buffers); all ranks must have created a bank of identical geometry
 * (M, D, template lengths): a small header is broadcast and the ranks' verdicts
 * are reduced first, so a mismatch returns TSR_ERR_INVALID on EVERY rank instead
 * of hanging or corrupting the collective.
 * No other collective exists on this path: scoring needs no cross-GPU exchange. */
int tsr_dist_unique_id(char id[128]);
int tsr_dist_init(tsr_ctx *ctx, const char id[128], int nranks, int rank);
int tsr_dist_finalize(tsr_ctx *ctx);
int tsr_bank_broadcast(tsr_ctx *ctx, tsr_bank *bank, int root);

#ifdef __cplusplus
}
#endif
#endif /* TSR_H */
