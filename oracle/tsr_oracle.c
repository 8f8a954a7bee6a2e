/*
 * tsr_oracle.c -- CPU restatement of the reference's sliding-window template
 * scoring path.  TEST INFRASTRUCTURE ONLY: this file is the parity checker
 * for the HIP library (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 * Nothing in the product path may link, import or call it.
 *
 * Every function restates, loop for loop and in the same summation order, one
 * function of /root/reference/template_speech_rec (file:line cited per
 * function).  Arithmetic types follow the reference exactly: f32 sequential
 * accumulation for detect*, f64 for code_parts* and get_max_detect_examples.
 * Build with -O2 -ffp-contract=off and WITHOUT -ffast-math so that the
 * summation order (and therefore every bit of the result) is preserved.
 *
 * Pinning: checked bit-for-bit against the compiled reference (oracle/_ref,
 * built by oracle/build_ref.py from the reference's own .pyx) and against the
 * golden vectors in tests/golden/ (tests/test_oracle.py).
 *
 * Layout contract: all arrays C-contiguous.  The reference's (frequency,
 * part) trailing axes of F/LF are flattened into one axis `D` here; the
 * reference iterates frequency-outer / part-inner, which is exactly the
 * flattened order, so results are bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define TSR_EXPORT __attribute__((visibility("default")))

/* compute_likelihood_linear_filter.pyx:135-176 (detect).
 * out[t] = sum_{tau<L} sum_{d<D} [F[t+tau,d] != 0] * LF[tau,d], t <= T-L,
 * accumulated directly into the f32 output in order tau -> d; out[t] = 0
 * for t > T-L.  Caller guarantees T >= L (the reference segfaults
 * otherwise: uint16 wrap at :157). */
TSR_EXPORT void tsr_oracle_detect_u8(const uint8_t *F, int64_t T, int64_t D,
                                     const float *LF, int64_t L, float *out)
{
    memset(out, 0, (size_t)T * sizeof(float));
    for (int64_t t = 0; t + L <= T; ++t) {
        for (int64_t tau = 0; tau < L; ++tau) {
            const uint8_t *f = F + (t + tau) * D;
            const float *w = LF + tau * D;
            for (int64_t d = 0; d < D; ++d)
                if (f[d]) out[t] += w[d]; /* RMW on the output, as :170-174 */
        }
    }
}

/* compute_likelihood_linear_filter.pyx:178-219 (detect_float), :281-318
 * (detect_float2d) and :322-359 (detect_spectral): identical loops on f32
 * features, multiply then add (two roundings: build with -ffp-contract=off). */
TSR_EXPORT void tsr_oracle_detect_f32(const float *F, int64_t T, int64_t D,
                                      const float *LF, int64_t L, float *out)
{
    memset(out, 0, (size_t)T * sizeof(float));
    for (int64_t t = 0; t + L <= T; ++t) {
        for (int64_t tau = 0; tau < L; ++tau) {
            const float *f = F + (t + tau) * D;
            const float *w = LF + tau * D;
            for (int64_t d = 0; d < D; ++d)
                out[t] += w[d] * f[d]; /* product rounded to f32, then added (:213-218) */
        }
    }
}

/* compute_likelihood_linear_filter.pyx:221-278 (detect_float_max).
 * Fused bank, all templates the same length L: accumulator starts at cs[m],
 * first maximum wins (strict '>' at :275). ids are uint16 in the reference. */
TSR_EXPORT void tsr_oracle_detect_float_max(const float *F, int64_t T, int64_t D,
                                            const float *LF, int64_t M, int64_t L,
                                            const float *cs, float *out, uint16_t *ids)
{
    memset(out, 0, (size_t)T * sizeof(float));
    memset(ids, 0, (size_t)T * sizeof(uint16_t));
    for (int64_t t = 0; t + L <= T; ++t) {
        for (int64_t m = 0; m < M; ++m) {
            float acc = cs[m];
            const float *lf = LF + m * L * D;
            for (int64_t tau = 0; tau < L; ++tau) {
                const float *f = F + (t + tau) * D;
                const float *w = lf + tau * D;
                for (int64_t d = 0; d < D; ++d) acc += w[d] * f[d];
            }
            if (m == 0 || acc > out[t]) {
                out[t] = acc;
                ids[t] = (uint16_t)m;
            }
        }
    }
}

/* compute_likelihood_linear_filter.pyx:21-76 (get_max_detect_examples).
 * Per example: max over t < lengths[i] of the f64 sum with the filter
 * TRUNCATED at the end of the example (min(T-t, L) filter frames, :58-59). */
TSR_EXPORT void tsr_oracle_get_max_detect_examples(const uint8_t *F, int64_t N, int64_t T,
                                                   int64_t D, const double *LF, int64_t L,
                                                   const int64_t *lengths, double *out)
{
    for (int64_t n = 0; n < N; ++n) {
        const uint8_t *Fn = F + n * T * D;
        double best = 0.0;
        for (int64_t t = 0; t < lengths[n]; ++t) {
            double acc = 0.0;
            int64_t mt = (T - t < L) ? (T - t) : L;
            for (int64_t tau = 0; tau < mt; ++tau) {
                const uint8_t *f = Fn + (t + tau) * D;
                const double *w = LF + tau * D;
                for (int64_t d = 0; d < D; ++d)
                    if (f[d]) acc += w[d];
            }
            if (t == 0 || acc > best) best = acc;
        }
        out[n] = best;
    }
}

/* get_train_data.py:962-992 (_compute_detection_E bank epilogue), restated
 * with the reference's NumPy-1.x dtype rules: detect() returns f32, `+ c`
 * with an np.float64 scalar stays f32 (the scalar is cast to f32 and the add
 * is done in f32), strict '>' keeps the lower template id on ties, and the
 * tail t > T-L_m holds 0 + c_m and takes part in the max.
 * W_cat: templates concatenated, template m is L[m]*D floats at woff[m]. */
TSR_EXPORT void tsr_oracle_score_bank_u8(const uint8_t *F, int64_t T, int64_t D,
                                         const float *W_cat, const int64_t *woff,
                                         const int32_t *L, const double *c, int64_t M,
                                         float *scores, int32_t *ids, float *scratch)
{
    for (int64_t m = 0; m < M; ++m) {
        tsr_oracle_detect_u8(F, T, D, W_cat + woff[m], L[m], scratch);
        const float cm = (float)c[m];
        for (int64_t t = 0; t < T; ++t) {
            float v = scratch[t] + cm;
            if (m == 0) {
                scores[t] = v;
                ids[t] = 0;
            } else {
                if (v > scores[t]) ids[t] = (int32_t)m;
                scores[t] = (v > scores[t]) ? v : scores[t]; /* np.maximum */
            }
        }
    }
}

/* All-template variant of the same epilogue: per_template[m*T + t] =
 * detect(F, W_m)[t] + (float)c[m]  (get_train_data.py:962-977 before the max). */
TSR_EXPORT void tsr_oracle_score_bank_all_u8(const uint8_t *F, int64_t T, int64_t D,
                                             const float *W_cat, const int64_t *woff,
                                             const int32_t *L, const double *c, int64_t M,
                                             float *per_template)
{
    for (int64_t m = 0; m < M; ++m) {
        float *o = per_template + m * T;
        tsr_oracle_detect_u8(F, T, D, W_cat + woff[m], L[m], o);
        const float cm = (float)c[m];
        for (int64_t t = 0; t < T; ++t) o[t] = o[t] + cm;
    }
}

/* code_parts.pyx:17-31 (_count_edges) over the 3-D patch [i0,i0+P0) x
 * [j0,j0+P1) x [0,E). */
static unsigned int count_edges(const uint8_t *X, int64_t X1, int64_t E, int64_t i0,
                                int64_t P0, int64_t j0, int64_t P1)
{
    unsigned int count = 0;
    for (int64_t i = i0; i < i0 + P0; ++i)
        for (int64_t j = j0; j < j0 + P1; ++j)
            for (int64_t z = 0; z < E; ++z)
                if (X[(i * X1 + j) * E + z]) count += 1;
    return count;
}

/* Shared body of code_parts.pyx:70-135 (code_parts, fast=0) and :247-295
 * (code_parts_fast, fast=1).  X u8 [X0,X1,E]; log_parts/log_invparts f64
 * [K,P0,P1,E]; out f64 [X0-P0+1, X1-P1+1, K+1], zero-initialised here.
 * Sum order per location: i -> j -> z outer, k inner (each out[k+1] is a
 * sequential f64 sum over (i,j,z)).  `count >= threshold` compares unsigned
 * count with the int threshold converted to unsigned, as the generated C
 * does (code_parts.pyx:24,120-121): a negative threshold makes every
 * location background. */
static void code_parts_impl(const uint8_t *X, int64_t X0, int64_t X1, int64_t E,
                            const double *lp, const double *lip, int64_t K, int64_t P0,
                            int64_t P1, int threshold, int fast, double *out)
{
    const int64_t N0 = X0 - P0 + 1, N1 = X1 - P1 + 1;
    const int64_t pstride = P0 * P1 * E;
    memset(out, 0, (size_t)(N0 * N1 * (K + 1)) * sizeof(double));
    for (int64_t i0 = 0; i0 < N0; ++i0) {
        for (int64_t j0 = 0; j0 < N1; ++j0) {
            double *o = out + (i0 * N1 + j0) * (K + 1);
            unsigned int count = count_edges(X, X1, E, i0, P0, j0, P1);
            if (count >= (unsigned int)threshold) {
                if (!fast) o[0] = -INFINITY;
                for (int64_t i = 0; i < P0; ++i)
                    for (int64_t j = 0; j < P1; ++j)
                        for (int64_t z = 0; z < E; ++z) {
                            const int64_t q = (i * P1 + j) * E + z;
                            const double *tab = X[((i0 + i) * X1 + (j0 + j)) * E + z] ? lp : lip;
                            for (int64_t k = 0; k < K; ++k) o[k + 1] += tab[k * pstride + q];
                        }
                if (fast)
                    for (int64_t k = 0; k < K; ++k) o[0] += o[k + 1] - 1;
            } else {
                o[0] = fast ? 1.0 : 0.0;
            }
        }
    }
}

TSR_EXPORT void tsr_oracle_code_parts(const uint8_t *X, int64_t X0, int64_t X1, int64_t E,
                                      const double *lp, const double *lip, int64_t K,
                                      int64_t P0, int64_t P1, int threshold, double *out)
{
    code_parts_impl(X, X0, X1, E, lp, lip, K, P0, P1, threshold, 0, out);
}

TSR_EXPORT void tsr_oracle_code_parts_fast(const uint8_t *X, int64_t X0, int64_t X1, int64_t E,
                                           const double *lp, const double *lip, int64_t K,
                                           int64_t P0, int64_t P1, int threshold, double *out)
{
    code_parts_impl(X, X0, X1, E, lp, lip, K, P0, P1, threshold, 1, out);
}

/* np.argmax(out, -1) as used at get_train_data.py:2621 and :184: index of
 * the FIRST maximum along the last axis (NaN never occurs: sums of finite
 * logs and a single -inf/0/1 in slot 0). */
TSR_EXPORT void tsr_oracle_argmax_last(const double *A, int64_t rows, int64_t C, int64_t *out)
{
    for (int64_t r = 0; r < rows; ++r) {
        const double *a = A + r * C;
        int64_t best = 0;
        for (int64_t k = 1; k < C; ++k)
            if (a[k] > a[best]) best = k;
        out[r] = best;
    }
}

/* code_parts.pyx:406-457 (spread_patches) == spread_waliji_patches.pyx:15-59.
 * out[i,j,p-1] = 1 iff some (x0,x1) in the clipped box around (i,j) has
 * X[x0,x1] = p > 0.  X int64 [X0,X1]; out u8 [X0,X1,K] zero-initialised. */
TSR_EXPORT void tsr_oracle_spread_patches(const int64_t *X, int64_t X0, int64_t X1, int r0,
                                          int r1, int64_t K, uint8_t *out)
{
    memset(out, 0, (size_t)(X0 * X1 * K));
    for (int64_t i = 0; i < X0; ++i) {
        int64_t lo0 = i - r0 > 0 ? i - r0 : 0;
        int64_t hi0 = i + r0 + 1 < X0 ? i + r0 + 1 : X0;
        for (int64_t j = 0; j < X1; ++j) {
            int64_t lo1 = j - r1 > 0 ? j - r1 : 0;
            int64_t hi1 = j + r1 + 1 < X1 ? j + r1 + 1 : X1;
            for (int64_t x0 = lo0; x0 < hi0; ++x0)
                for (int64_t x1 = lo1; x1 < hi1; ++x1) {
                    int64_t p = X[x0 * X1 + x1];
                    if (p > 0) out[(i * X1 + j) * K + (p - 1)] = 1;
                }
        }
    }
}

/* code_parts.pyx:459-503 (spread_graph_patches): in place, sequential over
 * part_id1 so that parts switched on earlier in the scan propagate further
 * within the same pass. */
TSR_EXPORT void tsr_oracle_spread_graph_patches(uint8_t *X, int64_t X0, int64_t X1, int64_t K,
                                                const uint8_t *graph)
{
    for (int64_t i = 0; i < X0; ++i)
        for (int64_t j = 0; j < X1; ++j) {
            uint8_t *x = X + (i * X1 + j) * K;
            for (int64_t a = 0; a < K; ++a)
                if (x[a] > 0)
                    for (int64_t b = 0; b < K; ++b)
                        if (graph[a * K + b]) x[b] = 1;
        }
}

/* Single-thread corpus loop used by bench.py's cpu_baseline leg: the
 * reference's serial `for` over utterances (get_train_data.py:1269) around the
 * bank epilogue above.  feats: utterances concatenated, utterance n is
 * frames [foff[n], foff[n+1]) of D bytes each; scores/ids use the same frame
 * offsets. */
TSR_EXPORT void tsr_oracle_score_corpus_u8(const uint8_t *feats, const int64_t *foff, int64_t N,
                                           int64_t D, const float *W_cat, const int64_t *woff,
                                           const int32_t *L, const double *c, int64_t M,
                                           float *scores, int32_t *ids, float *scratch)
{
    for (int64_t n = 0; n < N; ++n) {
        const int64_t T = foff[n + 1] - foff[n];
        tsr_oracle_score_bank_u8(feats + foff[n] * D, T, D, W_cat, woff, L, c, M,
                                 scores + foff[n], ids + foff[n], scratch);
    }
}

/* cluster_times.pyx:133-222 (cluster_times): partition of the ascending peak times of one utterance into
 * clusters.  partition has n + 1 codes: 2 = a finished cluster starts here (and the closing sentinel),
 * 0 = inside a cluster, 1 only transiently.  times are uint16, the running distances uint16 as in the
 * reference (they wrap there too); C0 and C1 are float32 compared with the promoted integer differences. */
TSR_EXPORT void tsr_oracle_cluster_times(const uint16_t *times, int64_t n64, float C0, float C1, uint8_t *part)
{
    const uint16_t n = (uint16_t)n64;
    memset(part, 0, (size_t)n + 1);
    part[0] = 1;
    part[n] = 2;
    for (uint16_t i = 1; i < n; ++i)
        if ((float)((int)times[i] - (int)times[i - 1]) >= C0) part[i] = 1;
    uint16_t start = 0, max_link = 0;
    while (start < n) {
        if (part[start + 1] > 0) {
            part[start] = 2;
            ++start;
            while (part[start] != 1 && start < n) ++start;
        } else {
            uint16_t max_dist = 0, total = 0, k = 1;
            while (part[start + k] == 0) {
                const uint16_t cur = (uint16_t)(times[start + k] - times[start + k - 1]);
                if (cur > max_dist) {
                    max_link = (uint16_t)(start + k);
                    max_dist = cur;
                }
                total = (uint16_t)(total + cur);
                ++k;
            }
            if ((float)total > C1) {
                if ((float)((int)times[start + k - 1] - (int)times[max_link]) <= C1)
                    part[max_link] = 2;
                else
                    part[max_link] = 1;
                if ((float)((int)times[max_link - 1] - (int)times[start]) <= C1) {
                    part[start] = 2;
                    if (part[max_link] == 1) start = max_link;
                } else {
                    part[start] = 1;
                }
            } else {
                part[start] = 2;
                start = (uint16_t)(start + k);
                while (part[start] != 1 && start < n) ++start;
            }
        }
    }
}

/* cluster_times.pyx:221-310 (cluster_times_template_lengths): cluster_times with the thresholds taken from
 * the detected template: a link opens a new cluster when it is at least as long as the template of the
 * detection before it (C0), a cluster is split when it is longer than 1.5 x the template of its first detection
 * (C1; `uint16 * 1.5` is a double in the generated C, every operand is exact in it).  times u16 [n] strictly
 * increasing, comps u16 [n] < n_lengths, lengths u16 [n_lengths] -> part u8 [n + 1]. */
TSR_EXPORT void tsr_oracle_cluster_times_template_lengths(const uint16_t *times, const uint16_t *lengths,
                                                          const uint16_t *comps, int64_t n64, uint8_t *part)
{
    const uint16_t n = (uint16_t)n64;
    memset(part, 0, (size_t)n + 1);
    part[0] = 1;
    part[n] = 2;
    for (uint16_t i = 1; i < n; ++i)
        if ((int)times[i] - (int)times[i - 1] >= (int)lengths[comps[i - 1]]) part[i] = 1;
    uint16_t start = 0, max_link = 0;
    while (start < n) {
        if (part[start + 1] > 0) {
            part[start] = 2;
            ++start;
            while (part[start] != 1 && start < n) ++start;
        } else {
            uint16_t max_dist = 0, total = 0, k = 1;
            while (part[start + k] == 0) {
                const uint16_t cur = (uint16_t)(times[start + k] - times[start + k - 1]);
                if (cur > max_dist) {
                    max_link = (uint16_t)(start + k);
                    max_dist = cur;
                }
                total = (uint16_t)(total + cur);
                ++k;
            }
            if ((double)total > lengths[comps[start]] * 1.5) {
                if ((double)((int)times[start + k - 1] - (int)times[max_link]) <= lengths[comps[max_link]] * 1.5)
                    part[max_link] = 2;
                else
                    part[max_link] = 1;
                if ((double)((int)times[max_link - 1] - (int)times[start]) <= lengths[comps[start]] * 1.5) {
                    part[start] = 2;
                    if (part[max_link] == 1) start = max_link;
                } else {
                    part[start] = 1;
                }
            } else {
                part[start] = 2;
                start = (uint16_t)(start + k);
                while (part[start] != 1 && start < n) ++start;
            }
        }
    }
}

/* cluster_times.pyx:73-131 (cluster_times_variable_length).  The function marks a cluster start wherever the
 * gap to the previous detection is at least that detection's length, and then runs a split loop that never
 * changes anything: longest_cluster_detect_id (:18-45) only ever selects clusters that END at a marker, and
 * split_cluster (:47-70) walks to exactly that marker and sets it again (its max_jump_loc is computed and never
 * used).  What it returns is therefore the link partition below (verified against the compiled module).
 * times, lengths i32 [n] -> part u8 [n]. */
TSR_EXPORT void tsr_oracle_cluster_times_variable_length(const int32_t *times, const int32_t *lengths, int64_t n,
                                                         uint8_t *part)
{
    for (int64_t i = 0; i < n; ++i) part[i] = (i == 0 || times[i] - times[i - 1] >= lengths[i - 1]) ? 1 : 0;
}

/* code_parts.pyx:138-242 (code_parts_mag_aux) with the unsigned-int arithmetic of the generated C under
 * Python-2 semantics (`/` on unsigned ints is integer division).  X_edges u8 [X0,X1,E], X_mags u8 [X0,MD],
 * X_aux u8 [X0,AD], lp/lip f64 [K,P0,P1,E], lmp/lmip f64 [K,P0+MD], lap/laip f64 [K,P0*AD], mask u8 [X0,X1]
 * -> out f64 [N0,N1,K+1].  Per coded location the terms are added per part in the order: for each patch
 * row i: its AD auxiliary terms; for each patch column j its E edge terms, after the column j == P1/2 the
 * MD "which magnitude channel" terms; then the row's "any magnitude in the band" term. */
TSR_EXPORT void tsr_oracle_code_parts_mag_aux(const uint8_t *Xe, const uint8_t *Xm, const uint8_t *Xa, int64_t X0,
                                              int64_t X1, int64_t E, int64_t MD, int64_t AD, const double *lp,
                                              const double *lip, const double *lmp, const double *lmip,
                                              const double *lap, const double *laip, int64_t K, int64_t P0,
                                              int64_t P1, const uint8_t *mask, double *out)
{
    const int64_t N0 = X0 - P0 + 1, N1 = X1 - P1 + 1;
    memset(out, 0, (size_t)(N0 * N1 * (K + 1)) * sizeof(double));
    for (int64_t i0 = 0; i0 < N0; ++i0)
        for (int64_t j0 = 0; j0 < N1; ++j0) {
            double *o = out + (i0 * N1 + j0) * (K + 1);
            if (!mask[(i0 + P0 / 2) * X1 + j0 + P1 / 2]) {
                o[0] = 0.0;
                continue;
            }
            o[0] = -INFINITY;
            for (int64_t i = 0; i < P0; ++i) {
                unsigned tmp_freq_mag = 0;
                for (int64_t j = 0; j < AD; ++j) {
                    const double *t = Xa[(i0 + i) * AD + j] ? lap : laip;
                    for (int64_t k = 0; k < K; ++k) o[k + 1] += t[k * (P0 * AD) + i * AD + j];
                }
                for (int64_t j = 0; j < P1; ++j) {
                    const unsigned c = (unsigned)(((unsigned)(j0 + j) * (unsigned)MD) / (unsigned)X1);
                    if (Xm[(i0 + i) * MD + c]) tmp_freq_mag = 1;
                    for (int64_t z = 0; z < E; ++z) {
                        const double *t = Xe[((i0 + i) * X1 + j0 + j) * E + z] ? lp : lip;
                        for (int64_t k = 0; k < K; ++k) o[k + 1] += t[((k * P0 + i) * P1 + j) * E + z];
                    }
                    if (j == P1 / 2)
                        for (unsigned mi = 0; mi < (unsigned)MD; ++mi) {
                            const unsigned d = c - mi;  /* wraps like the reference's unsigned ints */
                            const double *t = (d * d > 1u) ? lmip : lmp;
                            for (int64_t k = 0; k < K; ++k) o[k + 1] += t[k * (P0 + MD) + mi + P0];
                        }
                }
                {
                    const double *t = tmp_freq_mag ? lmp : lmip;
                    for (int64_t k = 0; k < K; ++k) o[k + 1] += t[k * (P0 + MD) + i];
                }
            }
        }
}
