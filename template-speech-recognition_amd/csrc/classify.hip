// classify.hip -- phone-window classification on top of the template scores:
//   _compute_classification_E            get_train_data.py:718-773
//   _compute_isolated_classification_E   get_train_data.py:838-925
//
// Both reduce, for every phone window, the per-template score rows detect*(E, W_m) + c_m to one
// triple (score, location, template):  per template the FIRST maximum inside the window
// (np.argmax), across templates the highest value with the EARLIER template on ties (strict '<',
// :768 / :907).  `window_argmax_kernel` does that reduction for both variants; the isolated
// variant's scores come from `float_windows_kernel`, detect_float's arithmetic (f32, product
// rounded before the add, taps in the order tau -> d) on the listed positions of the
// background-padded f32 windows only.
#include "tsr_internal.h"

namespace {

constexpr int kThreads = 256;

// One wave per window.  all: f32 [M][stride] score rows; window w covers columns
// [starts[w], ends[w]).  Lane-level: strict '>' while walking t upwards keeps the first maximum;
// wave-level: higher value, then lower t.  Across templates: strict '<' keeps the earlier one.
__global__ __launch_bounds__(kThreads) void window_argmax_kernel(
    const float *__restrict__ all, int64_t stride, int M, const int64_t *__restrict__ starts,
    const int64_t *__restrict__ ends, int64_t W, float *__restrict__ score, int64_t *__restrict__ loc,
    int32_t *__restrict__ tid)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (w >= W) return;
    const int64_t s = starts[w], e = ends[w];
    float best = 0.0f;
    int64_t best_t = s;
    int best_m = 0;
    for (int m = 0; m < M; ++m) {
        const float *row = all + (int64_t)m * stride;
        bool have = false;
        float v = 0.0f;
        int64_t vt = 0;
        for (int64_t t = s + lane; t < e; t += 64) {
            const float x = row[t];
            if (!have || x > v) {
                v = x;
                vt = t;
                have = true;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int64_t ot = __shfl_xor(vt, off);
            const bool oh = __shfl_xor((int)have, off) != 0;
            const bool take = oh && (!have || ov > v || (ov == v && ot < vt));
            v = take ? ov : v;
            vt = take ? ot : vt;
            have = have || oh;
        }
        if (m == 0 || best < v) {
            best = v;
            best_t = vt;
            best_m = m;
        }
    }
    if (lane == 0) {
        score[w] = best;
        loc[w] = best_t;
        tid[w] = best_m;
    }
}

// detect_float on listed positions.  One lane per position i, TSR_MG templates per lane (group
// blockIdx.y): out[m][i] = (L_m <= rows_left[i] ? sum_{tau<L_m} sum_d X[row[i]+tau][d]*W_m[tau][d] : 0)
// + f32(c_m), the sum sequential in f32 with the product rounded first
// (compute_likelihood_linear_filter.pyx:178-219: no fused multiply-add).  Weights arrive as
// wave-uniform scalar loads from the exact path's layout Wx[g][tau][d][m%8] (zero beyond L_m, so
// the shorter templates of a group simply add +0).
template <bool VEC4>
__global__ __launch_bounds__(kThreads) void float_windows_kernel(
    const float *__restrict__ X, int D, int WR32, const int32_t *__restrict__ item_row,
    const int32_t *__restrict__ item_left, int64_t NW, const float *__restrict__ Wx,
    const int64_t *__restrict__ gx_off, const int32_t *__restrict__ gx_L, const int32_t *__restrict__ Lm,
    const float *__restrict__ cf, int M, float *__restrict__ out)
{
#pragma clang fp contract(off)  // the reference rounds the product to f32 before the add
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int g = blockIdx.y;
    const bool on = i < NW;
    const int row = on ? item_row[i] : 0;
    const int left = on ? item_left[i] : 0;
    const float *__restrict__ wg = Wx + gx_off[g];
    const int Lg = gx_L[g];
    float acc[TSR_MG];
#pragma unroll
    for (int j = 0; j < TSR_MG; ++j) acc[j] = 0.0f;
    for (int tau = 0; tau < Lg; ++tau) {
        const bool have = tau < left;  // rows beyond the window: weights of the templates that reach here are 0
        const float *__restrict__ x = X + (int64_t)(row + (have ? tau : 0)) * D;
        const float *__restrict__ wt = wg + (int64_t)tau * WR32 * TSR_MG;
        if (VEC4) {
            for (int d = 0; d < D; d += 4) {
                float4 xv = *reinterpret_cast<const float4 *>(x + d);
                if (!have) xv = make_float4(0.f, 0.f, 0.f, 0.f);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < TSR_MG; ++j) {
                        const float prod = wt[(d + u) * TSR_MG + j] * xs[u];
                        acc[j] = acc[j] + prod;
                    }
            }
        } else {
            for (int d = 0; d < D; ++d) {
                const float xv = have ? x[d] : 0.0f;
#pragma unroll
                for (int j = 0; j < TSR_MG; ++j) {
                    const float prod = wt[d * TSR_MG + j] * xv;
                    acc[j] = acc[j] + prod;
                }
            }
        }
    }
    if (!on) return;
#pragma unroll
    for (int j = 0; j < TSR_MG; ++j) {
        const int m = g * TSR_MG + j;
        if (m < M) out[(int64_t)m * NW + i] = (Lm[m] <= left ? acc[j] : 0.0f) + cf[m];
    }
}

// The isolated-classification windows WITHOUT materialising them: window item i looks at the
// frames f = first[i] + tau of the resident binary corpus; a frame inside [lo[i], hi[i]) is real
// (bytes 0/1 -> 0.0f/1.0f), every other row of the window is the background vector `bgd`
// (front/tail padding of the utterance and the rows behind the phone's segment end,
// get_train_data.py:791-832).  Arithmetic and order as float_windows_kernel, so the scores are those
// of detect_float on the materialised window, bit for bit.
__global__ __launch_bounds__(kThreads) void isolated_windows_kernel(
    const uint8_t *__restrict__ X, int Dp, int D, int WR32, const float *__restrict__ bgd,
    const int64_t *__restrict__ item_first, const int64_t *__restrict__ item_lo, const int64_t *__restrict__ item_hi,
    const int32_t *__restrict__ item_left, int64_t NW, const float *__restrict__ Wx,
    const int64_t *__restrict__ gx_off, const int32_t *__restrict__ gx_L, const int32_t *__restrict__ Lm,
    const float *__restrict__ cf, int M, float *__restrict__ out)
{
#pragma clang fp contract(off)  // the reference rounds the product to f32 before the add
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int g = blockIdx.y;
    const bool on = i < NW;
    const int64_t first = on ? item_first[i] : 0, lo = on ? item_lo[i] : 0, hi = on ? item_hi[i] : 0;
    const int left = on ? item_left[i] : 0;
    const float *__restrict__ wg = Wx + gx_off[g];
    const int Lg = gx_L[g];
    float acc[TSR_MG];
#pragma unroll
    for (int j = 0; j < TSR_MG; ++j) acc[j] = 0.0f;
    for (int tau = 0; tau < Lg; ++tau) {
        const int64_t f = first + tau;
        const bool real = f >= lo && f < hi;
        const uint8_t *__restrict__ x = X + (real ? f : 0) * (int64_t)Dp;
        const float *__restrict__ wt = wg + (int64_t)tau * WR32 * TSR_MG;
        for (int d0 = 0; d0 < D; d0 += 16) {  // resident rows are 16-byte multiples (zero padded)
            const uint4 xv = *reinterpret_cast<const uint4 *>(x + d0);
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (d0 + u < D) {  // uniform
                    const float xb = (float)((xw[u >> 2] >> (8 * (u & 3))) & 0xffu);
                    const float xs = real ? xb : bgd[d0 + u];
#pragma unroll
                    for (int j = 0; j < TSR_MG; ++j) {
                        const float prod = wt[(d0 + u) * TSR_MG + j] * xs;
                        acc[j] = acc[j] + prod;
                    }
                }
            }
        }
    }
    if (!on) return;
#pragma unroll
    for (int j = 0; j < TSR_MG; ++j) {
        const int m = g * TSR_MG + j;
        if (m < M) out[(int64_t)m * NW + i] = (Lm[m] <= left ? acc[j] : 0.0f) + cf[m];
    }
}

}  // namespace

int tsr_launch_window_argmax(tsr_ctx *ctx, const float *d_all, int64_t stride, int M, const int64_t *d_starts,
                             const int64_t *d_ends, int64_t W, float *d_score, int64_t *d_loc, int32_t *d_tid)
{
    if (W <= 0) return TSR_OK;
    const int wpb = kThreads / 64;
    hipLaunchKernelGGL(window_argmax_kernel, dim3((unsigned)((W + wpb - 1) / wpb)), dim3(kThreads), 0, ctx->stream,
                       d_all, stride, M, d_starts, d_ends, W, d_score, d_loc, d_tid);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_float_windows(tsr_ctx *ctx, const tsr_bank *bank, const float *d_X, const int32_t *d_item_row,
                             const int32_t *d_item_left, int64_t NW, float *d_out)
{
    if (NW <= 0) return TSR_OK;
    const int D = (int)bank->D;
    dim3 grid((unsigned)((NW + kThreads - 1) / kThreads), (unsigned)bank->n_groups);
    if (D % 4 == 0)
        hipLaunchKernelGGL(float_windows_kernel<true>, grid, dim3(kThreads), 0, ctx->stream, d_X, D, bank->Dq * 32,
                           d_item_row, d_item_left, NW, bank->d_Wx, bank->d_gx_off, bank->d_gx_L, bank->d_L,
                           bank->d_cf, (int)bank->M, d_out);
    else
        hipLaunchKernelGGL(float_windows_kernel<false>, grid, dim3(kThreads), 0, ctx->stream, d_X, D, bank->Dq * 32,
                           d_item_row, d_item_left, NW, bank->d_Wx, bank->d_gx_off, bank->d_gx_L, bank->d_L,
                           bank->d_cf, (int)bank->M, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_isolated_windows(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_X, int Dp, const float *d_bgd,
                                const int64_t *d_first, const int64_t *d_lo, const int64_t *d_hi,
                                const int32_t *d_left, int64_t NW, float *d_out)
{
    if (NW <= 0) return TSR_OK;
    dim3 grid((unsigned)((NW + kThreads - 1) / kThreads), (unsigned)bank->n_groups);
    hipLaunchKernelGGL(isolated_windows_kernel, grid, dim3(kThreads), 0, ctx->stream, d_X, Dp, (int)bank->D,
                       bank->Dq * 32, d_bgd, d_first, d_lo, d_hi, d_left, NW, bank->d_Wx, bank->d_gx_off, bank->d_gx_L,
                       bank->d_L, bank->d_cf, (int)bank->M, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- peak picking on the score rows -------------------------------------------
namespace {
// roc_functions.py:177-187 (get_max_above_threshold): for utterance n with T_n scores and
// l = T_n - C0 + 1, the indices t in [1, l-1) with v[t] >= v[t-1], v[t] >= v[t+1] and
// v[t] >= threshold, ascending.  One wave per utterance, ballot-ordered compaction; utterance n's
// indices are written from times[offsets[n]] on, their number to counts[n].
__global__ __launch_bounds__(kThreads) void peaks_kernel(const float *__restrict__ v, const int64_t *__restrict__ offsets,
                                                         int64_t N, float threshold, int C0,
                                                         int64_t *__restrict__ counts, int32_t *__restrict__ times)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t o = offsets[n];
    const int64_t l = offsets[n + 1] - o - C0 + 1;
    const float *__restrict__ x = v + o;
    int64_t cnt = 0;
    for (int64_t t0 = 1; t0 < l - 1; t0 += 64) {
        const int64_t t = t0 + lane;
        bool hit = false;
        if (t < l - 1) {
            const float c = x[t];
            hit = c >= x[t - 1] && c >= x[t + 1] && c >= threshold;
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (hit) times[o + cnt + __builtin_popcountll(mask & ((1ull << lane) - 1))] = (int32_t)t;
        cnt += __builtin_popcountll(mask);
    }
    if (lane == 0) counts[n] = cnt;
}

// roc_functions.py:88-139 (get_max_detection_in_syllable_windows): window w is the slice [s, e) of
// row utt[w] of the zero-padded detection array (values beyond the utterance's T_n scores are 0);
// among its inner positions that are >= both neighbours inside the slice, the maximum; -inf if none.
__global__ __launch_bounds__(kThreads) void window_peak_max_kernel(
    const float *__restrict__ v, const int64_t *__restrict__ offsets, const int64_t *__restrict__ utt,
    const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t W, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (w >= W) return;
    const int64_t n = utt[w], o = offsets[n], T = offsets[n + 1] - o;
    const int64_t s = starts[w], e = ends[w];
    auto at = [&](int64_t t) { return t < T ? v[o + t] : 0.0f; };
    float best = -__builtin_inff();
    for (int64_t t = s + 1 + lane; t < e - 1; t += 64) {
        const float c = at(t);
        if (c >= at(t - 1) && c >= at(t + 1)) best = fmaxf(best, c);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if (lane == 0) out[w] = best;
}
}  // namespace

int tsr_launch_peaks(tsr_ctx *ctx, const float *d_v, const int64_t *d_offsets, int64_t N, float threshold, int C0,
                     int64_t *d_counts, int32_t *d_times)
{
    if (N <= 0) return TSR_OK;
    const int wpb = kThreads / 64;
    hipLaunchKernelGGL(peaks_kernel, dim3((unsigned)((N + wpb - 1) / wpb)), dim3(kThreads), 0, ctx->stream, d_v,
                       d_offsets, N, threshold, C0, d_counts, d_times);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_window_peak_max(tsr_ctx *ctx, const float *d_v, const int64_t *d_offsets, const int64_t *d_utt,
                               const int64_t *d_starts, const int64_t *d_ends, int64_t W, float *d_out)
{
    if (W <= 0) return TSR_OK;
    const int wpb = kThreads / 64;
    hipLaunchKernelGGL(window_peak_max_kernel, dim3((unsigned)((W + wpb - 1) / wpb)), dim3(kThreads), 0, ctx->stream,
                       d_v, d_offsets, d_utt, d_starts, d_ends, W, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- Bernoulli-mixture M-step sums: out[k][d] = sum_n aff[n][k] * x[n][d] ------------
// bernoulli_mixture.py:179 (np.dot(affinities.T, data_mat)) on the resident binary corpus: one lane per
// feature d (coalesced byte reads of frame n), KC components per lane, frames split over gridDim.y
// slices whose partial sums are added in slice order by the second kernel (deterministic).
namespace {
constexpr int kMK = 8;
__global__ __launch_bounds__(kThreads) void mstep_partial_kernel(const uint8_t *__restrict__ X, int64_t P, int D, int Dp,
                                                                 const float *__restrict__ aff /*[P][K]*/, int K, int k0,
                                                                 int64_t rows_per_slice, float *__restrict__ part /*[S][K][D]*/)
{
    const int d = blockIdx.x * kThreads + threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.y * rows_per_slice, n1 = min(P, n0 + rows_per_slice);
    float acc[kMK];
#pragma unroll
    for (int j = 0; j < kMK; ++j) acc[j] = 0.0f;
    if (d < D)
        for (int64_t n = n0; n < n1; ++n) {
            const float x = (float)X[n * Dp + d];
            const float *a = aff + n * K + k0;  // wave-uniform
#pragma unroll
            for (int j = 0; j < kMK; ++j)
                if (k0 + j < K) acc[j] = __builtin_fmaf(a[j], x, acc[j]);
        }
    if (d < D)
#pragma unroll
        for (int j = 0; j < kMK; ++j)
            if (k0 + j < K) part[((int64_t)blockIdx.y * K + k0 + j) * D + d] = acc[j];
}

__global__ __launch_bounds__(kThreads) void mstep_reduce_kernel(const float *__restrict__ part, int S, int64_t KD,
                                                                float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= KD) return;
    float s = 0.0f;
    for (int sl = 0; sl < S; ++sl) s += part[(int64_t)sl * KD + i];
    out[i] = s;
}
}  // namespace

int tsr_launch_mstep(tsr_ctx *ctx, const uint8_t *d_X, int64_t P, int D, int Dp, const float *d_aff, int K,
                     float *d_part, int S, float *d_out)
{
    if (P <= 0 || K <= 0) return TSR_OK;
    const int64_t rows_per_slice = (P + S - 1) / S;
    for (int k0 = 0; k0 < K; k0 += kMK) {
        hipLaunchKernelGGL(mstep_partial_kernel, dim3((unsigned)((D + kThreads - 1) / kThreads), (unsigned)S), dim3(kThreads), 0,
                           ctx->stream, d_X, P, D, Dp, d_aff, K, k0, rows_per_slice, d_part);
        TSR_HIP(hipGetLastError());
    }
    const int64_t KD = (int64_t)K * D;
    hipLaunchKernelGGL(mstep_reduce_kernel, dim3((unsigned)((KD + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                       d_part, S, KD, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- Bernoulli-mixture E-step epilogue (bernoulli_mixture.py:263-285) ----------------------------------
// From the per-component scores log P(x_n | k) (the bank pass, [K][P] f32) and the log mixture weights:
// per data point the log marginal by a max-shifted log-sum-exp and the posterior component probabilities
// ("affinities"; one-hot on the best-scoring component with hard assignment, :278-281), all in f64 with one
// rounding to f32 at the end.  One lane per data point; the scores of consecutive points are consecutive, so
// every load is coalesced.  The log-likelihood is summed per block in a fixed tree order and the block sums
// (and the per-block column sums of the affinities, i.e. the M-step's new weights) are added up by the host
// in block order: bit-reproducible from run to run.
namespace {
__global__ __launch_bounds__(kThreads) void mixture_estep_kernel(const float *__restrict__ scores, int64_t P, int K,
                                                                 const double *__restrict__ log_w, int hard,
                                                                 float *__restrict__ aff /*[P][K]*/,
                                                                 double *__restrict__ blk /*[blocks][1 + K]*/)
{
    __shared__ double red[kThreads];
    const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const bool live = n < P;
    double ll = 0.0;
    if (live) {
        double m = -INFINITY;
        float best_s = 0.0f;
        int best_k = 0;
        for (int k = 0; k < K; ++k) {
            const float s = scores[(int64_t)k * P + n];
            m = fmax(m, (double)s + log_w[k]);
            if (k == 0 || s > best_s) {  // np.argmax: first maximum of the template scores (weights not applied, :280)
                best_s = s;
                best_k = k;
            }
        }
        double sum = 0.0;
        for (int k = 0; k < K; ++k) sum += exp((double)scores[(int64_t)k * P + n] + log_w[k] - m);
        ll = log(sum) + m;
        for (int k = 0; k < K; ++k) {
            const double a = hard ? (k == best_k ? 1.0 : 0.0) : exp((double)scores[(int64_t)k * P + n] + log_w[k] - m) / sum;
            aff[n * K + k] = (float)a;
        }
    }
    // block sums: [0] log-likelihood, [1 + k] column k of the affinities (as stored, f32 values)
    for (int q = 0; q <= K; ++q) {
        double v = 0.0;
        if (live) v = q == 0 ? ll : (double)aff[n * K + q - 1];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = kThreads / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) blk[(int64_t)blockIdx.x * (K + 1) + q] = red[0];
        __syncthreads();
    }
}
}  // namespace

// log P(x_n | k) = c_k + sum_d x[n,d] odds[k,d] for the rows of a resident 0/1 data matrix: a [N, D] x [D, K] product
// with K of the order of ten -- far too narrow for the bank kernels (K one-tap "templates" are K of 256 columns of a
// workgroup's matrix tiles, and every EM iteration would re-encode the bank on the host).  One wave per row: the lanes
// stride over the features four at a time (one dword of 0/1 bytes, a float4 of each component's log-odds row), eight
// components per pass, f32 partial sums per lane, f64 across the lanes and for + c_k, one rounding to f32.  The tables
// (K x D floats, a few hundred KB) are shared by all waves through L1/L2.
namespace {
__global__ __launch_bounds__(256) void mixture_scores_kernel(const uint8_t *__restrict__ X, int64_t P, int D, int Dp,
                                                             const float *__restrict__ odds /*[K][Dq]*/, int Dq,
                                                             const double *__restrict__ cst /*[K]*/, int K,
                                                             float *__restrict__ scores /*[K][P]*/)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (n >= P) return;  // (whole wave)
    const uint32_t *xr = reinterpret_cast<const uint32_t *>(X + n * Dp);
    const int nd4 = Dq >> 2;  // features in fours (rows and tables are zero padded up to Dq)
    for (int k0 = 0; k0 < K; k0 += 8) {
        const int kn = min(8, K - k0);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = lane; i < nd4; i += 64) {
            const uint32_t x = xr[i];
            const float x0 = (float)(x & 1u), x1 = (float)((x >> 8) & 1u), x2 = (float)((x >> 16) & 1u), x3 = (float)(x >> 24);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < kn) {
                    const float4 w = *reinterpret_cast<const float4 *>(odds + (int64_t)(k0 + k) * Dq + 4 * i);
                    acc[k] = __builtin_fmaf(x0, w.x, acc[k]);
                    acc[k] = __builtin_fmaf(x1, w.y, acc[k]);
                    acc[k] = __builtin_fmaf(x2, w.z, acc[k]);
                    acc[k] = __builtin_fmaf(x3, w.w, acc[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double v = (double)acc[k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (k < kn && lane == 0) scores[(int64_t)(k0 + k) * P + n] = (float)(v + cst[k0 + k]);
        }
    }
}
}  // namespace

int tsr_launch_mixture_scores(tsr_ctx *ctx, const uint8_t *d_X, int64_t P, int D, int Dp, const float *d_odds, int Dq,
                              const double *d_c, int K, float *d_scores)
{
    if (P <= 0) return TSR_OK;
    hipLaunchKernelGGL(mixture_scores_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, ctx->stream, d_X, P, D, Dp, d_odds, Dq, d_c,
                       K, d_scores);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_mixture_estep(tsr_ctx *ctx, const float *d_scores, int64_t P, int K, const double *d_log_w, int hard,
                             float *d_aff, double *d_blk, int *n_blocks)
{
    const int blocks = (int)((P + kThreads - 1) / kThreads);
    *n_blocks = blocks;
    if (P <= 0) return TSR_OK;
    hipLaunchKernelGGL(mixture_estep_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, ctx->stream, d_scores, P, K, d_log_w,
                       hard, d_aff, d_blk);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- detection clustering: cluster_times.pyx:133-222 / :221-310 + roc_functions.py:189-210 --
// The reference's state machine over the ascending peak times of one utterance is inherently sequential and
// tiny (a handful of peaks); one LANE per utterance runs it on the peak lists tsr_launch_peaks left on the
// device, so that only the clustered (start, end) pairs go back to the host.  uint16 arithmetic for the
// running distances as in the Cython source; part is scratch of n + 1 bytes per utterance.  With `lens_all`
// (the template length of every detection) the thresholds are cluster_times_template_lengths': a link opens a
// cluster when it is >= the length of the detection before it, a cluster is split when it is longer than
// 1.5 x the length of its first detection (all of these are exact in f32).
namespace {
__global__ __launch_bounds__(64) void cluster_times_kernel(const int32_t *__restrict__ times_all, const int64_t *__restrict__ offsets,
                                                           const int64_t *__restrict__ counts, int64_t N, float C0, float C1,
                                                           const int32_t *__restrict__ lens_all, uint8_t *__restrict__ part_all, int64_t *__restrict__ n_clusters,
                                                           int32_t *__restrict__ starts, int32_t *__restrict__ ends)
{
    const int64_t u = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (u >= N) return;
    const int32_t *times = times_all + offsets[u];
    const int32_t *lens = lens_all ? lens_all + offsets[u] : nullptr;
    auto c0 = [&](unsigned i) { return lens ? (float)lens[i] : C0; };         // threshold of the link behind detection i
    auto c1 = [&](unsigned i) { return lens ? 1.5f * (float)lens[i] : C1; };  // longest cluster starting at detection i
    uint8_t *part = part_all + offsets[u] + u;  // n + 1 <= T_u + 1 bytes
    const unsigned n = (unsigned)counts[u] & 0xffffu;
    for (unsigned i = 0; i <= n; ++i) part[i] = 0;
    part[0] = 1;
    part[n] = 2;
    for (unsigned i = 1; i < n; ++i)
        if ((float)(times[i] - times[i - 1]) >= c0(i - 1)) part[i] = 1;
    unsigned start = 0, max_link = 0;
    while (start < n) {
        if (part[start + 1] > 0) {
            part[start] = 2;
            ++start;
            while (part[start] != 1 && start < n) ++start;
        } else {
            unsigned max_dist = 0, total = 0, k = 1;
            while (part[start + k] == 0) {
                const unsigned cur = (unsigned)(times[start + k] - times[start + k - 1]) & 0xffffu;
                if (cur > max_dist) {
                    max_link = start + k;
                    max_dist = cur;
                }
                total = (total + cur) & 0xffffu;
                ++k;
            }
            if ((float)total > c1(start)) {
                part[max_link] = ((float)(times[start + k - 1] - times[max_link]) <= c1(max_link)) ? 2 : 1;
                if ((float)(times[max_link - 1] - times[start]) <= c1(start)) {
                    part[start] = 2;
                    if (part[max_link] == 1) start = max_link;
                } else {
                    part[start] = 1;
                }
            } else {
                part[start] = 2;
                start += k;
                while (part[start] != 1 && start < n) ++start;
            }
        }
    }
    // get_cluster_starts_ends: (first time of the cluster, last time + 1)
    int64_t nc = 0;
    unsigned t = 0;
    while (t < n) {
        const int32_t s = times[t];
        ++t;
        while (part[t] != 2) ++t;
        starts[offsets[u] + nc] = s;
        ends[offsets[u] + nc] = times[t - 1] + 1;
        ++nc;
    }
    n_clusters[u] = nc;
}
}  // namespace

// cluster_times_variable_length (cluster_times.pyx:73-131): what its never-splitting split loop leaves behind
// is the link partition -- a cluster starts wherever the gap to the previous detection is at least that
// detection's length: longest_cluster_detect_id (:18-45) only selects clusters that end at a marker and
// split_cluster (:47-70) sets exactly that marker again.  One thread per detection.
namespace {
__global__ __launch_bounds__(256) void link_partition_kernel(const int32_t *__restrict__ times, const int32_t *__restrict__ lengths,
                                                             int64_t n, uint8_t *__restrict__ part)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) part[i] = (i == 0 || times[i] - times[i - 1] >= lengths[i - 1]) ? 1 : 0;
}
}  // namespace

int tsr_launch_link_partition(tsr_ctx *ctx, const int32_t *d_times, const int32_t *d_lengths, int64_t n, uint8_t *d_part)
{
    if (n <= 0) return TSR_OK;
    hipLaunchKernelGGL(link_partition_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_times, d_lengths,
                       n, d_part);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_cluster_times(tsr_ctx *ctx, const int32_t *d_times, const int64_t *d_offsets, const int64_t *d_counts, int64_t N,
                             float C0, float C1, const int32_t *d_lens, uint8_t *d_part, int64_t *d_n_clusters,
                             int32_t *d_starts, int32_t *d_ends)
{
    if (N <= 0) return TSR_OK;
    hipLaunchKernelGGL(cluster_times_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, d_times, d_offsets,
                       d_counts, N, C0, C1, d_lens, d_part, d_n_clusters, d_starts, d_ends);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
