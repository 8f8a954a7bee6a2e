// edges2.hip -- the rest of the edge front-end:
//   threshold_edgemap / threshold_edge_block   edge_signal_proc.py:1184-1307  (whole-array quantile threshold per row block)
//   spread_edgemap / _spread_edges             edge_signal_proc.py:1397-1417  (max along the block's orientation)
//   magnitude_features                         edge_signal_proc.py:600-670    (per-bin quantile in overlapping windows)
//   magnitude_features_whole_block             edge_signal_proc.py:673-747    (one quantile per window, square spread)
// These are per-utterance functions on small arrays (tens of kilobytes): one launch each, latency-bound; what
// matters is that every comparison is the reference's f64 comparison on the same operands, so the maps are
// identical.  The k-th order statistic is found by an 8-pass radix select on the f64 bit patterns (keys made
// monotone for negative values as well), never by sorting.
#include "tsr_internal.h"

namespace {

constexpr int kT = 1024;

// f64 -> u64 key with the same order as the doubles (no NaNs expected; -0.0 < +0.0 in key order, which the
// callers never distinguish: a quantile that IS zero compares equal to both)
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// k-th smallest (0-based) of the values x[i] with pred(i), i in [0, n): all threads of the block call this with
// the same arguments; hist [256] and state [2] are LDS scratch.  get(i) returns the value, use(i) the predicate.
template <typename Get, typename Use>
__device__ double block_select(int64_t n, long long k, Get get, Use use, int *hist, unsigned long long *state)
{
    const int tid = threadIdx.x, nthreads = blockDim.x;
    if (tid == 0) {
        state[0] = 0ull;
        state[1] = (unsigned long long)k;
    }
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = 8 * pass;
        for (int i = tid; i < 256; i += nthreads) hist[i] = 0;
        __syncthreads();
        const unsigned long long prefix = state[0];
        for (int64_t i = tid; i < n; i += nthreads)
            if (use(i)) {
                const unsigned long long key = order_key(get(i));
                if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
            }
        __syncthreads();
        if (tid == 0) {  // 256 bins, once per pass: a serial walk is shorter than a scan here
            long long kk = (long long)state[1];
            int b = 0;
            while (b < 255 && kk >= hist[b]) kk -= hist[b++];
            state[0] = prefix | ((unsigned long long)b << shift);
            state[1] = (unsigned long long)kk;
        }
        __syncthreads();
    }
    return key_value(state[0]);
}

// ---- threshold_edgemap: one workgroup per row block ---------------------------------------------------------
__global__ __launch_bounds__(kT) void threshold_blocks_kernel(double *__restrict__ E, const int64_t *__restrict__ breaks,
                                                              int64_t T, double q, const double *__restrict__ abst, int binary,
                                                              double *__restrict__ taus)
{
    __shared__ int hist[256];
    __shared__ unsigned long long state[2];
    __shared__ long long cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    double *B = E + breaks[b] * T;
    const int64_t n_el = (breaks[b + 1] - breaks[b]) * T;
    if (tid == 0) cnt = 0;
    __syncthreads();
    long long mine = 0;
    for (int64_t i = tid; i < n_el; i += kT) mine += B[i] > 0.0;
    if (mine) atomicAdd((unsigned long long *)&cnt, (unsigned long long)mine);
    __syncthreads();
    const long long n = cnt;
    if (n <= 0) {  // "E_block[:] = 0" (:1283-1285)
        for (int64_t i = tid; i < n_el; i += kT) B[i] = 0.0;
        if (tid == 0) taus[b] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const long long k = (long long)(q * (double)n);  // int(quantile_level * n), :1287
    const double tau = block_select(n_el, k, [&](int64_t i) { return B[i]; }, [&](int64_t i) { return B[i] > 0.0; }, hist, state);
    const double thr = tau > abst[b] ? tau : abst[b];
    for (int64_t i = tid; i < n_el; i += kT) {
        const double v = B[i];
        B[i] = (v > 0.0 && v >= thr) ? (binary ? 1.0 : v) : 0.0;
    }
    if (tid == 0) taus[b] = tau;
}

// ---- spread_edgemap: out[h][t] = max(0, max_k in[h + k d0][t + k d1]) inside the row block ---------------------
__global__ __launch_bounds__(256) void spread_blocks_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                            const int64_t *__restrict__ breaks, int nb, int64_t T,
                                                            const int32_t *__restrict__ orient, int s)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rows = breaks[nb];
    if (idx >= rows * T) return;
    const int64_t row = idx / T, t = idx - row * T;
    int b = 0;
    while (b + 1 < nb && row >= breaks[b + 1]) ++b;
    if (row < breaks[0]) {  // rows in front of the first break are not touched by the reference
        out[idx] = in[idx];
        return;
    }
    const int64_t r0 = breaks[b], F = breaks[b + 1] - r0, h = row - r0;
    double m = 0.0;
    if (!orient) {  // spread_type 'all' (edge_signal_proc.py:1623-1636): scipy maximum_filter, (2s+1) x (2s+1), constant 0 outside the block
        for (int k0 = -s; k0 <= s; ++k0)
            for (int k1 = -s; k1 <= s; ++k1) {
                const int64_t hh = h + k0, tt = t + k1;
                if (hh >= 0 && hh < F && tt >= 0 && tt < T) {
                    const double v = in[(r0 + hh) * T + tt];
                    m = v > m ? v : m;
                }
            }
        out[idx] = m;
        return;
    }
    const int d0 = orient[2 * b], d1 = orient[2 * b + 1];
    for (int k = -s; k <= s; ++k) {
        const int64_t hh = h + (int64_t)k * d0, tt = t + (int64_t)k * d1;
        if (hh >= 0 && hh < F && tt >= 0 && tt < T) {
            const double v = in[(r0 + hh) * T + tt];
            m = v > m ? v : m;   // np.maximum with a zero start: NaNs aside, max(0, ...)
        }
    }
    out[idx] = m;
}

// ---- magnitude features ------------------------------------------------------------------------------------
// smoothing (maximum over `size` bins, indices reflected) + decimation: S [F0][T] -> S2 [T][F] (time-major)
__global__ __launch_bounds__(256) void mag_prepare_kernel(const double *__restrict__ S, int F0, int64_t T, int size, int down,
                                                          int F, double *__restrict__ S2)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * F) return;
    const int64_t t = idx / F;
    const int f = (int)(idx - t * F);
    const int f0 = down > 0 ? f * down : f;
    double m;
    if (size > 0) {
        m = -INFINITY;
        const int lo = size / 2;
        for (int j = 0; j < size; ++j) {
            int i = f0 - lo + j;
            i %= 2 * F0;
            if (i < 0) i += 2 * F0;
            if (i >= F0) i = 2 * F0 - 1 - i;
            const double v = S[(int64_t)i * T + t];
            m = v > m ? v : m;
        }
    } else {
        m = S[(int64_t)f0 * T + t];
    }
    S2[idx] = m;
}

struct MagWin {  // estimation window [a, b), frames it sets [s0, s1), rank of the threshold in the sorted window
    int a, b, s0, s1;
    long long k;
};

// per-bin quantile: one lane per (window, bin); the k-th smallest of <= a few hundred values by counting
__global__ __launch_bounds__(64) void mag_perbin_kernel(const double *__restrict__ S2, int F, const MagWin *__restrict__ wins,
                                                        uint8_t *__restrict__ E)
{
    const MagWin w = wins[blockIdx.y];
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    const int n = w.b - w.a;
    double thr = 0.0;
    for (int i = 0; i < n; ++i) {
        const double v = S2[(int64_t)(w.a + i) * F + f];
        int less = 0, leq = 0;
        for (int j = 0; j < n; ++j) {
            const double u = S2[(int64_t)(w.a + j) * F + f];
            less += u < v;
            leq += u <= v;
        }
        if (less <= w.k && w.k < leq) {
            thr = v;
            break;
        }
    }
    for (int t = w.s0; t < w.s1; ++t) E[(int64_t)t * F + f] = S2[(int64_t)t * F + f] >= thr ? 1 : 0;
}

// one quantile per window over all bins: one workgroup per window
__global__ __launch_bounds__(kT) void mag_whole_kernel(const double *__restrict__ S2, int F, const MagWin *__restrict__ wins,
                                                       uint8_t *__restrict__ E)
{
    __shared__ int hist[256];
    __shared__ unsigned long long state[2];
    const MagWin w = wins[blockIdx.x];
    const double *base = S2 + (int64_t)w.a * F;
    const int64_t n = (int64_t)(w.b - w.a) * F;
    const double thr = block_select(n, w.k, [&](int64_t i) { return base[i]; }, [&](int64_t) { return true; }, hist, state);
    for (int64_t i = (int64_t)w.s0 * F + threadIdx.x; i < (int64_t)w.s1 * F; i += kT) E[i] = S2[i] >= thr ? 1 : 0;
}

// spreads: along time (+-m frames, OR) or the square one (centre set, or >= min_count of the other cells of the
// (2r+1)^2 box set)
__global__ __launch_bounds__(256) void mag_spread_kernel(const uint8_t *__restrict__ E, int64_t T, int F, int m, int r, int min_count,
                                                         uint8_t *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * F) return;
    const int64_t t = idx / F;
    const int f = (int)(idx - t * F);
    uint8_t v = E[idx];
    if (m > 0) {
        for (int64_t tt = t - m; tt <= t + m; ++tt)
            if (tt >= 0 && tt < T) v |= E[tt * F + f];
    } else if (r > 0 && !v) {
        int cnt = 0;
        for (int64_t tt = t - r; tt <= t + r; ++tt)
            for (int ff = f - r; ff <= f + r; ++ff)
                if (tt >= 0 && tt < T && ff >= 0 && ff < F) cnt += E[tt * F + ff];
        v = cnt >= min_count;
    }
    out[idx] = v;
}

}  // namespace

int tsr_launch_threshold_edgemap(tsr_ctx *ctx, double *d_E, const int64_t *d_breaks, int nb, int64_t T, double q,
                                 const double *d_abst, int binary, double *d_taus)
{
    if (nb <= 0) return TSR_OK;
    hipLaunchKernelGGL(threshold_blocks_kernel, dim3((unsigned)nb), dim3(kT), 0, ctx->stream, d_E, d_breaks, T, q, d_abst, binary, d_taus);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_spread_edgemap(tsr_ctx *ctx, const double *d_in, double *d_out, const int64_t *d_breaks, int nb, int64_t rows,
                              int64_t T, const int32_t *d_orient, int spread)
{
    const int64_t n = rows * T;
    if (n <= 0) return TSR_OK;
    hipLaunchKernelGGL(spread_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, d_out, d_breaks, nb, T,
                       d_orient, spread);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_magnitude(tsr_ctx *ctx, const double *d_S, int F0, int64_t T, int smooth, int down, int F, double *d_S2,
                         const void *d_wins, int n_wins, int whole, int m, int r, int min_count, uint8_t *d_E, uint8_t *d_out)
{
    const int64_t n = T * F;
    if (n <= 0) return TSR_OK;
    hipLaunchKernelGGL(mag_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_S, F0, T, smooth, down, F, d_S2);
    if (whole)
        hipLaunchKernelGGL(mag_whole_kernel, dim3((unsigned)n_wins), dim3(kT), 0, ctx->stream, d_S2, F, (const MagWin *)d_wins, d_E);
    else
        hipLaunchKernelGGL(mag_perbin_kernel, dim3((unsigned)((F + 63) / 64), (unsigned)n_wins), dim3(64), 0, ctx->stream, d_S2, F,
                           (const MagWin *)d_wins, d_E);
    hipLaunchKernelGGL(mag_spread_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_E, T, F, whole ? 0 : m,
                       whole ? r : 0, min_count, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
