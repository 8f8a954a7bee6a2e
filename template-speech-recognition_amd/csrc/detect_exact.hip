// detect_exact.hip -- sequential-order (bit-exact) scoring kernels for gfx950.
//
// The reference accumulates every window's score in f32, one tap at a time, in
// the order tau -> frequency -> part (compute_likelihood_linear_filter.pyx:162-175).
// Floating-point addition is not associative, so the only way to reproduce its
// bits is to keep that order.  These kernels give every lane one window and walk
// the taps in reference order; parallelism comes from the windows (one lane
// each) and the templates (TSR_MG accumulators per lane), never from splitting a
// sum.  `acc = fma(x, w, acc)` with x in {0.0f, 1.0f} and finite w is exactly
// the reference's `if F: acc += w` (x == 1: rn(acc + w); x == 0: acc).
//
// Roofline: VALU-bound (one v_fmac per tap and template).  Algorithmic work is
// P * Lmax * D * M fused adds; HBM traffic is the feature map once plus
// 4*M bytes per frame of raw scores.
#include "tsr_internal.h"

namespace {

constexpr int kThreads = 256;

// One block = TB consecutive frame positions of the stream x one group of TSR_MG templates.
// The frames are bit-packed on the way into LDS (one dword = 32 features), which shrinks the
// tile 8x against the resident bytes: ~15 KiB for TB = 256 at D = 360, so 8 waves fit per SIMD
// and the scalar weight loads (one wave-uniform 32-byte line per feature: 8 templates) are
// hidden by the other waves.  Per feature: v_bfe + v_cvt build x in {0.0, 1.0}, then 8 v_fmac.
// LDS rows are an odd number of dwords apart (conflict-free across lanes).
template <int TB>
__global__ __launch_bounds__(TB) void exact_bank_kernel(
    const uint8_t *__restrict__ X, int64_t P, int D, int Dp, int WR /*dwords of bits per row*/,
    int SD /*LDS row stride, odd*/, int TC, const float *__restrict__ Wx,
    const int64_t *__restrict__ gx_off, const int32_t *__restrict__ gx_L, int M,
    float *__restrict__ raw)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const int g = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * TB;
    const int Lg = gx_L[g];
    const float *__restrict__ Wg = Wx + gx_off[g];

    float acc[TSR_MG];
#pragma unroll
    for (int m = 0; m < TSR_MG; ++m) acc[m] = 0.0f;

    const int rows = TB + TC - 1;
    const uint32_t *__restrict__ X4 = reinterpret_cast<const uint32_t *>(X);
    const int Dp4 = Dp >> 2;
    for (int tau0 = 0; tau0 < Lg; tau0 += TC) {
        __syncthreads();
        // ---- stage rows [p0+tau0, p0+tau0+rows): 32 resident 0/1 bytes -> one dword of bits.
        //      (x * 0x01020408) >> 24 gathers the low bits of 4 bytes into a nibble.
        for (int idx = tid; idx < rows * WR; idx += TB) {
            const int r = idx / WR, w = idx - r * WR;
            const int64_t p = p0 + tau0 + r;
            uint32_t bits = 0;
            if (p < P) {
                const uint32_t *src = X4 + p * Dp4 + w * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t x = (w * 8 + q < Dp4) ? src[q] : 0u;  // rows are zero padded up to Dp
                    bits |= ((x * 0x01020408u) >> 24 & 0xfu) << (4 * q);
                }
            }
            lds[r * SD + w] = bits;
        }
        __syncthreads();
        // ---- accumulate taps tau0 .. min(tau0+TC, Lg) in reference order: tau -> feature ----
        const int tau1 = min(tau0 + TC, Lg);
        for (int tau = tau0; tau < tau1; ++tau) {
            const uint32_t *xr = lds + (tid + (tau - tau0)) * SD;
            const float *__restrict__ w = Wg + (int64_t)tau * WR * 32 * TSR_MG;
            for (int i = 0; i < WR; ++i) {
                const uint32_t bits = xr[i];
                const float *__restrict__ wi = w + i * (32 * TSR_MG);
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float x = (float)((bits >> j) & 1u);
#pragma unroll
                    for (int m = 0; m < TSR_MG; ++m) acc[m] = __builtin_fmaf(x, wi[j * TSR_MG + m], acc[m]);
                }
            }
        }
    }
    const int64_t p = p0 + tid;
    if (p < P) {
#pragma unroll
        for (int m = 0; m < TSR_MG; ++m) {
            const int mm = g * TSR_MG + m;
            if (mm < M) raw[(int64_t)mm * P + p] = acc[m];
        }
    }
}

// Fallback for feature rows too wide for the LDS tile: same order, rows read
// straight from global memory (L1/L2 absorb the window overlap).
__global__ __launch_bounds__(kThreads) void exact_bank_direct_kernel(
    const uint8_t *__restrict__ X, int64_t P, int D, int Dp, int WR,
    const float *__restrict__ Wx, const int64_t *__restrict__ gx_off,
    const int32_t *__restrict__ gx_L, int M, float *__restrict__ raw)
{
    const int g = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int Lg = gx_L[g];
    const float *__restrict__ Wg = Wx + gx_off[g];
    float acc[TSR_MG];
#pragma unroll
    for (int m = 0; m < TSR_MG; ++m) acc[m] = 0.0f;
    for (int tau = 0; tau < Lg; ++tau) {
        const int64_t row = p + tau;
        const float *__restrict__ w = Wg + (int64_t)tau * WR * 32 * TSR_MG;
        for (int d = 0; d < D; ++d) {
            const float x = (row < P && X[row * Dp + d]) ? 1.0f : 0.0f;
            const float *__restrict__ wi = w + (int64_t)d * TSR_MG;
#pragma unroll
            for (int m = 0; m < TSR_MG; ++m) acc[m] = __builtin_fmaf(x, wi[m], acc[m]);
        }
    }
    if (p < P) {
#pragma unroll
        for (int m = 0; m < TSR_MG; ++m) {
            const int mm = g * TSR_MG + m;
            if (mm < M) raw[(int64_t)mm * P + p] = acc[m];
        }
    }
}

// get_train_data.py:962-992: v = raw (0 on the tail t > T - L_m) + f32(c_m);
// running max with strict '>' (lowest template id wins ties).
__global__ __launch_bounds__(kThreads) void bank_epilogue_kernel(
    const float *__restrict__ raw, int64_t P, int M, const int32_t *__restrict__ L,
    const float *__restrict__ cf, const int32_t *__restrict__ utt_end, float *__restrict__ scores,
    int32_t *__restrict__ ids, float *__restrict__ all)
{
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p >= P) return;
    const int64_t end = utt_end[p];
    float best = 0.0f;
    int besti = 0;
    for (int m = 0; m < M; ++m) {
        const float r = (p + L[m] <= end) ? raw[(int64_t)m * P + p] : 0.0f;
        const float v = r + cf[m];
        if (all) all[(int64_t)m * P + p] = v;
        if (m == 0 || v > best) {
            best = v;
            besti = m;
        }
    }
    if (scores) scores[p] = best;
    if (ids) ids[p] = besti;
}

// compute_likelihood_linear_filter.pyx:178-219 / :221-278 / :281-359: f32
// features, product rounded to f32 then added (no FMA contraction), accumulator
// initialised to cs[m] when given.  One thread per (template, window).
__global__ __launch_bounds__(64) void detect_f32_kernel(
    const float *__restrict__ F, int T, int D, const float *__restrict__ LF, int M, int L,
    const float *__restrict__ cs, float *__restrict__ out)
{
#pragma clang fp contract(off)  // the reference rounds the product to f32 before the add
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int m = blockIdx.y;
    if (t >= T) return;
    float acc = 0.0f;
    if (t + L <= T) {
        acc = cs ? cs[m] : 0.0f;
        const float *__restrict__ w = LF + (int64_t)m * L * D;   // wave-uniform: scalar loads
        const float *__restrict__ f = F + (int64_t)t * D;
        const int n = L * D;
        int k = 0;
        if ((D & 3) == 0) {  // rows are 16-byte multiples: four features per load, 16 in flight
            for (; k + 16 <= n; k += 16) {
                float4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4 *>(f + k + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float prod = w[k + 4 * u + j] * xs[j];
                        acc = acc + prod;
                    }
                }
            }
        }
        for (; k < n; ++k) {
            const float prod = w[k] * f[k];
            acc = acc + prod;
        }
    }
    out[(int64_t)m * T + t] = acc;
}

// detect (compute_likelihood_linear_filter.pyx:135-176) for ONE utterance and ONE filter straight from
// the caller's packed bytes: the per-call drop-in path (no bank, no corpus object).  One lane per
// window, the reference's order tau -> d, acc = fma(x, w, acc) with x in {0, 1}.
__global__ __launch_bounds__(64) void detect_u8_single_kernel(const uint8_t *__restrict__ F, int T, int D,
                                                              const float *__restrict__ LF, int L,
                                                              float *__restrict__ out)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= T) return;
    float acc = 0.0f;
    if (t + L <= T) {
        const uint8_t *__restrict__ f = F + (int64_t)t * D;  // rows t .. t+L-1 are contiguous: L*D bytes
        const int n = L * D;
        int k = 0;
        if ((D & 3) == 0) {  // (vector loads for the bytes, scalar loads for the weights: separate wait counters)
            for (; k + 16 <= n; k += 16) {
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const uint32_t *>(f + k + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xs = ((x[u] >> (8 * j)) & 0xffu) ? 1.0f : 0.0f;
                        acc = __builtin_fmaf(xs, LF[k + 4 * u + j], acc);
                    }
            }
        }
        for (; k < n; ++k) acc = __builtin_fmaf(f[k] ? 1.0f : 0.0f, LF[k], acc);
    }
    out[t] = acc;
}

// The same with both operands in LDS (D a multiple of 8, everything fits): the 64 lanes first copy the
// block's rows (as 0/1 bytes) and the whole filter, then every lane walks its window's chain with the data
// of the next 8 features already requested while the current 8 are added -- the chain of L*D dependent
// f32 additions is latency-bound, so what matters is that nothing but the adds sits on it.
__global__ __launch_bounds__(64) void detect_u8_single_lds_kernel(const uint8_t *__restrict__ F, int T, int D,
                                                                  const float *__restrict__ LF, int L, int RS,
                                                                  float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t du_smem[];
    const int t0 = blockIdx.x * 64, lane = threadIdx.x;
    const int nrows = min(64 + L - 1, T - t0);
    uint8_t *xs = du_smem;                                                        // [nrows][RS]
    float *ws = reinterpret_cast<float *>(du_smem + (((size_t)(64 + L - 1) * RS + 15) & ~(size_t)15));  // [L*D], 16-byte aligned (b128 accesses)
    const int D4 = D >> 2;
    for (int i = lane; i < nrows * D4; i += 64) {
        const int r = i / D4, c4 = (i - r * D4) * 4;
        const uint32_t x = *reinterpret_cast<const uint32_t *>(F + (int64_t)(t0 + r) * D + c4);
        *reinterpret_cast<uint32_t *>(xs + (size_t)r * RS + c4) = ((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;
    }
    for (int i = lane; i < (L * D) >> 2; i += 64) reinterpret_cast<float4 *>(ws)[i] = reinterpret_cast<const float4 *>(LF)[i];
    __syncthreads();
    const int t = t0 + lane;
    if (t >= T) return;
    float acc = 0.0f;
    if (t + L <= T) {
        const int per_row = D >> 3, n8 = L * per_row;  // 8-feature groups (rows are multiples of 8 bytes)
        auto xaddr = [&](int gidx) { const int tau = gidx / per_row; return xs + (size_t)(lane + tau) * RS + (gidx - tau * per_row) * 8; };
        uint2 xn = *reinterpret_cast<const uint2 *>(xaddr(0));
        float4 wn[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) wn[u] = reinterpret_cast<const float4 *>(ws)[u];
        for (int gidx = 0; gidx < n8; ++gidx) {
            const uint2 xc = xn;
            float4 wc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) wc[u] = wn[u];
            if (gidx + 1 < n8) {
                xn = *reinterpret_cast<const uint2 *>(xaddr(gidx + 1));
#pragma unroll
                for (int u = 0; u < 2; ++u) wn[u] = reinterpret_cast<const float4 *>(ws)[(gidx + 1) * 2 + u];
            }
            const uint32_t xw[2] = {xc.x, xc.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float w4[4] = {wc[u].x, wc[u].y, wc[u].z, wc[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_fmaf((float)((xw[u] >> (8 * j)) & 0xffu), w4[j], acc);
            }
        }
    }
    out[t] = acc;
}

// compute_likelihood_linear_filter.pyx:21-76: f64 sum, filter truncated at the
// end of the example.  One thread per (example, t); the max over t is taken by
// the caller with the reference's first-t/strict-'>' rule.
__global__ __launch_bounds__(kThreads) void max_detect_examples_kernel(
    const uint8_t *__restrict__ F, int N, int T, int D, const double *__restrict__ LF, int L,
    double *__restrict__ tmp)
{
    const int t = blockIdx.x * kThreads + threadIdx.x;
    const int n = blockIdx.y;
    if (t >= T) return;
    const int mt = min(T - t, L);
    const uint8_t *__restrict__ f = F + ((int64_t)n * T + t) * D;
    double acc = 0.0;
    for (int k = 0; k < mt * D; ++k)
        if (f[k]) acc += LF[k];
    tmp[(int64_t)n * T + t] = acc;
}

}  // namespace

int tsr_launch_exact_bank(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_feats, int64_t P,
                          int Dp, float *d_raw)
{
    if (P == 0) return TSR_OK;
    const int WR = bank->Dq;  // dwords of feature bits per row
    const int SD = WR | 1;
    const int Lmax = bank->Lmax;
    const size_t row_bytes = (size_t)SD * 4;
    constexpr int TB = 256;
    const size_t budget = 64 * 1024;  // keep several blocks per CU resident
    if (budget / row_bytes < (size_t)TB + 1) {  // absurdly wide rows: stream from global memory
        dim3 grid((unsigned)((P + kThreads - 1) / kThreads), bank->n_groups);
        hipLaunchKernelGGL(exact_bank_direct_kernel, grid, dim3(kThreads), 0, ctx->stream, d_feats, P,
                           (int)bank->D, Dp, WR, bank->d_Wx, bank->d_gx_off, bank->d_gx_L, bank->M, d_raw);
        TSR_HIP(hipGetLastError());
        return TSR_OK;
    }
    const int TC = (int)std::min<size_t>(Lmax, budget / row_bytes - (TB - 1));
    const size_t lds = (size_t)(TB + TC - 1) * row_bytes;
    dim3 grid((unsigned)((P + TB - 1) / TB), bank->n_groups);
    hipLaunchKernelGGL(exact_bank_kernel<TB>, grid, dim3(TB), lds, ctx->stream, d_feats, P, (int)bank->D,
                       Dp, WR, SD, TC, bank->d_Wx, bank->d_gx_off, bank->d_gx_L, bank->M, d_raw);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_bank_epilogue(tsr_ctx *ctx, const tsr_bank *bank, const float *d_raw, int64_t P,
                             const int32_t *d_utt_end, int flags, float *d_scores, int32_t *d_ids,
                             float *d_all)
{
    (void)flags;
    if (P == 0) return TSR_OK;
    hipLaunchKernelGGL(bank_epilogue_kernel, dim3((unsigned)((P + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, ctx->stream, d_raw, P, bank->M, bank->d_L, bank->d_cf,
                       d_utt_end, d_scores, d_ids, d_all);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_detect_f32(tsr_ctx *ctx, const float *d_F, int64_t T, int64_t D, const float *d_LF,
                          int64_t M, int64_t L, const float *d_cs, float *d_out, uint16_t *d_ids)
{
    (void)d_ids;
    dim3 grid((unsigned)((T + 63) / 64), (unsigned)M);
    hipLaunchKernelGGL(detect_f32_kernel, grid, dim3(64), 0, ctx->stream, d_F, (int)T, (int)D,
                       d_LF, (int)M, (int)L, d_cs, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_detect_u8_single(tsr_ctx *ctx, const uint8_t *d_F, int64_t T, int64_t D, const float *d_LF, int64_t L,
                                float *d_out)
{
    if ((D & 7) == 0) {
        int RS = (int)D;
        if (((RS >> 3) & 1) == 0) RS += 8;  // odd number of 8-byte slots: consecutive rows start in different banks
        const size_t lds = (((size_t)(64 + L - 1) * RS + 15) & ~(size_t)15) + (size_t)(L * D) * 4;
        if (lds <= 128 * 1024) {
            TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(detect_u8_single_lds_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(detect_u8_single_lds_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), lds, ctx->stream, d_F,
                               (int)T, (int)D, d_LF, (int)L, RS, d_out);
            TSR_HIP(hipGetLastError());
            return TSR_OK;
        }
    }
    hipLaunchKernelGGL(detect_u8_single_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, ctx->stream, d_F, (int)T,
                       (int)D, d_LF, (int)L, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_max_detect_examples(tsr_ctx *ctx, const uint8_t *d_F, int64_t N, int64_t T,
                                   int64_t D, const double *d_LF, int64_t L,
                                   const int64_t *d_lengths, double *d_tmp, double *d_out)
{
    (void)d_lengths;
    (void)d_out;
    dim3 grid((unsigned)((T + kThreads - 1) / kThreads), (unsigned)N);
    hipLaunchKernelGGL(max_detect_examples_kernel, grid, dim3(kThreads), 0, ctx->stream, d_F, (int)N,
                       (int)T, (int)D, d_LF, (int)L, d_tmp);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// Corpus upload: the caller's [P, D] bytes -> the resident form [P, Dp] with
// Dp = round_up(D, 16), every byte normalised to (b != 0) and the row padding zeroed.
// The reference tests `if F[...]` (compute_likelihood_linear_filter.pyx:168), so any
// non-zero byte is an edge; normalising once here lets the scoring kernels use the
// bytes directly as int8 / ubyte->float operands.
namespace {
__global__ __launch_bounds__(kThreads) void corpus_pack_kernel(const uint8_t *__restrict__ src,
                                                               int64_t P, int D, int Dp,
                                                               uint8_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;  // one dword of dst
    const int Dp4 = Dp >> 2;
    if (i >= P * Dp4) return;
    const int64_t p = i / Dp4;
    const int d0 = (int)(i - p * Dp4) * 4;
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (d0 + b < D && src[p * D + d0 + b]) v |= 1u << (8 * b);
    reinterpret_cast<uint32_t *>(dst)[i] = v;
}
}  // namespace

int tsr_launch_corpus_pack(tsr_ctx *ctx, const uint8_t *d_src, int64_t P, int D, int Dp,
                           uint8_t *d_dst)
{
    const int64_t n = P * (Dp >> 2);
    if (n == 0) return TSR_OK;
    hipLaunchKernelGGL(corpus_pack_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, ctx->stream, d_src, P, D, Dp, d_dst);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
