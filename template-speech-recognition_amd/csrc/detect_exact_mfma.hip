// detect_exact_mfma.hip -- TSR_EXACT (the bit-identical path) on the matrix cores of gfx950.
//
// The reference adds a window's taps one at a time, in f32, in the order tau -> frequency -> part
// (compute_likelihood_linear_filter.pyx:162-175); detect_exact.hip reproduces that with one lane per window and
// one v_fmac per tap and template.  v_mfma_f32_16x16x4_f32 does the same arithmetic 1024 taps at a time:
// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j] is evaluated as the fmaf chain k = 0, 1, 2, 3 (ascending, one
// rounding per step; tools/ubench/mfma_f32_chain.hip: 0 of 256 outputs differ from fmaf over 2048-step chains),
// and consecutive MFMAs on the same accumulator continue the chain.  So with
//     row    i      = window at frame 32 T + 2 i            (16 windows at a stride of two frames)
//     column j      = (template m = j & 7, replica r = j >> 3)
//     K      step   = (tap row tau' = 0 .. L, features 4 n .. 4 n + 3)
//     A[i][k]       = X[32 T + 2 i + tau'][4 n + k]                      (0.0 or 1.0)
//     B[k][(m, r)]  = W_m[tau' - r][4 n + k]   (0 outside 0 <= tau' - r < L_m)
// column (m, r) of row i walks exactly the taps of the window at frame 32 T + 2 i + r in the reference's order
// (the zero products in front of / behind it leave the sum untouched: s + (+-0) = s), all sixteen columns of the
// instruction carry work for a group of eight templates, and one tile of accumulators covers 32 frames.
//
// Operand feed.  B fragments are precomputed per bank ([tau'][n / 4][lane] float4, one coalesced 16-byte load
// per lane and four K-steps).  A fragments never exist as data: the frames are kept as BIT PLANES
//     plane[q][n / 4][b][n % 4]  (64 bits)  bit 16 k + i  =  X[32 b + 2 i + q][4 n + k]
// (exact_planes_kernel, built once per resident corpus), i.e. for frames of one parity the lane mask of an
// aligned tile is one 64-bit word, and a tile that starts s rows into a block takes its lanes i < 16 - s from
// word b shifted down by s and the others from word b + 1 shifted up by 16 - s.  The words arrive by scalar
// loads (the blocks a wave needs for a step are contiguous), two scalar shifts and three v_cndmask_b32 (mask ?
// 1.0 : 0.0 twice, then the lane select) make the operand register; there is no LDS.  The scalar unit issues one
// instruction per SIMD every four cycles, eight per matrix instruction: the first version of this kernel,
// which merged the two words with and/andn2/or on the scalar side (8.1 scalar instructions per MFMA with its
// address arithmetic), was bound by exactly that (0.53 ms on config 2).
//
// Roofline: matrix-core bound.  P/32 tiles x (L+1) Dp/4 MFMAs x ceil(M/8) groups, 32 cycles each.
#include "tsr_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxNT = 3;       // tiles (of 32 frames) a wave carries through one pass over the taps
constexpr int kWavesPerCU = 28; // seven per SIMD, what 68 registers allow (measured on config 2: 12: 480 us, 20: 452, 28: 441)

// One wave per 64 frames (two blocks).  Lane l = 16 c + i reads row 32 (2 w + (c >> 1)) + 2 i + (c & 1), so that the
// ballot of feature k is four 16-bit fields (block half, parity) and lane c < 4 assembles word c from the four
// ballots of a dword.
__global__ __launch_bounds__(256) void exact_planes_kernel(const uint8_t *__restrict__ X, int64_t P, int Dp, int NK, int64_t nb,
                                                            unsigned long long *__restrict__ planes)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (2 * w >= nb) return;  // (whole wave)
    const int i = lane & 15, c = lane >> 4;
    const int64_t row = 32 * (2 * w + (c >> 1)) + 2 * i + (c & 1);
    const bool ok = row < P;
    const uint4 *xr = reinterpret_cast<const uint4 *>(X + row * Dp);
    const int sh = 16 * (lane & 3);
    const int q = lane & 1;
    const int64_t b = 2 * w + ((lane & 3) >> 1);
    const int NK4 = NK >> 2;
    unsigned long long *dst = planes + ((int64_t)q * NK4 * nb + b) * 4;  // + n4 * nb * 4 + u
    const bool wr = lane < 4 && b < nb;
    for (int n4 = 0; n4 < (NK >> 2); ++n4) {
        uint4 x = make_uint4(0, 0, 0, 0);
        if (ok) x = xr[n4];
        const uint32_t dw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned long long res = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long bal = __ballot((dw[u] >> (8 * k)) & 0xffu);
                res |= ((bal >> sh) & 0xffffull) << (16 * k);
            }
            if (wr) dst[(int64_t)n4 * nb * 4 + u] = res;
        }
    }
}

__device__ __forceinline__ float mask_to_operand(unsigned long long mask)
{
    // (an ordinary VALU instruction to the compiler -- v_cndmask_b32 with the scalar pair as its mask -- so that its
    // hazard recognizer keeps the distance a matrix instruction needs behind the write of its operand; the same
    // instruction issued through inline asm gave wrong sums)
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? 1.0f : 0.0f;
}

// NT tiles T0 .. T0+NT-1 of NG template groups (a pair shares the operand masks: half the scalar work per matrix
// instruction) through all taps; raw scores of the groups' templates out.
template <int NT, int NG>
__device__ __forceinline__ void exact_run(const unsigned long long *__restrict__ planes, int64_t nb, int NK,
                                          const f32x4 *__restrict__ wg0, const f32x4 *__restrict__ wg1, int Lg1, int64_t T0, int lane,
                                          float *__restrict__ raw_g, int64_t P, int Mg)
{
    const int NK4 = NK >> 2;
    f32x4 acc[NG][NT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // One step = four K-steps of every tile: a float4 of B fragments per group and four words of each of the NT + 1
    // blocks.  Two operand sets alternate: the loads of step it + 1 are issued in front of the matrix work of step it
    // (the scalar and vector loads have a whole step, 4 NT NG matrix instructions, to arrive).
    struct Step {
        unsigned long long Q[NT + 1][4];
        f32x4 bf[NG];
    };
    const f32x4 *wp0 = wg0 + lane, *wp1 = wg1 + lane;
    const int64_t step_words = nb * 4;  // from (q, n4) to (q, n4 + 1)
    int ltp = 0, ln4 = 0;               // load cursor
    const unsigned long long *lst = planes + T0 * 4;
    auto load = [&](Step &o) {
        o.bf[0] = *wp0;
        wp0 += 64;
        if (NG == 2) {
            o.bf[NG - 1] = *wp1;
            wp1 += 64;
        }
#pragma unroll
        for (int j = 0; j <= NT; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) o.Q[j][u] = lst[4 * j + u];
        lst += step_words;
        if (++ln4 == NK4) {  // next tap row
            ln4 = 0;
            ++ltp;
            lst = planes + ((int64_t)(ltp & 1) * NK4 * nb + T0 + (ltp >> 5)) * 4;
        }
    };
    int ctp = 0, cn4 = 0;  // compute cursor
    int sdn = 0, sup = 16;  // shifts of the current tap row: s = (tau' >> 1) & 15 and 16 - s
    unsigned long long lo = 0xffffffffffffffffull;  // lanes i < 16 - s of every 16-lane field take the lower block's word
    auto compute = [&](const Step &o) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float x = mask_to_operand(((o.Q[t][u] >> sdn) & lo) | ((o.Q[t + 1][u] << sup) & ~lo));
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, o.bf[g][u], acc[g][t], 0, 0, 0);
            }
        if (++cn4 == NK4) {
            cn4 = 0;
            ++ctp;
            sdn = (ctp >> 1) & 15;
            sup = 16 - sdn;
            lo = 0x0001000100010001ull * (unsigned long long)(0xffffu >> sdn);
        }
    };
    const int total = Lg1 * NK4;
    Step A, B;
    load(A);
    int it = 0;
    for (; it + 2 <= total; it += 2) {
        load(B);
        compute(A);
        load(A);  // (the step behind the last one is read and dropped: the planes and the fragments are padded for it)
        compute(B);
    }
    if (it < total) compute(A);
    // lane l holds rows i = 4 (l / 16) + r of column j = l % 16: template j & 7 at frame 32 T + 2 i + (j >> 3)
    const int j = lane & 15;
    const int m = j & 7;
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (g * TSR_MG + m < Mg) {
            float *dst = raw_g + (int64_t)(g * TSR_MG + m) * P;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t p = 32 * (T0 + t) + 2 * (4 * (lane >> 4) + r) + (j >> 3);
                    if (p < P) dst[p] = acc[g][t][r];
                }
        }
}

// Units u = pair * nTiles + T (pair = two consecutive template groups, the last one may be single) are dealt to the
// waves in contiguous, equally long ranges; a wave walks its range in runs of up to kMaxNT tiles of one pair.
template <bool PAIRS>  // (two instantiations: the pair form needs 100 registers, the single-group form 68 = one more wave per SIMD)
__global__ __launch_bounds__(256) void exact_mfma_kernel(const unsigned long long *__restrict__ planes, int64_t nb, int NK,
                                                         const f32x4 *__restrict__ wmf, const int64_t *__restrict__ gm_off,
                                                         const int32_t *__restrict__ gm_rows, int n_groups, int64_t nTiles, int64_t U,
                                                         int64_t nW, int M, int64_t P, float *__restrict__ raw)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6));
    if (w >= nW) return;
    int64_t u = (U * w) / nW;
    const int64_t hi = (U * (w + 1)) / nW;
    while (u < hi) {
        const int64_t pair = u / nTiles, T0 = u - pair * nTiles;
        const int g = PAIRS ? 2 * (int)pair : (int)pair;
        const int nt = (int)min((int64_t)kMaxNT, min(hi - u, nTiles - T0));
        const bool two = PAIRS && g + 1 < n_groups;
        const f32x4 *wg0 = wmf + gm_off[g], *wg1 = wmf + gm_off[two ? g + 1 : g];
        const int rows = gm_rows[g];  // (both groups of a pair are laid out with the same number of tap rows)
        float *raw_g = raw + (int64_t)g * TSR_MG * P;
        const int Mg = M - g * TSR_MG;  // templates from this pair's first one on
        if (PAIRS && two) {
            if (nt == 3)
                exact_run<3, 2>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
            else if (nt == 2)
                exact_run<2, 2>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
            else
                exact_run<1, 2>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
        } else {
            if (nt == 3)
                exact_run<3, 1>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
            else if (nt == 2)
                exact_run<2, 1>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
            else
                exact_run<1, 1>(planes, nb, NK, wg0, wg1, rows, T0, lane, raw_g, P, Mg);
        }
        u += nt;
    }
}

}  // namespace

// Bit planes of a resident corpus with room for tap rows up to Lmax (rebuilt larger when a longer bank comes).
static int ensure_planes(tsr_ctx *ctx, tsr_corpus *cp, int Lmax)
{
    const int64_t nTiles = (cp->P + 31) / 32;
    const int64_t nb = nTiles + ((Lmax + 1) >> 5) + kMaxNT + 2;
    if (cp->d_planes && cp->planes_nb >= nb) return TSR_OK;
    if (cp->d_planes) {
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(cp->d_planes);
        cp->d_planes = nullptr;
    }
    const int64_t nb2 = nb + 8;  // (a little headroom for the next, slightly longer bank)
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));  // (the planes are ballots of the byte rows)
    const int NK = cp->Dp / 4;
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&cp->d_planes), (size_t)2 * nb2 * NK * 8));
    cp->planes_nb = nb2;
    const int64_t waves = (nb2 + 1) / 2;
    hipLaunchKernelGGL(exact_planes_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream, cp->d_feats, cp->P, cp->Dp, NK,
                       nb2, reinterpret_cast<unsigned long long *>(cp->d_planes));
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// TSR_ERR_UNSUPPORTED: the bank has no matrix-core form (weights that are not finite or so small that partial sums
// could be subnormal); the caller then takes the vector kernel.
int tsr_launch_exact_bank_mfma(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, float *d_raw)
{
    if (!bank->d_Wm || cp->Dp != bank->wm_Dp) return TSR_ERR_UNSUPPORTED;
    if (cp->P == 0) return TSR_OK;
    TSR_TRY(ensure_planes(ctx, cp, bank->Lmax));
    const int64_t nTiles = (cp->P + 31) / 32;
    const bool pairs = bank->n_groups > 1;
    const int64_t U = nTiles * (pairs ? (bank->n_groups + 1) / 2 : 1);
    int wpc = pairs ? 16 : kWavesPerCU;  // (the pair form fits four waves per SIMD)
    const int64_t slots = (int64_t)wpc * ctx->cus;
    // as many waves as the chip is to hold, but none with less than one tile
    const int64_t nW = std::max<int64_t>(1, std::min<int64_t>(slots, U));
    const unsigned grid = (unsigned)((nW + 3) / 4);
    if (pairs)
        hipLaunchKernelGGL(exact_mfma_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long *>(cp->d_planes),
                           cp->planes_nb, cp->Dp / 4, reinterpret_cast<const f32x4 *>(bank->d_Wm), bank->d_gm_off, bank->d_gm_rows,
                           bank->n_groups, nTiles, U, nW, bank->M, cp->P, d_raw);
    else
        hipLaunchKernelGGL(exact_mfma_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long *>(cp->d_planes),
                           cp->planes_nb, cp->Dp / 4, reinterpret_cast<const f32x4 *>(bank->d_Wm), bank->d_gm_off, bank->d_gm_rows,
                           bank->n_groups, nTiles, U, nW, bank->M, cp->P, d_raw);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
