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

// ---- isolated-window classification (get_train_data.py:836-922 over detect_float, compute_likelihood_linear_filter.pyx:178-219)
// on the same matrix-core chains.  A window row is either a REAL frame of the binary corpus -- product w * x with x in
// {0, 1}: exact, so fma(x, w, acc) is the reference's "round the product, then add" -- or a PADDING row holding the
// background vector -- product rn(w * bgd[d]), precomputed (WB), added as fma(1, wb, acc).  One accumulator row is a PAIR of
// consecutive positions of a window (the two frame replicas of the B fragments: column (m, 1) walks the window one frame
// later through the same physical rows), a tile is 16 pairs that share their padding pattern -- physical rows
// [k, k + r) of the pair are real, all others padding (the host buckets the pairs by (k, r)) -- so that at every tap row
// the whole tile takes either its frames' bytes against W or the constant 1.0 against WB: one matrix instruction per
// K-step, all sixteen columns at work.  The A operand is gathered per lane (lane (i, k) takes byte 4 n + k of row i's
// frame: a 16-byte load per four K-steps, the sixteen rows of a tile are sixteen different frames); no planes, no LDS.
namespace {
// WB fragments of a call: the bank's W fragments times the background value of each element's feature, rounded once
__global__ __launch_bounds__(256) void iso_wb_kernel(const f32x4 *__restrict__ wm, int64_t total4, int NK4, const float *__restrict__ bgd, int D,
                                                     f32x4 *__restrict__ wb)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int lane = (int)(idx & 63), k = lane >> 4, n4 = (int)((idx >> 6) % NK4);
    const f32x4 w = wm[idx];
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int d = 16 * n4 + 4 * u + k;
        o[u] = d < D ? __fmul_rn(w[u], bgd[d]) : 0.0f;
    }
    wb[idx] = o;
}

__global__ __launch_bounds__(256) void iso_mfma_kernel(const uint8_t *__restrict__ X, int Dp, const tsr_iso_pair *__restrict__ pairs,
                                                       const int32_t *__restrict__ tile_kr, int64_t n_units, int n_groups,
                                                       const f32x4 *__restrict__ wm, const f32x4 *__restrict__ wb,
                                                       const int64_t *__restrict__ gm_off, const int32_t *__restrict__ gm_rows,
                                                       const int32_t *__restrict__ Lm, const float *__restrict__ cf, int M, int64_t NW,
                                                       float *__restrict__ out, const int *__restrict__ n_tiles_dev)
{
    const int lane = threadIdx.x & 63;
    const int64_t unit = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6));
    if (unit >= n_units) return;
    if (n_tiles_dev && unit >= (int64_t)n_tiles_dev[0] * n_groups) return;  // (tables made on the device: the grid covers their upper bound)
    const int64_t tile = unit / n_groups;
    const int g = (int)(unit - tile * n_groups);
    const int i = lane & 15, kk = lane >> 4;
    const int k0 = tile_kr[2 * tile], k1 = k0 + tile_kr[2 * tile + 1];  // physical rows [k0, k1) of every pair are real
    const int64_t f0 = pairs[tile * 16 + i].f0;
    const int NK4 = Dp >> 4;
    const int rows = gm_rows[g];
    const f32x4 *wg = wm + gm_off[g] + lane, *bg = wb + gm_off[g] + lane;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int sh = 8 * kk;
    for (int tp = 0; tp < rows; ++tp) {
        const int64_t fo = ((int64_t)tp * NK4) * 64;
        if (tp >= k0 && tp < k1) {  // (wave-uniform) real frames against W
            const uint4 *xr = reinterpret_cast<const uint4 *>(X + (f0 + tp) * (int64_t)Dp);
            const f32x4 *wp = wg + fo;
            for (int n4 = 0; n4 < NK4; ++n4) {
                const uint4 x = xr[n4];
                const f32x4 b = wp[(int64_t)n4 * 64];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((x.x >> sh) & 0xffu), b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((x.y >> sh) & 0xffu), b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((x.z >> sh) & 0xffu), b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)((x.w >> sh) & 0xffu), b[3], acc, 0, 0, 0);
            }
        } else {  // background rows: 1.0 against rn(w * bgd)
            const f32x4 *bp = bg + fo;
            for (int n4 = 0; n4 < NK4; ++n4) {
                const f32x4 b = bp[(int64_t)n4 * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b[u], acc, 0, 0, 0);
            }
        }
    }
    // lane l holds rows 4 (l / 16) + r of column j = l % 16: template j & 7 of the group, position item + (j >> 3)
    const int j = lane & 15, m = g * TSR_MG + (j & 7), rep = j >> 3;
    if (m >= M) return;
    const int L = Lm[m];
    const float c = cf[m];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const tsr_iso_pair q = pairs[tile * 16 + 4 * kk + r];
        const bool second = (q.left >> 30) & 1;
        const int left = (q.left & 0x3fffffff) - rep;
        if (q.item < 0 || (rep && !second)) continue;
        out[(int64_t)m * NW + q.item + rep] = (L <= left ? acc[r] : 0.0f) + c;
    }
}
}  // namespace

// ---- the tile tables of a call, made on the device.  A pair is two consecutive searched positions of a window; its
// bucket is its padding pattern (k, r): physical rows [k, k + r) real.  count -> scan (buckets padded to whole tiles of 16
// pairs, (k, r) of every tile) -> scatter (a block reserves its share of every bucket with one atomic) -> fillers.
namespace {
constexpr int kIsoBlock = 256;
constexpr int kIsoMaxBuckets = TSR_ISO_MAX_BUCKETS;  // (RMAX + 1)^2: templates of up to 62 taps (longer ones: the vector kernel)

struct IsoWin {  // per-window geometry as tsr_corpus_classify_isolated receives it, plus prefix sums
    const int64_t *first, *real_lo, *real_hi, *starts, *pair_start;  // [W] ... pair_start [W + 1]
    const int32_t *rows, *n_valid;
    int64_t W, NP;
    int RMAX;
};

__device__ __forceinline__ void iso_pair_of(const IsoWin &g, int64_t i, int64_t &w, int32_t &t, int &bucket)
{
    int64_t lo = 0, hi = g.W;  // last w with pair_start[w] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (g.pair_start[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    w = lo;
    t = (int32_t)(2 * (i - g.pair_start[lo]));
    const int64_t f0 = g.first[lo] + t;
    int k = (int)min(max(g.real_lo[lo] - f0, (int64_t)0), (int64_t)g.RMAX);
    const int e = (int)min(max(g.real_hi[lo] - f0, (int64_t)0), (int64_t)g.RMAX);
    const int r = max(e - k, 0);
    if (r == 0) k = 0;
    bucket = k * (g.RMAX + 1) + r;
}

__global__ __launch_bounds__(kIsoBlock) void iso_count_kernel(const IsoWin g, int NB, int *__restrict__ cnt)
{
    __shared__ int h[kIsoMaxBuckets];
    for (int b = threadIdx.x; b < NB; b += kIsoBlock) h[b] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kIsoBlock + threadIdx.x;
    if (i < g.NP) {
        int64_t w;
        int32_t t;
        int b;
        iso_pair_of(g, i, w, t, b);
        atomicAdd(&h[b], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += kIsoBlock)
        if (h[b]) atomicAdd(&cnt[b], h[b]);
}

// one block: first row of every bucket, the tile count, (k, r) of every tile (a tile's bucket: the last one that starts at or
// in front of its first row -- empty buckets share their successor's start); the fill counters start at zero
__global__ __launch_bounds__(1024) void iso_scan_kernel(int NB, int RMAX, const int *__restrict__ cnt, int *__restrict__ at, int *__restrict__ fill,
                                                        int32_t *__restrict__ tile_kr, int *__restrict__ n_tiles)
{
    __shared__ int part[1024];
    __shared__ int at_s[kIsoMaxBuckets];
    const int per = (NB + 1023) / 1024, b0 = threadIdx.x * per;
    int sum = 0;
    for (int b = b0; b < min(b0 + per, NB); ++b) sum += (cnt[b] + 15) / 16 * 16;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of the threads' sums
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int row = part[threadIdx.x] - sum;
    for (int b = b0; b < min(b0 + per, NB); ++b) {
        at[b] = row;
        at_s[b] = row;
        fill[b] = 0;
        row += (cnt[b] + 15) / 16 * 16;
    }
    const int nt = part[1023] >> 4;
    if (threadIdx.x == 1023) n_tiles[0] = nt;
    __syncthreads();
    for (int tile = threadIdx.x; tile < nt; tile += 1024) {
        int lo = 0, hi = NB;  // last b with at_s[b] <= 16 tile
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (at_s[mid] <= 16 * tile)
                lo = mid;
            else
                hi = mid;
        }
        tile_kr[2 * tile] = lo / (RMAX + 1);
        tile_kr[2 * tile + 1] = lo % (RMAX + 1);
    }
}

__global__ __launch_bounds__(kIsoBlock) void iso_scatter_kernel(const IsoWin g, int NB, const int *__restrict__ at, int *__restrict__ fill,
                                                                tsr_iso_pair *__restrict__ tiles)
{
    __shared__ int h[kIsoMaxBuckets];  // the block's pairs per bucket, then the block's first slot in the bucket
    for (int b = threadIdx.x; b < NB; b += kIsoBlock) h[b] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kIsoBlock + threadIdx.x;
    int64_t w = 0;
    int32_t t = 0;
    int b = 0, rank = 0;
    const bool on = i < g.NP;
    if (on) {
        iso_pair_of(g, i, w, t, b);
        rank = atomicAdd(&h[b], 1);
    }
    __syncthreads();
    for (int bb = threadIdx.x; bb < NB; bb += kIsoBlock)
        if (h[bb]) h[bb] = atomicAdd(&fill[bb], h[bb]);
    __syncthreads();
    if (on) {
        tsr_iso_pair q;
        q.f0 = g.first[w] + t;
        q.item = (int32_t)(g.starts[w] + t);
        q.left = (g.rows[w] - t) | (t + 1 < g.n_valid[w] ? (1 << 30) : 0);
        tiles[at[b] + h[b] + rank] = q;
    }
}

// the rows behind a bucket's pairs, up to the end of its last tile, shadow that tile's first pair (frames loaded, nothing stored)
__global__ __launch_bounds__(kIsoBlock) void iso_filler_kernel(int NB, const int *__restrict__ cnt, const int *__restrict__ at,
                                                               tsr_iso_pair *__restrict__ tiles)
{
    const int b = blockIdx.x * kIsoBlock + threadIdx.x;
    if (b >= NB) return;
    const int end = at[b] + (cnt[b] + 15) / 16 * 16;
    for (int x = at[b] + cnt[b]; x < end; ++x) {
        tsr_iso_pair q;
        q.f0 = tiles[x & ~15].f0;
        q.item = -1;
        q.left = 0;
        tiles[x] = q;
    }
}
}  // namespace

// d_geo: first | real_lo | real_hi | starts (i64 [W] each) | pair_start (i64 [W + 1]) | rows | n_valid (i32 [W] each), as
// tsr_corpus_classify_isolated uploads them; d_work: 2 NB + 1 ints + the tile (k, r) pairs; tiles: rows_bound pairs.
int tsr_launch_isolated_tables(tsr_ctx *ctx, const int64_t *d_geo, int64_t W, int64_t NP, int RMAX, int *d_work, tsr_iso_pair *d_tiles,
                               int64_t tiles_bound, const int32_t **d_tile_kr, const int **d_n_tiles)
{
    const int NB = (RMAX + 1) * (RMAX + 1);
    if (NB > kIsoMaxBuckets) return TSR_ERR_UNSUPPORTED;
    IsoWin g;
    g.first = d_geo;
    g.real_lo = d_geo + W;
    g.real_hi = d_geo + 2 * W;
    g.starts = d_geo + 3 * W;
    g.pair_start = d_geo + 4 * W;
    g.rows = reinterpret_cast<const int32_t *>(d_geo + 5 * W + 1);
    g.n_valid = g.rows + W;
    g.W = W;
    g.NP = NP;
    g.RMAX = RMAX;
    int *cnt = d_work, *at = cnt + NB, *fill = at + NB, *n_tiles = fill + NB;
    int32_t *tile_kr = n_tiles + 1;
    TSR_HIP(hipMemsetAsync(cnt, 0, (size_t)NB * 4, ctx->stream));
    const unsigned blocks = (unsigned)((NP + kIsoBlock - 1) / kIsoBlock);
    hipLaunchKernelGGL(iso_count_kernel, dim3(blocks), dim3(kIsoBlock), 0, ctx->stream, g, NB, cnt);
    hipLaunchKernelGGL(iso_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, NB, RMAX, (const int *)cnt, at, fill, tile_kr, n_tiles);
    hipLaunchKernelGGL(iso_scatter_kernel, dim3(blocks), dim3(kIsoBlock), 0, ctx->stream, g, NB, (const int *)at, fill, d_tiles);
    hipLaunchKernelGGL(iso_filler_kernel, dim3((unsigned)((NB + kIsoBlock - 1) / kIsoBlock)), dim3(kIsoBlock), 0, ctx->stream, NB, (const int *)cnt,
                       (const int *)at, d_tiles);
    TSR_HIP(hipGetLastError());
    (void)tiles_bound;
    *d_tile_kr = tile_kr;
    *d_n_tiles = n_tiles;
    return TSR_OK;
}

int tsr_launch_isolated_mfma(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *d_X, int Dp, const float *d_bgd,
                             const tsr_iso_pair *d_pairs, const int32_t *d_tile_kr, int64_t n_tiles, int64_t NW, void *d_wb_scratch,
                             float *d_out, const int *d_n_tiles)
{
    if (!bank->d_Wm || Dp != bank->wm_Dp || bank->wm_total4 <= 0) return TSR_ERR_UNSUPPORTED;
    if (n_tiles <= 0) return TSR_OK;
    const int NK4 = Dp >> 4;
    f32x4 *wb = static_cast<f32x4 *>(d_wb_scratch);
    hipLaunchKernelGGL(iso_wb_kernel, dim3((unsigned)((bank->wm_total4 + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const f32x4 *>(bank->d_Wm), bank->wm_total4, NK4, d_bgd, (int)bank->D, wb);
    const int64_t n_units = n_tiles * bank->n_groups;
    hipLaunchKernelGGL(iso_mfma_kernel, dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, ctx->stream, d_X, Dp, d_pairs, d_tile_kr, n_units,
                       (int)bank->n_groups, reinterpret_cast<const f32x4 *>(bank->d_Wm), (const f32x4 *)wb, bank->d_gm_off, bank->d_gm_rows,
                       bank->d_L, bank->d_cf, (int)bank->M, NW, d_out, d_n_tiles);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

