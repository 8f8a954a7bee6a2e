// edges.hip -- the edge front-end, spectrogram -> binary edge map:
//   _edge_map_no_threshold        edge_signal_proc.py:1697-1836
//   _edge_map_threshold_segments  edge_signal_proc.py:1537-1675  (unidirectional spread, binary)
//     threshold_edge_block        :1257-1307   quantile of the positive values, by sorting
//     _spread_edges               :1406-1418   max over shifts along the edge's orientation
//   reorg_part_for_fast_filtering get_train_data.py:41-51       -> [T, H, 8]
// as get_edge_features chains them (get_train_data.py:2547-2557, default branch).
//
// One workgroup per (utterance, block of block_length frames).  For each of the 8 orientations it
//   1. evaluates the stencil from the f64 spectrogram: the first difference along the orientation,
//      kept where it is >= both neighbouring differences and of the orientation's sign;
//   2. finds the reference's threshold -- element int(q n) of the sorted positive values -- by an
//      radix select (8-bit digits, at most 8 passes, stops when one candidate is left) on the f64 bit patterns
//      (positive doubles order like their bits), no sort;
//   3. thresholds at max(tau, abst[e]) and ORs the binary block along the orientation
//      (+-spread_length steps, zeros outside the BLOCK, like the reference's zero-padded Z);
// and finally writes the block's frames in the resident corpus layout [frame][h*8 + e] as whole rows.
// All comparisons are the reference's f64 comparisons on the same operands, so the map is identical.
#include "tsr_internal.h"

#include <stdlib.h>

#include <vector>

namespace {

constexpr int kThreads = 256;

struct EdgeArgs {
    const double *S;           // [P][H] time-major spectrogram stream
    const uint8_t *mask;       // [P][H] or null: edge strengths are zeroed where the mask is (get_train_data.py:2536-2539)
    const int64_t *utt_off;    // [N+1]
    const int32_t *item;       // [n_items][4]: utterance, first column of the window, width, first kept column
    int H, spread, Dp, max_w;
    double q;
    double abst[8];
    uint8_t *out;              // resident corpus rows [P][Dp]
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

__global__ __launch_bounds__(kThreads) void edge_block_kernel(const EdgeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H;
    const int32_t *it = a.item + (size_t)blockIdx.x * 4;
    const int u = it[0], c0 = it[1], W = it[2], keep0 = it[3];
    const int64_t o = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - o);
    const int n_el = H * W;
    double *vals = reinterpret_cast<double *>(esm);                          // [H][W]
    uint8_t *bin = reinterpret_cast<uint8_t *>(vals + (size_t)H * a.max_w);  // [H][W]
    uint8_t *tile = bin + (((size_t)H * a.max_w + 15) & ~(size_t)15);       // [W][Dp]
    int *hist = reinterpret_cast<int *>(tile + (size_t)a.max_w * a.Dp);      // [256]
    int *red = hist + 256;                                                   // [8]: scratch
    unsigned long long *sel = reinterpret_cast<unsigned long long *>(red + 8);  // [2]: prefix, k

    for (int i = tid * 16; i < W * a.Dp; i += kThreads * 16) *reinterpret_cast<uint4 *>(tile + i) = make_uint4(0, 0, 0, 0);
    auto Sv = [&](int h, int t) { return a.S[(o + t) * H + h]; };

    for (int e = 0; e < 8; ++e) {
        const int pair = e >> 1;
        const double sg = (e & 1) ? -1.0 : 1.0;
        // ---- 1. stencil ------------------------------------------------------------------------
        int npos = 0;
        for (int idx = tid; idx < n_el; idx += kThreads) {
            const int h = idx / W, c = idx - h * W, t = c0 + c;
            double v = 0.0;
            const bool tin = t >= 1 && t < T - 2, hin = h >= 1 && h < H - 2;
            double d = 0.0, lo = 0.0, hi = 0.0;
            bool ok = false;
            if (pair == 0 && tin) {  // along time
                d = Sv(h, t + 1) - Sv(h, t);
                lo = Sv(h, t) - Sv(h, t - 1);
                hi = Sv(h, t + 2) - Sv(h, t + 1);
                ok = true;
            } else if (pair == 1 && hin) {  // along frequency
                d = Sv(h + 1, t) - Sv(h, t);
                lo = Sv(h, t) - Sv(h - 1, t);
                hi = Sv(h + 2, t) - Sv(h + 1, t);
                ok = true;
            } else if (pair == 2 && hin && tin) {  // diagonal
                d = Sv(h + 1, t + 1) - Sv(h, t);
                lo = Sv(h, t) - Sv(h - 1, t - 1);
                hi = Sv(h + 2, t + 2) - Sv(h + 1, t + 1);
                ok = true;
            } else if (pair == 3 && hin && tin) {  // anti-diagonal
                d = Sv(h, t + 1) - Sv(h + 1, t);
                lo = Sv(h + 1, t) - Sv(h + 2, t - 1);
                hi = Sv(h - 1, t + 2) - Sv(h, t + 1);
                ok = true;
            }
            if (ok) {
                d *= sg;
                lo *= sg;
                hi *= sg;
                if (d >= lo && d >= hi && d >= 0.0) v = d;
            }
            if (a.mask && !a.mask[(o + t) * H + h]) v = 0.0;
            vals[idx] = v;
            npos += v > 0.0;
        }
        // ---- 2. n, k = int(q n), k-th smallest positive value ------------------------------------
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) npos += __shfl_xor(npos, off);
        if (lane == 0) red[wave] = npos;
        __syncthreads();
        const int n = red[0] + red[1] + red[2] + red[3];
        double thr = 0.0;
        if (n > 0) {
            if (tid == 0) {
                sel[0] = 0ull;
                sel[1] = (unsigned long long)(long long)(a.q * (double)n);
            }
            for (int pass = 7; pass >= 0; --pass) {
                const int shift = 8 * pass;
                hist[tid] = 0;
                __syncthreads();
                const unsigned long long prefix = sel[0];
                for (int idx = tid; idx < n_el; idx += kThreads) {
                    const double v = vals[idx];
                    if (v > 0.0) {
                        const unsigned long long key = (unsigned long long)__double_as_longlong(v);
                        if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                            atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
                    }
                }
                __syncthreads();
                if (wave == 0) {  // bin whose cumulative count passes k
                    const int kk = (int)sel[1];
                    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                    const int incl = wave_incl_scan(h0 + h1 + h2 + h3, lane);
                    const int excl = incl - (h0 + h1 + h2 + h3);
                    if (kk >= excl && kk < incl) {
                        int b = 0, cum = excl;
                        if (kk >= cum + h0) {
                            cum += h0;
                            b = 1;
                            if (kk >= cum + h1) {
                                cum += h1;
                                b = 2;
                                if (kk >= cum + h2) {
                                    cum += h2;
                                    b = 3;
                                }
                            }
                        }
                        sel[0] = prefix | ((unsigned long long)(4 * lane + b) << shift);
                        sel[1] = (unsigned long long)(kk - cum);
                    }
                }
                __syncthreads();
            }
            const double tau = __longlong_as_double((long long)sel[0]);
            thr = tau > a.abst[e] ? tau : a.abst[e];  // max(tau_quant, abst_threshold)
        }
        // ---- 3. threshold ------------------------------------------------------------------------
        for (int idx = tid; idx < n_el; idx += kThreads) {
            const double v = vals[idx];
            bin[idx] = (n > 0 && v > 0.0 && v >= thr) ? 1 : 0;
        }
        __syncthreads();
        // ---- 4. spread along the orientation inside the block, into the frame tile ----------------
        const int d0 = pair == 0 ? 0 : (pair == 3 ? -1 : 1), d1 = pair == 1 ? 0 : 1;
        for (int idx = tid; idx < n_el; idx += kThreads) {
            const int h = idx / W, c = idx - h * W;
            uint8_t r = bin[idx];
            for (int k = 1; k <= a.spread; ++k) {
                const int ha = h + k * d0, ca = c + k * d1, hb = h - k * d0, cb = c - k * d1;
                if (ha >= 0 && ha < H && ca >= 0 && ca < W) r |= bin[ha * W + ca];
                if (hb >= 0 && hb < H && cb >= 0 && cb < W) r |= bin[hb * W + cb];
            }
            tile[(size_t)c * a.Dp + h * 8 + e] = r;
        }
        __syncthreads();
    }
    // ---- the kept frames of the block, whole resident rows -------------------------------------------
    const int row16 = a.Dp >> 4;
    for (int i = tid; i < (W - keep0) * row16; i += kThreads) {
        const int c = keep0 + i / row16, s16 = i - (c - keep0) * row16;
        *reinterpret_cast<uint4 *>(a.out + (o + c0 + c) * (int64_t)a.Dp + s16 * 16) =
            *reinterpret_cast<const uint4 *>(tile + (size_t)c * a.Dp + s16 * 16);
    }
}

// The same block, one WAVE per orientation (8 waves): the block's values of an orientation stay in the
// lanes' registers (EPL per lane), the radix select keeps its histogram in a per-wave LDS strip and its
// state in wave-uniform registers, so no workgroup barrier is needed until the frame tile is complete.
// Used when H * block_length <= 64 * EPL; edge_block_kernel above serves larger blocks.
template <int EPL>
__global__ __launch_bounds__(512) void edge_block_wave_kernel(const EdgeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
    const int tid = threadIdx.x, lane = tid & 63, e = tid >> 6;  // wave = orientation
    const int H = a.H;
    const int32_t *it = a.item + (size_t)blockIdx.x * 4;
    const int u = it[0], c0 = it[1], W = it[2], keep0 = it[3];
    const int64_t o = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - o);
    const int n_el = H * W;
    const size_t hw = ((size_t)H * a.max_w + 15) & ~(size_t)15;
    uint8_t *tile = esm;                                            // [W][Dp]
    uint8_t *bin = tile + (size_t)a.max_w * a.Dp + (size_t)e * hw;  // this wave's [H][W]
    int *hist = reinterpret_cast<int *>(tile + (size_t)a.max_w * a.Dp + 8 * hw) + e * 256;
    // the block's spectrogram frames (one before, two behind: the stencil's reach) staged once with coalesced
    // loads: the eight orientation waves read every value up to six times each
    double *Sl = reinterpret_cast<double *>(tile + (size_t)a.max_w * a.Dp + 8 * hw + 8 * 256 * 4);
    const int f_lo = max(c0 - 1, 0), f_hi = min(c0 + W + 2, T);
    {
        const double *src = a.S + (o + f_lo) * H;
        for (int i = tid; i < (f_hi - f_lo) * H; i += 512) Sl[i] = src[i];
    }
    for (int i = tid * 16; i < W * a.Dp; i += 512 * 16) *reinterpret_cast<uint4 *>(tile + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    auto Sv = [&](int h, int t) { return Sl[(t - f_lo) * H + h]; };
    const int pair = e >> 1;
    const double sg = (e & 1) ? -1.0 : 1.0;
    // ---- 1. stencil into registers ------------------------------------------------------------------
    double v[EPL];
    int npos = 0;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int idx = lane + 64 * i;
        double x = 0.0;
        if (idx < n_el) {
            const int h = idx / W, c = idx - h * W, t = c0 + c;
            const bool tin = t >= 1 && t < T - 2, hin = h >= 1 && h < H - 2;
            double d = 0.0, lo = 0.0, hi = 0.0;
            bool ok = false;
            if (pair == 0 && tin) {
                d = Sv(h, t + 1) - Sv(h, t);
                lo = Sv(h, t) - Sv(h, t - 1);
                hi = Sv(h, t + 2) - Sv(h, t + 1);
                ok = true;
            } else if (pair == 1 && hin) {
                d = Sv(h + 1, t) - Sv(h, t);
                lo = Sv(h, t) - Sv(h - 1, t);
                hi = Sv(h + 2, t) - Sv(h + 1, t);
                ok = true;
            } else if (pair == 2 && hin && tin) {
                d = Sv(h + 1, t + 1) - Sv(h, t);
                lo = Sv(h, t) - Sv(h - 1, t - 1);
                hi = Sv(h + 2, t + 2) - Sv(h + 1, t + 1);
                ok = true;
            } else if (pair == 3 && hin && tin) {
                d = Sv(h, t + 1) - Sv(h + 1, t);
                lo = Sv(h + 1, t) - Sv(h + 2, t - 1);
                hi = Sv(h - 1, t + 2) - Sv(h, t + 1);
                ok = true;
            }
            if (ok) {
                d *= sg;
                lo *= sg;
                hi *= sg;
                if (d >= lo && d >= hi && d >= 0.0) x = d;
            }
            if (a.mask && !a.mask[(o + t) * H + h]) x = 0.0;
        }
        v[i] = x;
        npos += x > 0.0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) npos += __shfl_xor(npos, off);
    const int n = npos;
    // ---- 2. k-th smallest positive value: 8-pass radix select, state in wave-uniform registers ---------
    double thr = 0.0;
    if (n > 0) {
        unsigned long long prefix = 0;
        int kk = (int)(a.q * (double)n);
        bool done = false;  // (wave-uniform)
#pragma unroll
        for (int pass = 7; pass >= 0; --pass) {
            if (done) continue;
            const int shift = 8 * pass;
#pragma unroll
            for (int b = 0; b < 4; ++b) hist[4 * lane + b] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < EPL; ++i)
                if (v[i] > 0.0) {
                    const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
                    if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                        atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int incl = wave_incl_scan(h0 + h1 + h2 + h3, lane);
            const int excl = incl - (h0 + h1 + h2 + h3);
            const bool mine = kk >= excl && kk < incl;
            int b = 0, cum = excl;
            if (kk >= cum + h0) {
                cum += h0;
                b = 1;
                if (kk >= cum + h1) {
                    cum += h1;
                    b = 2;
                    if (kk >= cum + h2) {
                        cum += h2;
                        b = 3;
                    }
                }
            }
            const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine));  // exactly one lane holds k
            const int digit = __shfl(4 * lane + b, src);
            const int cnt = __shfl(b == 0 ? h0 : (b == 1 ? h1 : (b == 2 ? h2 : h3)), src);
            kk = __shfl(kk - cum, src);
            prefix |= (unsigned long long)digit << shift;
            __builtin_amdgcn_wave_barrier();  // everyone has read the histogram before it is cleared again
            if (cnt == 1 && pass > 0) {
                // one value left under this prefix: it is the k-th smallest, whatever its remaining digits are
                // (random spectrogram differences separate after three or four digits; ties run all eight passes)
                unsigned long long found = 0;
#pragma unroll
                for (int i = 0; i < EPL; ++i)
                    if (v[i] > 0.0) {
                        const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
                        if ((key >> shift) == (prefix >> shift)) found = key;
                    }
                const int owner = __builtin_ctzll(__builtin_amdgcn_ballot_w64(found != 0));
                prefix = ((unsigned long long)(unsigned)__shfl((int)(found >> 32), owner) << 32) | (unsigned)__shfl((int)found, owner);
                done = true;
            }
        }
        const double tau = __longlong_as_double((long long)prefix);
        thr = tau > a.abst[e] ? tau : a.abst[e];
    }
    // ---- 3. threshold -> this wave's binary block in LDS -----------------------------------------------
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int idx = lane + 64 * i;
        if (idx < n_el) bin[idx] = (n > 0 && v[i] > 0.0 && v[i] >= thr) ? 1 : 0;
    }
    __syncthreads();  // (also: the tile is zeroed)
    // ---- 4. spread along the orientation inside the block, into the frame tile -------------------------
    const int d0 = pair == 0 ? 0 : (pair == 3 ? -1 : 1), d1 = pair == 1 ? 0 : 1;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int idx = lane + 64 * i;
        if (idx < n_el) {
            const int h = idx / W, c = idx - h * W;
            uint8_t r = bin[idx];
            for (int k = 1; k <= a.spread; ++k) {
                const int ha = h + k * d0, ca = c + k * d1, hb = h - k * d0, cb = c - k * d1;
                if (ha >= 0 && ha < H && ca >= 0 && ca < W) r |= bin[ha * W + ca];
                if (hb >= 0 && hb < H && cb >= 0 && cb < W) r |= bin[hb * W + cb];
            }
            tile[(size_t)c * a.Dp + h * 8 + e] = r;
        }
    }
    __syncthreads();
    const int row16 = a.Dp >> 4;
    for (int i = tid; i < (W - keep0) * row16; i += 512) {
        const int c = keep0 + i / row16, s16 = i - (c - keep0) * row16;
        *reinterpret_cast<uint4 *>(a.out + (o + c0 + c) * (int64_t)a.Dp + s16 * 16) =
            *reinterpret_cast<const uint4 *>(tile + (size_t)c * a.Dp + s16 * 16);
    }
}

constexpr int kEdgeCap = 512;  // positive values of a block an orientation packs for its select (more: the register walk)

// The same block with ONE LANE PER FREQUENCY BIN (H <= 64) and one wave per orientation: the lane walks its bin through
// the block's W <= 64 frames, so that nothing is divided (the round-2 kernels spent most of their ~150 instructions per
// element on idx / W, bounds and LDS addresses: 0.445 ms for 1000 utterances, 4 % of the HBM rate, of which the radix
// select was the smaller part), the block's values of an orientation stay in the lane's registers, the thresholded block
// is ONE 64-bit mask per lane (bit c = frame c), the spread along the orientation is a handful of lane shuffles and
// shifts of that mask, and the frames are written straight from the eight orientations' masks as 8-byte groups
// [frame][h * 8 + e] -- no byte tile, no binary block in LDS.
template <int WMAX>
__global__ __launch_bounds__(256, WMAX <= 48 ? 4 : 2) void edge_block_bin_kernel(const EdgeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
    const int tid = threadIdx.x, lane = tid & 63, pair = tid >> 6;  // wave = orientation PAIR (both signs of one difference)
    const int H = a.H;
    const int32_t *it = a.item + (size_t)blockIdx.x * 4;
    const int u = it[0], c0 = it[1], W = it[2], keep0 = it[3];
    const int64_t o = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - o);
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(esm);   // [8][64]
    int *hist = reinterpret_cast<int *>(masks + 8 * 64) + pair * 256;           // this wave's [256]
    unsigned long long *comp = reinterpret_cast<unsigned long long *>(reinterpret_cast<int *>(masks + 8 * 64) + 4 * 256) + pair * kEdgeCap;
    double *Sl = reinterpret_cast<double *>(reinterpret_cast<unsigned long long *>(reinterpret_cast<int *>(masks + 8 * 64) + 4 * 256) + 4 * kEdgeCap);
    // the block's spectrogram frames (one before, two behind: the stencil's reach), staged once with coalesced loads
    const int f_lo = max(c0 - 1, 0), f_hi = min(c0 + W + 2, T);
    {
        const double *src = a.S + (o + f_lo) * H;
        for (int i = tid; i < (f_hi - f_lo) * H; i += 256) Sl[i] = src[i];
    }
    __syncthreads();
    const int h = lane;
    const bool hon = h < H, hin = h >= 1 && h < H - 2;
    // ---- 1. stencil: this lane's bin through the frames of the block.  For every orientation pair the three differences
    //      are d = S[p1] - S[p0], lo = S[p0] - S[pl], hi = S[ph] - S[p1] at fixed element offsets from S[h][t] (wave-
    //      uniform, set once), so the loop over the frames is one straight line for all orientations:
    //        along time       p0 = (0, 0)  p1 = (0, +1)  pl = (0, -1)   ph = (0, +2)     needs t in [1, T - 2)
    //        along frequency  p0 = (0, 0)  p1 = (+1, 0)  pl = (-1, 0)   ph = (+2, 0)     needs h in [1, H - 2)
    //        diagonal         p0 = (0, 0)  p1 = (+1,+1)  pl = (-1,-1)   ph = (+2,+2)     needs both
    //        anti-diagonal    p0 = (+1,0)  p1 = (0, +1)  pl = (+2,-1)   ph = (-1,+2)     needs both     ((dh, dt))
    const int o0 = pair == 3 ? 1 : 0;
    const int o1 = pair == 0 ? H : (pair == 1 ? 1 : (pair == 2 ? H + 1 : H));
    const int ol = pair == 0 ? -H : (pair == 1 ? -1 : (pair == 2 ? -H - 1 : 2 - H));
    const int oh = pair == 0 ? 2 * H : (pair == 1 ? 2 : (pair == 2 ? 2 * H + 2 : 2 * H - 1));
    const bool lane_ok = hon && (pair == 0 || hin);
    // frames of the block whose stencil reaches neither in front of the utterance nor behind it (pair 1 needs no neighbour in time)
    const int c_lo = pair == 1 ? 0 : max(1 - c0, 0), c_hi = pair == 1 ? W : min(W, T - 2 - c0);
    double v[WMAX];
    const double *sp = Sl + (c0 - f_lo) * H + min(h, H - 1);  // S[h][t] of the block's first frame (idle lanes shadow the last bin)
    const uint8_t *mk = a.mask ? a.mask + (o + c0) * H + min(h, H - 1) : nullptr;
#pragma unroll
    for (int c = 0; c < WMAX; ++c) {
        double x = 0.0;
        if (c >= c_lo && c < c_hi) {  // (wave-uniform)
            const double *q = sp + c * H;
            const double s0 = q[o0], s1 = q[o1];
            const double d = s1 - s0, lo = s0 - q[ol], hi = q[oh] - s1;
            // orientation 2 pair keeps d where it is >= both neighbours and >= 0; orientation 2 pair + 1 does the same on
            // the negated differences, i.e. keeps -d where d <= both and <= 0: one signed value serves both (they
            // exclude each other unless d == 0, which neither counts as an edge)
            bool keep = lane_ok && ((d >= lo && d >= hi && d >= 0.0) || (d <= lo && d <= hi && d <= 0.0));
            if (mk) keep = keep && mk[c * H] != 0;
            x = keep ? d : 0.0;
        }
        v[c] = x;
    }
    // ---- 2. k-th smallest positive value.  The positive values (a sixth to a quarter of the block) are first packed
    //      into a per-wave LDS strip (ballot + mbcnt: no atomics) and read back kEdgeCap / 64 to a lane; the radix select
    //      (8-bit digits on the f64 bit patterns -- positive doubles order like their bits -- most significant first,
    //      stops when one candidate is left) then walks n / 64 keys per lane and pass instead of all W frames of its bin.
    const int d0 = pair == 0 ? 0 : (pair == 3 ? -1 : 1), d1 = pair == 1 ? 0 : 1;
    const unsigned long long wmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
    for (int sgn = 0; sgn < 2; ++sgn) {  // the two orientations of the pair: the positive values, then the negated negative ones
        const int e = 2 * pair + sgn;
        if (sgn) {  // the second orientation works on the negated values (in place: the first is done with them)
#pragma unroll
            for (int c = 0; c < WMAX; ++c) v[c] = -v[c];
        }
        int n = 0;
    #pragma unroll
        for (int c = 0; c < WMAX; ++c) {
            const bool p = v[c] > 0.0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(p);
            const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if (p && pos < kEdgeCap) comp[pos] = (unsigned long long)__double_as_longlong(v[c]);
            n += __builtin_popcountll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double thr = 0.0;
        if (n > 0) {
            constexpr int KPL = kEdgeCap / 64;
            const bool packed = n <= kEdgeCap;  // (wave-uniform; denser blocks walk the registers as round 2 did)
            unsigned khi[KPL], klo[KPL];
            const int m = packed ? (n + 63) >> 6 : 0;
    #pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const unsigned long long key = (j < m && lane + 64 * j < n) ? comp[lane + 64 * j] : 0ull;  // (0: not a positive double)
                khi[j] = (unsigned)(key >> 32);
                klo[j] = (unsigned)key;
            }
            unsigned phi = 0, plo = 0;  // prefix found so far
            int kk = (int)(a.q * (double)n);
            bool done = false;  // (wave-uniform)
    #pragma unroll
            for (int pass = 7; pass >= 0; --pass) {
                if (done) continue;
                const int sh = 8 * (pass & 3);  // digit position inside its dword (passes 7..4: high dword)
    #pragma unroll
                for (int b = 0; b < 4; ++b) hist[4 * lane + b] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // does a key still match the prefix above this digit?  (pass 7: every key does)
                auto match = [&](unsigned kh, unsigned kl) -> bool {
                    if (pass >= 4) return pass == 7 || (kh >> (sh + 8)) == (phi >> (sh + 8));
                    return kh == phi && (pass == 3 || (kl >> (sh + 8)) == (plo >> (sh + 8)));
                };
                if (packed) {
    #pragma unroll
                    for (int j = 0; j < KPL; ++j)
                        if (j < m && (khi[j] | klo[j]) != 0u && match(khi[j], klo[j]))
                            atomicAdd(&hist[(int)(((pass >= 4 ? khi[j] : klo[j]) >> sh) & 255u)], 1);
                } else {
    #pragma unroll
                    for (int i = 0; i < WMAX; ++i)
                        if (v[i] > 0.0) {
                            const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
                            const unsigned kh = (unsigned)(key >> 32), kl = (unsigned)key;
                            if (match(kh, kl)) atomicAdd(&hist[(int)(((pass >= 4 ? kh : kl) >> sh) & 255u)], 1);
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const int incl = wave_incl_scan(h0 + h1 + h2 + h3, lane);
                const int excl = incl - (h0 + h1 + h2 + h3);
                const bool mine = kk >= excl && kk < incl;
                int b = 0, cum = excl;
                if (kk >= cum + h0) {
                    cum += h0;
                    b = 1;
                    if (kk >= cum + h1) {
                        cum += h1;
                        b = 2;
                        if (kk >= cum + h2) {
                            cum += h2;
                            b = 3;
                        }
                    }
                }
                const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine));  // exactly one lane holds k
                const unsigned digit = (unsigned)__shfl(4 * lane + b, src);
                const int cnt = __shfl(b == 0 ? h0 : (b == 1 ? h1 : (b == 2 ? h2 : h3)), src);
                kk = __shfl(kk - cum, src);
                if (pass >= 4)
                    phi |= digit << sh;
                else
                    plo |= digit << sh;
                __builtin_amdgcn_wave_barrier();  // everyone has read the histogram before it is cleared again
                if (cnt == 1 && pass > 0) {  // one value left under this prefix: it is the k-th smallest
                    // (a key matches "through this digit": the match of the NEXT pass, evaluated now)
                    auto through = [&](unsigned kh, unsigned kl) -> bool {
                        if (pass >= 4) return (kh >> sh) == (phi >> sh);
                        return kh == phi && (kl >> sh) == (plo >> sh);
                    };
                    unsigned fh = 0, fl = 0;
                    if (packed) {
    #pragma unroll
                        for (int j = 0; j < KPL; ++j)
                            if (j < m && (khi[j] | klo[j]) != 0u && through(khi[j], klo[j])) {
                                fh = khi[j];
                                fl = klo[j];
                            }
                    } else {
    #pragma unroll
                        for (int i = 0; i < WMAX; ++i)
                            if (v[i] > 0.0) {
                                const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
                                if (through((unsigned)(key >> 32), (unsigned)key)) {
                                    fh = (unsigned)(key >> 32);
                                    fl = (unsigned)key;
                                }
                            }
                    }
                    const int owner = __builtin_ctzll(__builtin_amdgcn_ballot_w64((fh | fl) != 0u));
                    phi = (unsigned)__shfl((int)fh, owner);
                    plo = (unsigned)__shfl((int)fl, owner);
                    done = true;
                }
            }
            const double tau = __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
            thr = tau > a.abst[e] ? tau : a.abst[e];
        }
        // ---- 3. threshold -> one mask per bin (bit c = frame c of the block) --------------------------------------
        unsigned mlo = 0, mhi = 0;
    #pragma unroll
        for (int c = 0; c < WMAX; ++c) {
            const unsigned bit = (n > 0 && v[c] > 0.0 && v[c] >= thr) ? 1u : 0u;
            if (c < 32)
                mlo |= bit << c;
            else
                mhi |= bit << (c - 32);
        }
        const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
        // ---- 4. spread along the orientation inside the block ------------------------------------------------------
        unsigned long long r = m;
        for (int k = 1; k <= a.spread; ++k) {
            const int la = h + k * d0, lb = h - k * d0;
            // (__shfl reads lane (index mod 64): out-of-range bins contribute nothing)
            unsigned long long ua = ((unsigned long long)(unsigned)__shfl((int)(m >> 32), la & 63) << 32) | (unsigned)__shfl((int)m, la & 63);
            unsigned long long ub = ((unsigned long long)(unsigned)__shfl((int)(m >> 32), lb & 63) << 32) | (unsigned)__shfl((int)m, lb & 63);
            if (la < 0 || la >= H) ua = 0;
            if (lb < 0 || lb >= H) ub = 0;
            const int sh = k * d1;
            if (sh < 64) {
                r |= ua >> sh;            // bin[h + k d0][c + k d1]
                r |= (ub << sh) & wmask;  // bin[h - k d0][c - k d1]
            }
        }
        masks[e * 64 + lane] = hon ? r : 0ull;
    }

    __syncthreads();
    // ---- 5. the kept frames, whole resident rows [frame][h * 8 + e]: lane = bin, wave w takes frames w, w + 8, ... ---
    const int groups = a.Dp >> 3;  // 8-byte groups of a row (the last one may be padding: zeros)
    if (lane < groups) {
        unsigned long long me[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) me[q] = masks[q * 64 + lane];  // (0 for bins >= H)
        for (int c = keep0 + pair; c < W; c += 4) {
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo |= (unsigned)((me[q] >> c) & 1ull) << (8 * q);
                hi |= (unsigned)((me[q + 4] >> c) & 1ull) << (8 * q);
            }
            *reinterpret_cast<uint2 *>(a.out + (o + c0 + c) * (int64_t)a.Dp + lane * 8) = make_uint2(lo, hi);
        }
    }
}

template <int WMAX>
int launch_edge_bin(tsr_ctx *ctx, const EdgeArgs &a, int64_t n_items, size_t lds)
{
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_block_bin_kernel<WMAX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(edge_block_bin_kernel<WMAX>, dim3((unsigned)n_items), dim3(256), lds, ctx->stream, a);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

template <int EPL>
int launch_edge_wave(tsr_ctx *ctx, const EdgeArgs &a, int64_t n_items, size_t lds)
{
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_block_wave_kernel<EPL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(edge_block_wave_kernel<EPL>, dim3((unsigned)n_items), dim3(512), lds, ctx->stream, a);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

}  // namespace

// (development switch: TSR_EDGEK=old at library load keeps the round-2 kernels, for A/B timing)
static const bool g_edge_kernel_old = [] {
    const char *e = getenv("TSR_EDGEK");
    return e && e[0] == 'o';
}();

int tsr_launch_edge_blocks(tsr_ctx *ctx, const double *d_S, const uint8_t *d_mask, const int64_t *d_utt_off, const int32_t *d_items,
                           int64_t n_items, int H, int block_length, int spread, double q, const double *abst,
                           int Dp, uint8_t *d_out)
{
    if (n_items <= 0) return TSR_OK;
    EdgeArgs a;
    a.S = d_S;
    a.mask = d_mask;
    a.utt_off = d_utt_off;
    a.item = d_items;
    a.H = H;
    a.spread = spread;
    a.Dp = Dp;
    a.max_w = block_length;
    a.q = q;
    for (int e = 0; e < 8; ++e) a.abst[e] = abst[e];
    a.out = d_out;
    const size_t hw = (size_t)H * block_length;
    if (H <= 64 && block_length <= 64 && (Dp >> 3) <= 64 && !g_edge_kernel_old) {  // one lane per frequency bin
        const size_t lds_b = (size_t)8 * 64 * 8 + 4 * 256 * 4 + (size_t)4 * kEdgeCap * 8 + (size_t)(block_length + 3) * H * 8;
        if (block_length <= 32) return launch_edge_bin<32>(ctx, a, n_items, lds_b);
        if (block_length <= 40) return launch_edge_bin<40>(ctx, a, n_items, lds_b);   // (the reference's default block length)
        if (block_length <= 48) return launch_edge_bin<48>(ctx, a, n_items, lds_b);
        return launch_edge_bin<64>(ctx, a, n_items, lds_b);
    }
    {  // one wave per orientation while the block fits the lanes' registers
        const size_t hw16 = (hw + 15) & ~(size_t)15;
        const size_t lds_w = (size_t)block_length * Dp + 8 * hw16 + 8 * 256 * 4 + (size_t)(block_length + 3) * H * 8;
        if (hw <= 64 * 32 && lds_w <= 64 * 1024) {
            if (hw <= 64 * 8) return launch_edge_wave<8>(ctx, a, n_items, lds_w);
            if (hw <= 64 * 16) return launch_edge_wave<16>(ctx, a, n_items, lds_w);
            return launch_edge_wave<32>(ctx, a, n_items, lds_w);
        }
    }
    const size_t lds = hw * 8 + ((hw + 15) & ~(size_t)15) + (size_t)block_length * Dp + 256 * 4 + 8 * 4 + 16 + 16;
    if (lds > 160 * 1024) return TSR_ERR_UNSUPPORTED;
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(edge_block_kernel, dim3((unsigned)n_items), dim3(kThreads), lds, ctx->stream, a);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
