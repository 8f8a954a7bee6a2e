// parts_mfma.hip -- the patch coder of the fused front-end on the matrix cores.
//
// What the reference computes (code_parts.pyx:70-135 / :247-295, then np.argmax at
// get_train_data.py:2621 / :184): for every patch location and part k the f64 sum, in the
// order i -> j -> z, of log_parts[k] where the edge is set and log_invparts[k] where it is
// not; the chain keeps only the INDEX of the first maximum.  The VALU kernel in parts.hip
// reproduces those sums bit for bit, one LDS read + one f64 add per (location, part, cell):
// 12 ms for BASELINE config 3.  Here the index is found without forming the f64 sums:
//
//   1. S_k = sum_q lip[k,q] + sum_q x_q (lp - lip)[k,q].  The differences are quantised to
//      16-bit fixed point (two int8 limbs, scale s) and the base sums to integers, so that
//      I_k = B_k + sum_q x_q dq[k,q] is an exact integer computed by v_mfma_i32_32x32x32_i8
//      with |I_k/s - S_k| <= (0.5*count + 0.75)/s, count = number of set edges in the patch
//      (the +0.75 covers the rounding of B_k and of the reference's own f64 sum).
//   2. A location whose best and second-best I differ by more than count + 2 has a unique
//      maximiser that no rounding can overturn: its index is final.  Every other location
//      (exact ties between duplicate parts included) goes, with the bit mask of the parts
//      inside the window, onto a list that `code_parts_recheck_kernel` re-evaluates with the
//      reference's sequential f64 sums -- so the part ids are the reference's in all cases.
//   3. The edge count comes out of the same GEMM (a row of ones); code_parts_fast's slot 0
//      (sum_k (S_k - 1), code_parts.pyx:289-293) is bounded from the same integers and sends
//      the location to the list in the (for log-probability tables impossible) case that it
//      could win.
//
// GEMM shape.  The resident corpus is a byte stream with row stride Dp (a multiple of 16) and
// E = 8 edge bytes per frequency bin, i.e. a stream of 8-byte "positions", S = Dp/8 per frame.
// For the patch at position p, cell (i, j, z) is byte (p + i*S + j)*8 + z: the same offset for
// every location, so 32 locations form the N dimension of one MFMA (B operand: lane n reads
// 16 bytes = 2 bins x 8 edges of its own location's patch), the K dimension runs over PAIRS OF
// PATCH BINS (bins 2g and 2g+1 of the P0*P1 in row-major order, wherever they lie: the two
// 8-byte halves of a lane's operand are separate LDS reads) and the M dimension over the parts
// (A operand: weights, held in registers for the life of the workgroup).  A tile is 32
// consecutive LIVE locations of the chunk (bins beyond X1 - P1 are never computed).  The result
// layout puts the parts of one location into the registers of one lane (and its partner lane
// + 32), so the argmax is per-lane VALU work on a packed key (I_k << 6 | k): the staged bytes
// are 0 / 64 instead of 0 / 1 and the accumulator of the low limb starts at (B_k << 6 | k), so
// that key = (high limb sum << 8) + low limb sum -- one instruction -- then v_med3 + v_max.
//
// Schedule.  A wave alternates between the two M-tiles (32 part slots each) with two
// accumulator pairs: while the matrix instructions of one half-tile issue, the keys of the
// previous half-tile are reduced (and, behind the second M-tile, the location is finished),
// dealt out between the K-block steps -- the vector work rides in the shadow of the wave's own
// matrix instructions instead of alternating with them.
//
// Frames are staged through LDS in chunks of CH frames (+ P0 - 1 rows of overlap), two buffers: a workgroup (2 per CU)
// owns an even share of the corpus' frames and copies its next chunk while it codes the current one.
#include "tsr_internal.h"

#include <cmath>
#include <vector>

namespace {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kWG = 256;  // 4 waves
constexpr int kWaves = kWG / 64;
constexpr int kPad = 256;             // zeroed bytes behind the staged rows (over-read of the last bin pairs)
constexpr int kVInv = -(1 << 25) + 1024;  // value of the empty part slots: below every real score
constexpr int kVMax = 1 << 24;            // |I_k| bound of the real slots
constexpr int kOnesKey = kVInv * 64 + 63; // start value of slot 63, the row of ones (edge count)

struct PmArgs {
    const uint8_t *X;  // resident corpus [P][Dp]
    int64_t P;
    int Dp, S, N1, P0, P1, CH, nchunks;
    const int32_t *utt_end;
    const i32x4 *Wq;    // [mt 2][kb NKB][limb 2][lane 64]
    const int32_t *Kc;  // [half 2][mt 2][r 16]: (B_m << 6) | m
    int K;
    unsigned threshold;
    float o0_const, o0_slack;  // code_parts_fast: slot 0 in units (see host prep)
    int n_inv;                 // empty part slots (incl. the row of ones)
    int *work;                 // [1] list length ([0] unused since the workgroups own fixed frame ranges)
    int32_t *argmax;           // [P][N1]
    int ids8;                  // the map is written as bytes (K <= 63: the dilation kernel of the corpus chain reads them so)
    uint32_t *list;            // [n][2]: location, full (slot 0 of code_parts_fast in play)
};

// NOTE: nothing that reads an accumulator of a matrix instruction may be inline asm -- the compiler's hazard
// recogniser does not count the wait states between a v_mfma and an asm statement that reads its result (a
// one-K-block build read the registers two cycles after the issue and got the old values); plain C++ is covered.
__device__ __forceinline__ int med3(int a, int b, int c)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

#ifndef TSR_PM_ABL
#define TSR_PM_ABL 0   // development ablations (never set in the shipped build): 1 keys, 2 MFMAs, 4 operand refetch, 8 staging, 16 finish
#endif
// FAST: code_parts_fast's slot 0 (the sum of all part scores) is tracked as well.
template <int NKB, bool FAST>
__global__ __launch_bounds__(kWG, 2) void code_parts_mfma_kernel(const PmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, half = lane >> 5;
    const int row_bytes = (a.CH + a.P0 - 1) * a.Dp;
    const int buf_bytes = row_bytes + kPad;                                // one staged chunk (two buffers: the next chunk is copied while this one is coded)
    int32_t *kcl = reinterpret_cast<int32_t *>(smem + 2 * buf_bytes);      // [2][2][16]
    int32_t *livel = kcl + 64;                                             // [2][CH]: the frame takes patches

    // weights: register-stationary for the life of the workgroup
    i32x4 W[2][NKB][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int l = 0; l < 2; ++l) W[mt][kb][l] = a.Wq[((mt * NKB + kb) * 2 + l) * 64 + lane];
    int boff[NKB][2];  // byte offsets of this lane's two patch bins of K-block kb (padding bins: weights are 0)
    const int nbins = a.P0 * a.P1;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 2 * (2 * kb + half) + h;
            boff[kb][h] = q < nbins ? ((q / a.P1) * a.S + q % a.P1) * 8 : 0;
        }
    if (tid < 64) kcl[tid] = a.Kc[tid];
    const i32x16 *const kc0 = reinterpret_cast<const i32x16 *>(kcl + (half * 2 + 0) * 16);
    const i32x16 *const kc1 = reinterpret_cast<const i32x16 *>(kcl + (half * 2 + 1) * 16);

    const int64_t body = a.P * (int64_t)a.Dp;
    const int dS = a.S - a.N1;                                     // position of location ll in frame fl: ll + fl * dS
    const int adv_fl = (32 * kWaves) / a.N1, adv_j = (32 * kWaves) % a.N1;  // a wave's next tile is 32 * kWaves locations on
    const bool wide = a.K > 56;  // part slots 56..62 in use (the keys of registers 12..15 of M-tile 1 count)

    // A workgroup owns a contiguous range of frames (P split evenly over the grid: two workgroups on every CU run the
    // same number of tiles -- chunks drawn from a counter left the last ones to a few workgroups, 13 % of the kernel)
    // and walks it in chunks of CH frames: while chunk j is coded out of one LDS buffer, chunk j + 1 is copied into
    // the other, one 16-byte unit per thread and tile (the load is issued in front of a tile's matrix instructions and
    // committed -- every byte x 64 -- behind them).
    const int copy_units = buf_bytes >> 4;
    const int64_t f_begin = a.P * (int64_t)blockIdx.x / (int64_t)gridDim.x, f_end = a.P * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x;
    if (f_begin >= f_end) return;  // uniform
    auto valid_bytes = [&](int64_t f_first) -> int {  // bytes of a chunk's rows that lie inside the corpus (zeros behind them)
        const int64_t left = body - f_first * a.Dp;
        return (int)(left < 0 ? 0 : (left < row_bytes ? left : row_bytes));
    };
    {  // the workgroup's first chunk: staged at once
        const uint8_t *src = a.X + f_begin * a.Dp;
        const int lim = valid_bytes(f_begin);
        for (int u = tid; u < copy_units; u += kWG) {
            i32x4 v = {0, 0, 0, 0};
            if (u * 16 + 16 <= lim) v = *reinterpret_cast<const i32x4 *>(src + u * 16);
            *reinterpret_cast<i32x4 *>(smem + u * 16) = v << 6;  // bytes are 0 / 1: no carry between them
        }
        // a frame takes patches when the whole patch lies inside its utterance
        if (tid < a.CH) livel[tid] = (f_begin + tid < a.P && f_begin + tid + a.P0 <= (int64_t)a.utt_end[min(f_begin + tid, a.P - 1)]) ? 1 : 0;
    }
    __syncthreads();  // (kcl visible as well)
    for (int64_t f0 = f_begin, cur = 0; f0 < f_end; f0 += a.CH, cur ^= 1) {
        const int64_t fn = f0 + a.CH;
        const bool has_next = fn < f_end && !(TSR_PM_ABL & 8);
        const int nloc = (int)min((int64_t)a.CH, f_end - f0) * a.N1;  // live locations of the chunk, frame-major: location ll = frame * N1 + bin
        const int ntiles = (nloc + 31) / 32;
        uint8_t *const rows = smem + cur * buf_bytes, *const rows_n = smem + (cur ^ 1) * buf_bytes;
        const int32_t *const livec = livel + cur * a.CH;
        const uint8_t *const src_n = a.X + fn * a.Dp;
        const int lim_n = has_next ? valid_bytes(fn) : 0;
        int uend_n = 0;  // utt_end of the next chunk's frame tid: asked for now, used behind the tiles
        if (has_next && tid < a.CH && fn + tid < a.P) uend_n = a.utt_end[fn + tid];
        int cu = has_next ? tid : copy_units;  // this thread's next unit of the copy
        auto copy_load = [&]() -> i32x4 {
            i32x4 v = {0, 0, 0, 0};
            if (cu < copy_units && cu * 16 + 16 <= lim_n) v = *reinterpret_cast<const i32x4 *>(src_n + cu * 16);
            return v;
        };
        auto copy_commit = [&](const i32x4 &v) {
            if (cu < copy_units) *reinterpret_cast<i32x4 *>(rows_n + cu * 16) = v << 6;
            cu += kWG;
        };
        const unsigned loc0 = (unsigned)(f0 * a.N1);  // (P * N1 < 2^31: checked at launch)

        int best = INT32_MIN, second = INT32_MIN;
        float vsum = 0.0f;
        auto slot = [&](const i32x16 (&acc)[2], int i) {
            if ((TSR_PM_ABL & 1) && i != 3) return;
            const int key = (int)(((unsigned)acc[1][i] << 8) + (unsigned)acc[0][i]);  // one v_lshl_add_u32
            second = med3(best, second, key);
            best = max(best, key);
            if (FAST) vsum += (float)(key >> 6);
        };
        // everything behind the second M-tile of the tile at locations ll (frame fl): margin test, list or id.
        // v_permlane32_swap(x, x) = {x of lane & 31, x of lane | 32} in every lane.
        auto finish = [&](int ll, int live_fl, int cnt_own) {  // live_fl: livel[] of the location's frame, fetched a half-tile ago
            if (TSR_PM_ABL & 16) return;
            const auto bs = __builtin_amdgcn_permlane32_swap(best, best, false, false);
            const auto ss = __builtin_amdgcn_permlane32_swap(second, second, false, false);
            const auto cs = __builtin_amdgcn_permlane32_swap(cnt_own, cnt_own, false, false);
            const unsigned count = cs[1];  // the row of ones is slot 63: half 1, M-tile 1, register 15
            const int mbest = max((int)bs[0], (int)bs[1]);
            const int msecond = max(min((int)bs[0], (int)bs[1]), max((int)ss[0], (int)ss[1]));
            const int vbest = mbest >> 6;
            const int window = (int)count + 2;
            const bool live = ll < nloc && live_fl != 0;
            const bool coded = count >= a.threshold;
            bool flag = live && coded && (vbest - (msecond >> 6) <= window);
            bool full = false;
            if (FAST) {
                const auto vs2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, vsum), __builtin_bit_cast(unsigned, vsum), false, false);
                const float vs = __builtin_bit_cast(float, (unsigned)vs2[0]) + __builtin_bit_cast(float, (unsigned)vs2[1]);
                // slot 0 = sum_k (S_k - 1) in units, pushed up by its error bound
                const float o0 = vs - (float)a.n_inv * (float)kVInv - (float)count + a.o0_const +
                                 (float)a.K * (0.5f * (float)count + 1.0f) + a.o0_slack;
                full = live && coded && o0 >= (float)(vbest - window);
                flag = flag || full;
            }
            const unsigned loc = loc0 + (unsigned)ll;
            if (flag && half == 0) {  // rare: hand the location to the f64 re-evaluation
                const int sl = atomicAdd(a.work + 1, 1);
                a.list[(size_t)sl * 2] = loc;
                a.list[(size_t)sl * 2 + 1] = full ? 1u : 0u;
            }
            if (live && half == 0 && !flag) {
                const int id = coded ? (mbest & 63) + 1 : 0;
                if (a.ids8)
                    reinterpret_cast<uint8_t *>(a.argmax)[loc] = (uint8_t)id;
                else
                    a.argmax[loc] = id;
            }
        };

        // Half-tile pipeline: step A = M-tile 0 of a tile into accA (while the keys of the previous tile's M-tile 1
        // are reduced out of accB and that tile is finished), step B = M-tile 1 into accB (while accA's keys are
        // reduced and the operands of the wave's next tile are fetched).  Each low-limb accumulator is re-armed with
        // its start keys straight from LDS as soon as its last key has been taken.
        constexpr int SL = NKB > 3 ? NKB - 2 : NKB;  // step A: K-block steps that carry keys (the finish follows them)
        constexpr int SB = NKB > 1 ? NKB - 1 : NKB;  // step B: K-block steps that carry keys
        i32x16 accA[2], accB[2];
        i32x4 bf[NKB];
        int tile = wave;
        int ll = tile * 32 + n, fl, j;  // this lane's location of the tile: frame fl, bin j of the chunk
        {
            fl = (int)((float)ll * (1.0f / (float)a.N1));
            j = ll - fl * a.N1;
            if (j < 0) {
                --fl;
                j += a.N1;
            } else if (j >= a.N1) {
                ++fl;
                j -= a.N1;
            }
        }
        auto patch_addr = [&]() -> const uint8_t * { return rows + (ll < nloc ? (ll + fl * dS) * 8 : 0); };
        if (tile < ntiles) {
            const uint8_t *xa = patch_addr();
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const i32x2 b0 = *reinterpret_cast<const i32x2 *>(xa + boff[kb][0]);
                const i32x2 b1 = *reinterpret_cast<const i32x2 *>(xa + boff[kb][1]);
                bf[kb] = (i32x4){b0.x, b0.y, b1.x, b1.y};
            }
            accA[0] = *kc0;
            accB[0] = *kc1;
        }
        bool have_prev = false;
        int ll_fin = 0, live_fin = 0;  // the tile whose M-tile 1 is in accB: its locations, the live flag of their frames
        for (; tile < ntiles; tile += kWaves) {
            const i32x4 cv = copy_load();  // (a unit of the next chunk rides through this tile)
            // ---- step A
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (TSR_PM_ABL & 2) {
                    asm volatile("" : "+v"(accA[1]), "+v"(accA[0]) : "v"(bf[kb]), "v"(W[0][kb][0]), "v"(W[0][kb][1]));
                } else {
                accA[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[0][kb][1], bf[kb], kb ? accA[1] : (i32x16){}, 0, 0, 0);
                accA[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[0][kb][0], bf[kb], accA[0], 0, 0, 0);
                }
                if (have_prev) {
#pragma unroll
                    for (int e = 0; e < 12; ++e)
                        if (e * SL / 16 == kb) slot(accB, e);
                    if (wide) {
#pragma unroll
                        for (int e = 12; e < 16; ++e)
                            if (e * SL / 16 == kb) slot(accB, e);
                    }
                    if (kb == (SL < NKB ? SL : NKB - 1)) {
                        const int cnt_own = (accB[0][15] - kOnesKey) >> 6;
                        accB[0] = *kc1;
                        finish(ll_fin, live_fin, cnt_own);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- step B
            best = INT32_MIN;
            second = INT32_MIN;
            vsum = 0.0f;
            ll_fin = ll;
            live_fin = livec[ll < nloc ? fl : 0];  // (lands under this step's matrix instructions)
            if (tile + kWaves < ntiles) {  // (the last tile fetches itself again)
                ll += 32 * kWaves;
                fl += adv_fl;
                j += adv_j;
                if (j >= a.N1) {
                    j -= a.N1;
                    ++fl;
                }
            }
            const uint8_t *xn = patch_addr();
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (TSR_PM_ABL & 2) {
                    asm volatile("" : "+v"(accB[1]), "+v"(accB[0]) : "v"(bf[kb]), "v"(W[1][kb][0]), "v"(W[1][kb][1]));
                } else {
                accB[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[1][kb][1], bf[kb], kb ? accB[1] : (i32x16){}, 0, 0, 0);
                accB[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[1][kb][0], bf[kb], accB[0], 0, 0, 0);
                }
                if (!(TSR_PM_ABL & 4))
                {  // this K-block's operand of the wave's next tile
                    const i32x2 b0 = *reinterpret_cast<const i32x2 *>(xn + boff[kb][0]);
                    const i32x2 b1 = *reinterpret_cast<const i32x2 *>(xn + boff[kb][1]);
                    bf[kb] = (i32x4){b0.x, b0.y, b1.x, b1.y};
                }
                // (the keys leave the last K-block step free: the accumulator's start keys, read from LDS behind its last key,
                //  then have a whole step to arrive before M-tile 0 of the next tile accumulates onto them)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (e * SB / 16 == kb) slot(accA, e);
                if (kb == SB - 1) accA[0] = *kc0;
                __builtin_amdgcn_sched_barrier(0);
            }
            copy_commit(cv);
            have_prev = true;
        }
        if (have_prev) {  // the wave's last tile: nothing left to overlap
#pragma unroll
            for (int e = 0; e < 16; ++e) slot(accB, e);
            finish(ll_fin, live_fin, (accB[0][15] - kOnesKey) >> 6);
        }
        while (cu < copy_units) copy_commit(copy_load());  // what the tiles did not carry (few tiles per wave, or none)
        if (has_next && tid < a.CH) livel[(cur ^ 1) * a.CH + tid] = (fn + tid < a.P && fn + tid + a.P0 <= (int64_t)uend_n) ? 1 : 0;
        __syncthreads();  // this buffer is done with, the other one complete
    }
}

// The reference's sequential f64 sums for the listed locations.  One wave takes kRcEnt entries
// at a time, one lane per part: the lane reads both table values of a cell once ([q][k][2], a
// coalesced 16 bytes per lane) and adds the one each entry's edge bit selects to that entry's
// accumulator, in the reference's cell order (i -> j -> z) -- every sum is the reference's bit
// for bit, and the table is streamed once per kRcEnt entries instead of once per entry.  The
// first maximum over the lanes is np.argmax's; for `full` entries slot 0 of code_parts_fast,
// sum_k (S_k - 1) in ascending k (code_parts.pyx:289-293), decides between it and index 0.
constexpr int kRcWaves = 1;   // (one-wave blocks: the few thousand groups spread evenly over the SIMDs)
constexpr int kRcEnt = 4;
constexpr int kRcBins = 32;  // P0 * P1 <= 2 * (16 K-groups)

__global__ __launch_bounds__(64 * kRcWaves) void code_parts_recheck_kernel(
    const uint8_t *__restrict__ X, int Dp, int P0, int P1, int N1, const double *__restrict__ tab /*[Q][K][2]*/,
    int K, int fast, const uint32_t *__restrict__ list, const int *__restrict__ work, int32_t *__restrict__ argmax, int ids8)
{
    static_assert(kRcEnt == 4 || kRcEnt == 8, "the bin words below hold 4 entries each");
    __shared__ __attribute__((aligned(8))) uint32_t xbits[kRcWaves][kRcBins][kRcEnt / 4];  // one BYTE = the 8 edge bits of one bin of one entry's patch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = work[1];
    const int nbins = P0 * P1;
    const int kl = min(lane, K - 1);  // idle lanes shadow the last part (their sums are ignored)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *t2 = reinterpret_cast<const d2 *>(tab) + kl;
    for (int e0 = (blockIdx.x * kRcWaves + wave) * kRcEnt; e0 < n; e0 += gridDim.x * kRcWaves * kRcEnt) {
        {  // E == 8: one bin = one aligned 8-byte word of 0/1 bytes.  Lane = (entry, bin group): its location is read once,
           // its bins are independent loads -- two round trips for the whole group
            constexpr int kGroups = 64 / kRcEnt, kPer = kRcBins / kGroups;
            const int e = lane % kRcEnt, g = lane / kRcEnt;
            const uint32_t loc = list[(size_t)min(e0 + e, n - 1) * 2];
            const int i0 = (int)(loc / (unsigned)N1), j0 = (int)(loc - (unsigned)i0 * (unsigned)N1);
            uint64_t v[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int bq = min(g + kGroups * q, nbins - 1);
                const int i = bq / P1, j = bq - i * P1;
                v[q] = *reinterpret_cast<const uint64_t *>(X + (int64_t)(i0 + i) * Dp + (j0 + j) * 8);
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q)
                if (g + kGroups * q < nbins)
                    reinterpret_cast<uint8_t *>(&xbits[wave][g + kGroups * q][0])[e] = (uint8_t)((v[q] * 0x0102040810204080ull) >> 56);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double acc[kRcEnt];
#pragma unroll
        for (int e = 0; e < kRcEnt; ++e) acc[e] = 0.0;
        // the table values of bin b + 1 are in flight while bin b is summed (a bin's 8 loads per lane are one round
        // trip to L2: unbuffered, the 25 bins of a 5 x 5 patch are 25 round trips in sequence -- all of the kernel's time)
        d2 tv[2][8];
#pragma unroll
        for (int z = 0; z < 8; ++z) tv[0][z] = t2[(int64_t)z * K];
        // The edge bits are the same for every lane (a lane is a part): they go to scalar registers, a bit becomes a scalar
        // mask 0 / ~0 (s_bfe_i32) and the select is two v_bfi_b32 on it -- three vector instructions per cell and entry
        // with the add, and no select waits for another one's condition register (the compiler's own forms: two
        // v_cndmask behind one vcc per select, every select waiting for the previous one's; or v_and + v_and_or).
        // (NB: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 with this compiler; hence the scalar copies.)
        auto sum_bin = [&](int b, const d2 (&t)[8]) {
            unsigned bw[kRcEnt / 4];  // the bin's bits of the entries
#pragma unroll
            for (int i = 0; i < kRcEnt / 4; ++i) bw[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)xbits[wave][b][i]);
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const double tx = t[z].x, ty = t[z].y;
                const int xl = __double2loint(tx), xh = __double2hiint(tx), yl = __double2loint(ty), yh = __double2hiint(ty);
#pragma unroll
                for (int e = 0; e < kRcEnt; ++e) {
                    const int m = (int)(bw[e >> 2] << (31 - (8 * (e & 3) + z))) >> 31;  // 0 or ~0
                    int vl, vh;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(vl) : "s"(m), "v"(yl), "v"(xl));
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(vh) : "s"(m), "v"(yh), "v"(xh));
                    acc[e] += __hiloint2double(vh, vl);
                }
            }
        };
        int b = 0;
        for (; b + 2 <= nbins; b += 2) {  // bins in pairs, nothing conditional inside (a load the compiler can sink into a branch is a round trip again)
            const int b2 = min(b + 2, nbins - 1);
#pragma unroll
            for (int z = 0; z < 8; ++z) tv[1][z] = t2[(int64_t)((b + 1) * 8 + z) * K];
            __builtin_amdgcn_sched_barrier(0);  // (the loads stay in front of the sums they overlap)
            sum_bin(b, tv[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int z = 0; z < 8; ++z) tv[0][z] = t2[(int64_t)(b2 * 8 + z) * K];
            __builtin_amdgcn_sched_barrier(0);
            sum_bin(b + 1, tv[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (b < nbins) sum_bin(b, tv[0]);  // odd count: the last bin's values are the ones in flight
#pragma unroll
        for (int e = 0; e < kRcEnt; ++e) {
            if (e0 + e >= n) break;  // uniform
            // first maximum over the parts (lanes >= K never win)
            double bv = acc[e];
            int bk = lane < K ? lane : 0x7fffffff;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(bv, off);
                const int ok = __shfl_xor(bk, off);
                const bool take = (ok != 0x7fffffff) && (bk == 0x7fffffff || ov > bv || (ov == bv && ok < bk));
                bv = take ? ov : bv;
                bk = take ? ok : bk;
            }
            int res = bk + 1;
            if (fast && list[(size_t)(e0 + e) * 2 + 1] != 0) {
                double slot0 = 0.0;
                for (int k = 0; k < K; ++k) slot0 += __shfl(acc[e], k) - 1.0;
                if (!(bv > slot0)) res = 0;
            }
            if (lane == 0) {
                if (ids8)
                    reinterpret_cast<uint8_t *>(argmax)[list[(size_t)(e0 + e) * 2]] = (uint8_t)res;
                else
                    argmax[list[(size_t)(e0 + e) * 2]] = res;
            }
        }
        __builtin_amdgcn_wave_barrier();  // xbits is rewritten by the next group
    }
}

template <int NKB, bool FAST>
int launch_nkb2(tsr_ctx *ctx, const PmArgs &a, size_t lds)
{
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(code_parts_mfma_kernel<NKB, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = std::min(a.nchunks, 2 * std::max(ctx->cus, 1));  // frames split evenly over the grid, walked in chunks of CH
    hipLaunchKernelGGL((code_parts_mfma_kernel<NKB, FAST>), dim3((unsigned)grid), dim3(kWG), lds, ctx->stream, a);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
template <int NKB>
int launch_nkb(tsr_ctx *ctx, const PmArgs &a, size_t lds, bool fast)
{
    return fast ? launch_nkb2<NKB, true>(ctx, a, lds) : launch_nkb2<NKB, false>(ctx, a, lds);
}

}  // namespace

void tsr_parts_mfma_free(tsr_parts *p)
{
    if (p->d_wq) (void)hipFree(p->d_wq);
    if (p->d_kc) (void)hipFree(p->d_kc);
    p->d_wq = nullptr;
    p->d_kc = nullptr;
    p->mfma_ok = false;
}

// Quantise the tables for the MFMA coder.  Leaves p->mfma_ok false (and says why in
// p->mfma_why) when the geometry or the value range is outside what the kernel covers; the
// caller then keeps to the VALU kernel.  lp / lip are [K][Q] as handed to tsr_parts_create.
int tsr_parts_mfma_prepare(tsr_ctx *ctx, tsr_parts *p, const double *lp, const double *lip)
{
    p->mfma_ok = false;
    auto no = [&](const char *why) {
        snprintf(p->mfma_why, sizeof(p->mfma_why), "%s", why);
        return TSR_OK;
    };
    const int64_t K = p->K, P0 = p->P0, P1 = p->P1, E = p->E, Q = P0 * P1 * E;
    if (E != 8) return no("E != 8 edge bytes per bin");
    if (K > 63) return no("more than 63 parts");
    const int64_t NB = P0 * P1, NG = (NB + 1) / 2;  // K-groups: pairs of patch bins in row-major order
    if (NG > 16) return no("patch larger than 32 bins");
    const int nkb = (int)((NG + 1) / 2);
    double maxd = 0.0, maxabs = 0.0;
    for (int64_t i = 0; i < K * Q; ++i) {
        if (!std::isfinite(lp[i]) || !std::isfinite(lip[i])) return no("non-finite table entry");
        maxd = std::max(maxd, std::fabs(lp[i] - lip[i]));
        maxabs = std::max(maxabs, std::max(std::fabs(lp[i]), std::fabs(lip[i])));
    }
    std::vector<long double> B((size_t)K, 0.0L);
    long double bmin = 0, bmax = 0;
    for (int64_t k = 0; k < K; ++k) {
        long double s = 0;
        for (int64_t q = 0; q < Q; ++q) s += (long double)lip[k * Q + q];
        B[(size_t)k] = s;
        bmin = k ? std::min(bmin, s) : s;
        bmax = k ? std::max(bmax, s) : s;
    }
    const long double shift_real = (bmin + bmax) / 2;
    const double spread = (double)(bmax - bmin) / 2;
    double s = maxd > 0 ? 32512.0 / maxd : 1.0;
    // |I_k| <= |B_k - shift| s + Q maxd s + 1 must stay below kVMax
    const double lim = (double)(kVMax - 4096);
    if ((spread + (double)Q * maxd) * s > lim) s = lim / (spread + (double)Q * maxd);
    if (!(s > 0) || !std::isfinite(s)) return no("degenerate value range");
    // the reference's own f64 rounding, in units, must vanish inside the 0.25 the window reserves for it
    if ((double)Q * 2.3e-16 * ((double)Q * maxabs) * s > 0.2) return no("tables too large for the f64 rounding allowance");
    const long long shift_units = llroundl(shift_real * (long double)s);

    std::vector<int8_t> wq((size_t)2 * nkb * 2 * 64 * 16, 0);
    auto wq_at = [&](int mt, int kb, int l, int lane, int byte) -> int8_t & {
        return wq[(size_t)((((mt * nkb + kb) * 2 + l) * 64 + lane) * 16 + byte)];
    };
    for (int m = 0; m < 64; ++m) {
        const int mt = m / 32, mr = m % 32;
        for (int64_t g = 0; g < NG; ++g) {
            const int kb = (int)(g / 2), kg = (int)(g % 2);
            const int lane = kg * 32 + mr;
            for (int t = 0; t < 16; ++t) {
                const int64_t bin = 2 * g + t / 8, z = t % 8;
                if (bin >= NB) continue;
                const int64_t q = bin * E + z;
                long long dq;
                if (m < K) {
                    dq = llround((lp[m * Q + q] - lip[m * Q + q]) * s);
                    if (dq > 32512) dq = 32512;   // cannot happen for s <= 32512/maxd; keeps the limbs in range
                    if (dq < -32512) dq = -32512;
                } else if (m == 63) {
                    dq = 1;  // the row of ones: edge count
                } else {
                    continue;
                }
                const long long hi = (dq + 128) >> 8, lo = dq - hi * 256;
                wq_at(mt, kb, 0, lane, t) = (int8_t)lo;
                wq_at(mt, kb, 1, lane, t) = (int8_t)hi;
            }
        }
    }
    std::vector<int32_t> kc(64);
    int n_inv = 0;
    for (int h = 0; h < 2; ++h)
        for (int mt = 0; mt < 2; ++mt)
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + 8 * (r / 4) + h * 4 + (r % 4);
                long long v;
                if (m < K) {
                    v = llroundl(B[(size_t)m] * (long double)s) - shift_units;
                    if (v > kVMax || v < -kVMax) return no("base sums out of range");
                } else {
                    v = kVInv;
                    ++n_inv;
                }
                kc[(size_t)((h * 2 + mt) * 16 + r)] = (int32_t)(v * 64 + m);
            }
    p->pm_NG = (int)NG;
    p->nkb = nkb;
    p->pm_n_inv = n_inv;
    // slot 0 of code_parts_fast, sum_k (I_k + shift) - K s, against best + shift: the kernel adds
    // this constant to sum_k I_k and compares with the (shifted) best
    p->pm_o0_const = (float)((double)(K - 1) * (double)shift_units - (double)K * s);
    p->pm_o0_slack = 4096.0f + 1e-6f * (float)((double)K * (double)kVMax);  // f32 summation of up to 64 values below 2^26
    TSR_HIP(hipSetDevice(ctx->device));
    TSR_HIP(hipMalloc(&p->d_wq, wq.size()));
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_kc), kc.size() * 4));
    TSR_HIP(hipMemcpyAsync(p->d_wq, wq.data(), wq.size(), hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(p->d_kc, kc.data(), kc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    p->mfma_ok = true;
    p->mfma_why[0] = 0;
    return TSR_OK;
}

// Part ids of every live location of a resident edge stream -> d_argmax [P][N1].
// Returns TSR_ERR_UNSUPPORTED (nothing enqueued) when this corpus geometry does not fit.
// d_list must hold 8 bytes per location, d_work two ints.
int tsr_launch_code_parts_mfma(tsr_ctx *ctx, const tsr_parts *p, const uint8_t *d_X, int64_t P, int64_t X1,
                               int Dp, const int32_t *d_utt_end, int threshold, int fast, int32_t *d_argmax,
                               uint32_t *d_list, int *d_work, int ids8)
{
    if (!p->mfma_ok) return TSR_ERR_UNSUPPORTED;
    const int64_t N1 = X1 - p->P1 + 1;
    if (P <= 0 || N1 <= 0) return TSR_OK;
    if (Dp % 16 != 0 || X1 * p->E > Dp || P * N1 >= ((int64_t)1 << 31)) return TSR_ERR_UNSUPPORTED;
    const int S = Dp / 8;
    int CH = 0;
    size_t lds = 0;
    for (int ch : {64, 32, 16}) {  // two chunk buffers per workgroup, two workgroups per CU where they fit
        const size_t need = 2 * ((size_t)(ch + p->P0 - 1) * Dp + kPad) + 256 + (size_t)2 * ch * 4 + 32;
        if (need <= 76 * 1024 || (ch == 16 && need <= 150 * 1024)) {
            CH = ch;
            lds = need;
            break;
        }
    }
    if (!CH) return TSR_ERR_UNSUPPORTED;
    // (d_work[0..1] = 0: the caller's setup kernel)
    PmArgs a;
    a.X = d_X;
    a.P = P;
    a.Dp = Dp;
    a.S = S;
    a.N1 = (int)N1;
    a.P0 = (int)p->P0;
    a.P1 = (int)p->P1;
    a.CH = CH;
    a.nchunks = (int)((P + CH - 1) / CH);
    a.utt_end = d_utt_end;
    a.Wq = reinterpret_cast<const i32x4 *>(p->d_wq);
    a.Kc = p->d_kc;
    a.K = (int)p->K;
    a.threshold = (unsigned)threshold;
    a.o0_const = p->pm_o0_const;
    a.o0_slack = p->pm_o0_slack;
    a.n_inv = p->pm_n_inv;
    a.work = d_work;
    a.argmax = d_argmax;
    a.ids8 = ids8;
    a.list = d_list;
    int s = TSR_OK;
    switch (p->nkb) {
    case 1: s = launch_nkb<1>(ctx, a, lds, fast != 0); break;
    case 2: s = launch_nkb<2>(ctx, a, lds, fast != 0); break;
    case 3: s = launch_nkb<3>(ctx, a, lds, fast != 0); break;
    case 4: s = launch_nkb<4>(ctx, a, lds, fast != 0); break;
    case 5: s = launch_nkb<5>(ctx, a, lds, fast != 0); break;
    case 6: s = launch_nkb<6>(ctx, a, lds, fast != 0); break;
    case 7: s = launch_nkb<7>(ctx, a, lds, fast != 0); break;
    case 8: s = launch_nkb<8>(ctx, a, lds, fast != 0); break;
    default: return TSR_ERR_UNSUPPORTED;
    }
    TSR_TRY(s);
    hipLaunchKernelGGL(code_parts_recheck_kernel, dim3((unsigned)(32 * std::max(ctx->cus, 1) / kRcWaves)), dim3(64 * kRcWaves), 0, ctx->stream,
                       d_X, Dp, (int)p->P0, (int)p->P1, (int)N1, p->d_tab, (int)p->K, fast ? 1 : 0,
                       (const uint32_t *)d_list, (const int *)d_work, d_argmax, ids8);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
