// detect_mfma.hip -- TSR_FAST: the template-bank scorer on the gfx950 matrix cores.
//
// What is computed (get_train_data.py:962-992 over compute_likelihood_linear_filter
// .pyx:135-176):  score[p, m] = sum_{tau < L_m} sum_d X[p+tau, d] * W_m[tau, d].
//
// Formulation -- "skew GEMM".  G = X . Wall^T is one dense product with
//     rows    = frames p of the whole corpus (utterances concatenated),
//     K       = the D features of one frame,
//     columns = every (template m, tap tau) pair, sum_m L_m of them,
// and the window sum is the diagonal ("skew") reduction
//     score[p, m] = sum_tau G[p + tau, (m, tau)].
// No im2col, no padding of the bank to a common length, every frame is read once.
//
// Arithmetic -- exact integers.  X is {0,1} bytes, i.e. already an int8 operand.
// Each template's f32 weights are quantised ONCE (bank build) to a signed fixed point
// with a per-template scale and split into NL signed int8 limbs (balanced base 256):
//   TSR_FAST            NL = 2, 16-bit weights (|q| <= 127*256): per-tap error
//                       <= 2^-16 max|W_m|, i.e. the same order as the rounding noise of
//                       the reference's own sequential f32 sum; int32 ring;
//   TSR_FAST | TSR_W24  NL = 3, 23-bit weights, power-of-two scale: per-tap error
//                       <= 2^-22 max|W_m|; int64 ring.
// NL v_mfma_i32_32x32x32_i8 per K-block accumulate the limbs in int32; all sums are exact,
// so the result does not depend on tiling, wave count or summation order (bit-reproducible)
// and the only deviation from the f32 reference is the weight quantisation.  int8 MFMA runs
// at 2x the bf16 rate: two limbs cost one bf16 pass and are ~100x more accurate than bf16.
//
// Data movement.
//   * Weights are REGISTER-STATIONARY: a workgroup is 8 waves = CT column tiles
//     (32 columns each) x KS K-slices; each wave keeps its NKB x 3 limb fragments
//     (NKB*12 VGPRs) for the whole launch.  Config 2 (D=360, 8 x 32 taps): the
//     entire bank, 288 KiB of limbs, lives in one workgroup's registers.
//   * Frames stream through LDS in 32-row tiles by LDS-DMA (global_load_lds,
//     16 B/lane), double buffered; the resident corpus rows are 16-byte multiples
//     so a tile is one flat copy.  A-fragments are ds_read_b128, conflict-free
//     because the row stride is an odd number of 16-byte slots.
//   * The skew reduction goes through LDS atomics (ds_add_u32) into a small ring
//     indexed by output frame; a finishing pass turns the ring into f32 scores,
//     adds c_m, takes the max / first arg-max over templates and writes 8 bytes
//     per frame.  HBM sees the feature map once and the result once.
//
// Roofline: MFMA-bound for M*L >~ 170 columns; algorithmic HBM
// bytes per frame = D + 6.
#include <math.h>

#include <algorithm>

#include "tsr_internal.h"

// Development ablation switches (never set in the shipped build): bit 0 drops the skew
// emission, bit 1 the MFMAs, bit 2 the finishing pass, bit 3 the frame staging.
#ifndef TSR_ABL
#define TSR_ABL 0
#endif
// Development stamps (never in the shipped build): -DTSR_STAMP accumulates s_memtime deltas of
// the steady-state iteration's phases per wave and dumps them to the file $TSR_STAMP_OUT.
#ifdef TSR_STAMP
#include <stdlib.h>
#define TSR_STAMP_AT(i)                                                                  \
    {                                                                                    \
        unsigned long long _t;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                               \
        st_sum[i] += _t - st_last;                                                       \
        st_last = _t;                                                                    \
    }
#else
#define TSR_STAMP_AT(i)
#endif

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxNKB = 12;   // K-blocks (of 32 features) a wave keeps in registers
constexpr int kMaxSlots = 64; // template fragments per workgroup (ring rows)

struct SlotMeta {  // one template fragment handled by a workgroup (LDS copy, 16 bytes)
    float inv_scale;  // 2^-e_m (exact power of two)
    float cf;         // f32(c_m)
    int L;            // template length
    int tf;           // template id (G == 1) or row of raw_frag (G > 1)
};

struct FastArgs {
    const uint8_t *X;        // [P + slack][Dp] bytes in {0,1}
    int Dp;                  // resident row stride (multiple of 16)
    int SB;                  // LDS row stride: odd multiple of 16 >= the bytes copied per row
    int src_slots;           // 16-byte slots copied per frame (whole row, or one K-phase of it)
    int tile_bytes;          // bytes one LDS tile buffer holds (multiple of 1024)
    int G, CT, KS;           // column groups; column tiles and K-slices per workgroup
    int nseg;                // frame-stream segments (workgroups per column group)
    const int32_t *seg;      // [3][nseg]: first output frame, end of outputs, end of rows
    const i32x4 *Wl;         // limb fragments [G][CT][KS][NKB][3][64]
    const int32_t *col_slot; // [G*CT*32] fragment slot of a column inside its group, -1 = padding
    const int32_t *col_tau;  // [G*CT*32] tap index of the column inside its template
    const SlotMeta *slots;   // [G][kMaxSlots]
    const int32_t *g_nslots; // [G] slots in use per column group
    const int32_t *utt_end;  // [P]
    int64_t P;
    int ring;                // ring length in frames (power of two)
    int lag;                 // tiles between emitting a tile and finishing it
    int lmax1;               // Lmax - 1
    float *scores;           // fused outputs (G == 1)
    int32_t *ids;
    float *all;              // [M][P] per-template scores (G == 1, optional)
    float *raw_frag;         // [n_frag][P] partial raw scores (G > 1)
    unsigned long long *dbg; // TSR_STAMP builds only
};

// LDS-DMA (global_load_lds): `size` bytes per lane straight into LDS, destination =
// wave-uniform LDS byte address + lane*size, no VGPR round trip.  Issued through inline asm
// on purpose: hipcc does not model an asm statement's memory operation, so it neither
// drains the DMA (s_waitcnt vmcnt(0)) in front of the next ds_read -- it cannot prove that
// the tile being filled is not the tile being read -- nor counts it in its own waits.  The
// kernel drains explicitly (dma_drain) in front of the barrier that publishes the tile.
// M0 carries the LDS address and is saved/restored inside the statement
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ void lds_dma16(const uint8_t *gsrc, unsigned lds_dst_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_wave_base)
                 : "memory");
}
__device__ __forceinline__ void lds_dma4(const void *gsrc, unsigned lds_dst_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_wave_base)
                 : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NL> struct RingT { using type = long long; };
template <> struct RingT<2> { using type = int; };

// NKB: K-blocks (of 32 features) per wave; NL: int8 limbs per weight (2: 16-bit fixed point,
// int32 ring; 3: 23-bit fixed point, int64 ring); FLAT: LDS row stride == resident row stride.
// MODE: 0 = max score + template id per frame, 1 = every template's score, 2 = partial raw
// scores per template fragment (several column groups; frag_epilogue_kernel finishes them).
// CPW: column tiles per wave.  1: eight waves of 32 columns, two per SIMD.  2: four waves of 64
// columns, ONE per SIMD with the whole 512-register file: every A fragment read from LDS feeds
// twice the MFMAs, and all of the wave's other work is issued in the shadow of its own MFMAs.
template <int NKB, int NL, bool FLAT, int MODE, int CPW>
__global__ __launch_bounds__(kThreads / CPW, CPW == 2 ? 1 : 2) void fast_bank_kernel(const FastArgs a)
{
    constexpr int kW = kWaves / CPW;  // waves per workgroup
    constexpr int kT = kW * 64;
    using ring_t = typename RingT<NL>::type;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.x % a.G;   // column group; neighbours share the same frames (L2 reuse)
    const int sg = blockIdx.x / a.G;   // frame-stream segment
    const int o0 = a.seg[sg], o1 = a.seg[a.nseg + sg], r1 = a.seg[2 * a.nseg + sg];
    const int nOut = o1 - o0;
    const int nT = (TSR_ABL & 16) ? 0 : (r1 - o0 + 31) >> 5;
    const int nslots = a.g_nslots[cg];

    // ---- LDS carve-up: [tile 0][tile 1][tile 2][slot meta][utt_end ring][max-key ring]
    //      [ring: (nslots+1) rows] -----------------------------------------------------------
    // A ring row holds one partial-sum cell per frame in flight (a.ring of them, power of
    // two) PLUS 32 overflow cells: a tile's 32 rows are emitted at cell (first & mask) + 0..31
    // without wrapping, so the 16 atomics of a lane share one address register and differ
    // only in their immediate offset; the finishing pass folds overflow cell i into cell i.
    const unsigned xb0 = lds_addr(smem), xb_end = xb0 + 3 * a.tile_bytes;
    auto xnext = [&](unsigned x) { return x + a.tile_bytes == xb_end ? xb0 : x + a.tile_bytes; };
    auto lds_ptr = [&](unsigned addr) { return smem + (addr - xb0); };
    SlotMeta *meta = reinterpret_cast<SlotMeta *>(smem + 3 * a.tile_bytes);
    int *uend_ring = reinterpret_cast<int *>(meta + kMaxSlots);  // utt_end of the frames in flight
    unsigned long long *maxring = reinterpret_cast<unsigned long long *>(uend_ring + a.ring + 64);
    ring_t *ring = reinterpret_cast<ring_t *>(maxring + a.ring);
    for (int i = tid; i < a.ring; i += kT) maxring[i] = 0;
    const int ring_mask = a.ring - 1;
    const int rstride = a.ring + 33;  // odd: a column of the ring hits distinct banks
    const int dummy = nslots * rstride;  // out-of-range emissions land in this spare row
    for (int i = tid; i < (nslots + 1) * rstride; i += kT) ring[i] = 0;
    if (tid < nslots) meta[tid] = a.slots[cg * kMaxSlots + tid];

    // ---- this wave's role and its register-stationary limb fragments --------------------
    const int half = lane >> 5;
    i32x4 b[CPW][NKB][NL];
    int kl[CPW];           // frame (relative to the tile's first row) of accumulator row 0
    ring_t *my_row[CPW];
    bool real_cols = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int coltile = cg * a.CT + wave * CPW + c;
        const i32x4 *wl = a.Wl + ((size_t)coltile * NKB * NL) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int l = 0; l < NL; ++l) b[c][kb][l] = wl[(kb * NL + l) * 64];
        const int col = coltile * 32 + (lane & 31);
        const int slot = a.col_slot[col];
        // padding columns get a tap index that pushes every target frame out of range
        const int tau = slot >= 0 ? a.col_tau[col] : (1 << 28);
        my_row[c] = ring + (slot >= 0 ? slot : nslots) * rstride;
        kl[c] = 4 * half - tau;
        real_cols = real_cols && slot >= 0;
    }
    ring_t *const dummy_row = ring + dummy;

    // ---- frame tiles: tile t = rows [o0 + 32 t, +32), copied by LDS-DMA --------------------
    // one DMA instruction moves 1 KiB (64 lanes x 16 B); wave w issues pieces w, w+8, ...
    const int n_pieces = a.tile_bytes >> 10;
    const int slots_per_row = a.SB >> 4, src_slots = a.src_slots;
    (void)slots_per_row;
    (void)src_slots;
    // per-lane byte offset of DMA piece `pc` inside a tile of the resident corpus
    auto piece_src_off = [&](int pc) -> int {
        const int chunk = pc * 64 + lane;  // 16-byte slot of the LDS tile
        if (FLAT) return chunk * 16;
        // LDS rows carry one extra (unused) slot to keep the stride odd
        const int row = chunk / slots_per_row, c16 = chunk - row * slots_per_row;
        return row * a.Dp + min(c16, src_slots - 1) * 16;
    };
    // (kept in registers only when there are registers to spare)
    constexpr int NP = 2 * CPW;  // DMA pieces per wave whose source offsets stay in registers
    int src_off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) src_off[i] = NL == 2 ? piece_src_off(wave + i * kW) : 0;
    const uint8_t *const seg_src = a.X + (int64_t)o0 * a.Dp;
    const int tile_src_bytes = 32 * a.Dp;
    auto stage = [&](int t, unsigned dst) {
        if (TSR_ABL & 8) return;
        const uint8_t *src_tile = seg_src + (int64_t)t * tile_src_bytes;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (wave + i * kW < n_pieces)
                lds_dma16(src_tile + (NL == 2 ? src_off[i] : piece_src_off(wave + i * kW)), dst + (wave + i * kW) * 1024);
        for (int pc = wave + NP * kW; pc < n_pieces; pc += kW)  // wide rows only
            lds_dma16(src_tile + piece_src_off(pc), dst + pc * 1024);
        // the utt_end values of the tile's 32 frames ride along (4-byte DMA, one wave): the
        // finishing pass must not issue register loads, or hipcc drains the tile DMA early
        if (wave == kW - 1 && lane < 32)
            lds_dma4(a.utt_end + (int64_t)o0 + 32 * (int64_t)t + lane,
                     __builtin_amdgcn_readfirstlane(lds_addr(uend_ring + ((32 * t) & ring_mask))));
    };

    // A-fragment address of this lane: row (lane & 31), K-group (lane >> 5) of its K-slice
    const int a_off = (lane & 31) * a.SB + half * 16;

    // ---- finishing pass -----------------------------------------------------------------------
    // Phase AB (every wave, its own slots: s = wave + kW (2 j + half); lane&31 is the frame):
    // read and reset the ring cell (plus its overflow cell when this tile owns them),
    // integer -> f32 score, zero tail, + c_m.  The max / first arg-max over templates is
    // taken by the LDS: one ds_max_u64 per (frame, slot) of the key
    //     (orderable(score) << 32) | ~template_id
    // (larger score wins; on equal scores the lower id has the larger ~id, i.e. the
    // reference's strict '>' = first maximum).
    // Phase C (wave f % 8, one iteration later): decode the tile's 32 keys, lanes 0..31 store
    // scores and lanes 32..63 ids -- two coalesced 128-byte stores.
    struct FinA {  // phase A: everything phase B needs from LDS, fetched a few steps ahead
        ring_t q;
        SlotMeta sm;
        int uend;
    };
    auto fin_a = [&](int f, int j, FinA &fa) {
        const int s = wave + kW * (2 * j + half);
        const int ridx = (f * 32 + (lane & 31)) & ring_mask;
        ring_t *cell = ring + min(s, nslots) * rstride + ridx;  // spare row when s is past the end
        fa.q = *cell;
        *cell = 0;
        if (((32 * f) & ring_mask) == 0) {  // wave-uniform: this tile owns the overflow cells
            fa.q += cell[a.ring];
            cell[a.ring] = 0;
        }
        fa.sm = meta[min(s, kMaxSlots - 1)];
        fa.uend = uend_ring[ridx];
    };
    auto fin_b = [&](int f, int j, const FinA &fa) {
        const int s = wave + kW * (2 * j + half);
        const int prel = f * 32 + (lane & 31);
        const int ridx = prel & ring_mask;
        const SlotMeta sm = fa.sm;
        const int64_t p = (int64_t)o0 + prel;
        float x;
        if (NL == 2) {
            x = (float)fa.q * sm.inv_scale;  // |q| < 2^31: one rounding
        } else {
            // exact int64 -> correctly rounded f32: q = hi*2^24 + lo, |hi| < 2^13, 0 <= lo < 2^24,
            // both exact in f32, a single rounding in the fma
            const long long qq = fa.q;
            x = __builtin_fmaf((float)(int)(qq >> 24), 16777216.0f, (float)((unsigned)qq & 0xffffffu)) *
                sm.inv_scale;
        }
        if (MODE == 2) {  // several column groups: partial sums, finished by the epilogue kernel
            if (prel < nOut && s < nslots) a.raw_frag[(int64_t)sm.tf * a.P + p] = x;
            return;
        }
        // tail t > T - L_m scores 0 (+ c_m), as the reference's zero tail does
        const float raw = (p + sm.L <= fa.uend) ? x : 0.0f;
        const float sc = raw + sm.cf;
        if (MODE == 1) {
            if (prel < nOut && s < nslots) a.all[(int64_t)sm.tf * a.P + p] = sc;
            return;
        }
        unsigned u = __builtin_bit_cast(unsigned, sc);
        u ^= (unsigned)((int)u >> 31) | 0x80000000u;  // monotone map f32 -> u32
        // idle lanes (slot past the end) contribute key 0, the identity of max: no branch
        const unsigned long long key = (s < nslots) ? (((unsigned long long)u << 32) | (unsigned)~sm.tf) : 0ull;
        atomicMax(&maxring[ridx], key);
    };
    auto fin_ab = [&](int f, int j) {
        FinA fa;
        fin_a(f, j, fa);
        fin_b(f, j, fa);
    };
    auto fin_c_read = [&](int f) -> unsigned long long {
        if (MODE != 0) return 0;
        const int ridx = (f * 32 + (lane & 31)) & ring_mask;
        const unsigned long long key = maxring[ridx];
        __builtin_amdgcn_wave_barrier();  // both halves have read the cell before it is reset
        if (half == 0) maxring[ridx] = 0;
        return key;
    };
    auto fin_c_store = [&](int f, unsigned long long key) {
        if (MODE != 0) return;
        const int prel = f * 32 + (lane & 31);
        unsigned u = (unsigned)(key >> 32);
        u ^= (~(unsigned)((int)u >> 31)) | 0x80000000u;  // inverse of the monotone map
        // lanes 0..31 store the score bits, lanes 32..63 the ids: one predicated dword store
        int *dst = (half == 0 ? reinterpret_cast<int *>(a.scores) : a.ids) + (int64_t)o0 + prel;
        const int val = half == 0 ? (int)u : (int)~(unsigned)key;
        if (prel < nOut) *dst = val;
    };
    const int njs = (nslots - wave + 2 * kW - 1) / (2 * kW);  // slot pairs this wave walks (wave-uniform)

    // ---- the pipelined tile loop ----------------------------------------------------------
    // Iteration t (one barrier, in the MIDDLE of the tile):
    //   * MFMAs of tile t: NL*NKB v_mfma_i32_32x32x32_i8 -> one exact int32 per element
    //     (|sum| <= NKB*32 * 2^(8 NL - 2) < 2^31).  A fragments come from a rolling queue that
    //     is refilled PF K-blocks ahead and runs across the tile boundary, so the matrix
    //     pipe never waits for LDS latency or for the barrier;
    //   * skew emission of tile t-1: G[row, (m, tau)] adds into frame row - tau of its
    //     template (ring, LDS atomics; accumulator register r of lane (col, half) is row
    //     (r&3) + 8 (r>>2) + 4 half), spread over the K-block steps;
    //   * mid-tile: drain the DMA of tile t+1, barrier, then the DMA of tile t+2 is issued
    //     (its buffer was last read in iteration t-1, which every wave has left);
    //   * second half: finishing phase AB of tile f = t - lag (every tile <= t-2 is fully
    //     emitted once all waves passed the barrier) and phase C of tile f - 1.
    // sched_barrier keeps hipcc from clustering the VALU/LDS work behind the last MFMA.
    constexpr int H = NKB / 2;                            // K-block step that opens with the barrier
    constexpr int PF = (NL == 2 ? 4 : 1) < (NKB - H) ? (NL == 2 ? 4 : 1) : (NKB - H);  // queue depth
    // finishing phases: A (LDS reads) two steps ahead of B (arithmetic + max-key atomic); C reads
    // its keys one step ahead of the stores
    // (three limbs leave no registers to carry phase A's values across steps: same step there)
    constexpr int kFA = H, kFB = (NL == 2 && H + 2 < NKB) ? H + 2 : H;
    constexpr int kCR = (H + 3 < NKB) ? H + 3 : NKB - 1, kCS = (NL == 2 && H + 4 < NKB) ? H + 4 : kCR;
    const int lag = a.lag + 2;
    // a wave whose 32 columns are all real taps can skip the range check on tiles whose
    // targets [32 te - (Lmax-1), 32 te + 35] all lie inside the segment's outputs
    const bool no_pad = __all(real_cols);
    const int lmax1 = a.lmax1;
    i32x4 aq[PF];

    unsigned x_cur = xb0;  // LDS address of tile t; t+1 and t+2 follow in the 3-buffer rotation
    // Two accumulator sets alternate between tiles: `acc` receives tile t's MFMAs while `pacc`
    // holds tile t-1's sums for the emission.  With two limbs pacc keeps the raw limb sums and
    // each element is combined (acc1*256 + acc0, exact int32) right before its atomic: no combine
    // burst waits for the last MFMA of a tile and the next tile's MFMAs can issue while the old
    // ones drain.  With three limbs two raw sets do not fit the register file, so the tile is
    // folded into limb 0 at its end (`fold`) and only those 16 registers stay live.
    auto combine = [&](const i32x16 (&pa)[NL], int r) -> int {
        if (NL != 2) return pa[0][r];
        return pa[NL - 1][r] * 256 + pa[0][r];
    };
    auto fold = [&](i32x16 (&acc)[NL]) {
        if (NL == 2) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int x = acc[NL - 1][r];
#pragma unroll
            for (int l = NL - 2; l >= 0; --l) x = x * 256 + acc[l][r];
            acc[0][r] = x;
        }
    };
    auto tile_iter = [&](int t, const i32x16 (&pacc)[CPW][NL], i32x16 (&acc)[CPW][NL], bool do_emit, bool interior,
                         int fin) {
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int l = 0; l < NL; ++l) acc[c][l] = (i32x16){};
        const unsigned x_nxt = xnext(x_cur);
        const uint8_t *xa = lds_ptr(x_cur) + a_off, *xn = lds_ptr(x_nxt) + a_off;
        int first[CPW];  // target frame of accumulator row 0 (tile t-1)
        ring_t *cell0[CPW], *dcell0[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            first[c] = kl[c] + 32 * (t - 1);
            cell0[c] = my_row[c] + (first[c] & ring_mask);
            dcell0[c] = dummy_row + (first[c] & ring_mask);
        }
        FinA fa = {};
        unsigned long long ckey = 0;
        const bool c_wave = fin >= 1 && wave == ((fin - 1) & (kW - 1));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb == H) {
                if (!(TSR_ABL & 128)) dma_drain();  // this wave's pieces of tile t+1 (and its stores) are done
                if (!(TSR_ABL & 64)) __syncthreads();  // tile t+1 landed; tiles <= t-2 fully emitted
                if (t + 2 < nT) stage(t + 2, xnext(x_nxt));
            }
            const i32x4 av = aq[0];
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i) aq[i] = aq[i + 1];
            // refill: K-block kb+PF of this tile, or the head of the next tile (landed:
            // kb+PF >= NKB only happens at steps >= H, behind the barrier)
            const int nk = kb + PF;
            aq[PF - 1] = *reinterpret_cast<const i32x4 *>(nk < NKB ? xa + nk * 32 : xn + (nk - NKB) * 32);
            if (TSR_ABL & 2) {
                asm volatile("" ::"v"(av), "v"(b[0][kb][0]), "v"(b[CPW - 1][kb][NL - 1]));
            } else {
#pragma unroll
                for (int c = 0; c < CPW; ++c)
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        acc[c][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b[c][kb][l], acc[c][l], 0, 0, 0);
            }
            if (do_emit && !(TSR_ABL & 1)) {
#pragma unroll
                for (int e = 0; e < 16 * CPW; ++e) {
                    if (e * NKB / (16 * CPW) != kb) continue;  // compile-time after unrolling
                    const int ci = e >> 4, r = e & 15;
                    const int c = (r & 3) + 8 * (r >> 2);
                    ring_t *cell = cell0[ci] + c;
                    if (!interior) cell = ((unsigned)(first[ci] + c) < (unsigned)nOut) ? cell : dcell0[ci] + c;
                    const int vr = combine(pacc[ci], r);
                    if (NL == 2)
                        atomicAdd(reinterpret_cast<unsigned int *>(cell), (unsigned int)vr);
                    else
                        atomicAdd(reinterpret_cast<unsigned long long *>(cell),
                                  (unsigned long long)(long long)vr);
                }
            }
            if (!(TSR_ABL & 4)) {
                if (kb == kFA && fin >= 0 && njs > 0) fin_a(fin, 0, fa);
                if (kb == kFB && fin >= 0) {
                    if (njs > 0) fin_b(fin, 0, fa);
                    for (int j = 1; j < njs; ++j) fin_ab(fin, j);  // more than 16 slots
                }
                if (kb == kCR && c_wave) ckey = fin_c_read(fin - 1);
                if (kb == kCS && c_wave) fin_c_store(fin - 1, ckey);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            if (TSR_ABL & 3) {
#pragma unroll
                for (int l = 0; l < NL; ++l) asm volatile("" : "+v"(acc[c][l]));
            }
            fold(acc[c]);
        }
        x_cur = x_nxt;
    };

    // ---- steady-state iteration: the same work as tile_iter for the common case -- interior
    //      tile, DMA of tile t+2 due, finishing phases A/B (tile t-lag) and C (tile t-lag-1) all
    //      active, at most 16 slots -- with every loop-invariant hoisted: this lane's slot
    //      metadata in registers, LDS byte addresses precomputed, no predication.  A wave issues
    //      about one instruction per four cycles, so the instruction COUNT of this loop is what
    //      decides whether the matrix pipe stays fed.
    typedef __attribute__((address_space(3))) int lds_i32;
    typedef __attribute__((address_space(3))) ring_t lds_ring;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) const i32x4 lds_frag;
    const int lane31 = lane & 31;
    constexpr int NJ = CPW;  // slots this lane finishes per tile: s_j = wave + kW (2 j + half)
    bool s_ok[NJ];
    SlotMeta my_sm[NJ];
    unsigned st_fin_row[NJ], st_key_lo[NJ], st_emit_row[CPW];
    int st_pL0[NJ];  // (frame + L) of this lane's frame in tile 0
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int my_s = wave + kW * (2 * j + half);
        s_ok[j] = my_s < nslots;
        my_sm[j] = a.slots[cg * kMaxSlots + min(my_s, kMaxSlots - 1)];
        st_fin_row[j] = lds_addr(ring + (s_ok[j] ? my_s : nslots) * rstride + lane31);
        st_pL0[j] = o0 + lane31 + my_sm[j].L;
        st_key_lo[j] = s_ok[j] ? (unsigned)~my_sm[j].tf : 0u;
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) st_emit_row[c] = lds_addr(my_row[c]);
    const unsigned st_uend = lds_addr(uend_ring + lane31);
    const unsigned st_max = lds_addr(maxring + lane31);
    int *const st_out = (half == 0 ? reinterpret_cast<int *>(a.scores) : a.ids) + o0 + lane31;
    const unsigned st_a_off = (unsigned)a_off;
    const bool st_ok = MODE == 0 && NL == 2 && nslots <= 2 * kW * NJ && n_pieces <= NP * kW;
#ifdef TSR_STAMP
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
    const uint8_t *st_src[NP];  // set where the steady range is known
    const int32_t *st_usrc = nullptr;
    unsigned st_udst0 = 0;
    // finishing steps of the steady iteration: phase A of slot j, phase B two steps later, phase C
    // (read, then decode + store) behind them; everything clamped into the tile's K-block steps
    auto kclamp = [](int k) constexpr { return k < NKB ? k : NKB - 1; };
    auto steady_iter = [&](int t, const i32x16 (&pacc)[CPW][NL], i32x16 (&acc)[CPW][NL]) {
#ifdef TSR_STAMP
        {
            unsigned long long _t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");
            if (st_last) st_sum[5] += _t - st_last;  // between iterations (loop control, combine)
            st_last = _t;
        }
#endif
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int l = 0; l < NL; ++l) acc[c][l] = (i32x16){};
        const unsigned x_nxt = xnext(x_cur);
        const unsigned xa = x_cur + st_a_off, xn = x_nxt + st_a_off;
        const int f = t - lag;                                     // phases A/B; phase C does f-1
        const unsigned fb4 = (unsigned)((32 * f) & ring_mask) * 4u;        // byte offset of tile f in a ring row
        const unsigned fc8 = (unsigned)((32 * (f - 1)) & ring_mask) * 8u;  // ... of tile f-1 in the key ring
        unsigned cell0[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c)
            cell0[c] = st_emit_row[c] + (unsigned)((kl[c] + 32 * (t - 1)) & ring_mask) * (unsigned)sizeof(ring_t);
        const bool c_wave = wave == ((f - 1) & (kW - 1));
        ring_t fq[NJ];
        int fuend = 0;
        unsigned long long ckey = 0;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb == H) {
                TSR_STAMP_AT(0)  // first half: steps 0..H-1
                dma_drain();
                TSR_STAMP_AT(1)  // wait for this wave's DMA
                __syncthreads();
                TSR_STAMP_AT(2)  // barrier
                // DMA of tile t+2: per-lane source pointers advance by one tile per iteration
                const unsigned dst = xnext(x_nxt);
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    if (wave + i * kW < n_pieces) lds_dma16(st_src[i], dst + (wave + i * kW) * 1024);
                    st_src[i] += tile_src_bytes;
                }
                if (wave == kW - 1 && lane < 32)
                    lds_dma4(st_usrc, st_udst0 + (unsigned)((32 * (t + 2)) & ring_mask) * 4u);
                st_usrc += 32;
                TSR_STAMP_AT(3)  // DMA issue
            }
            const i32x4 av = aq[0];
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i) aq[i] = aq[i + 1];
            const int nk = kb + PF;
            aq[PF - 1] = *(lds_frag *)(size_t)(nk < NKB ? xa + nk * 32 : xn + (nk - NKB) * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (kb == kclamp(H + j)) {  // finishing phase A: this lane's cell of tile f, reset; utt_end
                    lds_ring *cell = (lds_ring *)(size_t)(st_fin_row[j] + fb4 * (unsigned)(sizeof(ring_t) / 4));
                    fq[j] = *cell;
                    *cell = 0;
                    if (fb4 == 0) {  // wave-uniform: tile f owns the overflow cells
                        fq[j] += cell[a.ring];
                        cell[a.ring] = 0;
                    }
                    if (j == 0) fuend = *(lds_i32 *)(size_t)(st_uend + fb4);
                }
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int l = 0; l < NL; ++l)
                    acc[c][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b[c][kb][l], acc[c][l], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16 * CPW; ++e) {
                if (e * NKB / (16 * CPW) != kb) continue;  // compile-time after unrolling
                const int ci = e >> 4, r = e & 15;
                const int c = (r & 3) + 8 * (r >> 2);
                lds_ring *cell = (lds_ring *)(size_t)cell0[ci] + c;
                const int vr = combine(pacc[ci], r);
                if (NL == 2)
                    __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int *)cell, (unsigned int)vr,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_fetch_add((lds_u64 *)cell, (unsigned long long)(long long)vr, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (kb == kclamp(H + 2 + j)) {  // finishing phase B
                    float x;
                    if (NL == 2) {
                        x = (float)fq[j] * my_sm[j].inv_scale;
                    } else {
                        const long long qq = fq[j];
                        x = __builtin_fmaf((float)(int)(qq >> 24), 16777216.0f, (float)((unsigned)qq & 0xffffffu)) *
                            my_sm[j].inv_scale;
                    }
                    const float raw = (st_pL0[j] + 32 * f <= fuend) ? x : 0.0f;  // zero tail
                    const float sc = raw + my_sm[j].cf;
                    unsigned u = __builtin_bit_cast(unsigned, sc);
                    u ^= (unsigned)((int)u >> 31) | 0x80000000u;
                    const unsigned long long key = ((unsigned long long)(s_ok[j] ? u : 0u) << 32) | st_key_lo[j];
                    __hip_atomic_fetch_max((lds_u64 *)(size_t)(st_max + fb4 * 2u), key, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            if (kb == kclamp(H + 2 + NJ) && c_wave) {  // phase C: read + reset the keys
                lds_u64 *kc = (lds_u64 *)(size_t)(st_max + fc8);
                ckey = *kc;
                *kc = 0;  // (both halves read before either write: LDS ops of a wave are in order)
            }
            if (kb == kclamp(H + 3 + NJ) && c_wave) {  // phase C: decode + store
                unsigned u = (unsigned)(ckey >> 32);
                u ^= (~(unsigned)((int)u >> 31)) | 0x80000000u;
                st_out[32 * (f - 1)] = half == 0 ? (int)u : (int)~(unsigned)ckey;
            }
            if (CPW == 2) {
                // one wave per SIMD: nothing else fills the gap behind an MFMA, so the step's other
                // instructions are dealt out between its four MFMAs (in-order issue would otherwise
                // park the wave on the busy matrix pipe with all of them still to come)
#pragma unroll
                for (int i = 0; i < CPW * NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
                    __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);  // LDS
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) fold(acc[c]);
        x_cur = x_nxt;
        TSR_STAMP_AT(4)  // second half: steps H..NKB-1 + combine
    };

    if (nT > 0) stage(0, xb0);
    if (nT > 1) stage(1, xnext(xb0));
    dma_drain();
    __syncthreads();  // ring/meta initialised, tiles 0 and 1 landed
    if (nT > 0) {
#pragma unroll
        for (int i = 0; i < PF; ++i) aq[i] = *reinterpret_cast<const i32x4 *>(smem + a_off + i * 32);
    }
    // steady range [ts0, ts1): interior tile, DMA of t+2 due, phases A/B/C active and phase C's
    // tile t-lag-1 entirely inside the outputs (it stores unconditionally)
    int ts0 = max(lag + 1, (lmax1 + 31) / 32 + 1), ts1 = min(nT - 2, (nOut - 36) / 32 + 2);
    if (!st_ok || !no_pad) ts1 = ts0;
    while (ts1 > ts0 && !(32 * (ts1 - 2) + 35 < nOut && 32 * (ts1 - 1 - lag) <= nOut)) --ts1;
    // DMA source pointers of the first steady iteration (tile ts0 + 2)
#pragma unroll
    for (int i = 0; i < NP; ++i) st_src[i] = seg_src + (int64_t)(ts0 + 2) * tile_src_bytes + src_off[i];
    st_usrc = a.utt_end + (int64_t)o0 + 32 * (int64_t)(ts0 + 2) + lane;
    st_udst0 = __builtin_amdgcn_readfirstlane(lds_addr(uend_ring));
    i32x16 accA[CPW][NL], accB[CPW][NL];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int l = 0; l < NL; ++l) accA[c][l] = (i32x16){};
    auto one_iter = [&](int t, const i32x16 (&pacc)[CPW][NL], i32x16 (&acc)[CPW][NL]) {
        const bool interior = no_pad && 32 * (t - 1) >= lmax1 && 32 * (t - 1) + 35 < nOut;
        if (t >= ts0 && t < ts1)
            steady_iter(t, pacc, acc);
        else if (interior)
            tile_iter(t, pacc, acc, true, true, t - lag);
        else
            tile_iter(t, pacc, acc, t > 0, false, t - lag);
    };
    // the last tile's sums go through the skew reduction after the loop (nothing left to overlap)
    auto emit_last = [&](const i32x16 (&pacc)[CPW][NL]) {
        if (TSR_ABL & 1) return;
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            const int first = kl[ci] + 32 * (nT - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2);
                ring_t *cell =
                    (((unsigned)(first + c) < (unsigned)nOut) ? my_row[ci] : dummy_row) + (first & ring_mask) + c;
                const int vr = combine(pacc[ci], r);
                if (NL == 2)
                    atomicAdd(reinterpret_cast<unsigned int *>(cell), (unsigned int)vr);
                else
                    atomicAdd(reinterpret_cast<unsigned long long *>(cell), (unsigned long long)(long long)vr);
            }
        }
    };
    int t = 0;
    for (; t + 1 < nT; t += 2) {  // two tiles per trip: the accumulator sets swap roles
        one_iter(t, accA, accB);
        one_iter(t + 1, accB, accA);
    }
    if (t < nT) {
        one_iter(t, accA, accB);
        emit_last(accB);
    } else if (nT > 0) {
        emit_last(accA);
    }
    dma_drain();
    __syncthreads();
#ifdef TSR_STAMP
    if (a.dbg && lane == 0)
        for (int i = 0; i < 6; ++i) a.dbg[((size_t)blockIdx.x * kWaves + wave) * 6 + i] = st_sum[i];
#endif
    const int nF = (TSR_ABL & 48) ? 0 : (nOut + 31) >> 5;
    const int f0 = max(nT - lag, 0);  // tiles < f0 went through phase AB inside the loop
    if (!(TSR_ABL & 4))
        for (int f = f0; f < nF; ++f)
            for (int j = 0; j < njs; ++j) fin_ab(f, j);
    __syncthreads();
    if (!(TSR_ABL & 4))
        for (int f = max(f0 - 1, 0); f < nF; ++f)  // tile f0-1 still awaits its phase C
            if (wave == (f & (kW - 1))) fin_c_store(f, fin_c_read(f));
}

// (G > 1) sum the fragments of every template, then the reference epilogue.
__global__ __launch_bounds__(256) void frag_epilogue_kernel(
    const float *__restrict__ raw_frag, int64_t P, int M, const int32_t *__restrict__ frag_ptr,
    const int32_t *__restrict__ frag_idx, const int32_t *__restrict__ L, const float *__restrict__ cf,
    const int32_t *__restrict__ utt_end, float *__restrict__ scores, int32_t *__restrict__ ids,
    float *__restrict__ all)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t end = utt_end[p];
    float best = 0.0f;
    int besti = 0;
    for (int m = 0; m < M; ++m) {
        float r = 0.0f;
        for (int f = frag_ptr[m]; f < frag_ptr[m + 1]; ++f) r += raw_frag[(int64_t)frag_idx[f] * P + p];
        if (p + L[m] > end) r = 0.0f;
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

struct FastBank {
    int NL = 0;  // int8 limbs per weight
    int NKB = 0, KS = 0, CT = 0, G = 0, nkb_total = 0, n_coltiles = 0, n_frag = 0;
    i32x4 *d_Wl = nullptr;
    int n_phase = 1;       // K-phases: feature ranges of NKB*32, one kernel pass each
    size_t wl_phase = 0;   // i32x4 elements of Wl per phase
    int32_t *d_col_slot = nullptr, *d_col_tau = nullptr, *d_g_nslots = nullptr;
    int32_t *d_frag_ptr = nullptr, *d_frag_idx = nullptr;
    SlotMeta *d_slots = nullptr;
    int max_slots = 0;
};

template <typename T>
int upload(tsr_ctx *ctx, T **dptr, const std::vector<T> &v)
{
    *dptr = nullptr;
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(dptr), std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty()) TSR_HIP(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return TSR_OK;
}

template <int NKB, int NL, bool FLAT, int MODE, int CPW>
int launch5(tsr_ctx *ctx, const FastArgs &args, int grid, size_t lds)
{
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fast_bank_kernel<NKB, NL, FLAT, MODE, CPW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((fast_bank_kernel<NKB, NL, FLAT, MODE, CPW>), dim3(grid), dim3(kThreads / CPW), lds,
                       ctx->stream, args);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// CPW = 1 (eight waves of one column tile, two per SIMD) is what ships.  CPW = 2 -- four waves of
// two column tiles, one per SIMD -- passes every test but is slower on MI355X (config 2: 47 us vs
// 36.5: with nothing else on the SIMD every LDS wait of the finishing pass and every DMA issue
// stalls the matrix pipe); build with -DTSR_CPW2 and run with TSR_CPW=2 to try it.
template <int NKB, int NL, bool FLAT, int MODE>
int launch4(tsr_ctx *ctx, const FastArgs &args, int grid, size_t lds)
{
#ifdef TSR_CPW2
    if constexpr (NL == 2) {
        static const bool wide = [] {
            const char *e = getenv("TSR_CPW");
            return e && e[0] == '2';
        }();
        if (wide) return launch5<NKB, NL, FLAT, MODE, 2>(ctx, args, grid, lds);
    }
#endif
    return launch5<NKB, NL, FLAT, MODE, 1>(ctx, args, grid, lds);
}

template <int NKB, int NL, bool FLAT>
int launch3(tsr_ctx *ctx, const FastArgs &args, int grid, size_t lds)
{
    if (args.raw_frag) return launch4<NKB, NL, FLAT, 2>(ctx, args, grid, lds);
    if (args.all) return launch4<NKB, NL, FLAT, 1>(ctx, args, grid, lds);
    return launch4<NKB, NL, FLAT, 0>(ctx, args, grid, lds);
}

template <int NKB>
int launch(tsr_ctx *ctx, const FastArgs &args, int grid, size_t lds, int NL, bool flat)
{
    if (NL == 2) return flat ? launch3<NKB, 2, true>(ctx, args, grid, lds) : launch3<NKB, 2, false>(ctx, args, grid, lds);
    return flat ? launch3<NKB, 3, true>(ctx, args, grid, lds) : launch3<NKB, 3, false>(ctx, args, grid, lds);
}

}  // namespace

static void fast_bank_free_one(tsr_bank *bank, int which)
{
    FastBank *fb = static_cast<FastBank *>(bank->fast[which]);
    if (!fb) return;
    (void)hipSetDevice(bank->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(fb->d_Wl);
    (void)hipFree(fb->d_col_slot);
    (void)hipFree(fb->d_col_tau);
    (void)hipFree(fb->d_g_nslots);
    (void)hipFree(fb->d_frag_ptr);
    (void)hipFree(fb->d_frag_idx);
    (void)hipFree(fb->d_slots);
    delete fb;
    bank->fast[which] = nullptr;
}

void tsr_fast_bank_free(tsr_bank *bank)
{
    fast_bank_free_one(bank, 0);
    fast_bank_free_one(bank, 1);
}

// NL = 2: 16-bit fixed point (two int8 limbs, int32 ring); NL = 3: 23-bit (three limbs,
// int64 ring).  Built lazily, cached in bank->fast[NL - 2].
int tsr_fast_bank_build(tsr_ctx *ctx, tsr_bank *bank, int NL)
{
    const int M = bank->M;
    const int64_t D = bank->D;
    const int nkb_total = (int)((D + 31) / 32);
    // Feature rows wider than one wave's register budget (kMaxNKB K-blocks) are split into
    // K-phases: each phase is a complete kernel pass over its own feature range with its own
    // register-stationary limbs; the phases' partial raw scores are summed by the epilogue.
    const int n_phase = (nkb_total + kMaxNKB - 1) / kMaxNKB;
    const int NKB = (nkb_total + n_phase - 1) / n_phase;
    const int KS = 1;
    const int CT = kWaves / KS;
    const int64_t C = bank->sumL;
    const int n_coltiles = (int)((C + 31) / 32);
    const int G = (n_coltiles + CT - 1) / CT;

    // ---- per-template scale and the int8 limbs ------------------------------------------------
    // NL = 3: q = rint(W * 2^e) with |q| <= 2^22 (power-of-two scale, exactly invertible).
    // NL = 2: q = rint(W * qmax / max|W|), qmax = min(127*256, (2^31-1) / (L_m * D)) so that the
    //         int32 ring cannot overflow even if every feature of every tap is set.
    std::vector<double> inv_scale(M);
    std::vector<int8_t> limbs((size_t)NL * C * D);  // [limb][column][d]
    std::vector<int32_t> col_tmpl((size_t)C), col_tau_v((size_t)C);
    {
        int64_t woff = 0, c0 = 0;
        for (int m = 0; m < M; ++m) {
            const int64_t n = (int64_t)bank->L[m] * D;
            const float *w = bank->W_cat.data() + woff;
            float mx = 0.0f;
            for (int64_t i = 0; i < n; ++i) mx = std::max(mx, fabsf(w[i]));
            double sc = 1.0;
            if (mx > 0.0f) {
                if (NL == 3) {
                    int ex;
                    frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)  =>  mx * 2^(22-ex) in [2^21, 2^22)
                    sc = ldexp(1.0, std::min(22 - ex, 120));  // keep 2^-e a normal f32
                } else {
                    const double qmax = std::min<double>(127.0 * 256.0, floor(2147483647.0 / (double)n));
                    if (qmax < 1.0) {
                        tsr_set_error("TSR_FAST: template %d has %lld taps; too many for the int32 ring", m, (long long)n);
                        return TSR_ERR_UNSUPPORTED;
                    }
                    sc = qmax / (double)mx;
                }
            }
            inv_scale[m] = 1.0 / sc;
            for (int tau = 0; tau < bank->L[m]; ++tau) {
                const int64_t c = c0 + tau;
                col_tmpl[(size_t)c] = m;
                col_tau_v[(size_t)c] = tau;
                for (int64_t d = 0; d < D; ++d) {
                    int q = (int)llrint((double)w[tau * D + d] * sc);
                    for (int l = 0; l < NL; ++l) {  // balanced base-256 digits, last one keeps the rest
                        int digit = (l == NL - 1) ? q : ((q + 128) & 255) - 128;
                        limbs[((size_t)l * C + c) * D + d] = (int8_t)digit;
                        q = (q - digit) >> 8;
                    }
                }
            }
            woff += n;
            c0 += bank->L[m];
        }
    }
    // ---- column groups -> fragment slots ------------------------------------------------------
    // A fragment = (K-phase, column group, template): the unit of partial raw score a workgroup
    // produces when the bank does not fit one pass of one workgroup.  Fragment rows are numbered
    // phase-major; the epilogue sums a template's fragments in that fixed order.
    const int64_t Cpad = (int64_t)G * CT * 32;
    std::vector<int32_t> col_slot((size_t)Cpad, -1), col_tau((size_t)Cpad, 0);
    std::vector<SlotMeta> slots((size_t)n_phase * G * kMaxSlots);
    memset(slots.data(), 0, slots.size() * sizeof(SlotMeta));
    std::vector<int32_t> g_nslots(G, 0);
    std::vector<std::vector<int>> frags_of(M);
    int max_slots = 0, n_frag = 0;
    const bool single = (n_phase == 1 && G == 1);
    for (int g = 0; g < G; ++g) {
        int nslots = 0, last_m = -1;
        for (int64_t c = (int64_t)g * CT * 32; c < std::min<int64_t>(C, (int64_t)(g + 1) * CT * 32); ++c) {
            const int m = col_tmpl[(size_t)c];
            if (m != last_m) {
                if (nslots == kMaxSlots) {
                    tsr_set_error("TSR_FAST: more than %d templates share one column group", kMaxSlots);
                    return TSR_ERR_UNSUPPORTED;
                }
                ++nslots;
                last_m = m;
            }
            col_slot[(size_t)c] = nslots - 1;
            col_tau[(size_t)c] = col_tau_v[(size_t)c];
        }
        g_nslots[g] = nslots;
        max_slots = std::max(max_slots, nslots);
    }
    for (int ph = 0; ph < n_phase; ++ph)
        for (int g = 0; g < G; ++g) {
            int sidx = -1, last_m = -1;
            for (int64_t c = (int64_t)g * CT * 32; c < std::min<int64_t>(C, (int64_t)(g + 1) * CT * 32); ++c) {
                const int m = col_tmpl[(size_t)c];
                if (m == last_m) continue;
                last_m = m;
                ++sidx;
                SlotMeta &sm = slots[((size_t)ph * G + g) * kMaxSlots + sidx];
                sm.inv_scale = (float)inv_scale[m];
                sm.cf = (float)bank->c[m];
                sm.L = bank->L[m];
                sm.tf = single ? m : n_frag;  // template id (fused outputs) or fragment row
                frags_of[m].push_back(n_frag);
                ++n_frag;
            }
        }
    std::vector<int32_t> frag_ptr(M + 1, 0), frag_idx;
    for (int m = 0; m < M; ++m) {
        for (int f : frags_of[m]) frag_idx.push_back(f);
        frag_ptr[m + 1] = (int32_t)frag_idx.size();
    }
    // ---- limb fragments in register order: [phase][coltile][ks][kb][limb][lane] x 16 bytes ------
    const size_t wl_phase = (size_t)G * CT * KS * NKB * NL * 64;
    std::vector<i32x4> Wl(wl_phase * n_phase);
    memset(Wl.data(), 0, Wl.size() * sizeof(i32x4));
    for (int ph = 0; ph < n_phase; ++ph)
        for (int64_t tile = 0; tile < (int64_t)G * CT; ++tile)
            for (int kb = 0; kb < NKB; ++kb)
                for (int l = 0; l < NL; ++l)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int64_t c = tile * 32 + (lane & 31);
                        if (c >= C) continue;
                        int8_t *dst = reinterpret_cast<int8_t *>(
                            &Wl[ph * wl_phase + (((size_t)tile * NKB + kb) * NL + l) * 64 + lane]);
                        const int64_t k0 = ((int64_t)ph * NKB + kb) * 32 + 16 * (lane >> 5);
                        for (int e = 0; e < 16; ++e)
                            if (k0 + e < D) dst[e] = limbs[((size_t)l * C + c) * D + k0 + e];
                    }
    FastBank *fb = new FastBank;
    fb->NL = NL;
    fb->NKB = NKB;
    fb->KS = KS;
    fb->CT = CT;
    fb->G = G;
    fb->nkb_total = nkb_total;
    fb->n_coltiles = n_coltiles;
    fb->n_frag = n_frag;
    fb->max_slots = max_slots;
    fb->n_phase = n_phase;
    fb->wl_phase = wl_phase;
    bank->fast[NL - 2] = fb;
    int s = upload(ctx, &fb->d_Wl, Wl);
    if (s == TSR_OK) s = upload(ctx, &fb->d_col_slot, col_slot);
    if (s == TSR_OK) s = upload(ctx, &fb->d_col_tau, col_tau);
    if (s == TSR_OK) s = upload(ctx, &fb->d_g_nslots, g_nslots);
    if (s == TSR_OK) s = upload(ctx, &fb->d_frag_ptr, frag_ptr);
    if (s == TSR_OK) s = upload(ctx, &fb->d_frag_idx, frag_idx);
    if (s == TSR_OK) s = upload(ctx, &fb->d_slots, slots);
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    if (s != TSR_OK) fast_bank_free_one(bank, NL - 2);
    return s;
}

// Partition the frame stream into `nseg` segments (one per workgroup of a column group).  A cut
// inside an utterance makes the segment also read the Lmax-1 frames its last windows overhang;
// a cut at an utterance boundary needs no halo (windows never span utterances).  The launch
// lasts as long as its longest segment, so cuts start uniform and are moved to a boundary only
// when that does not lengthen the longest segment.
static int build_plan(tsr_ctx *ctx, tsr_corpus *cp, int nseg, int Lmax)
{
    const int key = nseg * 65536 + Lmax;
    if (cp->d_plan && cp->plan_key == key) return TSR_OK;
    const int64_t P = cp->P;
    std::vector<int32_t> plan((size_t)3 * nseg);
    const std::vector<int64_t> &off = cp->offsets;
    auto is_boundary = [&](int64_t p) { return std::binary_search(off.begin(), off.end(), p); };
    auto tiles_of = [&](int64_t o0, int64_t o1) -> int64_t {  // row tiles a segment costs
        if (o1 <= o0) return 0;
        const int64_t rows = is_boundary(o1) ? o1 - o0 : std::min<int64_t>(P, o1 + Lmax - 1) - o0;
        return (rows + 31) / 32;
    };
    // 1) uniform cuts: the launch lasts as long as its longest segment (in 32-row tiles)
    std::vector<int64_t> cuts(nseg + 1, 0);
    for (int i = 1; i < nseg; ++i) cuts[i] = (int64_t)llrint((double)P * i / nseg);
    cuts[nseg] = P;
    int64_t tmax = 0;
    for (int i = 0; i < nseg; ++i) tmax = std::max(tmax, tiles_of(cuts[i], cuts[i + 1]));
    // 2) snap a cut to a nearby utterance boundary (no halo rows needed there) whenever that
    //    keeps both neighbouring segments within the uniform plan's longest segment
    for (int i = 1; i < nseg; ++i) {
        auto it = std::lower_bound(off.begin(), off.end(), cuts[i]);
        int64_t cand[2] = {-1, -1};
        if (it != off.end()) cand[0] = *it;
        if (it != off.begin()) cand[1] = *(it - 1);
        int64_t best = -1;
        for (int64_t c : cand) {
            if (c < 0 || c <= cuts[i - 1] || c >= cuts[i + 1]) continue;
            if (tiles_of(cuts[i - 1], c) > tmax || tiles_of(c, cuts[i + 1]) > tmax) continue;
            if (best < 0 || llabs(c - cuts[i]) < llabs(best - cuts[i])) best = c;
        }
        if (best >= 0) cuts[i] = best;
    }
    for (int i = 0; i < nseg; ++i) {
        const int64_t o0 = cuts[i], o1 = cuts[i + 1];
        plan[(size_t)i] = (int32_t)o0;
        plan[(size_t)nseg + i] = (int32_t)o1;
        plan[(size_t)2 * nseg + i] =
            (int32_t)((o1 > o0 && !is_boundary(o1)) ? std::min<int64_t>(P, o1 + Lmax - 1) : o1);
    }
    if (cp->d_plan) {
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        TSR_HIP(hipFree(cp->d_plan));
        cp->d_plan = nullptr;
    }
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&cp->d_plan), plan.size() * 4));
    TSR_HIP(hipMemcpyAsync(cp->d_plan, plan.data(), plan.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    cp->plan_key = key;
    cp->plan_nseg = nseg;
    return TSR_OK;
}

int tsr_launch_fast_bank(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, int flags,
                         float *d_scores, int32_t *d_ids, float *d_all)
{
    const int NL = (flags & TSR_W24) ? 3 : 2;
    const FastBank *fb = static_cast<const FastBank *>(bank->fast[NL - 2]);
    if (!fb) return TSR_ERR_UNSUPPORTED;
    const int64_t P = cp->P;
    if (P == 0) return TSR_OK;
    const int Dp = cp->Dp;
    // LDS row: the bytes one pass needs of a frame (the whole resident row, or one K-phase of
    // it), as an odd number of 16-byte slots so that b128 fragment reads are conflict-free
    const int k_span = fb->KS * fb->NKB * 32;
    int row_bytes = fb->n_phase == 1 ? Dp : k_span;  // source bytes copied per frame and pass
    int SB = row_bytes;
    if (((SB >> 4) & 1) == 0) SB += 16;
    // a tile buffer: 32 rows, rounded up to whole 1 KiB DMA pieces, plus the bytes the last
    // row's K-block reads may run past it (zero weights there, any finite byte is fine)
    int tile_bytes = 32 * SB + std::max(0, k_span + 16 - SB);
    tile_bytes = (tile_bytes + 1023) / 1024 * 1024;
    // a frame is final once the tile holding its last tap (Lmax-1 rows further) is emitted;
    // emission trails the MFMAs by one tile, finishing trails emission by one barrier and the
    // utt_end row of tile t+2 is DMA'd while tile t-lag-2 is finished: lag + 5 tiles
    int lag = (bank->Lmax - 1 + 31) / 32;
    int ring = 32;
    while (ring < 32 * (lag + 5)) ring *= 2;
    const size_t lds = (size_t)3 * tile_bytes + (size_t)kMaxSlots * sizeof(SlotMeta) +
                       (size_t)(ring + 64) * 4 + (size_t)ring * 8 +
                       (size_t)(fb->max_slots + 1) * (ring + 33) * (NL == 2 ? 4 : 8) + 16;
    if (lds > 160 * 1024) {
        tsr_set_error("TSR_FAST: D=%lld, Lmax=%d need %zu bytes of LDS", (long long)bank->D, bank->Lmax, lds);
        return TSR_ERR_UNSUPPORTED;
    }
    // segments: one workgroup per (segment, column group); one wave of workgroups per launch
    int nseg = std::max(1, ctx->cus / fb->G);
    const int64_t min_rows = 64;  // do not cut the stream finer than two tiles per segment
    nseg = (int)std::max<int64_t>(1, std::min<int64_t>(nseg, (P + min_rows - 1) / min_rows));
    TSR_TRY(build_plan(ctx, cp, nseg, bank->Lmax));

    const bool fragments = fb->n_phase > 1 || fb->G > 1;
    void *raw_frag = nullptr;
    if (fragments) TSR_TRY(tsr_scratch(ctx, 0, (size_t)fb->n_frag * P * sizeof(float), &raw_frag));
    const int grid = nseg * fb->G;
    const bool flat = (SB == Dp) && fb->n_phase == 1;
    unsigned long long *d_dbg = nullptr;
#ifdef TSR_STAMP
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&d_dbg), (size_t)grid * kWaves * 6 * 8));
    TSR_HIP(hipMemsetAsync(d_dbg, 0, (size_t)grid * kWaves * 6 * 8, ctx->stream));
#endif
    for (int ph = 0; ph < fb->n_phase; ++ph) {
        FastArgs a;
        memset(&a, 0, sizeof a);
        a.X = cp->d_feats + (size_t)ph * k_span;  // this phase's feature range of every row
        a.Dp = Dp;
        a.SB = SB;
        a.src_slots = row_bytes >> 4;
        a.tile_bytes = tile_bytes;
        a.G = fb->G;
        a.CT = fb->CT;
        a.KS = fb->KS;
        a.nseg = nseg;
        a.seg = cp->d_plan;
        a.Wl = fb->d_Wl + (size_t)ph * fb->wl_phase;
        a.col_slot = fb->d_col_slot;
        a.col_tau = fb->d_col_tau;
        a.slots = fb->d_slots + (size_t)ph * fb->G * kMaxSlots;
        a.g_nslots = fb->d_g_nslots;
        a.utt_end = cp->d_utt_end;
        a.P = P;
        a.ring = ring;
        a.lag = lag;
        a.lmax1 = bank->Lmax - 1;
        a.dbg = d_dbg;
        if (fragments) {
            a.raw_frag = (float *)raw_frag;
        } else {
            a.scores = d_scores;
            a.ids = d_ids;
            a.all = d_all;
        }
        int s;
        switch (fb->NKB) {
        case 1: s = launch<1>(ctx, a, grid, lds, NL, flat); break;
        case 2: s = launch<2>(ctx, a, grid, lds, NL, flat); break;
        case 3: s = launch<3>(ctx, a, grid, lds, NL, flat); break;
        case 4: s = launch<4>(ctx, a, grid, lds, NL, flat); break;
        case 5: s = launch<5>(ctx, a, grid, lds, NL, flat); break;
        case 6: s = launch<6>(ctx, a, grid, lds, NL, flat); break;
        case 7: s = launch<7>(ctx, a, grid, lds, NL, flat); break;
        case 8: s = launch<8>(ctx, a, grid, lds, NL, flat); break;
        case 9: s = launch<9>(ctx, a, grid, lds, NL, flat); break;
        case 10: s = launch<10>(ctx, a, grid, lds, NL, flat); break;
        case 11: s = launch<11>(ctx, a, grid, lds, NL, flat); break;
        default: s = launch<12>(ctx, a, grid, lds, NL, flat); break;
        }
        TSR_TRY(s);
    }
#ifdef TSR_STAMP
    if (const char *path = getenv("TSR_STAMP_OUT")) {
        std::vector<unsigned long long> h((size_t)grid * kWaves * 6);
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        TSR_HIP(hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *fp = fopen(path, "w")) {
            for (size_t w = 0; w < (size_t)grid * kWaves; ++w)
                fprintf(fp, "%zu %zu %llu %llu %llu %llu %llu %llu\n", w / kWaves, w % kWaves, h[w * 6], h[w * 6 + 1],
                        h[w * 6 + 2], h[w * 6 + 3], h[w * 6 + 4], h[w * 6 + 5]);
            fclose(fp);
        }
    }
    (void)hipFree(d_dbg);
#endif
    if (fragments) {
        hipLaunchKernelGGL(frag_epilogue_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                           ctx->stream, (const float *)raw_frag, P, bank->M, fb->d_frag_ptr, fb->d_frag_idx,
                           bank->d_L, bank->d_cf, cp->d_utt_end, d_scores, d_ids, d_all);
        TSR_HIP(hipGetLastError());
    }
    return TSR_OK;
}
