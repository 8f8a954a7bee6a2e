// detect_mx.hip -- TSR_FAST | TSR_MX: the bank pass as a skew-free Toeplitz product on the block-scaled matrix
// instruction v_mfma_scale_f32_16x16x128_f8f6f4, eight templates per launch.
// (reference: compute_likelihood_linear_filter.pyx:162-175 for the sums, get_train_data.py:962-992 for the bank epilogue)
//
// Formulation.  A tile is 16 *pairs* of window starts: pair i stands for the starts t0_i and t0_i + 1 (t0_i even
// inside its utterance), n = 8 s + m for (shift s, template m) -- the weights are the matrix instruction's A operand
// (rows n), the frames its B operand (columns i) -- and
//     acc[(s, m)][i] = sum over tap rows rho = 0 .. Lmax and features d of  X[t0_i + rho][d] * W_m[rho - s][d]
// (weights zero outside 0 <= rho - s < L_m): the score of template m at start t0_i + s, complete in ONE accumulator
// element -- no skew reduction, no ring, no atomics; the price is Lmax + 1 tap rows instead of Lmax.  Rows are only
// the starts that some template can score (t <= T - Lmin), so the 31 tail frames of a 200-frame utterance cost nothing.
//
// Operands.  X is binary, so it is an exact fp4 (e2m1) operand: nibble 0x2 = 1.0.  The weights are fixed point with a
// per-template scale, |q| <= 4095, split into three radix-16 digits in -16 .. 16, each an exact fp6 (e2m3) value
// digit / 8; the instruction's block scales (E8M0, one per lane and 32 K-elements) carry 2^3, 2^7 and 2^11, so the three
// limb products meet in ONE f32 accumulator as the exact integer sum (exact while it stays below 2^24; banks that can go
// beyond take their K-blocks in fixed shares, so that the rounding is the same in every launch).
// fp4 x fp6 issues at four times the bf16 rate (MI355X_MICROARCH.md; tools/ubench/mx_probe.hip measured 16.4 cycles per
// instruction and SIMD), three limbs therefore cost 0.75 of the two int8 limbs of the tap-pair kernel -- at 13 instead of
// 16 bits of weight (include/tsr.h says what that means for the error).
//
// Work split.  One workgroup per CU takes MX_NT = 21 tiles (672 window starts) and holds ALL their frames in LDS
// as nibbles (expanded from the corpus' resident bit rows: 48 B per frame from HBM instead of 368), column-major:
// [frame parity][16-byte chunk of the row][row] so that the 16 columns of a B fragment are 16 consecutive 16-byte slots
// (conflict-free ds_read_b128 at any alignment when the rows per array are a multiple of 16) and a fragment's address
// is (per-lane tile base) + (offset of the K-block's K-quarter, from a table).  The eight waves split K: a K-block (any
// four 32-feature chunks: features of a tap row, or the leftover features of four tap rows packed into one chunk) goes to
// one wave -- drawn from a counter in LDS, or in fixed shares -- which streams its three digit fragments from L2 straight
// into registers and walks all 21 tiles with them.  The eight partial accumulators of a tile meet once, at the end,
// through LDS; the tile's owner applies scale, tail rule, + c and the max / first arg-max over templates.
//
// Wider forms of the same kernel: rows of more than 512 features go through staging and K loop once per feature phase
// with one accumulator set (PHASES; the phases are those of the corpora's resident bit form); banks of more than eight
// templates take one launch per slice of eight, the later ones merging into the result arrays.
#include <algorithm>
#include <cmath>

#include "mx_plan.h"
#include "tsr_internal.h"

namespace {

using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int MX_NT = 21;         // row tiles per workgroup (accumulators: 4 registers each)
constexpr int MX_WAVES = 8;
constexpr int MX_LDS = 163840;    // the whole LDS of a CU
constexpr int MX_MAX_KBF = 4;     // feature rows of up to 512 features
constexpr int MX_MAX_SLICES = 32;  // banks of up to 256 templates (eight per launch)
constexpr int MX_QMAX = 4095;     // |fixed-point weight|

struct MxCol {     // one of the 16 columns: (shift s = n >> 3, template m = n & 7)
    float inv_scale, c;
    int32_t L, id;
};
struct MxArgs {
    const uint32_t *bits;     // [rows][DW] resident bit rows at virtual positions
    const unsigned char *B;   // this launch's slice: [phase][NKB][3 digits][1536]: 64 lanes x 16 B, then 64 lanes x 8 B
    const MxGroup *groups;
    const int4 *wg;           // per workgroup: first tile, tiles, first virtual row, rows to stage
    const MxCol *cols;        // [16] of this launch's slice
    float *scores;
    int32_t *ids;
    float *all;               // all-scores pass: [M][P] per-template scores instead of the maximum (scores / ids unused)
    int64_t P;
    const int32_t *kq;        // [NKB][4] LDS byte offset of the chunk that feeds K-quarter q of K-block kb (frame row 0 of a pair)
    int n_groups, NKB, KBF, NR, Lmin, idmax;
    int CH, nfull, packed, M;   // chunk arrays per frame parity; full 32-feature chunks of a row; leftover features of four tap rows share a chunk; templates of this launch
    int slow_cap;             // K-blocks a wave of the slower half (waves 4..7) takes at most
    int merge;                // max mode, second and later slices of a bank of more than eight templates: keep the better of old and new
    int fixed;                // K-blocks in fixed shares (wave w: w, w + 8, ...): banks whose sums may pass 2^24, where the order of the f32 additions shows
    int n_phase, DWr;         // feature phases (rows wider than 512 features: one staging + K loop per phase, one accumulator set); dwords per bit row
    long long bits_phase, B_phase;   // dwords / bytes from one phase's bit rows / fragments to the next
    float cmax;
    unsigned long long *dbg;  // -DTSR_STAMP builds only: [workgroup][wave][16] s_memtime stamps
};

// Development stamps (never in the shipped build): -DTSR_STAMP records s_memtime at the phase boundaries of every wave
// and the launch dumps them to the file $TSR_STAMP_OUT (tools/mx_stamp_report.py).
#ifdef TSR_STAMP
#define MX_STAMP(i)                                                                            \
    {                                                                                          \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        if (lane == 0) a.dbg[((size_t)blockIdx.x * MX_WAVES + wave) * 16 + (i)] = t_;           \
    }
#else
#define MX_STAMP(i)
#endif

struct MxBank {
    int KBF = 0, ROWS = 0, NR = 0, Lmin = 0, Lmax = 0, idmax = 0, CH = 0, nfull = 0, packed = 0, n_phase = 1, pw = 0;
    int n_slice = 1;          // groups of eight templates
    std::vector<int> slice_NKB;             // K-blocks of a slice (its longest filter + 1 tap rows)
    std::vector<size_t> slice_kq, slice_B;  // where the slice's offset table (ints) / fragments (bytes) begin
    bool exact = true;        // no sum of the bank's weights over any set of taps reaches 2^24
    int32_t *d_kq = nullptr;
    float cmax = 0;
    unsigned char *d_B = nullptr;
    MxCol *d_cols = nullptr;
};

struct MxCorpus {   // derived forms of a corpus: bit rows (once), the tile plan of one (Lmin, ROWS, NR)
    uint32_t *d_bits = nullptr;
    const uint32_t *ext_bits = nullptr;   // the corpus' own bit form (wide rows), not owned
    int DW = 0, DWr = 0;
    long long bits_phase = 0;
    int64_t Pv = 0, rows_alloc = 0;
    std::vector<int64_t> voff;
    int plan_Lmin = -1, plan_ROWS = -1, plan_NR = -1;
    int plan_target = MX_NT;   // tiles per workgroup the plan aimed at (MX_NT when the corpus fits one round)
    MxGroup *d_groups = nullptr;
    int4 *d_wg = nullptr;
    int n_groups = 0, n_wg = 0;
    bool plan_unsupported = false;
};

// 32 resident bytes (0 / 1) -> one dword of bits: eight bytes become eight bits by one multiply (byte i lands on bit 56 + i,
// no two partial products share a bit)
__device__ inline uint32_t mx_pack32(const uint8_t *row, int nb /* bytes inside the row: a multiple of 16 */)
{
    if (nb <= 0) return 0;
    const uint4 x0 = *reinterpret_cast<const uint4 *>(row);
    uint4 x1 = make_uint4(0, 0, 0, 0);
    if (nb > 16) x1 = *reinterpret_cast<const uint4 *>(row + 16);
    auto pack8 = [](unsigned lo32, unsigned hi32) {
        const unsigned long long x = ((unsigned long long)hi32 << 32) | lo32;
        return (uint32_t)((x * 0x0102040810204080ull) >> 56);
    };
    return pack8(x0.x, x0.y) | (pack8(x0.z, x0.w) << 8) | (pack8(x1.x, x1.y) << 16) | (pack8(x1.z, x1.w) << 24);
}

__device__ inline int64_t mx_utt_of(const int64_t *voff, int64_t N, int64_t v)   // last u with voff[u] <= v
{
    int64_t lo = 0, hi = N;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (voff[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// bit rows at virtual positions from the resident byte rows (one-off per corpus)
__global__ void mx_pack_bits_kernel(const uint8_t *feats, int Dp, const int64_t *off, const int64_t *voff, int64_t N, int DW,
                                    int64_t Pv, uint32_t *bits)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Pv * DW) return;
    const int64_t v = idx / DW;
    const int w = (int)(idx - v * DW);
    const int64_t u = mx_utt_of(voff, N, v);
    const int64_t t = v - voff[u], T = off[u + 1] - off[u];
    bits[idx] = t < T ? mx_pack32(feats + (off[u] + t) * (int64_t)Dp + 32 * w, Dp - 32 * w) : 0u;
}

// Wide rows, corpora with an utterance of odd length: this file's own copy of the bit form, phase-major like the corpus'
// ([phase][row][pw dwords]) with every utterance on an even row -- from the corpus' bit form (same geometry) or its byte rows.
__global__ void mx_pack_phase_kernel(const uint8_t *feats, int Dp, const uint32_t *src_bits, int64_t src_rows, const int64_t *off,
                                     const int64_t *voff, int64_t N, int pw, int n_phase, int64_t Pv, int64_t rows_alloc, uint32_t *bits)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = Pv * pw;
    if (idx >= plane * n_phase) return;
    const int ph = (int)(idx / plane);
    const int64_t r = idx - (int64_t)ph * plane;
    const int64_t v = r / pw;
    const int w = (int)(r - v * pw);
    const int64_t u = mx_utt_of(voff, N, v);
    const int64_t t = v - voff[u], T = off[u + 1] - off[u];
    uint32_t word = 0;
    if (t < T) {
        const int64_t p = off[u] + t;
        const int gw = ph * pw + w;   // dword of the whole row
        word = src_bits ? src_bits[((int64_t)ph * src_rows + p) * pw + w] : mx_pack32(feats + p * (int64_t)Dp + 32 * gw, Dp - 32 * gw);
    }
    bits[((int64_t)ph * rows_alloc + v) * pw + w] = word;
}

constexpr int MX_STASH = 4096;                    // behind the LDS image: the workgroup's group records
constexpr int MX_GROUPS = MX_NT * 16;             // 336 x 3 ints = 4032 bytes
constexpr int MX_IMAGE = MX_LDS - MX_STASH;       // bytes the frame image may take
#ifndef MX_ABL
#define MX_ABL 0
#endif
constexpr int MX_DEPTH = 7;                       // A fragments in flight per wave
static_assert(MX_NT % MX_DEPTH == 0 && MX_NT % 3 == 0, "the fragment ring must line up across K-blocks: tile j lives in A[j % MX_DEPTH]");

// FULL: the row has exactly 4 KBF chunk arrays (every 16-byte piece of a bit row fills four of them: no per-chunk checks)
// PHASES: rows wider than 512 features, staged and multiplied in feature phases (only <3, true, true, .> is built)
// PARTIAL: the launch's workgroups hold fewer than MX_NT tiles (a corpus spread over more than one round of workgroups):
// the matrix work of absent tile groups is skipped -- a branch per tile group, which costs full workgroups ~4 %, hence a variant
template <int KBF, bool FULL, bool PHASES = false, bool PARTIAL = false>
__global__ __launch_bounds__(MX_WAVES * 64, 2) void mx_bank_kernel(const MxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int CH = a.CH, nfull = a.nfull;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if MX_ABL & 32   // development ablation: config 2's plan computed instead of loaded (what the dependent load of the plan costs)
    int4 wg;
    {
        const int b = blockIdx.x, g0 = 336 * b, g1 = min(336 * b + 335, a.n_groups - 1);
        const int v0 = 200 * (g0 / 85) + 2 * (g0 % 85), v1 = 200 * (g1 / 85) + 2 * (g1 % 85);
        wg = make_int4(21 * b, min(21, (a.n_groups + 15) / 16 - 21 * b), v0, v1 + 33 - v0);
    }
#else
    const int4 wg = a.wg[blockIdx.x];
#endif
    const int tile_beg = wg.x, nt = wg.y, v_lo = wg.z, nfr = wg.w;
    const int NR = a.NR;
    int *const g_row = reinterpret_cast<int *>(lds + MX_IMAGE);   // [336] LDS row of the group's first frame
    int *const g_p0 = g_row + MX_GROUPS, *const g_rem = g_p0 + MX_GROUPS;
    unsigned *const next_kb = reinterpret_cast<unsigned *>(g_rem + MX_GROUPS);   // the next K-block nobody has taken yet

    MX_STAMP(0)
#ifdef TSR_STAMP
    {
        unsigned long long r_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_)::"memory");
        if (lane == 0) a.dbg[((size_t)blockIdx.x * MX_WAVES + wave) * 16 + 14] = r_;
    }
#endif
    // ---- everything this workgroup reads from memory besides the bank is requested at once: the group records (one
    //      per thread), the bit rows (five 16-byte pieces per thread), the first K-block's B fragments ----
    // accumulator register g of this lane is (shift s = lane >> 5, template m = 4 h + g, h = (lane >> 4) & 1) of the
    // window pair lane & 15: the weights are the A operand, the frames the B operand, so that the templates of a window
    // lie in registers and across ONE lane pair (lane ^ 16) and the maximum over them is three in-lane compares and one exchange
    const int s = lane >> 5, h = (lane >> 4) & 1;
    MxCol col[4];
    if constexpr (!PHASES) {   // (requested with everything else; the phase loop would only spill them: it loads them behind the loop)
#pragma unroll
        for (int g = 0; g < 4; ++g) col[g] = a.cols[4 * (lane >> 4) + g];
    }
    MxGroup G = {v_lo, 0, 0, 0};
    if (tid < MX_GROUPS) {
        const int gi = tile_beg * 16 + tid;
        if (tid < nt * 16 && gi < a.n_groups) G = a.groups[gi];
    }
    f32x4 acc[MX_NT];
#pragma unroll
    for (int j = 0; j < MX_NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nrows = (nfr + 1) >> 1;
    // Feature phases: rows of up to 512 features are one phase; wider rows (part corpora: D = 2050 as six phases of 352) go
    // through staging and K loop once per phase with the SAME accumulators, so that nothing but the final sums is exchanged.
    const int n_phase = PHASES ? a.n_phase : 1, DWr = PHASES ? a.DWr : 4 * KBF;
    for (int ph = 0; ph < n_phase; ++ph) {
    if (ph) __syncthreads();   // (everybody is done with the previous phase's image and K-block counter)
    const uint32_t *const bits = PHASES ? a.bits + (size_t)ph * a.bits_phase : a.bits;
    const unsigned char *const Bph = PHASES ? a.B + (size_t)ph * a.B_phase : a.B;
    // frames -> LDS: lane = row of one (parity, K-block-of-the-row) pass, so that a wave's stores are consecutive 16-byte
    // slots; at most 2 * KBF * NR <= 2496 pieces = five per thread
    u32x4 piece[5];
    int dst[5], cpiece[5];
    bool inside[5];
    int tidp = tid;
    if constexpr (PHASES) asm volatile("" : "+v"(tidp));   // (keeps the staging addresses from being hoisted out of the phase loop, where they would be spilled)
    // Every thread takes five pieces: past the last one it takes the last one again (the same bits to the same slot), and a
    // row past the last frame is loaded from the first one and zeroed on the way -- loads and stores are unconditional, i.e.
    // straight-line code, in which the compiler's wait counts are exact: the first piece is expanded while the others are
    // still on their way (with branches around the pieces every one of them waited for all five).
    const int n_piece = 2 * KBF * nrows;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int idx = min(tidp + it * (MX_WAVES * 64), n_piece - 1);
        const int pass = idx / nrows, r = idx - pass * nrows;
        const int plane = pass & 1, kbf = pass >> 1;
        const int fr = 2 * r + plane;
        inside[it] = fr < nfr;
        cpiece[it] = kbf;
        dst[it] = ((plane * CH + 4 * kbf) * NR + r) << 4;
        piece[it] = *reinterpret_cast<const u32x4 *>(bits + (size_t)(v_lo + (inside[it] ? fr : 0)) * DWr + 4 * kbf);
    }
    const unsigned char *bp = Bph + (size_t)wave * (3 * 1536) + lane * 16;
    i32x4 b4[3];
    i32x2 b2[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        b4[l] = *reinterpret_cast<const i32x4 *>(bp + l * 1536);
        b2[l] = *reinterpret_cast<const i32x2 *>(bp + l * 1536 + 1024 - lane * 8);
    }
    if (tid == 0) *next_kb = 3 * MX_WAVES;   // (wave w starts with the K-blocks w, w + 8 and w + 16)
    if (tid < MX_GROUPS && ph == 0) {
        g_row[tid] = (G.v0 - v_lo) >> 1;
        g_p0[tid] = G.p0;
        g_rem[tid] = G.rem;
    }
#ifdef TSR_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MX_STAMP(6)
#endif
    // bit rows expanded to fp4 nibbles (0x2 = 1.0): element 8 k + j of a 32-feature chunk is feature 4 j + k (the bank's
    // fragments use the same order), which makes a chunk four shift-and-mask pairs
#pragma unroll
    for (int it = 0; it < 5; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned x = inside[it] ? piece[it][q] : 0u;
            const u32x4 o = {(x << 1) & 0x22222222u, x & 0x22222222u, (x >> 1) & 0x22222222u, (x >> 2) & 0x22222222u};
            if (FULL || 4 * cpiece[it] + q < CH) *reinterpret_cast<u32x4 *>(lds + dst[it] + ((q * NR) << 4)) = o;
        }
    }
    MX_STAMP(1)
    if (a.packed) {
        // Leftover array: the staging above left slot r = the row's own (at most eight) leftover features in the chunk order,
        // i.e. in the low byte of each of its four dwords.  A pair whose first frame is row b reads slot b + k0 for the tap
        // rows 2 (k0 .. k0 + 3) + parity, so slot r has to hold the leftovers of rows r .. r + 3: the chunk expansion of the
        // word lo(r) | lo(r+1) << 8 | lo(r+2) << 16 | lo(r+3) << 24, which is dword by dword the OR of the four rows' slots
        // shifted by 0 / 8 / 16 / 24 bits (a nibble is four bits, a feature's index moves by 8 per row).  In place: everybody
        // reads, barrier, everybody writes.  Rows past the staged ones hold whatever was there: their weights are zero and
        // every nibble pattern is a finite fp4 value.
        __syncthreads();
        u32x4 o[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) {   // (2 * nrows <= 2 * NR <= 2560 slots)
            const int idx = tid + it * (MX_WAVES * 64);
            const int plane = idx >= nrows, r = idx - plane * nrows;
            o[it] = u32x4{0, 0, 0, 0};
            if (it * (MX_WAVES * 64) >= 2 * nrows) break;   // (wave-uniform: the rounds that have no slot at all)
            if (idx < 2 * nrows) {
                const unsigned char *slot = lds + (((plane * CH + nfull) * NR + r) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(slot + 16 * t);
                    o[it] |= (x & 0xffu) << (8 * t);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int idx = tid + it * (MX_WAVES * 64);
            const int plane = idx >= nrows, r = idx - plane * nrows;
            if (it * (MX_WAVES * 64) >= 2 * nrows) break;
            if (idx < 2 * nrows) *reinterpret_cast<u32x4 *>(lds + (((plane * CH + nfull) * NR + r) << 4)) = o[it];
        }
    }
    __syncthreads();
    MX_STAMP(2)
    // per-lane LDS offset of (tile j, pair lane & 15): the pair's first frame row
    int tb[MX_NT];
#pragma unroll
    for (int j = 0; j < MX_NT; ++j) tb[j] = g_row[j * 16 + (lane & 15)] << 4;
#ifdef MX_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    // ---- K loop: K-blocks (a K-block = 128 features of one tap row) over all tiles; the A fragments of the next
    //      MX_DEPTH tiles are in flight, across the K-block boundary too.  A wave draws its next K-block from a counter
    //      in LDS while it works on this one: the two waves of a SIMD only fill the matrix pipe together (one wave alone
    //      issues at half its rate, tools/ubench/mx_rate.hip), and with fixed shares the wave that loses the issue
    //      arbitration is left to finish alone ----
    // the chunk that feeds this lane's K-quarter of a K-block (any chunk of any tap row: a K-block is four of them)
    auto kq_of = [&](int kb) { return a.kq[(kb < a.NKB ? kb : a.NKB - 1) * 4 + (lane >> 4)]; };
    i32x4 A[MX_DEPTH];
    int koff = kq_of(wave), koff_1 = kq_of(wave + MX_WAVES);   // this K-block's and the next one's (requested two K-blocks ahead)
#pragma unroll
    for (int j = 0; j < MX_DEPTH; ++j) A[j] = *reinterpret_cast<const i32x4 *>(lds + tb[j] + koff);
    // Three register sets for the B fragments in rotation (the loop body exists three times so that no set is ever copied):
    // one K-block's MFMAs read one set, the next K-block's fragments sit in the second, and those of the K-block after
    // that land in the third, requested piece by piece between the tile groups -- one K-block of lead (the last piece
    // ~700 cycles in front of its use) still showed as a wait at every K-block.  The draw of a K-block is one more
    // K-block ahead, so that nobody waits for the atomic's return (a wave that waits issues nothing, and its SIMD
    // partner alone runs at half rate).
    int kb_1 = wave + MX_WAVES, kb_2 = wave + 2 * MX_WAVES;   // the two K-blocks after this one
    int taken = 0;
    auto kblock = [&](const i32x4 (&c4)[3], const i32x2 (&c2)[3], i32x4 (&n4)[3], i32x2 (&n2)[3]) -> bool {
        // c: this K-block's fragments; n: the set that is free now and takes those of the K-block after the next
        int kb_3 = 0;
#if MX_ABL & 8   // development ablation: fixed shares (wave w takes K-blocks w, w + 8, ...)
        kb_3 = kb_2 + MX_WAVES;
#else
        // Wide rows: fixed shares.  Their sums pass 2^24, where an f32 addition rounds, so the result depends on which wave
        // added which K-block in which order; with wave w taking K-blocks w, w + 8, ... every launch adds in the same order
        // and the scores are the same bit for bit from run to run (measured on config 3: no slower than drawing).
        if constexpr (PHASES) kb_3 = kb_2 + MX_WAVES;
        else {
        // The two waves of a SIMD do not get equal shares of its issue slots (the older one, waves 0..3, wins the
        // arbitration: tools/ubench/mx_rate.hip), so the younger half needs ~1.4 x as long for a K-block; with K-blocks drawn
        // two ahead it would still hold two of them when the pool is empty.  It stops drawing after slow_cap K-blocks and
        // the older half finishes the pool.
        ++taken;
        // (a.fixed: a bank whose sums can pass 2^24 takes fixed shares here too -- same bits from launch to launch)
        if (lane == 0 && !a.fixed && (wave < 4 || taken + 2 < a.slow_cap))
            kb_3 = (int)__hip_atomic_fetch_add(next_kb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (lane == 0) kb_3 = a.fixed ? kb_2 + MX_WAVES : a.NKB;
        }
#endif
        const bool more = kb_1 < a.NKB;
        const int koff_n = koff_1;
        const int koff_2 = kq_of(kb_2);
        const int kb_ld = kb_2 < a.NKB ? kb_2 : a.NKB - 1;
        i32x8 B[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) B[l] = i32x8{c4[l][0], c4[l][1], c4[l][2], c4[l][3], c2[l][0], c2[l][1], 0, 0};
        // three tiles at a time, their limbs interleaved: a matrix instruction that adds to the accumulator of the one
        // before it waits for its result (a lone wave then runs at half the pipe's rate); with two other tiles' between
        // them the three accumulations of a tile are 48 cycles apart
#pragma unroll
        for (int j = 0; j < MX_NT; j += 3) {
            // one piece of the fragments of the K-block after the next per tile group: a wave that issues its six loads
            // back to back sits in that sequence for as long as the data takes (vector-memory instructions are issued at
            // the rate the path delivers) and issues no matrix work meanwhile
            // (unconditional -- behind the last K-block it re-reads that one -- so that the sets are plain registers, not
            // merges of old and new values)
            if (j / 3 < 6 && !(MX_ABL & 4)) {
                const unsigned char *nb = Bph + (size_t)kb_ld * (3 * 1536) + lane * 16 + (j / 6) * 1536;
                if ((j / 3) & 1) n2[j / 6] = *reinterpret_cast<const i32x2 *>(nb + 1024 - lane * 8);
                else n4[j / 6] = *reinterpret_cast<const i32x4 *>(nb);
            }
            // (PARTIAL: a workgroup skips the matrix work of the tile groups it does not have; the ring below keeps turning: its
            // reads are cheap and harmless)
            if (!PARTIAL || j == 0 || j < nt) {
                i32x8 Af[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const i32x4 x = A[(j + u) % MX_DEPTH];
                    Af[u] = i32x8{x[0], x[1], x[2], x[3], 0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[0], Af[u], acc[j + u], 2, 4, 0, (int)0x82828282u, 0, 0x7f7f7f7f);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[1], Af[u], acc[j + u], 2, 4, 0, (int)0x86868686u, 0, 0x7f7f7f7f);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    acc[j + u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B[2], Af[u], acc[j + u], 2, 4, 0, (int)0x8a8a8a8au, 0, 0x7f7f7f7f);
            }
            // (behind the group's matrix instructions, which read the ring slots these reads refill: no copies of the slots)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int jn = j + u + MX_DEPTH;
#if MX_ABL & 1   // development ablation: no fragment reads inside the loop (wrong sums)
                (void)jn;
#elif MX_ABL & 2   // development ablation: no address arithmetic per read (wrong sums)
                A[(j + u) % MX_DEPTH] = *reinterpret_cast<const i32x4 *>(lds + (jn < MX_NT ? tb[jn] : tb[jn - MX_NT]));
#else
                A[(j + u) % MX_DEPTH] = *reinterpret_cast<const i32x4 *>(lds + (jn < MX_NT ? tb[jn] + koff : tb[jn - MX_NT] + koff_n));
#endif
            }
#if !(MX_ABL & 16)
            __builtin_amdgcn_sched_barrier(0);   // (the compiler otherwise sinks the reads to just in front of their use)
#endif
        }
        koff = koff_n;
        koff_1 = koff_2;
        kb_1 = kb_2;
        kb_2 = __builtin_amdgcn_readfirstlane(kb_3);
        return more;
    };
    i32x4 d4[3], e4[3] = {b4[0], b4[1], b4[2]};
    i32x2 d2[3], e2[3] = {b2[0], b2[1], b2[2]};
    {   // the second K-block's fragments (the first one's were requested at the top of the kernel)
        const unsigned char *nb = Bph + (size_t)(kb_1 < a.NKB ? kb_1 : a.NKB - 1) * (3 * 1536) + lane * 16;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            d4[l] = *reinterpret_cast<const i32x4 *>(nb + l * 1536);
            d2[l] = *reinterpret_cast<const i32x2 *>(nb + l * 1536 + 1024 - lane * 8);
        }
    }
    while (kblock(b4, b2, e4, e2) && kblock(d4, d2, b4, b2) && kblock(e4, e2, d4, d2)) {
    }
    }   // (feature phases)
    if constexpr (PHASES) {
#pragma unroll
        for (int g = 0; g < 4; ++g) col[g] = a.cols[4 * (lane >> 4) + g];
    }

    MX_STAMP(3)
    // ---- the eight partial sums of a tile meet in LDS (the frames are done with): tile j belongs to wave j & 7 ----
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MX_NT; ++j) {
        const int owner = j & 7;
        if (PARTIAL && j >= nt) continue;   // (nobody reads the sums of a tile the workgroup does not have)
        if (wave != owner) {
            const int r = wave - (wave > owner ? 1 : 0);
            *reinterpret_cast<f32x4 *>(lds + ((size_t)(j * 7 + r) << 10) + lane * 16) = acc[j];
        }
    }
    __syncthreads();
    MX_STAMP(4)
    // the wave's own partial sums of its (at most three) tiles, picked by wave-uniform branches so that the finishing code
    // below exists three times instead of once per tile (it runs once per wave: every copy is a cold instruction fetch)
    f32x4 own[3] = {acc[0], acc[8], acc[16]};
#pragma unroll
    for (int w = 1; w < MX_WAVES; ++w) {
        if (wave == w) {
            asm volatile("" ::: "memory");   // (keeps this a branch: as selects it is 84 conditional moves for every wave)
            own[0] = acc[w];
            own[1] = acc[w + 8];
            if (w + 16 < MX_NT) own[2] = acc[w + 16];
        }
    }
    // The wave's (at most three) tiles are finished together, stage by stage: every stage is a round trip to LDS, and a
    // tile's result write may alias the next tile's reads as far as the compiler can tell, so tile after tile would be
    // nine dependent round trips instead of three.
    int jj[3];
    bool ok[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        jj[u] = wave + 8 * u;
        ok[u] = jj[u] < nt && jj[u] < MX_NT;
        if (!ok[u]) jj[u] = 0;   // (reads of tile 0's slots are harmless; nothing is written or stored for it)
    }
    MX_STAMP(7)
    f32x4 sum[3] = {own[0], own[1], own[2]};
    int gp0[3], grem[3];   // the lane's window pair (lane & 15) of each tile: stream frame of its first start, frames left
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        gp0[u] = g_p0[jj[u] * 16 + (lane & 15)];
        grem[u] = g_rem[jj[u] * 16 + (lane & 15)];
    }
#pragma unroll
    for (int r = 0; r < 7; ++r)   // (reads only: the compiler keeps as many in flight as it has registers for)
#pragma unroll
        for (int u = 0; u < 3; ++u) sum[u] += *reinterpret_cast<const f32x4 *>(lds + ((size_t)(jj[u] * 7 + r) << 10) + lane * 16);
    MX_STAMP(8)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        // scale, tail rule (0 + c_m where the window leaves the utterance), + c_m; the first maximum over the lane's four
        // templates, then against the other four of the partner lane: the lower id wins a tie
        float bv = 0;
        int bm = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool valid = s + col[g].L <= grem[u];
            const float val = (valid ? sum[u][g] * col[g].inv_scale : 0.0f) + col[g].c;
            if (a.all && ok[u] && s < grem[u] && col[g].L < (1 << 30)) a.all[(int64_t)col[g].id * a.P + gp0[u] + s] = val;
            if (g == 0 || val > bv) {
                bv = val;
                bm = col[g].id;   // (ids grow with the slot: 4 h + g + the slice's first template)
            }
        }
        const float ov = __shfl_xor(bv, 16);
        const int om = __shfl_xor(bm, 16);
        if (h ? ov >= bv : ov > bv) {
            bv = ov;
            bm = om;
        }
        if (a.all) continue;
        if (ok[u] && h == 0 && s < grem[u]) {   // lanes 0..15: the first start of a pair, lanes 32..47: the second
            // (merge: a later slice of a bank of more than eight templates replaces what the earlier ones left only where it
            // is better -- the lowest id keeps a tie, as within a slice)
            // (slices are dealt by filter length, so ids interleave: an equal score keeps the lower id)
            const float old = a.merge ? a.scores[gp0[u] + s] : 0.0f;
            if (!a.merge || bv > old || (bv == old && bm < a.ids[gp0[u] + s])) {
                a.scores[gp0[u] + s] = bv;
                a.ids[gp0[u] + s] = bm;
            }
        }
    }
    // the last pair of an utterance: the frames behind its two starts score max_m c_m (every template is in its tail
    // there).  Pairs are flagged by lanes 0..15, the whole wave then fills one flagged pair after the other.
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        unsigned long long todo = __ballot(ok[u] && lane < 16 && grem[u] > 2 && grem[u] < a.Lmin + 2 && (a.all || !a.merge));   // (max mode: the first slice fills the tails with the whole bank's max_m c_m)
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int bp = __builtin_amdgcn_readlane(gp0[u], i), br = __builtin_amdgcn_readlane(grem[u], i);
            for (int t = 2 + lane; t < br; t += 64) {
                if (a.all) {
                    for (int m = 0; m < a.M; ++m) a.all[(int64_t)a.cols[m].id * a.P + bp + t] = a.cols[m].c;
                } else {
                    a.scores[bp + t] = a.cmax;
                    a.ids[bp + t] = a.idmax;
                }
            }
        }
    }
    MX_STAMP(5)
#ifdef TSR_STAMP
    {
        unsigned long long r_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_)::"memory");
        if (lane == 0) a.dbg[((size_t)blockIdx.x * MX_WAVES + wave) * 16 + 15] = r_;
    }
#endif
}

inline MxBank *mxb(const tsr_bank *b) { return static_cast<MxBank *>(b->mx); }
inline MxCorpus *mxc(const tsr_corpus *c) { return static_cast<MxCorpus *>(c->mx); }

}  // namespace

void tsr_mx_bank_free(tsr_bank *bank)
{
    MxBank *mb = mxb(bank);
    if (!mb) return;
    (void)hipSetDevice(bank->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(mb->d_B);
    (void)hipFree(mb->d_kq);
    (void)hipFree(mb->d_cols);
    delete mb;
    bank->mx = nullptr;
    bank->mx_unsupported = false;
}

void tsr_mx_corpus_free(tsr_corpus *cp)
{
    MxCorpus *mc = mxc(cp);
    if (!mc) return;
    (void)hipFree(mc->d_bits);
    (void)hipFree(mc->d_groups);
    (void)hipFree(mc->d_wg);
    delete mc;
    cp->mx = nullptr;
}

// The bank's fragments.  TSR_ERR_UNSUPPORTED: more than 256 templates, fewer than 24 K-blocks (per phase), or filters so long
// that not even one tile's frames fit the LDS image.
static int mx_bank_build(tsr_ctx *ctx, tsr_bank *bank)
{
    if (bank->mx) return TSR_OK;
    if (bank->mx_unsupported) return TSR_ERR_UNSUPPORTED;
    const int M = bank->M;
    const int64_t D = bank->D;
    // Rows of up to 512 features are one phase in this file's own bit rows.  Wider rows (part corpora) go in feature phases
    // with the geometry of the corpora's resident bit form (tsr_bit_geometry: D = 2050 -> six phases of 11 dwords = 352
    // features), which the kernel reads as it lies.
    int n_phase = 1, pw = 0;
    int64_t Dph = D;
    if (D > 128 * MX_MAX_KBF) {
        tsr_bit_geometry(D, &pw, &n_phase);
        Dph = 32 * (int64_t)pw;
    }
    const int KBF = n_phase > 1 ? 3 : (int)((Dph + 127) / 128);   // (phases: always three staged pieces per row, pw <= 12)
    const int n_slice = (M + 7) / 8;   // more than eight templates: slices of eight, one launch each, merged in the result arrays
    if (M < 1 || n_slice > MX_MAX_SLICES || KBF > MX_MAX_KBF || n_phase > 64) {
        bank->mx_unsupported = true;
        return TSR_ERR_UNSUPPORTED;
    }
    int Lmin = bank->L[0];
    for (int m = 0; m < M; ++m) Lmin = std::min(Lmin, bank->L[m]);
    // Chunks (32 features of one frame = 16 bytes of nibbles): a row has nfull full ones and, for D % 32 != 0, one that is
    // partly filled.  Up to eight leftover features are not worth a chunk per tap row: the leftovers of FOUR tap rows of one
    // frame parity (rho, rho + 2, rho + 4, rho + 6) share a chunk (D = 360: 33 x 11 + 9 = 372 chunks = 93 K-blocks instead of
    // 33 x 12 = 396 = 99 -- 6 % less matrix work, weight stream and fragment reads).  A K-block is any four chunks.
    const int nfull = (int)(Dph / 32), lo = (int)(Dph % 32);
    const int packed = lo >= 1 && lo <= 8;
    const int CHw = nfull + (lo ? 1 : 0);                      // chunks of a row that carry weights
    const int CH = n_phase > 1 ? 4 * KBF : CHw;                // chunk arrays per frame parity in the LDS image (phases: four per
                                                               // staged piece, so that the staging needs no per-chunk checks)
    // rows per array: a multiple of 16 (conflict-free fragment reads), as many as the image holds -- and no more than the
    // staging covers (five 16-byte pieces per thread: 2 KBF NR <= 2560; narrow rows would otherwise get arrays longer than
    // a workgroup of short utterances can fill in one staging pass)
    const int NR = std::min(MX_IMAGE / (32 * CH), 5 * MX_WAVES * 64 / (2 * KBF)) / 16 * 16;
    struct ChunkRef { int plane, c, rho, k0; };                // c < nfull: features 32 c .. of tap row rho; c == nfull && packed: leftovers of
                                                               // the tap rows plane + 2 (k0 .. k0 + 3); c == nfull && !packed: the partly filled chunk
    const int ROWS = bank->Lmax + 1;                           // (the plan's: the longest filter of the whole bank)
    if (ROWS + 34 > 2 * NR) {   // (a tile of one long utterance spans 32 + ROWS frames)
        bank->mx_unsupported = true;
        return TSR_ERR_UNSUPPORTED;
    }
    // Slices of eight templates.  A slice pays for ITS longest filter (K runs over Lmax_slice + 1 tap rows), so the templates
    // are dealt to the slices by length -- the eight shortest, the next eight, ... -- and ordered by id inside a slice
    // (config 4: L_m = 24 .. 48 over 64 templates: 38.5 tap rows on average instead of 49).  One slice: the bank as it is.
    std::vector<int> order(M);
    for (int m = 0; m < M; ++m) order[m] = m;
    if (n_slice > 1) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return bank->L[x] < bank->L[y]; });
    for (int sl = 0; sl < n_slice; ++sl) std::sort(order.begin() + 8 * sl, order.begin() + std::min(M, 8 * sl + 8));
    TSR_HIP(hipSetDevice(ctx->device));
    MxBank *mb = new MxBank;
    mb->KBF = KBF; mb->ROWS = ROWS; mb->NR = NR; mb->Lmin = Lmin; mb->Lmax = bank->Lmax;
    mb->CH = CH; mb->nfull = nfull; mb->packed = packed; mb->n_phase = n_phase; mb->pw = pw; mb->n_slice = n_slice;
    std::vector<MxCol> cols((size_t)16 * n_slice);
    std::vector<double> scale(M);
    std::vector<int64_t> woff(M + 1, 0);
    for (int m = 0; m < M; ++m) woff[m + 1] = woff[m] + (int64_t)bank->L[m] * D;
    float cmax = 0;
    int idmax = 0;
    for (int m = 0; m < M; ++m) {
        double mx = 0;
        for (int64_t i = woff[m]; i < woff[m + 1]; ++i) mx = std::max(mx, (double)fabsf(bank->W_cat[(size_t)i]));
        scale[m] = mx > 0 ? (double)MX_QMAX / mx : 1.0;
        const float cf = (float)bank->c[m];
        if (m == 0 || cf > cmax) { cmax = cf; idmax = m; }
    }
    mb->cmax = cmax; mb->idmax = idmax;
    // Can a sum of this bank's fixed-point weights over any set of taps reach 2^24?  Below that every partial sum is an
    // exact integer in f32 and the order of the additions cannot show; from there on the kernel takes its K-blocks in fixed
    // shares, so that every launch rounds alike.
    mb->exact = true;
    for (int m = 0; m < M; ++m) {
        double pos = 0, neg = 0;
        for (int64_t i = woff[m]; i < woff[m + 1]; ++i) {
            const double q = nearbyint((double)bank->W_cat[(size_t)i] * scale[m]);
            (q > 0 ? pos : neg) += fabs(q);
        }
        if (std::max(pos, neg) >= 16777216.0) mb->exact = false;
    }
    std::vector<unsigned char> Bf;     // [slice][phase][K-block][digit][1536]
    std::vector<int32_t> kq;           // [slice][K-block][4]
    for (int sl = 0; sl < n_slice; ++sl) {
        const int m0 = 8 * sl, ms = std::min(8, M - m0);
        int rows_s = 1;
        for (int i = 0; i < ms; ++i) rows_s = std::max(rows_s, bank->L[order[m0 + i]] + 1);
        std::vector<ChunkRef> refs;
        for (int plane = 0; plane < 2; ++plane) {
            for (int rho = plane; rho < rows_s; rho += 2)
                for (int c = 0; c < (packed ? nfull : CHw); ++c) refs.push_back(ChunkRef{plane, c, rho, 0});
            if (packed)
                for (int k0 = 0; plane + 2 * k0 < rows_s; k0 += 4) refs.push_back(ChunkRef{plane, nfull, -1, k0});
        }
        const int NKB = std::max(3 * MX_WAVES, (int)((refs.size() + 3) / 4));   // (every wave starts with three K-blocks)
        while ((int)refs.size() < 4 * NKB) refs.push_back(ChunkRef{0, 0, -2, 0});   // filler: zero weights, any valid chunk
        if (n_slice == 1 && (int)((size_t)rows_s * (packed ? nfull : CHw)) < 4 * 3 * MX_WAVES) {   // (short filters on narrow rows: mostly filler -- the int8 kernels serve those)
            delete mb;
            bank->mx_unsupported = true;
            return TSR_ERR_UNSUPPORTED;
        }
        mb->slice_NKB.push_back(NKB);
        mb->slice_kq.push_back(kq.size());
        mb->slice_B.push_back(Bf.size());
        for (int n = 0; n < 16; ++n) {
            const int i = n & 7;
            if (i < ms) {
                const int m = order[m0 + i];
                cols[(size_t)16 * sl + n] = MxCol{(float)(1.0 / scale[m]), (float)bank->c[m], bank->L[m], m};
            } else cols[(size_t)16 * sl + n] = MxCol{0.0f, -INFINITY, 1 << 30, 0x7ffffffe};
        }
        for (int kb = 0; kb < NKB; ++kb)
            for (int q = 0; q < 4; ++q) {
                const ChunkRef &cr = refs[(size_t)kb * 4 + q];
                const int row = cr.rho >= 0 ? (cr.rho >> 1) : cr.rho == -1 ? cr.k0 : 0;   // rows behind the pair's first frame row
                kq.push_back(((cr.plane * CH + cr.c) * NR + row) << 4);
            }
        const size_t base = Bf.size();
        Bf.resize(base + (size_t)n_phase * NKB * 3 * 1536, 0);
        for (int ph = 0; ph < n_phase; ++ph)
        for (int kb = 0; kb < NKB; ++kb) {
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, q = lane >> 4, s = n >> 3, i = n & 7;
                const ChunkRef &cr = refs[(size_t)kb * 4 + q];
                uint32_t str[3][6] = {{0}};
                if (i < ms && cr.rho != -2) {
                    const int m = order[m0 + i];
                    for (int e = 0; e < 32; ++e) {
                        int tau;
                        int64_t f;
                        if (cr.rho >= 0) {          // element 8 k + j = feature 4 j + k of the chunk (the order the bit rows expand to)
                            tau = cr.rho - s;
                            f = 32 * (int64_t)cr.c + 4 * (e & 7) + (e >> 3);
                        } else {                    // leftover chunk: bit v = 4 j + k of lo(r) | lo(r+1) << 8 | ...: tap row k0 + (v >> 3), feature v & 7
                            const int v = 4 * (e & 7) + (e >> 3);
                            tau = cr.plane + 2 * (cr.k0 + (v >> 3)) - s;
                            f = 32 * (int64_t)nfull + (v & 7);
                        }
                        if (f >= Dph) continue;
                        f += (int64_t)ph * Dph;     // (the phase's features)
                        if (f >= D || tau < 0 || tau >= bank->L[m]) continue;
                        const float w = bank->W_cat[(size_t)(woff[m] + (int64_t)tau * D + f)];
                        const long v = lrint((double)w * scale[m]);
                        int dg[3];
                        mx_digits(v, dg);   // (csrc/mx_plan.h)
                        for (int l = 0; l < 3; ++l) {
                            const uint32_t code = (uint32_t)mx_fp6_code(dg[l]);
                            const int bit = 6 * e;
                            str[l][bit >> 5] |= code << (bit & 31);
                            if ((bit & 31) > 26) str[l][(bit >> 5) + 1] |= code >> (32 - (bit & 31));
                        }
                    }
                }
                for (int l = 0; l < 3; ++l) {
                    unsigned char *dst = Bf.data() + base + (((size_t)ph * NKB + kb) * 3 + l) * 1536;
                    memcpy(dst + lane * 16, &str[l][0], 16);
                    memcpy(dst + 1024 + lane * 8, &str[l][4], 8);
                }
            }
        }
    }
    int st = TSR_OK;
    auto up = [&](void **d, const void *h, size_t bytes) {
        if (st != TSR_OK) return;
        if (hipMalloc(d, bytes) != hipSuccess || hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = TSR_ERR_HIP;
    };
    up(reinterpret_cast<void **>(&mb->d_B), Bf.data(), Bf.size());
    up(reinterpret_cast<void **>(&mb->d_cols), cols.data(), cols.size() * sizeof(MxCol));
    up(reinterpret_cast<void **>(&mb->d_kq), kq.data(), kq.size() * 4);
    if (st == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = TSR_ERR_HIP;
    bank->mx = mb;
    if (st != TSR_OK) {
        tsr_mx_bank_free(bank);
        tsr_set_error("TSR_MX: bank upload failed");
        return st;
    }
    return TSR_OK;
}

// Bit rows of the corpus at virtual positions (every utterance starts on an even row), built once.  `pw` = 0: one phase,
// rows of DW dwords from the resident byte rows; `pw` > 0: phase-major [phase][row][pw dwords] from the corpus' own bit form
// (src_bits) or its byte rows.  On failure nothing stays behind that a later call could take for a finished form.
static int mx_build_rows(tsr_ctx *ctx, tsr_corpus *cp, MxCorpus *mc, int DW, int pw, int n_phase, const uint32_t *src_bits)
{
    const int64_t N = cp->N;
    if (mc->d_bits) {
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        TSR_HIP(hipFree(mc->d_bits));
        mc->d_bits = nullptr;
    }
    mc->DW = 0;
    mx_virtual_offsets(cp->offsets.data(), N, mc->voff);   // (csrc/mx_plan.h)
    mc->Pv = mc->voff[(size_t)N];
    mc->rows_alloc = mc->Pv + 2 * (MX_LDS / 128 / 16 * 16) + 2;   // read-ahead of the last workgroup: at most one LDS image of rows
    const int row_dw = pw ? pw : DW, planes = pw ? n_phase : 1;
    const size_t bytes = (size_t)planes * mc->rows_alloc * row_dw * 4 + 256;   // (+ 256: the last piece of a phase row reads up to 12 dwords)
    uint32_t *d_rows = nullptr;
    int64_t *d_tab = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_rows), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_rows, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_tab), (size_t)(N + 1) * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, cp->offsets.data(), (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab + (N + 1), mc->voff.data(), (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && mc->Pv > 0) {
        const int64_t total = mc->Pv * row_dw * planes;
        if (pw)
            hipLaunchKernelGGL(mx_pack_phase_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, cp->d_feats, cp->Dp, src_bits,
                               cp->bit_rows, d_tab, d_tab + (N + 1), N, pw, n_phase, mc->Pv, mc->rows_alloc, d_rows);
        else
            hipLaunchKernelGGL(mx_pack_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, cp->d_feats, cp->Dp, d_tab,
                               d_tab + (N + 1), N, DW, mc->Pv, d_rows);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tab);
    if (e != hipSuccess) {
        (void)hipFree(d_rows);
        tsr_set_error("TSR_MX: building the bit rows failed: %s", hipGetErrorString(e));
        return TSR_ERR_HIP;
    }
    mc->d_bits = d_rows;
    mc->DW = pw ? -pw : DW;   // (DW < 0: the phase-major copy)
    return TSR_OK;
}

static int mx_corpus_bits(tsr_ctx *ctx, tsr_corpus *cp, const MxBank *mb)
{
    MxCorpus *mc = mxc(cp);
    if (!mc) {
        mc = new MxCorpus;
        cp->mx = mc;
    }
    if (mb->n_phase > 1) {
        // wide rows: the corpus' own resident bit form, read as it lies -- phase-major [phase][frame][pw dwords] -- when its
        // geometry is the bank's and every utterance has an even number of frames (a pair of starts must begin in the same
        // frame parity of the LDS image).  Otherwise a copy of that form with every utterance moved to an even row, built once.
        bool even = true;
        for (int64_t u = 0; u < cp->N; ++u) even = even && ((cp->offsets[(size_t)u + 1] - cp->offsets[(size_t)u]) & 1) == 0;
        const bool same_geometry = cp->d_bits && cp->bit_pw == mb->pw && cp->bit_nph == mb->n_phase;
        mc->DWr = mb->pw;
        if (same_geometry && even) {
            mc->ext_bits = reinterpret_cast<const uint32_t *>(cp->d_bits);
            mc->bits_phase = (long long)cp->bit_rows * cp->bit_pw;
            mc->voff.assign(cp->offsets.begin(), cp->offsets.end());
            mc->Pv = cp->P;
            mc->rows_alloc = cp->bit_rows;
            return TSR_OK;
        }
        mc->ext_bits = nullptr;
        if (!(mc->d_bits && mc->DW == -mb->pw)) {
            if (!same_geometry) TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
            TSR_TRY(mx_build_rows(ctx, cp, mc, 0, mb->pw, mb->n_phase, same_geometry ? reinterpret_cast<const uint32_t *>(cp->d_bits) : nullptr));
        }
        mc->bits_phase = (long long)mc->rows_alloc * mb->pw;
        return TSR_OK;
    }
    const int DW = 4 * mb->KBF;
    mc->ext_bits = nullptr;
    mc->DWr = DW;
    mc->bits_phase = 0;
    if (mc->d_bits && mc->DW == DW) return TSR_OK;
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    return mx_build_rows(ctx, cp, mc, DW, 0, 1, nullptr);
}

// Groups (pairs of window starts), tiles of 16 groups, workgroups of at most MX_NT tiles whose frames fit the LDS image.
static int mx_corpus_plan(tsr_ctx *ctx, tsr_corpus *cp, const MxBank *mb)
{
    MxCorpus *mc = mxc(cp);
    if (mc->plan_Lmin == mb->Lmin && mc->plan_ROWS == mb->ROWS && mc->plan_NR == mb->NR)
        return mc->plan_unsupported ? TSR_ERR_UNSUPPORTED : TSR_OK;
    if (mc->d_groups || mc->d_wg) {
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(mc->d_groups);
        (void)hipFree(mc->d_wg);
        mc->d_groups = nullptr;
        mc->d_wg = nullptr;
    }
    mc->plan_Lmin = mc->plan_ROWS = mc->plan_NR = -1;   // (set again below: a failure on the way leaves no plan behind)
    mc->plan_unsupported = false;
    auto keep = [&](bool unsupported) {
        mc->plan_Lmin = mb->Lmin; mc->plan_ROWS = mb->ROWS; mc->plan_NR = mb->NR;
        mc->plan_unsupported = unsupported;
        return unsupported ? TSR_ERR_UNSUPPORTED : TSR_OK;
    };
    static_assert(sizeof(MxWg) == sizeof(int4), "the kernel reads a workgroup record as one int4");
    MxPlan pl = mx_make_plan(cp->offsets.data(), mc->voff.data(), cp->N, mb->Lmin, mb->ROWS, mb->NR, ctx->cus, MX_NT);   // (csrc/mx_plan.h)
    if (pl.unsupported) return keep(true);
    mc->plan_target = pl.target;
    const std::vector<MxGroup> &groups = pl.groups;
    const std::vector<MxWg> &wgs = pl.wgs;
    const int64_t n_groups = (int64_t)groups.size();
    for (const MxWg &w : wgs)
        if ((int64_t)w.v_lo + w.nfr > mc->rows_alloc) {   // (cannot happen: the slack behind the bit rows is one LDS image)
            return keep(true);
        }
    mc->n_groups = (int)n_groups;
    mc->n_wg = (int)wgs.size();
    if (n_groups) {
        TSR_HIP(hipMalloc(reinterpret_cast<void **>(&mc->d_groups), groups.size() * sizeof(MxGroup)));
        TSR_HIP(hipMalloc(reinterpret_cast<void **>(&mc->d_wg), wgs.size() * sizeof(MxWg)));
        TSR_HIP(hipMemcpyAsync(mc->d_groups, groups.data(), groups.size() * sizeof(MxGroup), hipMemcpyHostToDevice, ctx->stream));
        TSR_HIP(hipMemcpyAsync(mc->d_wg, wgs.data(), wgs.size() * sizeof(MxWg), hipMemcpyHostToDevice, ctx->stream));
        TSR_HIP(hipStreamSynchronize(ctx->stream));
    }
    return keep(false);
}

// TSR_FAST | TSR_MX: the maximum over the templates with its id, or (d_all) every template's scores [M][P].
// TSR_ERR_UNSUPPORTED when the bank or the corpus does not fit (the caller then takes
// the int8 kernels).
int tsr_launch_mx_bank(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, float *d_scores, int32_t *d_ids, float *d_all)
{
    if (!d_all && (!d_scores || !d_ids)) return TSR_ERR_UNSUPPORTED;
    tsr_bank *wb = const_cast<tsr_bank *>(bank);
    TSR_TRY(mx_bank_build(ctx, wb));
    const MxBank *mb = mxb(bank);
    if (cp->P == 0) return TSR_OK;
    if (cp->P > (int64_t)INT32_MAX - 4096) return TSR_ERR_UNSUPPORTED;
    TSR_TRY(mx_corpus_bits(ctx, cp, mb));
    TSR_TRY(mx_corpus_plan(ctx, cp, mb));
    const MxCorpus *mc = mxc(cp);
    if (mc->n_wg == 0) return TSR_OK;
    const bool full = mb->CH == 4 * mb->KBF;
    const bool partial = mc->plan_target < MX_NT - 2;   // (most workgroups a good deal short of MX_NT tiles: see the plan)
    const void *kern = nullptr;
#define MX_KERN(K, F, P) (partial ? reinterpret_cast<const void *>(mx_bank_kernel<K, F, P, true>) : reinterpret_cast<const void *>(mx_bank_kernel<K, F, P, false>))
    switch (mb->n_phase > 1 ? 0 : mb->KBF * 2 + (full ? 1 : 0)) {
    case 0: kern = MX_KERN(3, true, true); break;
    case 2: kern = MX_KERN(1, false, false); break;
    case 3: kern = MX_KERN(1, true, false); break;
    case 4: kern = MX_KERN(2, false, false); break;
    case 5: kern = MX_KERN(2, true, false); break;
    case 6: kern = MX_KERN(3, false, false); break;
    case 7: kern = MX_KERN(3, true, false); break;
    case 8: kern = MX_KERN(4, false, false); break;
    default: kern = MX_KERN(4, true, false); break;
    }
#undef MX_KERN
    // (per launch: the attribute belongs to the function on the CURRENT device, and a process may hold several contexts)
    TSR_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS));
    MxArgs a;
    a.bits = mc->ext_bits ? mc->ext_bits : mc->d_bits; a.B = mb->d_B;
    a.n_phase = mb->n_phase; a.DWr = mc->DWr; a.bits_phase = mc->bits_phase; a.groups = mc->d_groups; a.wg = mc->d_wg; a.cols = mb->d_cols;
    // (development override, read once per process: TSR_MX_SLOWCAP)
    static const int slow_cap_env = [] { const char *e = getenv("TSR_MX_SLOWCAP"); return e ? atoi(e) : 0; }();
    a.fixed = mb->exact ? 0 : 1;
    a.scores = d_scores; a.ids = d_ids; a.all = d_all; a.P = cp->P; a.M = bank->M;
    a.CH = mb->CH; a.nfull = mb->nfull; a.packed = mb->packed;
    a.n_groups = mc->n_groups; a.KBF = mb->KBF; a.NR = mb->NR; a.Lmin = mb->Lmin; a.idmax = mb->idmax; a.cmax = mb->cmax;
    TSR_REQUIRE((size_t)32 * mb->CH * mb->NR <= (size_t)MX_IMAGE && (size_t)MX_NT * 7 * 1024 <= (size_t)MX_IMAGE, "TSR_MX: LDS image too large");
    TSR_REQUIRE(2 * mb->KBF * mb->NR <= 5 * MX_WAVES * 64, "TSR_MX: more rows than the staging covers");
    const size_t lds = MX_LDS;
    a.dbg = nullptr;
#ifdef TSR_STAMP
    const size_t nst = (size_t)mc->n_wg * MX_WAVES * 16;
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&a.dbg), nst * 8));
    TSR_HIP(hipMemsetAsync(a.dbg, 0, nst * 8, ctx->stream));
#endif
    // one launch per slice of eight templates; the second and later ones merge into the result arrays (max mode)
    for (int sl = 0; sl < mb->n_slice; ++sl) {
        a.NKB = mb->slice_NKB[(size_t)sl];
        a.B = mb->d_B + mb->slice_B[(size_t)sl];
        a.B_phase = (long long)a.NKB * 3 * 1536;
        a.kq = mb->d_kq + mb->slice_kq[(size_t)sl];
        a.slow_cap = slow_cap_env > 0 ? slow_cap_env : (a.NKB * 13 + 63) / 64 / 2 + 1;
        a.cols = mb->d_cols + 16 * sl;
        a.M = std::min(8, bank->M - 8 * sl);
        a.merge = sl > 0;
        void *kargs[] = {&a};
        TSR_HIP(hipLaunchKernel(kern, dim3((unsigned)mc->n_wg), dim3(MX_WAVES * 64), kargs, lds, ctx->stream));
    }
#ifdef TSR_STAMP
    if (const char *path = getenv("TSR_STAMP_OUT")) {
        std::vector<unsigned long long> h(nst);
        TSR_HIP(hipMemcpyAsync(h.data(), a.dbg, nst * 8, hipMemcpyDeviceToHost, ctx->stream));
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        if (FILE *f = fopen(path, "w")) {
            for (size_t i = 0; i < nst / 16; ++i) {
                fprintf(f, "%zu %zu", i / MX_WAVES, i % MX_WAVES);
                for (int k = 0; k < 16; ++k) fprintf(f, " %llu", h[i * 16 + k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(a.dbg);
#endif
    return TSR_OK;
}
