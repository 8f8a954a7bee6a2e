// detect_toep.hip -- TSR_FAST for banks whose templates are all multiples of 32 taps long: the
// skew GEMM of detect_mfma_kernel.inc (same exact-integer limb arithmetic, same ring, same
// segment plan, bit-identical results) re-cut so that less of it goes through the LDS and none of
// the bookkeeping goes through the matrix waves.
//
// What is computed (get_train_data.py:962-992 over compute_likelihood_linear_filter.pyx:135-176):
//     score[p, m] = sum_{tau < L_m} sum_d X[p+tau, d] * W_m[tau, d],   max / first arg-max over m.
//
// 1. Block-Toeplitz accumulation.  A worker wave owns 32 consecutive taps tau0..tau0+31 of one
//    template as TWO column tiles of 16 (v_mfma_i32_16x16x64_i8).  For the 16-frame row tile R_s
//        N += X[R_s] . W[taps tau0 + 0..15]      (starts a diagonal)
//        O += X[R_s] . W[taps tau0 + 16..31]     (completes the diagonal R_{s-1} started)
//    element (frame i, tap j) of the first product and (frame i+16, tap j+16) of the second belong
//    to the same output frame i - j, and they sit in the same lane and register of the two MFMA
//    results -- so the matrix core adds them for free.  Every 16 frames one 16x16 accumulator
//    (4 registers) is complete: half the skew atomics and limb combines of the 32x32 kernel.
// 2. Wave specialisation.  Eight worker waves do nothing but A-fragment reads, MFMAs and the skew
//    emission (combine the two limbs, one ds_add_u32 per element).  A ninth wave, the helper, moves
//    the frame tiles (LDS-DMA, three tiles ahead, five buffers) and does the whole finishing pass
//    (ring -> f32 score, zero tail, + c_m, max / first arg-max over templates in registers, stores).
//    Its LDS and global-memory latencies stall nobody.
// 3. No workgroup barrier in the loop.  Workers and helper meet through two LDS words:
//    `landed` (tiles the helper has seen arrive) and `prog[t & 7]` (workers that have emitted
//    everything up to tile t and stopped reading it).  The LDS executes one wave's operations in
//    order, so a counter bump issued behind the data operations is ordered behind them.
//
// Registers: 96 for the limb fragments + 16 accumulators + 4 pending + the A queue: the kernel fits
// the 168 registers that three waves per SIMD (9 waves per CU) allow.
#include <type_traits>

#include "detect_toep.h"

// Development ablation switches (never set in the shipped build): bit 0 drops the skew emission,
// bit 1 the finishing pass, bit 2 the MFMAs, bit 3 the workers' landed poll, bit 4 the A reads.
#ifndef TSR_TABL
#define TSR_TABL 0
#endif

namespace tsr_toep {

using i32x4 = __attribute__((ext_vector_type(4))) int;
typedef __attribute__((address_space(3))) const i32x4 lds_frag;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) unsigned int lds_u32;

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}
// LDS-DMA, 16 bytes per lane; see detect_mfma_kernel.inc for why this is inline asm.
__device__ __forceinline__ void lds_dma16(const uint8_t *gsrc, unsigned lds_dst_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_wave_base)
                 : "memory");
}
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// wait until at most n vector-memory operations (the youngest n) are outstanding
__device__ __forceinline__ void vm_wait_pieces(int n)
{
    switch (n) {
#define TSR_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    TSR_VMW(8) TSR_VMW(9) TSR_VMW(10) TSR_VMW(11) TSR_VMW(12) TSR_VMW(13) TSR_VMW(14) TSR_VMW(15) TSR_VMW(16)
    TSR_VMW(17) TSR_VMW(18) TSR_VMW(19) TSR_VMW(20)
#undef TSR_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// NKB: K-blocks of 64 features per worker; FLAT: LDS row stride == resident row stride;
// MODE: 0 = max score + template id, 1 = every template's score, 2 = partial raw scores per
// template fragment (several column groups or K-phases; frag_epilogue_kernel finishes them).
template <int NKB, bool FLAT, int MODE>
__global__ __launch_bounds__(kThreads) void toep_bank_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.x % a.G;  // column group; neighbours share the same frames (L2 reuse)
    const int sg = blockIdx.x / a.G;  // frame-stream segment
    const int o0 = a.seg[sg], o1 = a.seg[a.nseg + sg], r1 = a.seg[2 * a.nseg + sg];
    const int nOut = o1 - o0;
    const int nT = (r1 - o0 + 31) >> 5;  // 32-frame tiles of rows
    const int nF = (nOut + 31) >> 5;     // 32-frame tiles of outputs
    const int nslots = a.g_nslots[cg];
    if (nT <= 0) return;  // (uniform) empty segment

    // ---- LDS: [kBuffers tiles][slot meta][landed, prog[8]][ring: (nslots+1) rows] ---------------
    // ring row = a.ring cells + 33: a diagonal's 4 rows are emitted at cell (first & mask) + 0..3
    // without wrapping; the finishing pass folds the (at most 3) overflow cells back.
    const unsigned xb0 = lds_addr(smem);
    SlotMeta *meta = reinterpret_cast<SlotMeta *>(smem + kBuffers * a.tile_bytes);
    int *ctrl = reinterpret_cast<int *>(meta + kMaxSlots);  // [0] landed tiles, [8..15] prog
    int *ring = ctrl + 16;
    const int ring_mask = a.ring - 1, rstride = a.ring + 33;
    for (int i = tid; i < (nslots + 1) * rstride; i += kThreads) ring[i] = 0;
    if (tid < 16) ctrl[tid] = 0;
    if (tid < nslots) meta[tid] = a.slots[cg * kMaxSlots + tid];

    auto ld_ctrl = [&](int idx) -> int {
        return __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(ctrl + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    };
    auto buf_addr = [&](int t) -> unsigned { return xb0 + (unsigned)(t % kBuffers) * (unsigned)a.tile_bytes; };

    // ---- helper: tile t = rows [o0 + 32 t, +32) by LDS-DMA, 1 KiB per instruction ---------------
    const int n_pieces = a.tile_bytes >> 10;
    auto stage = [&](int t) {
        const uint8_t *src_tile = a.X + ((int64_t)o0 + 32 * (int64_t)t) * a.Dp;
        const unsigned dst = __builtin_amdgcn_readfirstlane(buf_addr(t));
        for (int pc = 0; pc < n_pieces; ++pc) {
            const int chunk = pc * 64 + lane;  // 16-byte slot of the LDS tile
            int off;
            if (FLAT) {
                off = chunk * 16;
            } else {  // LDS rows carry extra slots (odd stride, K padding): clamp into the source row
                const int spr = a.SB >> 4;
                const int row = chunk / spr, c16 = chunk - row * spr;
                off = row * a.Dp + min(c16, a.src_slots - 1) * 16;
            }
            lds_dma16(src_tile + off, dst + pc * 1024);
        }
    };
    if (wave == kWorkers) {
        for (int t = 0; t < kAhead && t < nT; ++t) stage(t);
        vm_drain();
    }
    __syncthreads();  // ring/meta/ctrl initialised, the first tiles landed

    if (wave < kWorkers) {
        // =============================== worker ================================================
        const int coltile = cg * a.CT + wave;
        int slot = a.col_slot[coltile * 32], tau0 = a.col_tau[coltile * 32];
        if (slot < 0) {  // padding tile of the last group: zero weights, sums go to the spare row
            slot = nslots;
            tau0 = 0;
        }
        i32x4 b[2][NKB][2];
        {
            const i32x4 *wt = reinterpret_cast<const i32x4 *>(a.Wt) + (size_t)coltile * (2 * NKB * 2 * 64) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int l = 0; l < 2; ++l) b[c][kb][l] = wt[((c * NKB + kb) * 2 + l) * 64];
        }
        const int j = lane & 15, g = lane >> 4;
        const unsigned row_addr = lds_addr(ring + slot * rstride);
        const unsigned dummy_addr = lds_addr(ring + nslots * rstride);
        // accumulator register r of lane (j, g) completed after row step s' is output frame
        // 16 (s' - 1) + 4 g + r - (tau0 + j); it is emitted during step s = s' + 1
        const int lane_pb = 4 * g - j - tau0 - 32;
        const unsigned a_off = (unsigned)(lane & 15) * (unsigned)a.SB + (unsigned)(lane >> 4) * 16u;
        const unsigned step_off = 16u * (unsigned)a.SB;
        constexpr int U = 2 * NKB;           // K-block steps per 32-frame tile
        constexpr int PF = U < 3 ? U : 3;    // A fragments in flight
        auto frag = [&](unsigned xb, int u) -> i32x4 {
            return *(lds_frag *)(size_t)(xb + a_off + (u >= NKB ? step_off : 0u) + (unsigned)((u % NKB) * 64));
        };
        i32x4 acc[2][2] = {{(i32x4){}, (i32x4){}}, {(i32x4){}, (i32x4){}}};
        i32x4 aq[PF];
        int pend[4] = {0, 0, 0, 0};
        unsigned cellb = dummy_addr, dcellb = dummy_addr;
        int pb = 0;
        unsigned xcur = xb0;
        int bcur = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) aq[i] = frag(xcur, i);
        auto bump = [&](int tile) {
            compiler_fence();
            if (lane == 0)
                __hip_atomic_fetch_add(ctrl + 8 + (tile & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            compiler_fence();
        };
        // one skew atomic: element r of the pending diagonal (range-checked outside interior steps)
        auto emit1 = [&](int r, bool interior) {
            if (TSR_TABL & 1) return;
            unsigned cell = cellb + 4u * r;
            if (!interior) cell = ((unsigned)(pb + r) < (unsigned)nOut) ? cell : dcellb + 4u * r;
            __hip_atomic_fetch_add((lds_u32 *)(size_t)cell, (unsigned)pend[r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // the diagonal in `done` is complete: combine its limbs; it targets frames pb + 0..3.
        // Returns whether every frame this step's emission can touch is an output of the segment.
        auto open_emission = [&](const i32x4 (&done)[2], int s) -> bool {
#pragma unroll
            for (int r = 0; r < 4; ++r) pend[r] = done[1][r] * 256 + done[0][r];
            pb = 16 * s + lane_pb;
            const unsigned ci = (unsigned)(pb & ring_mask) * 4u;
            cellb = row_addr + ci;
            dcellb = dummy_addr + ci;
            const int lo = 16 * s - tau0 - 47;
            return lo >= 0 && lo + 30 < nOut;
        };
        // one 16-frame row step of tile t; RHO (its parity) selects the accumulator roles
        auto step = [&](auto rho_c, auto int_c, int t, unsigned xnxt, bool have_next) {
            constexpr int rho = decltype(rho_c)::value;
            constexpr bool interior = decltype(int_c)::value;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int u = rho * NKB + kb;  // compile-time after unrolling
                if (u == U - PF) {  // the queue is about to run into tile t+1: has it landed?
                    while (!(TSR_TABL & 8) && have_next && ld_ctrl(0) < t + 2) __builtin_amdgcn_s_sleep(1);
                    compiler_fence();
                }
                const i32x4 av = aq[0];
#pragma unroll
                for (int i = 0; i + 1 < PF; ++i) aq[i] = aq[i + 1];
                const int nu = u + PF;
                if (!(TSR_TABL & 16)) aq[PF - 1] = nu < U ? frag(xcur, nu) : frag(xnxt, nu - U);
                if (TSR_TABL & 4) {
                    asm volatile("" ::"v"(av), "v"(b[0][kb][0]), "v"(b[1][kb][1]), "v"(b[0][kb][1]), "v"(b[1][kb][0]));
                } else {
                    // O: the diagonal started one row step ago takes this step's taps 16..31
#pragma unroll
                    for (int l = 0; l < 2; ++l)
                        acc[1 - rho][l] =
                            __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[1][kb][l], acc[1 - rho][l], 0, 0, 0);
                    // N: a new diagonal, taps 0..15
#pragma unroll
                    for (int l = 0; l < 2; ++l)
                        acc[rho][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                            av, b[0][kb][l], kb == 0 ? (i32x4){} : acc[rho][l], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((r + 1 < NKB ? r + 1 : NKB - 1) == kb) emit1(r, interior);
                if (rho == 0 && kb == NKB - 1 && t > 0) bump(t - 1);  // tile t-1: read, and emitted up to its row 15
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using std::integral_constant;
        for (int t = 0; t < nT; ++t) {
            const bool have_next = t + 1 < nT;
            const int bnxt = bcur + 1 == kBuffers ? 0 : bcur + 1;
            const unsigned xnxt = have_next ? xb0 + (unsigned)bnxt * (unsigned)a.tile_bytes : xcur;
            // acc[rho] was O in the previous step: it is complete
            if (open_emission(acc[0], 2 * t))
                step(integral_constant<int, 0>{}, integral_constant<bool, true>{}, t, xnxt, have_next);
            else
                step(integral_constant<int, 0>{}, integral_constant<bool, false>{}, t, xnxt, have_next);
            if (open_emission(acc[1], 2 * t + 1))
                step(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t, xnxt, have_next);
            else
                step(integral_constant<int, 1>{}, integral_constant<bool, false>{}, t, xnxt, have_next);
            xcur = xnxt;
            bcur = bnxt;
        }
        // the last two diagonals (after the last step: O = acc[0], N = acc[1])
        open_emission(acc[0], 2 * nT);
#pragma unroll
        for (int r = 0; r < 4; ++r) emit1(r, false);
        bump(nT - 1);
        open_emission(acc[1], 2 * nT + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) emit1(r, false);
        bump(nT);
    } else {
        // =============================== helper ================================================
        const int i = lane & 31, h = lane >> 5;
        if (lane == 0)
            __hip_atomic_store(ctrl, min(kAhead, nT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // this lane finishes slots h, h+2, ...: four at a time, the first four's constants in registers
        constexpr int KS = 4;
        SlotMeta msm[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) msm[k] = a.slots[cg * kMaxSlots + min(h + 2 * k, kMaxSlots - 1)];
        const int Tend = max(nF + 1 + a.lagT, nT - 2);  // finishes the last output tile, publishes the last row tile
        for (int T = 0; T <= Tend; ++T) {
            const int f = T - 2 - a.lagT;
            const bool fin = f >= 0 && f < nF;
            const int prel = 32 * f + i;
            const int64_t p = (int64_t)o0 + prel;
            // The DMA of tile T+2 (issued an iteration ago) may still be in flight; everything older
            // has landed: publish tiles <= T+1.  (vmcnt counts in issue order.)
            if (T > 0) {
                if (T + 2 < nT)
                    vm_wait_pieces(n_pieces);
                else
                    vm_drain();
                compiler_fence();
                if (lane == 0)
                    __hip_atomic_store(ctrl, min(T + 2, nT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            int uend = 0;
            if (fin && MODE != 2) uend = a.utt_end[p];
            if (T >= 2) {  // workers are done with tile T-2: its buffer is free, its sums are in the ring
                const int tile = min(T - 2, nT);
                const int target = kWorkers * ((tile >> 3) + 1);
                while (ld_ctrl(8 + (tile & 7)) < target) __builtin_amdgcn_s_sleep(1);
            }
            compiler_fence();
            if (fin && !(TSR_TABL & 2)) {
                const int ridx = prel & ring_mask;
                const bool wrap = ((32 * f) & ring_mask) == 0;  // this tile owns the overflow cells
                unsigned long long key = 0;
                for (int s0 = 0; s0 < nslots; s0 += 2 * KS) {  // (one trip unless > 8 slots)
                    int q[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) {  // absent slots read (and clear) the spare row
                        int *cell = ring + min(s0 + h + 2 * k, nslots) * rstride + ridx;
                        q[k] = *cell;
                        *cell = 0;
                        if (wrap && i < 3) {
                            q[k] += cell[a.ring];
                            cell[a.ring] = 0;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const int s = s0 + h + 2 * k;
                        if (s >= nslots) continue;
                        const SlotMeta sm = s0 == 0 ? msm[k] : meta[s];
                        const float x = (float)q[k] * sm.inv_scale;  // |q| < 2^31: one rounding
                        if (MODE == 2) {
                            if (prel < nOut) a.raw_frag[(int64_t)sm.tf * a.P + p] = x;
                            continue;
                        }
                        // tail t > T - L_m scores 0 (+ c_m), as the reference's zero tail does
                        const float sc = ((p + sm.L <= uend) ? x : 0.0f) + sm.cf;
                        if (MODE == 1) {
                            if (prel < nOut) a.all[(int64_t)sm.tf * a.P + p] = sc;
                            continue;
                        }
                        unsigned u = __builtin_bit_cast(unsigned, sc);
                        u ^= (unsigned)((int)u >> 31) | 0x80000000u;  // monotone map f32 -> u32
                        // larger score wins; on equal scores the lower id has the larger ~id: the
                        // reference's strict '>' = first maximum
                        const unsigned long long kk = ((unsigned long long)u << 32) | (unsigned)~sm.tf;
                        key = kk > key ? kk : key;
                    }
                }
                if (MODE == 0) {
                    const unsigned olo = __shfl_xor((unsigned)key, 32), ohi = __shfl_xor((unsigned)(key >> 32), 32);
                    const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
                    key = other > key ? other : key;
                    unsigned u = (unsigned)(key >> 32);
                    u ^= (~(unsigned)((int)u >> 31)) | 0x80000000u;  // inverse of the monotone map
                    // lanes 0..31 store the score bits, lanes 32..63 the ids
                    int *dst = (h == 0 ? reinterpret_cast<int *>(a.scores) : a.ids) + p;
                    if (prel < nOut) *dst = h == 0 ? (int)u : (int)~(unsigned)key;
                }
            }
            // the DMA goes last: the next iteration's wait leaves exactly these pieces in flight
            if (T + kAhead < nT) stage(T + kAhead);
        }
    }
}

template <int NKB, bool FLAT, int MODE>
static int launch3(tsr_ctx *ctx, const Args &args, int grid, size_t lds)
{
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(toep_bank_kernel<NKB, FLAT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((toep_bank_kernel<NKB, FLAT, MODE>), dim3(grid), dim3(kThreads), lds, ctx->stream, args);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

template <int NKB>
static int launch1(tsr_ctx *ctx, const Args &args, bool flat, int grid, size_t lds)
{
    const int mode = args.raw_frag ? 2 : args.all ? 1 : 0;
    if (flat) {
        if (mode == 2) return launch3<NKB, true, 2>(ctx, args, grid, lds);
        if (mode == 1) return launch3<NKB, true, 1>(ctx, args, grid, lds);
        return launch3<NKB, true, 0>(ctx, args, grid, lds);
    }
    if (mode == 2) return launch3<NKB, false, 2>(ctx, args, grid, lds);
    if (mode == 1) return launch3<NKB, false, 1>(ctx, args, grid, lds);
    return launch3<NKB, false, 0>(ctx, args, grid, lds);
}

int launch(tsr_ctx *ctx, const Args &args, int nkb, bool flat, int grid, size_t lds)
{
    switch (nkb) {
    case 1: return launch1<1>(ctx, args, flat, grid, lds);
    case 2: return launch1<2>(ctx, args, flat, grid, lds);
    case 3: return launch1<3>(ctx, args, flat, grid, lds);
    case 4: return launch1<4>(ctx, args, flat, grid, lds);
    case 5: return launch1<5>(ctx, args, flat, grid, lds);
    case 6: return launch1<6>(ctx, args, flat, grid, lds);
    default: return TSR_ERR_UNSUPPORTED;
    }
}

}  // namespace tsr_toep
