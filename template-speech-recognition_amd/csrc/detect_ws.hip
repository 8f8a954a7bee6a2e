// detect_ws.hip -- TSR_FAST (two limbs), wave-specialised: the skew GEMM of detect_mfma_kernel.inc
// (same exact-integer limb arithmetic, same limb fragments, same ring, same segment plan,
// bit-identical results) with the bookkeeping taken out of the matrix waves.
//
// What is computed (get_train_data.py:962-992 over compute_likelihood_linear_filter.pyx:135-176):
//     score[p, m] = sum_{tau < L_m} sum_d X[p+tau, d] * W_m[tau, d],   max / first arg-max over m.
//
// Roles.  A workgroup is TEN waves.  Eight WORKERS own one 32-column tile of the bank each
// (register-stationary limb fragments) and do nothing but: A-fragment reads, v_mfma_i32_32x32x32_i8,
// and at the end of every 32-frame tile the skew emission (combine the two limbs, one ds_add_u32
// per element into the ring); they start staggered by an eighth of a tile so that their emission
// bursts do not meet in the LDS queue.  The MOVER wave brings the frame tiles in (LDS-DMA, three
// tiles ahead, five buffers, utt_end of the tile riding along).  The FINISHER wave does the whole
// finishing pass: ring -> f32 score, zero tail, + c_m, max / first arg-max over templates in
// registers, coalesced stores.  Their LDS and global-memory latencies stall nobody, and the
// workers' instruction stream is short enough to hide behind their own MFMAs.
//
// No workgroup barrier in the loop.  The waves meet through LDS words: `landed` (tiles the mover
// has seen arrive), `prog[t & 7]` (workers that have emitted tile t and stopped reading it) and
// `finished` (output tiles the finisher is done with; the mover stays within the ring of it).  The LDS executes one wave's operations in order, so a counter bump issued behind
// the data operations is ordered behind them; nobody waits for an event that is not strictly
// earlier in every other wave's program order, so the protocol cannot deadlock.
//
// Registers: 96 limb fragments + 32 accumulators + the A queue: the kernel fits the 168 registers
// that three waves per SIMD (10 waves per CU) allow.
#include "detect_ws.h"

// Development ablation switches (never set in the shipped build): bit 0 drops the skew emission,
// bit 1 the finishing pass, bit 2 the MFMAs, bit 3 the workers' landed poll.
#ifndef TSR_WABL
#define TSR_WABL 0
#endif

namespace tsr_ws {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
typedef __attribute__((address_space(3))) const i32x4 lds_frag;
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

// NKB: K-blocks of 32 features per worker; FLAT: LDS row stride == resident row stride;
// MODE: 0 = max score + template id, 1 = every template's score, 2 = partial raw scores per
// template fragment (several column groups or K-phases; frag_epilogue_kernel finishes them).
template <int NKB, bool FLAT, int MODE>
__global__ __launch_bounds__(kThreads) void ws_bank_kernel(const Args a)
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
    // ring row = a.ring cells + 33: a tile's 32 rows are emitted at cell (first & mask) + 0..31
    // without wrapping; the finishing pass folds overflow cell i into cell i.
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

    // ---- helper: tile t = rows [o0 + 32 t, +32) by LDS-DMA, 1 KiB per instruction ---------------
    const int n_pieces = a.tile_bytes >> 10;
    auto stage = [&](int t) {
        const uint8_t *src_tile = a.X + ((int64_t)o0 + 32 * (int64_t)t) * a.Dp;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xb0 + (unsigned)(t % kBuffers) * (unsigned)a.tile_bytes);
        for (int pc = 0; pc < n_pieces; ++pc) {
            const int chunk = pc * 64 + lane;  // 16-byte slot of the LDS tile
            int off;
            if (FLAT) {
                off = chunk * 16;
            } else {  // LDS rows carry an extra slot (odd stride): clamp into the source row
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
        const int half = lane >> 5;
        i32x4 b[NKB][2];
        {
            const i32x4 *wl = reinterpret_cast<const i32x4 *>(a.Wl) + (size_t)coltile * (NKB * 2 * 64) + lane;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int l = 0; l < 2; ++l) b[kb][l] = wl[(kb * 2 + l) * 64];
        }
        const int col = coltile * 32 + (lane & 31);
        const int slot = a.col_slot[col];
        // padding columns (zero weights) add their zeros to the spare row
        const int tau = slot >= 0 ? a.col_tau[col] : 0;
        const bool all_real = __all(slot >= 0);
        const unsigned row_addr = lds_addr(ring + (slot >= 0 ? slot : nslots) * rstride);
        const unsigned dummy_addr = lds_addr(ring + nslots * rstride);
        // accumulator register r of lane (col, half) is frame row (r&3) + 8 (r>>2) + 4 half of the
        // tile; it adds into output frame row - tau
        const int lane_pb = 4 * half - tau;
        const unsigned a_off = (unsigned)(lane & 31) * (unsigned)a.SB + (unsigned)half * 16u;
        constexpr int PF = NKB < 4 ? NKB : 4;  // A fragments in flight
        i32x4 aq[PF];
        unsigned xcur = xb0;
        int bcur = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) aq[i] = *(lds_frag *)(size_t)(xcur + a_off + i * 32);
        // stagger: an eighth of a tile (NKB/4 MFMA pairs of 64 cycles) per wave, so that the emission
        // bursts are spread over the tile; waves w and w+4 (one SIMD) end up half a tile apart
        for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(NKB / 4 > 0 ? NKB / 4 : 1);
        for (int t = 0; t < nT; ++t) {
            const bool have_next = t + 1 < nT;
            const int bnxt = bcur + 1 == kBuffers ? 0 : bcur + 1;
            const unsigned xnxt = have_next ? xb0 + (unsigned)bnxt * (unsigned)a.tile_bytes : xcur;
            i32x16 acc[2];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb == NKB - PF) {  // the queue is about to run into tile t+1: has it landed?
                    while (!(TSR_WABL & 8) && have_next && ld_ctrl(0) < t + 2) __builtin_amdgcn_s_sleep(1);
                    compiler_fence();
                }
                const i32x4 av = aq[0];
#pragma unroll
                for (int i = 0; i + 1 < PF; ++i) aq[i] = aq[i + 1];
                const int nk = kb + PF;
                aq[PF - 1] = *(lds_frag *)(size_t)(nk < NKB ? xcur + a_off + nk * 32 : xnxt + a_off + (nk - NKB) * 32);
                if (TSR_WABL & 4) {
                    asm volatile("" ::"v"(av), "v"(b[kb][0]), "v"(b[kb][1]));
                    if (kb == 0) acc[0] = acc[1] = (i32x16){};
                } else {
#pragma unroll
                    for (int l = 0; l < 2; ++l)
                        acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b[kb][l], kb == 0 ? (i32x16){} : acc[l], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- skew emission of the tile: 16 elements per lane ----------------------------------
            if (!(TSR_WABL & 1)) {
                const int pb = 32 * t + lane_pb;  // output frame of accumulator row 0
                const unsigned ci = (unsigned)(pb & ring_mask) * 4u;
                const unsigned cellb = row_addr + ci;
                const bool interior = all_real && 32 * t >= a.lmax1 && 32 * t + 31 < nOut;
                if (interior) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2);
                        __hip_atomic_fetch_add((lds_u32 *)(size_t)(cellb + 4u * c), (unsigned)(acc[1][r] * 256 + acc[0][r]),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    const unsigned dcellb = dummy_addr + ci;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2);
                        const unsigned cell = ((unsigned)(pb + c) < (unsigned)nOut ? cellb : dcellb) + 4u * c;
                        __hip_atomic_fetch_add((lds_u32 *)(size_t)cell, (unsigned)(acc[1][r] * 256 + acc[0][r]),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            // tile t: read and emitted
            compiler_fence();
            if (lane == 0)
                __hip_atomic_fetch_add(ctrl + 8 + (t & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            compiler_fence();
            xcur = xnxt;
            bcur = bnxt;
        }
    } else {
        // =============================== helper ================================================
        const int i = lane & 31, h = lane >> 5;
        if (lane == 0)
            __hip_atomic_store(ctrl, min(kAhead, nT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        constexpr int KS = 4;  // this lane finishes slots h, h+2, ...: four at a time
        // iteration T: publishes tile T+1, waits for the workers to leave tile T-2, finishes the
        // output tile whose sums that completes, stages tile T+kAhead into the freed buffer
        const int Tend = max(nF + 1 + a.lag, nT - 2);
        for (int T = 0; T <= Tend; ++T) {
            const int f = T - 2 - a.lag;
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
                const int tile = min(T - 2, nT - 1);
                const int target = kWorkers * ((tile >> 3) + 1);
                while (ld_ctrl(8 + (tile & 7)) < target) __builtin_amdgcn_s_sleep(1);
            }
            compiler_fence();
            if (fin && !(TSR_WABL & 2)) {
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
                        if (wrap) {
                            q[k] += cell[a.ring];
                            cell[a.ring] = 0;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const int s = s0 + h + 2 * k;
                        if (s >= nslots) continue;
                        const SlotMeta sm = meta[s];
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
    TSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ws_bank_kernel<NKB, FLAT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ws_bank_kernel<NKB, FLAT, MODE>), dim3(grid), dim3(kThreads), lds, ctx->stream, args);
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
    case 7: return launch1<7>(ctx, args, flat, grid, lds);
    case 8: return launch1<8>(ctx, args, flat, grid, lds);
    case 9: return launch1<9>(ctx, args, flat, grid, lds);
    case 10: return launch1<10>(ctx, args, flat, grid, lds);
    case 11: return launch1<11>(ctx, args, flat, grid, lds);
    case 12: return launch1<12>(ctx, args, flat, grid, lds);
    default: return TSR_ERR_UNSUPPORTED;
    }
}

}  // namespace tsr_ws
