// parts.hip -- the parts-coding front-end on gfx950:
//   code_parts / code_parts_fast  (code_parts.pyx:70-135, :247-295)
//   np.argmax(out, -1)            (get_train_data.py:2621, :184)
//   spread_patches == spread_waliji_patches (code_parts.pyx:406-457,
//                                            spread_waliji_patches.pyx:15-59)
//   spread_graph_patches          (code_parts.pyx:459-503)
//
// code_parts: one lane per patch location.  For part k the reference adds, in
// the order i -> j -> z over the P0 x P1 x E patch, either log_parts[k,i,j,z]
// (edge present) or log_invparts[k,i,j,z] -- a sequential f64 sum per (location,
// part).  We keep that order per part (bit-identical f64 results, hence an
// identical argmax) and take the parallelism from locations (lanes) and parts
// (KC independent accumulators per lane).  The two tables are interleaved in
// LDS as [q][k][{inv, part}] so that the lane's edge bit selects the operand by
// ADDRESS (one ds_read_b64, lanes hit one of two adjacent 8-byte slots: a
// broadcast, conflict-free) instead of by v_cndmask on a 64-bit value.
#include "tsr_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int KC = 16;   // parts per pass (accumulators per lane)
constexpr int QC = 128;  // patch cells per LDS table chunk: QC*KC*16 B = 32 KiB

// `row_bytes` is the byte stride between consecutive rows of X (X1*E for a plain map, the
// resident stride Dp for a corpus).  With `utt_end` the rows are a concatenated frame stream:
// a patch is live only if its P0 rows belong to one utterance.
__global__ __launch_bounds__(kThreads) void code_parts_kernel(
    const uint8_t *__restrict__ X, int X0, int X1, int E, int P0, int P1, int row_bytes,
    const int32_t *__restrict__ utt_end, const double *__restrict__ tab /*[Q][K][2]*/, int K,
    unsigned threshold, int fast, double *__restrict__ out /*[N0*N1][K+1] or null*/,
    int32_t *__restrict__ argmax /*or null*/)
{
    __shared__ __attribute__((aligned(16))) double lt[QC * KC * 2];
    const int N0 = X0 - P0 + 1, N1 = X1 - P1 + 1;
    const int Q = P0 * P1 * E;
    const int64_t loc = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    bool live = loc < (int64_t)N0 * N1;
    const int i0 = live ? (int)(loc / N1) : 0;
    const int j0 = live ? (int)(loc - (int64_t)i0 * N1) : 0;
    if (live && utt_end) live = i0 + P0 <= utt_end[i0];

    unsigned count = 0;
    double slot0 = 0.0;       // code_parts_fast: sum_k (out[k+1] - 1), k ascending
    double bestv = 0.0;       // first maximum over the part slots
    int bestk = -1;

    for (int kc0 = 0; kc0 < K; kc0 += KC) {
        double acc[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) acc[kk] = 0.0;
        for (int qc0 = 0; qc0 < Q; qc0 += QC) {
            const int nq = min(QC, Q - qc0);
            __syncthreads();
            for (int idx = threadIdx.x; idx < nq * KC * 2; idx += kThreads) {
                const int q = idx / (KC * 2), r = idx - q * (KC * 2);
                const int kk = r >> 1, sel = r & 1;
                const int k = kc0 + kk;
                lt[idx] = (k < K) ? tab[((int64_t)(qc0 + q) * K + k) * 2 + sel] : 0.0;
            }
            __syncthreads();
            if (live) {
                for (int q = 0; q < nq; ++q) {
                    const int qq = qc0 + q;
                    const int z = qq % E, ij = qq / E;
                    const int j = ij % P1, i = ij / P1;
                    const unsigned x = X[(int64_t)(i0 + i) * row_bytes + (j0 + j) * E + z] ? 1u : 0u;
                    if (kc0 == 0) count += x;
                    const double *row = lt + q * (KC * 2) + x;
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk) acc[kk] += row[2 * kk];
                }
            }
        }
        // count is complete after the first pass over q (kc0 == 0), i.e. here.
        const bool coded = count >= threshold;
        if (live) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const int k = kc0 + kk;
                if (k < K) {
                    const double v = coded ? acc[kk] : 0.0;
                    if (out) out[loc * (K + 1) + k + 1] = v;
                    if (coded) {
                        slot0 += v - 1.0;
                        if (bestk < 0 || v > bestv) {
                            bestv = v;
                            bestk = k;
                        }
                    }
                }
            }
        }
    }
    if (!live) return;
    const bool coded = count >= threshold;
    double s0;
    if (fast)
        s0 = coded ? slot0 : 1.0;
    else
        s0 = coded ? -__builtin_inf() : 0.0;
    if (out) out[loc * (K + 1)] = s0;
    if (argmax) argmax[loc] = (coded && bestk >= 0 && bestv > s0) ? bestk + 1 : 0;
}

// One thread per (sub-sampled) output location: dilate the part-id map over the
// clipped (2*r0+1) x (2*r1+1) box.  out rows must be zeroed beforehand.
__global__ __launch_bounds__(kThreads) void spread_kernel(const int32_t *__restrict__ A, int X0,
                                                          int X1, int r0, int r1, int K, int g0,
                                                          int g1, int O0, int O1,
                                                          uint8_t *__restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (o >= (int64_t)O0 * O1) return;
    const int oi = (int)(o / O1), oj = (int)(o - (int64_t)oi * O1);
    const int i = oi * g0, j = oj * g1;
    const int lo0 = max(i - r0, 0), hi0 = min(i + r0 + 1, X0);
    const int lo1 = max(j - r1, 0), hi1 = min(j + r1 + 1, X1);
    uint8_t *row = out + o * K;
    for (int x0 = lo0; x0 < hi0; ++x0)
        for (int x1 = lo1; x1 < hi1; ++x1) {
            const int p = A[(int64_t)x0 * X1 + x1];
            if (p > 0) row[p - 1] = 1;
        }
}

// In place, one thread per location, sequential over the source part so that a
// part switched on earlier in the scan propagates further (reference order).
__global__ __launch_bounds__(kThreads) void spread_graph_kernel(uint8_t *__restrict__ X,
                                                                int64_t n_loc, int K,
                                                                const uint8_t *__restrict__ graph)
{
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (o >= n_loc) return;
    uint8_t *x = X + o * K;
    for (int a = 0; a < K; ++a)
        if (x[a] > 0)
            for (int b = 0; b < K; ++b)
                if (graph[a * K + b]) x[b] = 1;
}

}  // namespace

int tsr_launch_code_parts(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *d_X, int64_t X0,
                          int64_t X1, int threshold, int fast, double *d_out, int32_t *d_argmax)
{
    const int64_t n_loc = (X0 - parts->P0 + 1) * (X1 - parts->P1 + 1);
    if (n_loc <= 0) return TSR_OK;
    hipLaunchKernelGGL(code_parts_kernel, dim3((unsigned)((n_loc + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, ctx->stream, d_X, (int)X0, (int)X1, (int)parts->E,
                       (int)parts->P0, (int)parts->P1, (int)(X1 * parts->E), (const int32_t *)nullptr,
                       parts->d_tab, (int)parts->K, (unsigned)threshold, fast, d_out, d_argmax);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_spread(tsr_ctx *ctx, const int32_t *d_A, int64_t X0, int64_t X1, int r0, int r1,
                      int64_t K, int g0, int g1, const uint8_t *d_graph, uint8_t *d_out)
{
    const int64_t O0 = (X0 + g0 - 1) / g0, O1 = (X1 + g1 - 1) / g1;
    const int64_t n = O0 * O1;
    if (n <= 0 || K <= 0) return TSR_OK;
    TSR_HIP(hipMemsetAsync(d_out, 0, (size_t)(n * K), ctx->stream));
    hipLaunchKernelGGL(spread_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, ctx->stream, d_A, (int)X0, (int)X1, r0, r1, (int)K, g0, g1, (int)O0,
                       (int)O1, d_out);
    TSR_HIP(hipGetLastError());
    if (d_graph) TSR_TRY(tsr_launch_spread_graph(ctx, d_out, n, K, d_graph));
    return TSR_OK;
}

int tsr_launch_spread_graph(tsr_ctx *ctx, uint8_t *d_X, int64_t n_loc, int64_t K,
                            const uint8_t *d_graph)
{
    if (n_loc <= 0 || K <= 0) return TSR_OK;
    hipLaunchKernelGGL(spread_graph_kernel, dim3((unsigned)((n_loc + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, ctx->stream, d_X, n_loc, (int)K, d_graph);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- corpus -> corpus: the front-end over a whole resident frame stream ---------------------
namespace {
// One thread per (output frame, output column): dilate the part-id map of the SOURCE stream over
// the box clipped to the frame's own utterance, write the K bytes of the output row (resident
// layout: row stride Dp_out, 0/1 bytes).  Rows must be zeroed beforehand.
__global__ __launch_bounds__(kThreads) void spread_stream_kernel(
    const int32_t *__restrict__ A /*[P_src][N1] part ids of the source stream*/, int N1, int r0, int r1,
    int K, int g0, int g1, int O1, int64_t P_out, const int32_t *__restrict__ out_src_row /*[P_out]*/,
    const int32_t *__restrict__ out_src_lo /*[P_out]*/, const int32_t *__restrict__ out_src_hi /*[P_out]*/,
    const uint8_t *__restrict__ graph, int Dp_out, uint8_t *__restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (o >= P_out * O1) return;
    const int64_t pf = o / O1;
    const int oj = (int)(o - pf * O1);
    const int i = out_src_row[pf], j = oj * g1;
    const int lo0 = max(i - r0, out_src_lo[pf]), hi0 = min(i + r0 + 1, out_src_hi[pf]);
    const int lo1 = max(j - r1, 0), hi1 = min(j + r1 + 1, N1);
    uint8_t *row = out + pf * Dp_out + (int64_t)oj * K;
    for (int x0 = lo0; x0 < hi0; ++x0)
        for (int x1 = lo1; x1 < hi1; ++x1) {
            const int p = A[(int64_t)x0 * N1 + x1];
            if (p > 0) row[p - 1] = 1;
        }
    if (graph)
        for (int a = 0; a < K; ++a)
            if (row[a] > 0)
                for (int b = 0; b < K; ++b)
                    if (graph[a * K + b]) row[b] = 1;
}
}  // namespace

int tsr_launch_code_parts_stream(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *d_X, int64_t P,
                                 int64_t X1, int Dp, const int32_t *d_utt_end, int threshold, int fast,
                                 int32_t *d_argmax)
{
    const int64_t N1 = X1 - parts->P1 + 1;
    const int64_t n_loc = (P - parts->P0 + 1) * N1;
    if (n_loc <= 0) return TSR_OK;
    hipLaunchKernelGGL(code_parts_kernel, dim3((unsigned)((n_loc + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, ctx->stream, d_X, (int)P, (int)X1, (int)parts->E,
                       (int)parts->P0, (int)parts->P1, Dp, d_utt_end, parts->d_tab, (int)parts->K,
                       (unsigned)threshold, fast, (double *)nullptr, d_argmax);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

namespace {
// K <= 64: the same dilation on 64-bit part masks.  One wave per output frame:
//   1. the (<= 2*r0+1) source rows of part ids become rows of one-hot masks in LDS (coalesced);
//   2. lane oj ORs the clipped box of its column (LDS reads only), applies the graph closure in
//      the reference's order (sequential over the source part, code_parts.pyx:459-503) and ORs its
//      K bits into the frame's bit stream in LDS (bit oj*K + k);
//   3. the lanes expand the stream to bytes and write the frame's output row as whole 16-byte
//      chunks (zero padding up to Dp_out included) -- no memset beforehand, no scattered byte stores.
constexpr int kSpreadWaves = 4;

__global__ __launch_bounds__(64 * kSpreadWaves) void spread_mask_kernel(
    const int32_t *__restrict__ A, int N1, int r0, int r1, int K, int g1, int O1, int64_t P_out,
    const int32_t *__restrict__ out_src_row, const int32_t *__restrict__ out_src_lo,
    const int32_t *__restrict__ out_src_hi, const uint64_t *__restrict__ graph_masks /*[K] or null*/, int Dp_out,
    uint8_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sp_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = 2 * r0 + 1;
    const size_t per_wave = ((size_t)R * N1 * 8 + ((size_t)(Dp_out >> 5) + 3) * 4 + 7) / 8 * 8;
    uint64_t *bits = reinterpret_cast<uint64_t *>(sp_smem + (size_t)wave * per_wave);  // [R][N1]
    uint32_t *stream = reinterpret_cast<uint32_t *>(bits + (size_t)R * N1);  // the row as a bit stream: bit oj*K + k
    const int nwords = (Dp_out >> 5) + 3;                                     // 1 bit per output byte (+ spill-over)
    const int64_t pf = (int64_t)blockIdx.x * kSpreadWaves + wave;
    const bool on = pf < P_out;
    int lo0 = 0, hi0 = 0;
    if (on) {
        const int i = out_src_row[pf];
        lo0 = max(i - r0, out_src_lo[pf]);
        hi0 = min(i + r0 + 1, out_src_hi[pf]);
        for (int x0 = lo0; x0 < hi0; ++x0)
            for (int c = lane; c < N1; c += 64) {
                const int p = A[(int64_t)x0 * N1 + c];
                bits[(x0 - lo0) * N1 + c] = p > 0 ? 1ull << (p - 1) : 0ull;
            }
        for (int w = lane; w < nwords; w += 64) stream[w] = 0;
    }
    __syncthreads();
    if (on)
        for (int oj = lane; oj < O1; oj += 64) {
            const int j = oj * g1;
            const int lo1 = max(j - r1, 0), hi1 = min(j + r1 + 1, N1);
            uint64_t m = 0;
            for (int x0 = 0; x0 < hi0 - lo0; ++x0)
                for (int x1 = lo1; x1 < hi1; ++x1) m |= bits[x0 * N1 + x1];
            if (graph_masks)
                for (int a = 0; a < K; ++a)
                    if ((m >> a) & 1) m |= graph_masks[a];
            // drop the K bits at bit offset oj*K of the stream (they straddle up to three words)
            const int off = oj * K, sh = off & 31;
            const uint64_t lo = m << sh;
            uint32_t *w = stream + (off >> 5);
            if ((uint32_t)lo) atomicOr(w, (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(w + 1, (uint32_t)(lo >> 32));
            if (sh && (uint32_t)(m >> (64 - sh))) atomicOr(w + 2, (uint32_t)(m >> (64 - sh)));
        }
    __syncthreads();
    if (!on) return;
    // one stream bit per output byte: a 16-byte chunk is 16 consecutive bits; 4 bits -> 4 bytes by
    // (nibble * 0x00204081) & 0x01010101 (bit i lands on bit 8 i, the other partial products are masked)
    uint8_t *row = out + pf * Dp_out;
    const uint16_t *stream16 = reinterpret_cast<const uint16_t *>(stream);
    for (int ci = lane; ci * 16 < Dp_out; ci += 64) {
        const uint32_t b = stream16[ci];
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (((b >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u;
        *reinterpret_cast<uint4 *>(row + ci * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// The same dilation, FB consecutive output frames per wave: their source rows overlap (frame q needs rows
// src_row[q] +- r0), so the one-hot rows are loaded once per group and first dilated along the columns into
// registers (cm[row] per lane = column), after which every frame ORs its own <= 2*r0+1 rows.  Waves work
// independently (own LDS strip, wave-level fences only).  A group whose frames straddle an utterance
// boundary so that more than kSpreadRows rows would be needed is processed frame by frame.
constexpr int kSpreadFB = 4;
constexpr int kSpreadRows = 12;

__global__ __launch_bounds__(64 * kSpreadWaves) void spread_mask_multi_kernel(
    const int32_t *__restrict__ A, int N1, int r0, int r1, int K, int g1, int O1, int64_t P_out,
    const int32_t *__restrict__ out_src_row, const int32_t *__restrict__ out_src_lo,
    const int32_t *__restrict__ out_src_hi, const uint64_t *__restrict__ graph_masks /*[K] or null*/, int Dp_out,
    uint8_t *__restrict__ out, uint8_t *__restrict__ out_bits /* the phase-major bit form instead of byte rows */, int bit_pw,
    int bit_nph, int64_t bit_rows, int nwords, int ids8 /* A holds bytes (the MFMA coder's map) */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sp_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = ((size_t)kSpreadRows * N1 * 8 + (size_t)kSpreadFB * nwords * 4 + 7) / 8 * 8;
    uint64_t *bits = reinterpret_cast<uint64_t *>(sp_smem + (size_t)wave * per_wave);  // [kSpreadRows][N1]
    uint32_t *streams = reinterpret_cast<uint32_t *>(bits + (size_t)kSpreadRows * N1);  // [FB][nwords]
    const int64_t q0 = ((int64_t)blockIdx.x * kSpreadWaves + wave) * kSpreadFB;
    if (q0 >= P_out) return;
    const int nf = (int)min((int64_t)kSpreadFB, P_out - q0);
    // per-frame row ranges (lane f holds frame f's), wave-uniform copies via readlane
    int my_lo = 0, my_hi = 0;
    if (lane < nf) {
        const int i = out_src_row[q0 + lane];
        my_lo = max(i - r0, out_src_lo[q0 + lane]);
        my_hi = min(i + r0 + 1, out_src_hi[q0 + lane]);
        if (my_hi < my_lo) my_hi = my_lo;
    }
    int flo[kSpreadFB], fhi[kSpreadFB];
    int rmin = 0x7fffffff, rmax = -0x7fffffff;
#pragma unroll
    for (int f = 0; f < kSpreadFB; ++f) {
        flo[f] = __builtin_amdgcn_readlane(my_lo, f);
        fhi[f] = __builtin_amdgcn_readlane(my_hi, f);
        if (f < nf && fhi[f] > flo[f]) {
            rmin = min(rmin, flo[f]);
            rmax = max(rmax, fhi[f]);
        }
    }
    const bool together = rmax - rmin <= kSpreadRows;
    const int ngroups = together ? 1 : nf;
    for (int g = 0; g < ngroups; ++g) {
        const int f_beg = together ? 0 : g, f_end = together ? nf : g + 1;
        int gmin = rmin, gmax = rmax;
        if (!together) {
            gmin = 0;
            gmax = 0;
#pragma unroll
            for (int f = 0; f < kSpreadFB; ++f)
                if (f == g) {
                    gmin = flo[f];
                    gmax = fhi[f];
                }
        }
        const int nrows = max(gmax - gmin, 0);
        // ---- one-hot rows of the group -> LDS; zero the frames' bit streams ---------------------------
        for (int x0 = 0; x0 < nrows; ++x0)
            for (int c = lane; c < N1; c += 64) {
                const int64_t at = (int64_t)(gmin + x0) * N1 + c;
                const int p = ids8 ? (int)reinterpret_cast<const uint8_t *>(A)[at] : A[at];
                bits[x0 * N1 + c] = p > 0 ? 1ull << (p - 1) : 0ull;
            }
        for (int w = lane; w < kSpreadFB * nwords; w += 64) streams[w] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int oj0 = 0; oj0 < O1; oj0 += 64) {
            const int oj = oj0 + lane;
            const bool col_on = oj < O1;
            const int j = oj * g1;
            const int lo1 = max(j - r1, 0), hi1 = min(j + r1 + 1, N1);
            uint64_t cm[kSpreadRows];  // column-dilated rows of this lane's output column
#pragma unroll
            for (int x0 = 0; x0 < kSpreadRows; ++x0) {
                uint64_t m = 0;
                if (col_on && x0 < nrows)
                    for (int x1 = lo1; x1 < hi1; ++x1) m |= bits[x0 * N1 + x1];
                cm[x0] = m;
            }
#pragma unroll
            for (int f = 0; f < kSpreadFB; ++f) {
                if (f < f_beg || f >= f_end) continue;
                const int a0 = flo[f] - gmin, a1 = fhi[f] - gmin;
                uint64_t m = 0;
#pragma unroll
                for (int x0 = 0; x0 < kSpreadRows; ++x0)
                    if (x0 >= a0 && x0 < a1) m |= cm[x0];
                if (graph_masks)
                    for (int a = 0; a < K; ++a)
                        if ((m >> a) & 1) m |= graph_masks[a];
                if (col_on) {  // the K bits at bit offset oj*K of frame f's stream
                    const int off = oj * K, sh = off & 31;
                    const uint64_t lo = m << sh;
                    uint32_t *w = streams + f * nwords + (off >> 5);
                    if ((uint32_t)lo) atomicOr(w, (uint32_t)lo);
                    if ((uint32_t)(lo >> 32)) atomicOr(w + 1, (uint32_t)(lo >> 32));
                    if (sh && (uint32_t)(m >> (64 - sh))) atomicOr(w + 2, (uint32_t)(m >> (64 - sh)));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- bit rows: the streams as they stand (bit oj*K + k of a frame = its feature oj*K + k), whole rows ----
        if (out_bits) {
            for (int f = f_beg; f < f_end; ++f) {
                uint32_t *o32 = reinterpret_cast<uint32_t *>(out_bits);
                for (int w = lane; w < bit_nph * bit_pw; w += 64) {  // dword w = ph * bit_pw + j of the frame -> plane ph
                    const int ph = w / bit_pw, j = w - ph * bit_pw;
                    o32[((int64_t)ph * bit_rows + (q0 + f)) * bit_pw + j] = streams[f * nwords + w];
                }
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // ---- expand the streams to bytes: whole 16-byte chunks of the frames' rows ----------------------
        for (int f = f_beg; f < f_end; ++f) {
            uint8_t *row = out + (q0 + f) * Dp_out;
            const uint16_t *stream16 = reinterpret_cast<const uint16_t *>(streams + f * nwords);
            for (int ci = lane; ci * 16 < Dp_out; ci += 64) {
                const uint32_t b = stream16[ci];
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = (((b >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u;
                *reinterpret_cast<uint4 *>(row + ci * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next group
    }
}
}  // namespace

// Geometry of the output stream: output frame q of utterance n is source row beg[n] + (q - ooff[n]) * g0,
// its dilation window is clipped to the utterance's rows [beg[n], beg[n] + n0[n]).
namespace {
__global__ __launch_bounds__(256) void out_geometry_kernel(const int64_t *__restrict__ ooff, const int32_t *__restrict__ beg,
                                                           const int32_t *__restrict__ n0, int64_t N, int g0, int64_t P_out,
                                                           int32_t *__restrict__ src_row, int32_t *__restrict__ src_lo,
                                                           int32_t *__restrict__ src_hi)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= P_out) return;
    int64_t lo = 0, hi = N;  // last n with ooff[n] <= q
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ooff[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    src_row[q] = beg[lo] + (int32_t)(q - ooff[lo]) * g0;
    src_lo[q] = beg[lo];
    src_hi[q] = beg[lo] + n0[lo];
}

// Everything the part chain needs per frame before its first real kernel, in ONE launch: the source-row geometry of every
// output frame (as out_geometry_kernel), the utterance tables of the new corpus (as tsr_api.hip's utt_tables_kernel: both
// come out of the same search over the output offsets), the zeroed read-ahead rows behind the bit planes and the zeroed
// work counters of the coder.
__global__ __launch_bounds__(256) void part_chain_setup_kernel(const int64_t *__restrict__ ooff, const int32_t *__restrict__ beg,
                                                               const int32_t *__restrict__ n0, int64_t N, int g0, int64_t P_out,
                                                               int32_t *__restrict__ src_row, int32_t *__restrict__ src_lo,
                                                               int32_t *__restrict__ src_hi, int32_t *__restrict__ utt_beg,
                                                               int32_t *__restrict__ utt_end, uint32_t *__restrict__ bits,
                                                               int64_t plane_dwords, int64_t tail_first, int tail_count, int n_planes,
                                                               int *__restrict__ work)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < 2 && work) work[q] = 0;
    if (bits && q < (int64_t)n_planes * tail_count) bits[(q / tail_count) * plane_dwords + tail_first + q % tail_count] = 0;
    if (q >= P_out) {
        if (q < P_out + 256) utt_end[q] = 0;  // the 256 ints behind utt_end that tiles read past the last frame
        return;
    }
    int64_t lo = 0, hi = N;  // last n with ooff[n] <= q
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ooff[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    src_row[q] = beg[lo] + (int32_t)(q - ooff[lo]) * g0;
    src_lo[q] = beg[lo];
    src_hi[q] = beg[lo] + n0[lo];
    utt_beg[q] = (int32_t)ooff[lo];
    utt_end[q] = (int32_t)ooff[lo + 1];
}
}  // namespace

int tsr_launch_out_geometry(tsr_ctx *ctx, const int64_t *d_ooff, const int32_t *d_beg, const int32_t *d_n0, int64_t N,
                            int g0, int64_t P_out, int32_t *d_src_row, int32_t *d_src_lo, int32_t *d_src_hi)
{
    if (P_out <= 0) return TSR_OK;
    hipLaunchKernelGGL(out_geometry_kernel, dim3((unsigned)((P_out + 255) / 256)), dim3(256), 0, ctx->stream, d_ooff, d_beg,
                       d_n0, N, g0, P_out, d_src_row, d_src_lo, d_src_hi);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

int tsr_launch_part_chain_setup(tsr_ctx *ctx, const int64_t *d_ooff, const int32_t *d_beg, const int32_t *d_n0, int64_t N, int g0,
                                int64_t P_out, int32_t *d_src_row, int32_t *d_src_lo, int32_t *d_src_hi, int32_t *d_utt_beg,
                                int32_t *d_utt_end, uint32_t *d_bits, int64_t plane_dwords, int64_t tail_first, int tail_count,
                                int n_planes, int *d_work)
{
    const int64_t n = std::max<int64_t>(P_out + 256, (int64_t)n_planes * tail_count);
    hipLaunchKernelGGL(part_chain_setup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_ooff, d_beg, d_n0, N, g0,
                       P_out, d_src_row, d_src_lo, d_src_hi, d_utt_beg, d_utt_end, d_bits, plane_dwords, tail_first, tail_count, n_planes,
                       d_work);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// `d_out` rows need no zeroing when this returns with *wrote_all = true (mask kernel); the
// byte-scatter kernel (K > 64) needs them zeroed and does that itself.
// The dilation straight into the bit form, in registers (round 3): one LANE per frequency bin (N1, O1 <= 64), one wave
// walks kLaneFrames consecutive output frames.  Per frame: the lane ORs the one-hot masks of its bin over the <= 2 r0 + 1
// source rows (byte ids: a coalesced N1-byte read per row, L1/L2 hits after the first frame), the output column takes the
// masks of the source bins of its box by lane shuffles, the graph closure runs on the 64-bit mask, and the frame's bit
// stream (bit oj K + k) is assembled per output DWORD from the one or two (K >= 32) lanes that overlap it -- no LDS, no
// atomics -- and stored into the phase planes.  Same bits as spread_mask_multi_kernel (test_part_corpus_bit_rows,
// test_config3_*, tools/fuzz_parts.py).
constexpr int kLaneFrames = 8;

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src)
{
    return ((unsigned long long)(unsigned)__shfl((int)(v >> 32), src) << 32) | (unsigned)__shfl((int)v, src);
}

// wavefront shifts by one lane (DPP, zero fill at the ends): with a symmetric box the direction does not matter
__device__ __forceinline__ unsigned long long wave_shift1(unsigned long long v, bool up)
{
    const int lo = (int)v, hi = (int)(v >> 32);
    const int l2 = up ? __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xF, 0xF, true);
    const int h2 = up ? __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xF, 0xF, true);
    return ((unsigned long long)(unsigned)h2 << 32) | (unsigned)l2;
}

// R0: the row radius as a compile-time constant (0..3), or -1 for any radius.  With it a wave whose frames' windows are
// whole and consecutive (no utterance edge inside them, grid step 1) reads every source row ONCE, all loads in flight
// together, and slides the window over masks in registers (static indices); other waves read per frame.
template <bool IDS8, int R0>
__global__ __launch_bounds__(256) void spread_lane_bits_kernel(
    const void *__restrict__ Aany, int N1, int r0, int r1, int K, int g1, int O1, int64_t P_out,
    const int32_t *__restrict__ out_src_row, const int32_t *__restrict__ out_src_lo, const int32_t *__restrict__ out_src_hi,
    const uint64_t *__restrict__ graph_masks /*[K] or null*/, uint32_t *__restrict__ o32, int bit_pw, int bit_nph, int64_t bit_rows)
{
    __shared__ unsigned long long buf[4][kLaneFrames][64];  // per wave: the frames' column masks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = bit_nph * bit_pw;                      // dwords of a frame's bit form
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * kLaneFrames;
    if (q0 >= P_out) return;
    const int nf = (int)min((int64_t)kLaneFrames, P_out - q0);
    const int j = lane * g1;                              // source bin of this lane's output column
    const int lo1 = max(j - r1, 0), hi1 = min(j + r1 + 1, N1);
    // one-hot mask of a part id (0 = none) as two dwords
    auto one_hot = [](int p) -> unsigned long long {
        const int b = p - 1;
        const unsigned lo = (unsigned)b < 32u ? 1u << (b & 31) : 0u, hi = (unsigned)(b - 32) < 32u ? 1u << (b & 31) : 0u;
        return ((unsigned long long)hi << 32) | lo;
    };
    auto part_at = [&](int64_t at) -> int {
        return IDS8 ? (int)static_cast<const uint8_t *>(Aany)[at] : static_cast<const int32_t *>(Aany)[at];
    };
    // ---- A. per frame: the rows of its window OR-ed per bin -------------------------------------------------------
    unsigned long long rowor[kLaneFrames];
    bool sliding = false;
    if (R0 >= 0) {  // (wave-uniform) whole consecutive windows inside one utterance?
        const int i0 = out_src_row[q0], i7 = out_src_row[q0 + nf - 1];
        sliding = nf == kLaneFrames && i7 == i0 + kLaneFrames - 1 && out_src_lo[q0] == out_src_lo[q0 + nf - 1] &&
                  i0 - R0 >= out_src_lo[q0] && i7 + R0 + 1 <= out_src_hi[q0 + nf - 1];
        if (sliding) {
            constexpr int NR = kLaneFrames + 2 * (R0 >= 0 ? R0 : 0);
            int pk[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) pk[k] = lane < N1 ? part_at((int64_t)(i0 - R0 + k) * N1 + lane) : 0;
            unsigned long long rm[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) rm[k] = one_hot(pk[k]);
#pragma unroll
            for (int f = 0; f < kLaneFrames; ++f) {
                unsigned long long m = rm[f];
#pragma unroll
                for (int d = 1; d <= 2 * (R0 >= 0 ? R0 : 0); ++d) m |= rm[f + d];
                rowor[f] = m;
            }
        }
    }
    if (!sliding) {
#pragma unroll
        for (int f = 0; f < kLaneFrames; ++f) {
            unsigned long long m = 0;
            if (f < nf && lane < N1) {
                const int64_t q = q0 + f;
                const int i = out_src_row[q];
                const int lo = max(i - r0, out_src_lo[q]), hi = max(min(i + r0 + 1, out_src_hi[q]), lo);  // (wave-uniform)
                for (int x0 = lo; x0 < hi; ++x0) m |= one_hot(part_at((int64_t)x0 * N1 + lane));
            }
            rowor[f] = m;
        }
    }
    // ---- B. box over the bins, closure -> the frame's column masks ---------------------------------------------------
#pragma unroll
    for (int f = 0; f < kLaneFrames; ++f) {
        const unsigned long long m = rowor[f];
        unsigned long long mc = m;                        // this output column's box over the source bins
        if (g1 == 1) {                                    // neighbours by wavefront shifts (bins >= N1 hold no parts)
            unsigned long long up = m, dn = m;
            for (int d = 0; d < r1; ++d) {
                up = wave_shift1(up, true);
                dn = wave_shift1(dn, false);
                mc |= up | dn;
            }
        } else {
            mc = 0;
            for (int x1 = lo1; x1 < lo1 + 2 * r1 + 1; ++x1) {  // (every lane runs the same number of shuffles)
                const unsigned long long v = shfl64(m, x1 & 63);
                if (x1 < hi1) mc |= v;
            }
        }
        if (lane >= O1) mc = 0;
        if (graph_masks)
            for (int a = 0; a < K; ++a)
                if ((mc >> a) & 1) mc |= graph_masks[a];
        buf[wave][f][lane] = mc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- C. the bit streams (bit oj K + k), one output DWORD per lane: bits [32 w, 32 w + 32) come from the columns
    //      first .. first + cmax - 1 at fixed shifts (all of them 32-bit: v_alignbit brings down what comes from the upper
    //      half of a mask).  Whole rounds of 64 dwords run frame by frame with everything but the frame hoisted; the
    //      dwords left over (nw mod 64: 2 of 66 at config 3) of ALL the wave's frames share one last round ----------------
    const int cmax = (31 + K) / K + 1;                    // output columns that can overlap one dword
    auto dword_of = [&](int f, int w, int first) -> unsigned {
        unsigned acc = 0;
        for (int c = 0; c < cmax; ++c) {
            const int oj = first + c;
            const int pos = oj * K - 32 * w;           // bit of the dword where this column's mask starts
            if (oj < O1 && pos < 32) {
                const unsigned long long v = buf[wave][f][oj];
                const unsigned vl = (unsigned)v, vh = (unsigned)(v >> 32);
                const int sdown = -pos;                 // (pos < 0) the mask starts sdown bits below the dword
                const unsigned down = __builtin_amdgcn_alignbit(sdown >= 32 ? 0u : vh, sdown >= 32 ? vh : vl, (unsigned)sdown & 31u);
                acc |= pos >= 0 ? vl << (pos & 31) : down;
            }
        }
        return acc;
    };
    const int full = nw & ~63, rem = nw - full;
    const bool packed = rem > 0 && rem * kLaneFrames <= 64;   // the leftover dwords of every frame fit one round
    for (int w0 = 0; w0 < (packed ? full : nw); w0 += 64) {
        const int w = w0 + lane;
        if (w >= nw) continue;
        const int first = (32 * w) / K;
        const int ph = w / bit_pw, jj = w - ph * bit_pw;
        uint32_t *dst = o32 + ((int64_t)ph * bit_rows + q0) * bit_pw + jj;
        for (int f = 0; f < nf; ++f) dst[(int64_t)f * bit_pw] = dword_of(f, w, first);
    }
    if (packed) {
        const int f = lane / rem, w = full + lane - f * rem;
        if (f < nf) {
            const int ph = w / bit_pw, jj = w - ph * bit_pw;
            o32[((int64_t)ph * bit_rows + q0 + f) * bit_pw + jj] = dword_of(f, w, (32 * w) / K);
        }
    }
}

// (development switch: TSR_SPREADK=old at library load keeps the LDS-stream kernel for the bit form, for A/B timing)
static const bool g_spread_old = [] {
    const char *e = getenv("TSR_SPREADK");
    return e && e[0] == 'o';
}();

static int spread_multi_words(int Dp_out, int bit_words) { return std::max((Dp_out >> 5) + 3, bit_words); }
static size_t spread_multi_lds(int64_t N1, int Dp_out, int bit_words)
{
    return (size_t)kSpreadWaves * (((size_t)kSpreadRows * N1 * 8 + (size_t)kSpreadFB * spread_multi_words(Dp_out, bit_words) * 4 + 7) / 8 * 8);
}

// Can the dilation write the part features as bit rows (tsr_corpus.d_bits)?  Only the mask kernel that handles several
// frames per wave does: at most 64 parts, dilation window of at most kSpreadRows source rows, its strips within 64 KB.
bool tsr_spread_packs_bits(int64_t N1, int r0, int64_t K, int Dp_out, int bit_words)
{
    return K <= 64 && 2 * r0 + 1 <= kSpreadRows && spread_multi_lds(N1, Dp_out, bit_words) <= 64 * 1024 && Dp_out % 16 == 0;
}

int tsr_launch_spread_stream(tsr_ctx *ctx, const int32_t *d_A, int64_t N1, int r0, int r1, int64_t K,
                             int g0, int g1, int64_t O1, int64_t P_out, const int32_t *d_src_row,
                             const int32_t *d_src_lo, const int32_t *d_src_hi, const uint8_t *d_graph,
                             const uint64_t *d_graph_masks, int Dp_out, uint8_t *d_out, uint8_t *d_out_bits, int bit_pw, int bit_nph,
                             int64_t bit_rows, int ids8)
{
    const int Bp = bit_pw * bit_nph;  // dwords of a frame's bit form
    const int64_t n = P_out * O1;
    if (n <= 0) return TSR_OK;
    if (d_out_bits && !tsr_spread_packs_bits(N1, r0, K, Dp_out, Bp)) {
        tsr_set_error("spread: bit rows asked for a geometry the mask kernel does not cover");
        return TSR_ERR_UNSUPPORTED;
    }
    if (d_out_bits && N1 <= 64 && O1 <= 64 && K <= 64 && !g_spread_old) {  // the register kernel writes the bit form directly
        const int64_t fpb = 4 * kLaneFrames;
        const dim3 grid((unsigned)((P_out + fpb - 1) / fpb));
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, dim3(256), 0, ctx->stream, (const void *)d_A, (int)N1, r0, r1, (int)K, g1, (int)O1, P_out, d_src_row,
                               d_src_lo, d_src_hi, d_graph ? d_graph_masks : nullptr, reinterpret_cast<uint32_t *>(d_out_bits), bit_pw, bit_nph,
                               bit_rows);
        };
        const int r0c = (g0 == 1 && r0 <= 3) ? r0 : -1;  // row radius as a constant: windows slide over masks in registers
        if (ids8) {
            switch (r0c) {
            case 0: go(spread_lane_bits_kernel<true, 0>); break;
            case 1: go(spread_lane_bits_kernel<true, 1>); break;
            case 2: go(spread_lane_bits_kernel<true, 2>); break;
            case 3: go(spread_lane_bits_kernel<true, 3>); break;
            default: go(spread_lane_bits_kernel<true, -1>); break;
            }
        } else {
            switch (r0c) {
            case 0: go(spread_lane_bits_kernel<false, 0>); break;
            case 1: go(spread_lane_bits_kernel<false, 1>); break;
            case 2: go(spread_lane_bits_kernel<false, 2>); break;
            case 3: go(spread_lane_bits_kernel<false, 3>); break;
            default: go(spread_lane_bits_kernel<false, -1>); break;
            }
        }
        TSR_HIP(hipGetLastError());
        return TSR_OK;
    }
    const size_t per_wave = ((size_t)(2 * r0 + 1) * N1 * 8 + ((size_t)(Dp_out >> 5) + 3) * 4 + 7) / 8 * 8;
    const size_t lds = (size_t)kSpreadWaves * per_wave;
    const int bp_eff = d_out_bits ? Bp : 0;
    if (K <= 64 && 2 * r0 + 1 <= kSpreadRows && spread_multi_lds(N1, Dp_out, bp_eff) <= 64 * 1024 && Dp_out % 16 == 0) {
        const int64_t frames_per_block = (int64_t)kSpreadWaves * kSpreadFB;
        hipLaunchKernelGGL(spread_mask_multi_kernel, dim3((unsigned)((P_out + frames_per_block - 1) / frames_per_block)),
                           dim3(64 * kSpreadWaves), spread_multi_lds(N1, Dp_out, bp_eff), ctx->stream, d_A, (int)N1, r0, r1, (int)K, g1,
                           (int)O1, P_out, d_src_row, d_src_lo, d_src_hi, d_graph ? d_graph_masks : nullptr, Dp_out, d_out,
                           d_out_bits, bit_pw, bit_nph, bit_rows, spread_multi_words(Dp_out, bp_eff), ids8);
        TSR_HIP(hipGetLastError());
        return TSR_OK;
    }
    if (ids8) {
        tsr_set_error("spread: byte part ids asked for a geometry the mask kernel does not cover");
        return TSR_ERR_UNSUPPORTED;
    }
    if (K <= 64 && lds <= 48 * 1024 && Dp_out % 16 == 0) {
        hipLaunchKernelGGL(spread_mask_kernel, dim3((unsigned)((P_out + kSpreadWaves - 1) / kSpreadWaves)),
                           dim3(64 * kSpreadWaves), lds, ctx->stream, d_A, (int)N1, r0, r1, (int)K, g1, (int)O1, P_out,
                           d_src_row, d_src_lo, d_src_hi, d_graph ? d_graph_masks : nullptr, Dp_out, d_out);
        TSR_HIP(hipGetLastError());
        return TSR_OK;
    }
    TSR_HIP(hipMemsetAsync(d_out, 0, (size_t)P_out * Dp_out, ctx->stream));
    hipLaunchKernelGGL(spread_stream_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, ctx->stream, d_A, (int)N1, r0, r1, (int)K, g0, g1, (int)O1, P_out, d_src_row,
                       d_src_lo, d_src_hi, d_graph, Dp_out, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

// ---- code_parts_mag_aux (code_parts.pyx:138-242) -------------------------------------------------------------
// Edges + magnitude channels + auxiliary channels.  One lane per patch location, KC parts per pass; the
// host lays the six tables out as ONE sequence of {inv, part} pairs in exactly the order in which the
// reference adds its terms (per patch row: AD auxiliary terms, P1 x E edge terms with the MD channel terms
// behind column P1/2, then the row's band term), so the kernel only has to produce the selector bit of
// every term -- the data bit, the unsigned "(c - mag_i)^2 > 1" test, the row's tmp_freq_mag flag -- and add.
// The unsigned arithmetic is the generated C's (Python-2 `/` on unsigned ints).
namespace {
constexpr int kMagKC = 8;
__global__ __launch_bounds__(kThreads) void code_parts_mag_aux_kernel(
    const uint8_t *__restrict__ Xe, const uint8_t *__restrict__ Xm, const uint8_t *__restrict__ Xa, int X0, int X1, int E,
    int MD, int AD, int P0, int P1, const uint8_t *__restrict__ mask, const double *__restrict__ tab /*[term][K][2]*/, int K,
    double *__restrict__ out)
{
    const int N0 = X0 - P0 + 1, N1 = X1 - P1 + 1;
    const int64_t loc = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (loc >= (int64_t)N0 * N1) return;
    const int i0 = (int)(loc / N1), j0 = (int)(loc - (int64_t)i0 * N1);
    double *o = out + loc * (K + 1);
    const bool coded = mask[(i0 + P0 / 2) * X1 + j0 + P1 / 2] != 0;
    o[0] = coded ? -__builtin_inf() : 0.0;
    if (!coded) {
        for (int k = 0; k < K; ++k) o[k + 1] = 0.0;
        return;
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *t2 = reinterpret_cast<const d2 *>(tab);
    for (int k0 = 0; k0 < K; k0 += kMagKC) {
        double acc[kMagKC];
#pragma unroll
        for (int kk = 0; kk < kMagKC; ++kk) acc[kk] = 0.0;
        int64_t term = 0;
        auto add = [&](bool sel) {
#pragma unroll
            for (int kk = 0; kk < kMagKC; ++kk) {
                const d2 tv = t2[term * K + min(k0 + kk, K - 1)];
                acc[kk] += sel ? tv.y : tv.x;
            }
            ++term;
        };
        for (int i = 0; i < P0; ++i) {
            unsigned tmp_freq_mag = 0;
            for (int j = 0; j < AD; ++j) add(Xa[(i0 + i) * AD + j] != 0);
            for (int j = 0; j < P1; ++j) {
                const unsigned c = ((unsigned)(j0 + j) * (unsigned)MD) / (unsigned)X1;
                if (Xm[(i0 + i) * MD + c]) tmp_freq_mag = 1;
                for (int z = 0; z < E; ++z) add(Xe[((int64_t)(i0 + i) * X1 + j0 + j) * E + z] != 0);
                if (j == P1 / 2)
                    for (unsigned mi = 0; mi < (unsigned)MD; ++mi) {
                        const unsigned d = c - mi;
                        add(!(d * d > 1u));
                    }
            }
            add(tmp_freq_mag != 0);
        }
#pragma unroll
        for (int kk = 0; kk < kMagKC; ++kk)
            if (k0 + kk < K) o[k0 + kk + 1] = acc[kk];
    }
}
}  // namespace

int tsr_launch_code_parts_mag_aux(tsr_ctx *ctx, const uint8_t *d_Xe, const uint8_t *d_Xm, const uint8_t *d_Xa, int64_t X0,
                                  int64_t X1, int64_t E, int64_t MD, int64_t AD, int64_t P0, int64_t P1, const uint8_t *d_mask,
                                  const double *d_tab, int64_t K, double *d_out)
{
    const int64_t n_loc = (X0 - P0 + 1) * (X1 - P1 + 1);
    if (n_loc <= 0) return TSR_OK;
    hipLaunchKernelGGL(code_parts_mag_aux_kernel, dim3((unsigned)((n_loc + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       ctx->stream, d_Xe, d_Xm, d_Xa, (int)X0, (int)X1, (int)E, (int)MD, (int)AD, (int)P0, (int)P1, d_mask, d_tab,
                       (int)K, d_out);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}
