// detect_mfma.hip -- host side of the MFMA bank scorer: limb encoding of the bank, segment plan,
// launch.  The kernel itself and its documentation live in detect_mfma_kernel.inc; its
// instantiations in detect_mfma_i0/i1/i2.hip.
#include "detect_mfma_kernel.inc"

namespace tsr_fast {
#define TSR_FAST_EXTERN(N) extern template int launch<N>(tsr_ctx *, const FastArgs &, int, size_t, int, bool);
TSR_FAST_EXTERN(1) TSR_FAST_EXTERN(2) TSR_FAST_EXTERN(3) TSR_FAST_EXTERN(4) TSR_FAST_EXTERN(5) TSR_FAST_EXTERN(6)
TSR_FAST_EXTERN(7) TSR_FAST_EXTERN(8) TSR_FAST_EXTERN(9) TSR_FAST_EXTERN(10) TSR_FAST_EXTERN(11) TSR_FAST_EXTERN(12)
#undef TSR_FAST_EXTERN
#define TSR_BITS_EXTERN(N) extern template int launch_bits<N>(tsr_ctx *, const FastArgs &, int, size_t);
TSR_BITS_EXTERN(1) TSR_BITS_EXTERN(2) TSR_BITS_EXTERN(3) TSR_BITS_EXTERN(4) TSR_BITS_EXTERN(5) TSR_BITS_EXTERN(6)
TSR_BITS_EXTERN(7) TSR_BITS_EXTERN(8) TSR_BITS_EXTERN(9) TSR_BITS_EXTERN(10) TSR_BITS_EXTERN(11) TSR_BITS_EXTERN(12)
#undef TSR_BITS_EXTERN
}  // namespace tsr_fast

namespace tsr_fast {
constexpr int kTap2MaxNKB = 23;  // (detect_mfma_tap2.inc)
template <int NKB2> int launch_tap2(tsr_ctx *, const FastArgs &, int, size_t);
#define TSR_TAP2_EXTERN(N) extern template int launch_tap2<N>(tsr_ctx *, const FastArgs &, int, size_t);
TSR_TAP2_EXTERN(1) TSR_TAP2_EXTERN(3) TSR_TAP2_EXTERN(5) TSR_TAP2_EXTERN(7) TSR_TAP2_EXTERN(9) TSR_TAP2_EXTERN(11)
TSR_TAP2_EXTERN(13) TSR_TAP2_EXTERN(15) TSR_TAP2_EXTERN(17) TSR_TAP2_EXTERN(19) TSR_TAP2_EXTERN(21) TSR_TAP2_EXTERN(23)
#undef TSR_TAP2_EXTERN
}  // namespace tsr_fast

using namespace tsr_fast;

namespace {
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
    // four templates at a time: their fragment loads of one round are independent of each other (a lane has four loads
    // in flight instead of one); every template's own sum keeps the order of its fragments
    for (int m0 = 0; m0 < M; m0 += 4) {
        float r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int f0[4], nf[4], nmax = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + i, M - 1);
            f0[i] = frag_ptr[m];
            nf[i] = m0 + i < M ? frag_ptr[m + 1] - f0[i] : 0;
            nmax = max(nmax, nf[i]);
        }
        for (int f = 0; f < nmax; ++f) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f < nf[i] ? raw_frag[(int64_t)frag_idx[f0[i] + f] * P + p] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (f < nf[i]) r[i] += v[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i;
            if (m >= M) break;
            if (p + L[m] > end) r[i] = 0.0f;
            const float v = r[i] + cf[m];
            if (all) all[(int64_t)m * P + p] = v;
            if (m == 0 || v > best) {
                best = v;
                besti = m;
            }
        }
    }
    if (scores) scores[p] = best;
    if (ids) ids[p] = besti;
}

// Several column groups of whole templates: every group's workgroups have taken the max of their keys
// (orderable(score) << 32 | ~template id) into keys[p]; scores and ids out (the decode of the kernels' phase C).
__global__ __launch_bounds__(256) void key_decode_kernel(const unsigned long long *__restrict__ keys, int64_t P,
                                                         float *__restrict__ scores, int32_t *__restrict__ ids)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const unsigned long long key = keys[p];
    unsigned u = (unsigned)(key >> 32);
    u ^= (~(unsigned)((int)u >> 31)) | 0x80000000u;
    if (scores) reinterpret_cast<unsigned *>(scores)[p] = u;
    if (ids) ids[p] = (int)~(unsigned)key;
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
    bool packed = false;   // tap pairs: every template lies inside one column group (no fragments: a group's
                           // slots are whole templates, several groups meet in a global max-key array)
};

template <typename T>
int upload(tsr_ctx *ctx, T **dptr, const std::vector<T> &v)
{
    *dptr = nullptr;
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(dptr), std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty()) TSR_HIP(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return TSR_OK;
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
    for (int k = 0; k < 5; ++k) fast_bank_free_one(bank, k);
    bank->tap2_unsupported = false;
    tsr_mx_bank_free(bank);
}

static uint16_t bf16_rn(float x)  // round to nearest even (finite inputs)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_value(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// kind 0: 16-bit fixed point (two int8 limbs, int32 ring); 1: 23-bit (three limbs, int64 ring); 2: bf16
// weights; 3: bf16 hi + lo.  Built lazily, cached in bank->fast[kind].
int tsr_fast_bank_build(tsr_ctx *ctx, tsr_bank *bank, int kind)
{
    const bool tap2 = kind == 4;   // two int8 limbs, two taps per column (detect_mfma_tap2.inc)
    const bool bf = kind == 2 || kind == 3;
    const int NL = tap2 ? 2 : (bf ? kind - 1 : kind + 2);
    const int M = bank->M;
    const int64_t D = bank->D;
    const int64_t Dp = (D + 15) / 16 * 16;      // the resident row of a corpus with this D
    if (tap2 && (Dp / 16 > kTap2MaxNKB || (Dp / 16) % 2 == 0)) {
        tsr_set_error("TSR_FAST (tap pairs): resident rows of %lld bytes do not fit (need an odd number <= %d of 16-byte slots)",
                      (long long)Dp, kTap2MaxNKB);
        return TSR_ERR_UNSUPPORTED;
    }
    if (bf && D > 16 * kBfSteps) {
        tsr_set_error("TSR_BF16: D=%lld exceeds the %d features of the bf16 variant", (long long)D, 16 * kBfSteps);
        return TSR_ERR_UNSUPPORTED;
    }
    const int nkb_total = tap2 ? (int)(Dp / 16) : (bf ? kBfSteps : (int)((D + 31) / 32));
    // Feature rows wider than one wave's register budget (kMaxNKB K-blocks) are split into
    // K-phases: each phase is a complete kernel pass over its own feature range with its own
    // register-stationary limbs; the phases' partial raw scores are summed by the epilogue.
    // (bf16: one pass of kBfSteps 16-feature steps, the kernel is instantiated for exactly that)
    const int n_phase = (bf || tap2) ? 1 : (nkb_total + kMaxNKB - 1) / kMaxNKB;
    const int NKB = (bf || tap2) ? nkb_total : (nkb_total + n_phase - 1) / n_phase;
    const int KS = 1;
    const int CT = tap2 ? 4 : kWaves / KS;       // tap pairs: both wave groups hold the same four column tiles
    const int64_t K2 = 2 * Dp;                   // tap pairs: K runs over (tap b of the pair, byte of the resident row)
    auto ncols = [&](int L) { return tap2 ? (L + 1) / 2 : L; };
    int64_t C = 0;
    for (int m = 0; m < M; ++m) C += ncols(bank->L[m]);
    // Column start of every template.  By default the templates follow each other and a column group is simply the
    // next CT tiles of columns (templates cut at group boundaries become fragments, summed by frag_epilogue_kernel).
    // Tap pairs with several groups: whole templates are packed into the groups (first fit, longest first, ids
    // ascending inside a group), the rest of a group is padding -- no fragments, no fragment traffic, the groups meet
    // in a max-key array -- as long as that costs no more than one group over the contiguous layout.
    std::vector<int64_t> col0((size_t)M, 0);
    bool packed = false;
    {
        const int64_t gcols = (int64_t)CT * 32;
        const int G_contig = (int)((C + gcols - 1) / gcols);
        if (tap2 && G_contig > 1) {
            std::vector<int> order((size_t)M);
            for (int m = 0; m < M; ++m) order[(size_t)m] = m;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return bank->L[x] > bank->L[y]; });
            std::vector<int64_t> fill;
            std::vector<int> grp_of((size_t)M, -1);
            bool ok = true;
            for (int m : order) {
                const int64_t n = ncols(bank->L[m]);
                if (n > gcols) {
                    ok = false;
                    break;
                }
                size_t g = 0;
                while (g < fill.size() && fill[g] + n > gcols) ++g;
                if (g == fill.size()) fill.push_back(0);
                grp_of[(size_t)m] = (int)g;
                fill[g] += n;
            }
            if (ok && (int)fill.size() <= G_contig + G_contig / 8) {  // (at most an eighth more groups than cutting templates would need)
                packed = true;
                std::vector<int64_t> next(fill.size());
                for (size_t g = 0; g < fill.size(); ++g) next[g] = (int64_t)g * gcols;
                for (int m = 0; m < M; ++m) {  // ids ascending inside a group
                    col0[(size_t)m] = next[(size_t)grp_of[(size_t)m]];
                    next[(size_t)grp_of[(size_t)m]] += ncols(bank->L[m]);
                }
                C = (int64_t)fill.size() * gcols;  // (padding columns between the groups' templates included)
            }
        }
        if (!packed) {
            int64_t c0 = 0;
            for (int m = 0; m < M; ++m) {
                col0[(size_t)m] = c0;
                c0 += ncols(bank->L[m]);
            }
        }
    }
    const int n_coltiles = (int)((C + 31) / 32);
    const int G = (n_coltiles + CT - 1) / CT;

    // ---- per-template scale and the int8 limbs ------------------------------------------------
    // NL = 3: q = rint(W * 2^e) with |q| <= 2^22 (power-of-two scale, exactly invertible).
    // NL = 2: q = rint(W * qmax / max|W|), qmax = min(127*256, (2^31-1) / (L_m * D)) so that the
    //         int32 ring cannot overflow even if every feature of every tap is set.
    std::vector<double> inv_scale(M);
    std::vector<int8_t> limbs(bf ? 0 : (size_t)NL * C * (tap2 ? K2 : D));   // [limb][column][d]  (tap pairs: [b Dp + d])
    std::vector<uint16_t> blimbs(bf ? (size_t)NL * C * D : 0);   // bf16: hi (and lo = rn(W - hi))
    std::vector<int32_t> col_tmpl((size_t)C, -1), col_tau_v((size_t)C, 0);   // (-1: a padding column of the packed layout)
    {
        int64_t woff = 0;
        for (int m = 0; m < M; ++m) {
            const int64_t c0 = col0[(size_t)m];
            const int64_t n = (int64_t)bank->L[m] * D;
            const float *w = bank->W_cat.data() + woff;
            float mx = 0.0f;
            for (int64_t i = 0; i < n; ++i) mx = std::max(mx, fabsf(w[i]));
            double sc = 1.0;
            if (mx > 0.0f && !bf) {
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
            if (mx > 0.0f && bf) {  // fixed point of the tile sums in the int32 ring: 2^e * (all taps set) < 2^30
                const int e = (int)floor(log2(1073741824.0 / ((double)n * (double)mx)));
                sc = ldexp(1.0, std::max(-60, std::min(e, 60)));
            }
            inv_scale[m] = 1.0 / sc;
            for (int a = 0; tap2 && a < ncols(bank->L[m]); ++a) {
                const int64_t c = c0 + a;
                col_tmpl[(size_t)c] = m;
                col_tau_v[(size_t)c] = 2 * a;            // the pair's first tap: the column's rows belong to frame row - 2a
                for (int b2 = 0; b2 < 2 && 2 * a + b2 < bank->L[m]; ++b2)
                    for (int64_t d = 0; d < D; ++d) {
                        int q = (int)llrint((double)w[(int64_t)(2 * a + b2) * D + d] * sc);
                        for (int l = 0; l < NL; ++l) {
                            int digit = (l == NL - 1) ? q : ((q + 128) & 255) - 128;
                            limbs[((size_t)l * C + c) * K2 + b2 * Dp + d] = (int8_t)digit;
                            q = (q - digit) >> 8;
                        }
                    }
            }
            for (int tau = 0; !tap2 && tau < bank->L[m]; ++tau) {
                const int64_t c = c0 + tau;
                col_tmpl[(size_t)c] = m;
                col_tau_v[(size_t)c] = tau;
                for (int64_t d = 0; bf && d < D; ++d) {
                    const float x = w[tau * D + d];
                    const uint16_t hi = bf16_rn(x);
                    blimbs[((size_t)0 * C + c) * D + d] = hi;
                    if (NL == 2) blimbs[((size_t)1 * C + c) * D + d] = bf16_rn(x - bf16_value(hi));
                }
                for (int64_t d = 0; !bf && d < D; ++d) {
                    int q = (int)llrint((double)w[tau * D + d] * sc);
                    for (int l = 0; l < NL; ++l) {  // balanced base-256 digits, last one keeps the rest
                        int digit = (l == NL - 1) ? q : ((q + 128) & 255) - 128;
                        limbs[((size_t)l * C + c) * D + d] = (int8_t)digit;
                        q = (q - digit) >> 8;
                    }
                }
            }
            woff += n;
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
    const bool single = (n_phase == 1 && (G == 1 || packed));
    for (int g = 0; g < G; ++g) {
        int nslots = 0, last_m = -1;
        for (int64_t c = (int64_t)g * CT * 32; c < std::min<int64_t>(C, (int64_t)(g + 1) * CT * 32); ++c) {
            const int m = col_tmpl[(size_t)c];
            if (m < 0) continue;
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
                if (m < 0 || m == last_m) continue;
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
                        if (bf) {  // v_mfma_f32_32x32x16_bf16 B operand: lane (col, h) holds k = 16 kb + 8 h + 0..7
                            uint16_t *d16 = reinterpret_cast<uint16_t *>(dst);
                            const int64_t k0 = (int64_t)kb * 16 + 8 * (lane >> 5);
                            for (int e = 0; e < 8; ++e)
                                if (k0 + e < D) d16[e] = blimbs[((size_t)l * C + c) * D + k0 + e];
                            continue;
                        }
                        const int64_t k0 = ((int64_t)ph * NKB + kb) * 32 + 16 * (lane >> 5);
                        if (tap2) {
                            for (int e = 0; e < 16; ++e) dst[e] = limbs[((size_t)l * C + c) * K2 + k0 + e];
                            continue;
                        }
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
    fb->packed = packed;
    bank->fast[kind] = fb;
    int s = upload(ctx, &fb->d_Wl, Wl);
    if (s == TSR_OK) s = upload(ctx, &fb->d_col_slot, col_slot);
    if (s == TSR_OK) s = upload(ctx, &fb->d_col_tau, col_tau);
    if (s == TSR_OK) s = upload(ctx, &fb->d_g_nslots, g_nslots);
    if (s == TSR_OK) s = upload(ctx, &fb->d_frag_ptr, frag_ptr);
    if (s == TSR_OK) s = upload(ctx, &fb->d_frag_idx, frag_idx);
    if (s == TSR_OK) s = upload(ctx, &fb->d_slots, slots);
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    if (s != TSR_OK) fast_bank_free_one(bank, kind);
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

// TSR_FAST on narrow rows: tap pairs folded into K (detect_mfma_tap2.inc).  TSR_ERR_UNSUPPORTED when the
// geometry does not fit; the caller then takes the interleaved kernel.
int tsr_launch_tap2_bank(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, float *d_scores, int32_t *d_ids, float *d_all)
{
    const FastBank *fb = static_cast<const FastBank *>(bank->fast[4]);
    if (!fb) return TSR_ERR_UNSUPPORTED;
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));  // (this kernel stages byte rows)
    const int64_t P = cp->P;
    if (P == 0) return TSR_OK;
    const int Dp = cp->Dp;
    if (Dp != fb->NKB * 16 || fb->NKB > kTap2MaxNKB || (fb->NKB & 1) == 0 || fb->CT != 4 || fb->n_phase != 1) {
        tsr_set_error("TSR_FAST (tap pairs): corpus rows of %d bytes do not match the bank's %d K-blocks", Dp, fb->NKB);
        return TSR_ERR_UNSUPPORTED;
    }
    const int lag0 = (bank->Lmax - 1 + 31) / 32;
    int ring = 32;
    while (ring < 32 * (lag0 + 13)) ring *= 2;
    const int NBT = 6;
    const size_t frames_b = ((size_t)NBT * 32 * Dp + Dp + 48 + 15) & ~(size_t)15;
    const size_t lds = frames_b + (size_t)kMaxSlots * sizeof(SlotMeta) + (size_t)(ring + 64) * 4 + (size_t)ring * 8 +
                       (size_t)(fb->max_slots + 1) * (ring + 33) * 4 + 64;
    if (lds > 160 * 1024) {
        tsr_set_error("TSR_FAST (tap pairs): Lmax=%d needs %zu bytes of LDS", bank->Lmax, lds);
        return TSR_ERR_UNSUPPORTED;
    }
    int nseg = std::max(1, ctx->cus / fb->G);
    const int64_t min_rows = 128;
    nseg = (int)std::max<int64_t>(1, std::min<int64_t>(nseg, (P + min_rows - 1) / min_rows));
    int spx = 0;
    if (fb->G > 1 && fb->G <= ctx->cus / 8) {  // (one workgroup per CU: 256 registers per wave)
        const int want = (ctx->cus / 8) / fb->G;
        if (ctx->opt_xcd_map && want >= 1 && (int64_t)8 * want * min_rows * 2 <= P) {
            spx = want;
            nseg = 8 * spx;
        }
    }
    TSR_TRY(build_plan(ctx, cp, nseg, bank->Lmax));
    // several column groups: template fragments summed by frag_epilogue_kernel, or -- whole templates packed into the
    // groups -- per-template rows written directly (all-scores pass) / a max-key array the groups meet in
    const bool fragments = fb->G > 1 && !fb->packed;
    const bool keyed = fb->G > 1 && fb->packed && !d_all;
    void *raw_frag = nullptr;
    if (fragments) TSR_TRY(tsr_scratch(ctx, 0, (size_t)fb->n_frag * P * sizeof(float), &raw_frag));
    if (keyed) {
        TSR_TRY(tsr_scratch(ctx, 0, (size_t)P * sizeof(unsigned long long), &raw_frag));
        TSR_HIP(hipMemsetAsync(raw_frag, 0, (size_t)P * sizeof(unsigned long long), ctx->stream));
    }
    const int grid = nseg * fb->G;
    FastArgs a;
    memset(&a, 0, sizeof a);
    a.X = cp->d_feats;
    a.Dp = Dp;
    a.SB = Dp;
    a.src_slots = Dp >> 4;
    a.tile_bytes = 32 * Dp;
    a.G = fb->G;
    a.CT = fb->CT;
    a.KS = 1;
    a.nseg = nseg;
    a.spx = spx;
    a.seg = cp->d_plan;
    a.Wl = fb->d_Wl;
    a.col_slot = fb->d_col_slot;
    a.col_tau = fb->d_col_tau;
    a.slots = fb->d_slots;
    a.g_nslots = fb->d_g_nslots;
    a.utt_end = cp->d_utt_end;
    a.P = P;
    a.ring = ring;
    a.lag = lag0;
    a.lmax1 = bank->Lmax - 1;
    unsigned long long *d_dbg = nullptr;
#ifdef TSR_STAMP
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&d_dbg), (size_t)grid * kWaves * 16 * 8));
    TSR_HIP(hipMemsetAsync(d_dbg, 0, (size_t)grid * kWaves * 16 * 8, ctx->stream));
#endif
    a.dbg = d_dbg;
    if (fragments) {
        a.raw_frag = (float *)raw_frag;
    } else if (keyed) {
        a.keys = (unsigned long long *)raw_frag;
    } else {
        a.scores = d_scores;
        a.ids = d_ids;
        a.all = d_all;
    }
    {  // (tsr_internal.h: tsr_staging_read_end; the tap-pair kernel stages flat tiles of exactly 32 rows)
        const size_t end = tsr_staging_read_end(P, Dp, Dp, Dp >> 4, 32 * Dp, true, 0);
        TSR_REQUIRE(end <= cp->feats_alloc, "TSR_FAST: frame staging would read %zu bytes of a %zu-byte feature array", end, cp->feats_alloc);
    }
    int s;
    switch (fb->NKB) {
    case 1: s = launch_tap2<1>(ctx, a, grid, lds); break;
    case 3: s = launch_tap2<3>(ctx, a, grid, lds); break;
    case 5: s = launch_tap2<5>(ctx, a, grid, lds); break;
    case 7: s = launch_tap2<7>(ctx, a, grid, lds); break;
    case 9: s = launch_tap2<9>(ctx, a, grid, lds); break;
    case 11: s = launch_tap2<11>(ctx, a, grid, lds); break;
    case 13: s = launch_tap2<13>(ctx, a, grid, lds); break;
    case 15: s = launch_tap2<15>(ctx, a, grid, lds); break;
    case 17: s = launch_tap2<17>(ctx, a, grid, lds); break;
    case 19: s = launch_tap2<19>(ctx, a, grid, lds); break;
    case 21: s = launch_tap2<21>(ctx, a, grid, lds); break;
    default: s = launch_tap2<23>(ctx, a, grid, lds); break;
    }
    TSR_TRY(s);
#ifdef TSR_STAMP
    if (const char *path = getenv("TSR_STAMP_OUT")) {
        std::vector<unsigned long long> h((size_t)grid * kWaves * 16);
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        TSR_HIP(hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *fp = fopen(path, "w")) {
            for (size_t w = 0; w < (size_t)grid * kWaves; ++w)
            {
                fprintf(fp, "%zu %zu", w / kWaves, w % kWaves);
                for (int i = 0; i < 16; ++i) fprintf(fp, " %llu", h[w * 16 + i]);
                fprintf(fp, "\n");
            }
            fclose(fp);
        }
    }
    (void)hipFree(d_dbg);
#endif
    if (fragments) {
        hipLaunchKernelGGL(frag_epilogue_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float *)raw_frag, P, bank->M, fb->d_frag_ptr, fb->d_frag_idx, bank->d_L, bank->d_cf,
                           cp->d_utt_end, d_scores, d_ids, d_all);
        TSR_HIP(hipGetLastError());
    }
    if (keyed) {
        hipLaunchKernelGGL(key_decode_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long *)raw_frag, P, d_scores, d_ids);
        TSR_HIP(hipGetLastError());
    }
    return TSR_OK;
}

int tsr_launch_fast_bank(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, int flags,
                         float *d_scores, int32_t *d_ids, float *d_all)
{
    const bool bf = (flags & TSR_BF16) != 0;
    const int kind = (bf ? 2 : 0) + ((flags & TSR_W24) ? 1 : 0);
    const int NL = bf ? kind - 1 : kind + 2;
    const FastBank *fb = static_cast<const FastBank *>(bank->fast[kind]);
    if (!fb) return TSR_ERR_UNSUPPORTED;
    const int64_t P = cp->P;
    if (P == 0) return TSR_OK;
    // A corpus resident as bit rows (a part corpus) is read as such by the two-limb interleaved kernel (one K-slice per
    // workgroup); every other kernel reads byte rows, made from the bits on first use.
    const bool bits = cp->d_bits != nullptr && !bf && NL == 2 && fb->KS == 1 && fb->NKB == cp->bit_pw && fb->n_phase == cp->bit_nph;
    if (!bits) TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    if (bf) TSR_TRY(tsr_corpus_bf16_copy(ctx, cp));
    // (bf16: the "row" is the bf16 copy's, its stride already an odd number of 16-byte slots)
    const int Dp = bf ? cp->bf16_stride : cp->Dp;
    // LDS row: the bytes one pass needs of a frame (the whole resident row, or one K-phase of
    // it), as an odd number of 16-byte slots so that b128 fragment reads are conflict-free
    const int k_span = fb->KS * fb->NKB * 32;  // bytes of a row one pass reads (int8: 32 features per K-block; bf16: 16)
    int row_bytes = (fb->n_phase == 1 && !bits) ? Dp : k_span;  // source bytes copied per frame and pass (bit rows: expanded)
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
    // (alt_bank_kernel copies tiles four ahead into five buffers and finishes one tile later: lag + 8)
    int ring = 32;
    while (ring < 32 * (lag + 8)) ring *= 2;
    const size_t lds = (size_t)5 * tile_bytes + (size_t)kMaxSlots * sizeof(SlotMeta) +
                       (size_t)(ring + 64) * 4 + (size_t)ring * 8 +
                       (size_t)(fb->max_slots + 1) * (ring + 33) * ((bf || NL == 2) ? 4 : 8) + 64;
    if (lds > 160 * 1024) {
        tsr_set_error("TSR_FAST: D=%lld, Lmax=%d need %zu bytes of LDS", (long long)bank->D, bank->Lmax, lds);
        return TSR_ERR_UNSUPPORTED;
    }
    // segments: one workgroup per (segment, column group); one wave of workgroups per launch
    // K-phases of the two-limb kernel share ONE launch: a block is (phase, segment, column group), the segments six
    // (n_phase) times as long -- five launch ramps fewer, and a tile stream long enough to run in its steady state
    const bool phase_grid = fb->n_phase > 1 && NL == 2 && !bf;
    int nseg = std::max(1, ctx->cus / (fb->G * (phase_grid ? fb->n_phase : 1)));
    const int64_t min_rows = 64;  // do not cut the stream finer than two tiles per segment
    nseg = (int)std::max<int64_t>(1, std::min<int64_t>(nseg, (P + min_rows - 1) / min_rows));
    // Several column groups: the G workgroups of a segment go to one XCD (so that its L2 serves the
    // re-reads), in two rounds of workgroups per CU slot: spx = floor(2 * CUs-per-XCD / G) segments per XCD
    // keep >= 98 % of the slots busy for G <= 9 where one round would leave up to G - 1 of 32 idle.
    int spx = 0;
    if (fb->G > 1 && fb->G <= 2 * (ctx->cus / 8) && !phase_grid) {
        const int want = (2 * (ctx->cus / 8)) / fb->G;
        if (ctx->opt_xcd_map && want >= 1 && (int64_t)8 * want * min_rows * 4 <= P) {
            spx = want;
            nseg = 8 * spx;
        }
    }
    TSR_TRY(build_plan(ctx, cp, nseg, bank->Lmax));

    const bool fragments = fb->n_phase > 1 || fb->G > 1;
    void *raw_frag = nullptr;
    if (fragments) TSR_TRY(tsr_scratch(ctx, 0, (size_t)fb->n_frag * P * sizeof(float), &raw_frag));
    const int grid = nseg * fb->G * (phase_grid ? fb->n_phase : 1);
    const bool flat = (SB == Dp) && fb->n_phase == 1 && !bits;
    unsigned long long *d_dbg = nullptr;
#ifdef TSR_STAMP
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(&d_dbg), (size_t)grid * kWaves * 6 * 8));
    TSR_HIP(hipMemsetAsync(d_dbg, 0, (size_t)grid * kWaves * 6 * 8, ctx->stream));
#endif
    for (int ph = 0; ph < (phase_grid ? 1 : fb->n_phase); ++ph) {
        FastArgs a;
        memset(&a, 0, sizeof a);
        if (phase_grid) {
            a.n_phase = fb->n_phase;
            a.x_phase_bytes = bits ? (long long)cp->bit_rows * cp->bit_pw * 4 : (long long)k_span;
            a.wl_phase = (long long)fb->wl_phase;
            a.slot_phase = fb->G * kMaxSlots;
        }
        a.X = bf ? reinterpret_cast<const uint8_t *>(cp->d_feats_bf16) : cp->d_feats + (size_t)ph * k_span;  // this phase's feature range of every row
        a.Dp = Dp;
        a.SB = SB;
        a.src_slots = row_bytes >> 4;
        if (bits) {  // the bit form is phase-major: this phase's plane, rows of NKB dwords
            a.X = cp->d_bits + (size_t)ph * (size_t)cp->bit_rows * cp->bit_pw * 4;
            a.Dp = cp->bit_pw * 4;
            a.src_slots = cp->bit_pw;
            a.bit_dword0 = 0;
        }
        a.tile_bytes = tile_bytes;
        a.G = fb->G;
        a.CT = fb->CT;
        a.KS = fb->KS;
        a.nseg = nseg;
        a.spx = spx;
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
        const int ph_last = phase_grid ? fb->n_phase - 1 : ph;  // the furthest feature range this launch reads
        if (bits) {  // whole bit rows of the tiles' 32 frames: the last tile starts at a frame of the corpus
            TSR_REQUIRE(P + 31 <= cp->bit_rows && (size_t)(ph_last + 1) * (size_t)cp->bit_rows * cp->bit_pw * 4 <= cp->bits_alloc,
                        "TSR_FAST: frame staging would read past the %zu-byte bit form", cp->bits_alloc);
        } else {  // the staging's read-ahead stays inside the array it reads (tsr_internal.h: tsr_staging_read_end)
            const size_t end = tsr_staging_read_end(P, Dp, SB, a.src_slots, tile_bytes, flat, bf ? 0 : (size_t)ph_last * k_span);
            const size_t have = bf ? cp->bf16_alloc : cp->feats_alloc;
            TSR_REQUIRE(end <= have, "TSR_FAST: frame staging would read %zu bytes of a %zu-byte %s array (P=%lld, row %d, tile %d)",
                        end, have, bf ? "bf16" : "feature", (long long)P, Dp, tile_bytes);
        }
        int s;
        if (bf) {
            // the kernel is instantiated for exactly this geometry: refuse anything else rather than launch
            if (fb->NKB != kBfSteps || fb->n_phase != 1 || !flat || a.Dp != cp->bf16_stride || tile_bytes < 32 * SB + 32) {
                tsr_set_error("TSR_BF16: internal geometry mismatch (NKB=%d, phases=%d, flat=%d)", fb->NKB, fb->n_phase, (int)flat);
                return TSR_ERR_UNSUPPORTED;
            }
            TSR_TRY(launch_bf16(ctx, a, grid, lds, NL));
            continue;
        }
        if (bits) {
            switch (fb->NKB) {
            case 1: s = launch_bits<1>(ctx, a, grid, lds); break;
            case 2: s = launch_bits<2>(ctx, a, grid, lds); break;
            case 3: s = launch_bits<3>(ctx, a, grid, lds); break;
            case 4: s = launch_bits<4>(ctx, a, grid, lds); break;
            case 5: s = launch_bits<5>(ctx, a, grid, lds); break;
            case 6: s = launch_bits<6>(ctx, a, grid, lds); break;
            case 7: s = launch_bits<7>(ctx, a, grid, lds); break;
            case 8: s = launch_bits<8>(ctx, a, grid, lds); break;
            case 9: s = launch_bits<9>(ctx, a, grid, lds); break;
            case 10: s = launch_bits<10>(ctx, a, grid, lds); break;
            case 11: s = launch_bits<11>(ctx, a, grid, lds); break;
            default: s = launch_bits<12>(ctx, a, grid, lds); break;
            }
            TSR_TRY(s);
            continue;
        }
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
