// Invariants of the TSR_MX host plan and of its weight arithmetic (template-speech-recognition_amd/csrc/mx_plan.h) -- built and run
// by tests/test_mx_plan_cpu.py with g++ (-fsanitize=address,undefined), no device involved.
//   ./mx_plan_check <cases> <seed>      prints "ok <cases> cases" or the first violated invariant
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "mx_plan.h"

static int fail(const char *what, long long a = 0, long long b = 0)
{
    std::printf("FAILED: %s (%lld, %lld)\n", what, a, b);
    return 1;
}

static int check(const std::vector<int64_t> &offs, int Lmin, int ROWS, int NR, int n_cu, int nt_max)
{
    const int64_t N = (int64_t)offs.size() - 1;
    std::vector<int64_t> voff;
    mx_virtual_offsets(offs.data(), N, voff);
    for (int64_t u = 0; u < N; ++u) {
        if (voff[u] & 1) return fail("utterance on an odd virtual row", u, voff[u]);
        if (voff[u + 1] - voff[u] < offs[u + 1] - offs[u]) return fail("virtual rows shorter than the utterance", u);
    }
    const MxPlan pl = mx_make_plan(offs.data(), voff.data(), N, Lmin, ROWS, NR, n_cu, nt_max);
    if (pl.unsupported) return fail("plan refused a small corpus");
    const int64_t P = offs[N];
    // (1) every frame that starts a window of the shortest filter -- and frame 0 of every non-empty utterance -- belongs to
    //     exactly one real record, as its first or second start; records lie inside their utterance and on even rows
    std::vector<int> owner((size_t)P, 0);
    int64_t u = 0;
    int64_t real = 0;
    for (size_t i = 0; i < pl.groups.size(); ++i) {
        const MxGroup &g = pl.groups[i];
        if (g.rem == 0) {   // filler: repeats the record before it, never at the head of a tile
            if (i % 16 == 0) return fail("filler record at the head of a tile", (long long)i);
            if (g.v0 != pl.groups[i - 1].v0) return fail("filler record off its predecessor's row", (long long)i);
            continue;
        }
        ++real;
        if (g.v0 & 1) return fail("pair on an odd virtual row", (long long)i, g.v0);
        while (u < N && !(offs[u] <= g.p0 && g.p0 < offs[u + 1])) ++u;   // (records come in stream order)
        if (u == N) return fail("record outside every utterance", (long long)i, g.p0);
        const int64_t T = offs[u + 1] - offs[u], t0 = g.p0 - offs[u];
        if (t0 & 1) return fail("pair on an odd frame of its utterance", (long long)i, t0);
        if (g.rem != T - t0) return fail("rem is not the rest of the utterance", (long long)i, g.rem);
        if (g.v0 - voff[u] != t0) return fail("virtual row and frame disagree", (long long)i);
        if (t0 > std::max<int64_t>(T - Lmin, 0)) return fail("pair behind the last window start", (long long)i, t0);
        owner[(size_t)g.p0] += 1;
        if (t0 + 1 < T) owner[(size_t)g.p0 + 1] += 1;
    }
    for (int64_t v = 0; v < N; ++v) {
        const int64_t T = offs[v + 1] - offs[v];
        if (T <= 0) continue;
        const int64_t last = std::max<int64_t>(T - Lmin, 0);
        for (int64_t t = 0; t < T; ++t) {
            const int want = t <= (last | 1) && t < T ? 1 : 0;   // the starts 0 .. last, rounded up to whole pairs
            if (owner[(size_t)(offs[v] + t)] != want) return fail("frame covered a wrong number of times", offs[v] + t, owner[(size_t)(offs[v] + t)]);
        }
    }
    // (2) tiles and workgroups fit the image; workgroups partition the tiles in order; the slots of the rounds suffice
    const int64_t n_tiles = ((int64_t)pl.groups.size() + 15) / 16;
    for (int64_t t = 0; t < n_tiles; ++t) {
        const int64_t a = t * 16, b = std::min<int64_t>((int64_t)pl.groups.size(), a + 16) - 1;
        if (pl.groups[(size_t)b].v0 + ROWS - pl.groups[(size_t)a].v0 > 2 * NR) return fail("tile outgrows the image", t);
        for (int64_t i = a; i < b; ++i)
            if (pl.groups[(size_t)i + 1].v0 < pl.groups[(size_t)i].v0) return fail("rows not ascending inside a tile", i);
    }
    int64_t next = 0;
    for (const MxWg &w : pl.wgs) {
        if (w.tile_beg != next || w.nt < 1 || w.nt > nt_max || w.nt > pl.target) return fail("workgroups do not partition the tiles", w.tile_beg, w.nt);
        next += w.nt;
        const int64_t gl = std::min<int64_t>((int64_t)pl.groups.size(), (int64_t)(w.tile_beg + w.nt) * 16) - 1;
        if (w.v_lo != pl.groups[(size_t)w.tile_beg * 16].v0) return fail("v_lo is not the first pair's row", w.tile_beg);
        if (w.nfr != pl.groups[(size_t)gl].v0 + ROWS - w.v_lo) return fail("nfr is not the span of the workgroup", w.tile_beg);
        if (w.nfr > 2 * NR || w.nfr < 1) return fail("workgroup outgrows the image", w.tile_beg, w.nfr);
        if ((int64_t)w.v_lo + w.nfr > voff[N] + 2 * NR + 2) return fail("workgroup reads past the slack behind the bit rows", w.tile_beg);
    }
    if (next != n_tiles) return fail("tiles left over", next, n_tiles);
    if (n_tiles && pl.target > 1 && (int64_t)pl.wgs.size() > (int64_t)n_cu * pl.rounds)
        return fail("more workgroups than the slots of the rounds the plan settled on", (long long)pl.wgs.size(), pl.rounds);
    // (3) fillers are few: at most 15 per closed tile, and only where a tile was closed for its span
    if ((int64_t)pl.groups.size() - real > 15 * n_tiles) return fail("too many filler records");
    return 0;
}

// e2m3 as the matrix instruction reads it: sign, exponent (bias 1), three mantissa bits
static double fp6_value(int code)
{
    const int e = (code >> 3) & 3, m = code & 7;
    const double mag = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
    return (code & 0x20) ? -mag : mag;
}

static int check_digits()
{
    for (long v = -4095; v <= 4095; ++v) {
        int d[3];
        mx_digits(v, d);
        if (d[0] < -8 || d[0] > 7 || d[1] < -8 || d[1] > 8 || d[2] < -16 || d[2] > 16) return fail("digit out of range", v);
        if (d[0] + 16L * d[1] + 256L * d[2] != v) return fail("digits do not recompose", v);
        // what the instruction computes: fp6 value x block scale (2^3, 2^7, 2^11), summed
        const double got = fp6_value(mx_fp6_code(d[0])) * 8.0 + fp6_value(mx_fp6_code(d[1])) * 128.0 + fp6_value(mx_fp6_code(d[2])) * 2048.0;
        if (got != (double)v) return fail("fp6 codes x block scales do not give the weight", v);
    }
    for (int d = -16; d <= 16; ++d)
        if ((mx_fp6_code(d) & ~0x3f) || fp6_value(mx_fp6_code(d)) != d / 8.0) return fail("fp6 code of a digit", d);
    return 0;
}

int main(int argc, char **argv)
{
    if (check_digits()) return 1;
    const int cases = argc > 1 ? std::atoi(argv[1]) : 200;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    // the shapes of the tree first: config 2 (253 workgroups of 21 tiles on 256 CUs) and its neighbour with two rounds
    {
        std::vector<int64_t> offs(1001);
        for (int i = 0; i <= 1000; ++i) offs[(size_t)i] = 200 * (int64_t)i;
        if (check(offs, 32, 33, 416, 256, 21)) return 1;
        std::vector<int64_t> voff;
        mx_virtual_offsets(offs.data(), 1000, voff);
        const MxPlan c2 = mx_make_plan(offs.data(), voff.data(), 1000, 32, 33, 416, 256, 21);
        if (c2.wgs.size() != 253 || c2.target != 21) return fail("config 2 is not 253 workgroups of 21 tiles", (long long)c2.wgs.size(), c2.target);
        const MxPlan two = mx_make_plan(offs.data(), voff.data(), 1000, 28, 37, 416, 256, 21);
        if (two.target != 11 || two.wgs.size() > 512) return fail("5438 tiles are not two rounds of 11", (long long)two.wgs.size(), two.target);
        if (check(offs, 28, 37, 416, 256, 21)) return 1;
    }
    for (int c = 0; c < cases; ++c) {
        const int N = uni(0, 120), Lmin = uni(2, 60), ROWS = Lmin + uni(1, 40);
        int NR = 16 * uni(1, 40);
        while (ROWS + 34 > 2 * NR) NR += 16;   // (the bank build's own condition)
        const int n_cu = (c % 3 == 0) ? 256 : uni(1, 12), nt_max = (c % 5 == 0) ? uni(1, 6) : 21;
        const int mode = uni(0, 3);   // lengths: anything / mostly shorter than the filters / barely longer / long
        std::vector<int64_t> offs(1, 0);
        for (int i = 0; i < N; ++i) {
            int T = mode == 0 ? uni(0, 300) : mode == 1 ? uni(0, Lmin + 3) : mode == 2 ? Lmin + uni(0, 3) : uni(200, 3000);
            if (uni(0, 19) == 0) T = 0;
            offs.push_back(offs.back() + T);
        }
        if (check(offs, Lmin, ROWS, NR, n_cu, nt_max)) {
            std::printf("case %d: N=%d Lmin=%d ROWS=%d NR=%d n_cu=%d nt_max=%d mode=%d\n", c, N, Lmin, ROWS, NR, n_cu, nt_max, mode);
            return 1;
        }
    }
    std::printf("ok %d cases\n", cases);
    return 0;
}
