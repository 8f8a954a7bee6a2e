// mx_plan.h -- the host-side plan of TSR_FAST | TSR_MX (csrc/detect_mx.hip): pair records, tiles, workgroups.
// Plain C++ without a device call, so that tests/test_mx_plan_cpu.py can compile it with g++ and hold it to its
// invariants on random corpora (every window start in exactly one record, every tile and workgroup inside an LDS image).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

struct MxGroup {   // one tile row: the window starts t0 and t0 + 1 of one utterance
    int32_t v0;    // virtual position of t0 (utterances padded to even lengths: v0 is even)
    int32_t p0;    // stream frame of t0
    int32_t rem;   // frames the utterance has left from t0 on (T - t0); 0: a filler record that scores nothing
    int32_t pad;
};
struct MxWg {      // one workgroup (read by the kernel as an int4)
    int32_t tile_beg, nt, v_lo, nfr;   // first tile, tiles, first virtual row, rows to stage
};
struct MxPlan {
    std::vector<MxGroup> groups;
    std::vector<MxWg> wgs;
    int target = 0;            // tiles per workgroup the plan aimed at (nt_max when the corpus fits one round of workgroups)
    int rounds = 0;            // rounds of n_cu workgroups it settled on
    bool unsupported = false;  // positions beyond 31 bits
};

// Virtual rows: every utterance starts on an even row (a pair of starts must begin in frame parity 0 of the LDS image).
inline void mx_virtual_offsets(const int64_t *offsets, int64_t N, std::vector<int64_t> &voff)
{
    voff.assign((size_t)N + 1, 0);
    for (int64_t u = 0; u < N; ++u) voff[(size_t)u + 1] = voff[(size_t)u] + ((offsets[u + 1] - offsets[u] + 1) & ~(int64_t)1);
}

// offsets / voff: [N + 1] stream frames / virtual rows of the utterances; Lmin: the bank's shortest filter; ROWS: its longest
// + 1 (tap rows a pair reads); NR: rows per LDS array (an image holds 2 NR frames); n_cu: workgroup slots per round;
// nt_max: tiles a workgroup can hold.
inline MxPlan mx_make_plan(const int64_t *offsets, const int64_t *voff, int64_t N, int Lmin, int ROWS, int NR, int n_cu, int nt_max)
{
    MxPlan pl;
    // A tile is 16 consecutive records.  Where 16 pairs would span more rows than an LDS image holds (utterances barely longer
    // than the filters give one pair per T rows), the tile is closed early and filled up with records that score nothing
    // (rem = 0: no window, no frame of their own), so that every tile fits a workgroup whatever the utterance lengths are.
    std::vector<MxGroup> &groups = pl.groups;
    const int64_t span_max = 2 * (int64_t)NR;
    int in_tile = 0;
    int64_t tile_v0 = 0;
    for (int64_t u = 0; u < N; ++u) {
        const int64_t T = offsets[u + 1] - offsets[u];
        if (T <= 0) continue;
        const int64_t ng = (std::max<int64_t>(T - Lmin + 1, 1) + 1) / 2;
        for (int64_t g = 0; g < ng; ++g) {
            const int64_t v0 = voff[u] + 2 * g, p0 = offsets[u] + 2 * g;
            if (v0 > INT32_MAX - 4096 || p0 > INT32_MAX - 4096 || groups.size() > (size_t)INT32_MAX / 32) {
                pl.unsupported = true;
                return pl;
            }
            if (in_tile == 16) in_tile = 0;
            if (in_tile > 0 && v0 + ROWS - tile_v0 > span_max) {
                const MxGroup last = groups.back();
                for (; in_tile < 16; ++in_tile) groups.push_back(MxGroup{last.v0, last.p0, 0, 0});
                in_tile = 0;
            }
            if (in_tile == 0) tile_v0 = v0;
            groups.push_back(MxGroup{(int32_t)v0, (int32_t)p0, (int32_t)std::min<int64_t>(T - 2 * g, INT32_MAX), 0});
            ++in_tile;
        }
    }
    const int64_t n_groups = (int64_t)groups.size(), n_tiles = (n_groups + 15) / 16;
    auto span_end = [&](int64_t tile_last) {   // one past the last virtual row the tiles up to tile_last read
        const int64_t gl = std::min<int64_t>(n_groups, (tile_last + 1) * 16) - 1;
        return (int64_t)groups[(size_t)gl].v0 + ROWS;
    };
    // Tiles per workgroup.  A launch takes as long as its slowest CU: with one workgroup per CU and round, the tiles are
    // spread evenly over (CUs x rounds) workgroups instead of filling each to nt_max -- 5438 tiles are two rounds of 11 per
    // workgroup (512 slots), not one round of 21 and a second one for the three workgroups that did not fit (which costs a
    // whole round: measured 44 vs 24 us at config 2's size with Lmin = 28).  Where the image's span cuts workgroups short,
    // the count can still exceed the slots: one more round, smaller shares, until it fits.
    const int64_t cus = std::max(1, n_cu);
    int64_t rounds = std::max<int64_t>(1, (n_tiles + cus * nt_max - 1) / (cus * nt_max));
    pl.target = nt_max;
    for (; n_tiles > 0; ++rounds) {
        const int64_t target = std::min<int64_t>(nt_max, std::max<int64_t>(1, (n_tiles + cus * rounds - 1) / (cus * rounds)));
        pl.wgs.clear();
        for (int64_t t = 0; t < n_tiles;) {
            const int64_t v_lo = groups[(size_t)(t * 16)].v0;
            int64_t cnt = 1;   // (one tile always fits: tiles are closed before they outgrow the image)
            while (cnt < target && t + cnt < n_tiles && span_end(t + cnt) - v_lo <= span_max) ++cnt;
            pl.wgs.push_back(MxWg{(int32_t)t, (int32_t)cnt, (int32_t)v_lo, (int32_t)(span_end(t + cnt - 1) - v_lo)});
            t += cnt;
        }
        pl.target = (int)target;
        pl.rounds = (int)rounds;
        if ((int64_t)pl.wgs.size() <= cus * rounds || target == 1) break;
    }
    return pl;
}

// ---- the weights' arithmetic (host side of the operand build) ------------------------------------------------------
// A fixed-point weight |v| <= 4095 as three balanced radix-16 digits, v = d0 + 16 d1 + 256 d2 with d0 in -8 .. 7,
// d1 in -8 .. 8, d2 in -16 .. 16: every digit / 8 is an exact fp6 (e2m3) value, the instruction's block scales carry
// 2^3, 2^7 and 2^11.
inline void mx_digits(long v, int d[3])
{
    const long l2 = (v + 128) >> 8;   // (arithmetic shift: floor)
    const long r = v - 256 * l2;      // -128 .. 127
    const long l1 = (r + 8) >> 4;
    const long l0 = r - 16 * l1;
    d[0] = (int)l0; d[1] = (int)l1; d[2] = (int)l2;
}
// digit / 8 as e2m3 (sign, two exponent bits of bias 1, three mantissa bits): magnitudes 0 .. 15 are their own code
// (subnormals 0 .. 7 / 8, then 1.xxx), 16 (= 2.0) is exponent 2 with mantissa 0
inline int mx_fp6_code(int d)
{
    const int mag = d < 0 ? -d : d;
    return (mag & 31) | (d < 0 ? 0x20 : 0);
}
