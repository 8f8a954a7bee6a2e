"""The kernel summary table of DESIGN.md section 4, generated from the evidence files under profiles/ so that the prose
cannot drift from them (round-2 review, item 7/9):
    python tools/design_table.py [round=r03] > /tmp/table.md        # or: --write to splice it into DESIGN.md
Every cell names the file it comes from; a missing file yields "n/a" (never a remembered number)."""
import csv, json, os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
R = next((a for a in sys.argv[1:] if re.fullmatch(r"r\d\d", a)), "r04")
ROUNDS = ["r%02d" % k for k in range(int(R[1:]), 0, -1)]   # newest first: a file missing in round R is taken from the round before


def P(name):
    """Path of the newest profiles/rNN_<name> at or before round R (a row whose evidence was not re-collected this round
    keeps citing the file it was measured in)."""
    for r in ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_%s" % (r, name))
        if os.path.exists(path):
            return path
    return os.path.join(ROOT, "profiles", "%s_%s" % (R, name))


def tag(name):
    """The file name a cell cites: `rNN_<name>` of the round the file was found in."""
    return os.path.basename(P(name))


def kavg(csv_name, needle):
    """(average us, calls) of the first kernel whose name contains `needle` in a rocprofv3 kernel_stats.csv."""
    try:
        for row in csv.DictReader(open(P(csv_name))):
            if needle in row["Name"]:
                return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    except OSError:
        pass
    return None, 0


def pmc(txt_name, kernel, counter):
    try:
        name = None
        for line in open(P(txt_name)):
            if not line.startswith(" "):
                name = line
            m = re.match(r"\s+%s\s+n=(\d+)\s+mean\s+([0-9.e+]+)" % counter, line)
            if m and name and kernel in name:
                return float(m.group(2))
    except OSError:
        pass
    return None


def jload(name):
    try:
        txt = open(P(name)).read().strip().splitlines()[-1]
        return json.loads(txt) if txt.startswith("{") else json.load(open(P(name)))
    except (OSError, ValueError, IndexError):
        try:
            return json.load(open(P(name)))
        except (OSError, ValueError):
            return None


def f(x, fmt="%.1f", na="n/a"):
    return na if x is None else fmt % x


rows = []
bench = jload("bench_line.json")
ALG_FLOPS, ALG_BYTES = 31.15008e9, 73.2e6
# ---- the headline kernel of round 4: the Toeplitz form on the block-scaled matrix instruction ----------------------
usx, callsx = kavg("bench_mx_kernel_stats.csv", "mx_bank_kernel")
if bench and bench.get("config", {}).get("mode") == "mx":
    evx = bench["roofline"]["kernel_ms_per_launch"] * 1e3
    trx = jload("traffic_mx.json")
    busyx, sqbx = pmc("pmc_sq1_mx.txt", "mx_bank_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"), pmc("pmc_sq1_mx.txt", "mx_bank_kernel", "SQ_BUSY_CYCLES")
    rows.append(("`mx_bank_kernel<3>` (`TSR_FAST|TSR_MX`: 13-bit weights, fp6 x fp4, Toeplitz form)", "config 2, every launch reads its corpus' bit rows from HBM",
                 "31.15 Gflop, 73.2 MB (366 B/frame)",
                 "%s us hipEvents in `bench.py` (`%s`); %s us rocprofv3 average of %d launches of a profiled run (`%s`)"
                 % (f(evx, "%.2f"), tag("bench_line.json"), f(usx, "%.2f"), callsx, tag("bench_mx_kernel_stats.csv")),
                 "%s TB/s of algorithmic bytes = %s of the 8 TB/s HBM peak; %s Tflop/s = %s of the 10 PF fp6 peak (%s counting the three digit products, K padding and the 33rd tap row); "
                 "PMC traffic %s MB = %sx algorithmic (`%s`: the kernel reads bit rows, 48 B per frame); matrix pipe busy %s of the launch at the nominal 16 cycles per instruction (`%s`)"
                 % (f(ALG_BYTES / evx / 1e6, "%.2f"), f(ALG_BYTES / evx / 1e6 / 8, "%.3f"), f(ALG_FLOPS / evx / 1e6, "%.0f"), f(ALG_FLOPS / evx / 1e6 / 10000, "%.3f"),
                    f(bench["roofline"]["mfma"]["issued"]["frac_of_peak"], "%.2f"),
                    f(trx["hbm_bytes_per_launch"] / 1e6 if trx else None), f(trx["ratio_to_algorithmic"] if trx else None, "%.2f"), tag("traffic_mx.json"),
                    f(100.0 * busyx / (1024.0 * sqbx / 32.0) if busyx and sqbx else None, "%.0f %%"), tag("pmc_sq1_mx.txt")),
                 "per-wave issue of the scaled matrix instruction (two waves per SIMD end at ~16 cycles per instruction and SIMD, `r04_mx_rate_ubench.txt`) + the un-overlapped staging / exchange / finishing phases (section 4.0)"))
    x10 = bench.get("configs", {}).get("config2_x10_one_launch")
    if x10:
        rows.append(("`mx_bank_kernel<3>`, 10 000 utterances in one launch", "config 2 ten times over", "311.5 Gflop, 732 MB",
                     "%s us per launch = %s us per 1000 utterances (`%s`, `configs.config2_x10_one_launch`)"
                     % (f(x10["kernel_ms_per_launch"] * 1e3, "%.1f"), f(x10["kernel_us_per_1000_utterances"], "%.2f"), tag("bench_line.json")),
                     "%s TB/s of algorithmic bytes" % f(x10["hbm_gbs_alg"] / 1e3, "%.2f"),
                     "as above: one workgroup per CU and round, nothing to amortise over a longer launch"))
# ---- the dominant kernel ------------------------------------------------------------------------------------------
us, calls = kavg("bench_fast_kernel_stats.csv", "tap2_bank_kernel")
ev = None
if bench:
    ev = (bench["modes"]["fast"]["kernel_ms_per_launch"] if bench.get("config", {}).get("mode") == "mx" and "fast" in bench.get("modes", {})
          else bench["roofline"]["kernel_ms_per_launch"]) * 1e3
tr = jload("traffic_fast.json")
busy, sqb = pmc("pmc_sq1.txt", "tap2_bank_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"), pmc("pmc_sq1.txt", "tap2_bank_kernel", "SQ_BUSY_CYCLES")
wavecyc = pmc("pmc_sq1.txt", "tap2_bank_kernel", "SQ_WAVE_CYCLES")
rows.append(("`tap2_bank_kernel<23,0>` (`TSR_FAST`, narrow rows)", "config 2, every launch streams its frames from HBM",
             "31.15 Gflop, 73.2 MB (366 B/frame)",
             "%s us hipEvents in `bench.py` (`%s`); %s us rocprofv3 average of %d launches of a profiled run (`%s`)"
             % (f(ev, "%.2f"), tag("bench_line.json"), f(us, "%.2f"), calls, tag("bench_fast_kernel_stats.csv")),
             "%s Tflop/s = %s of the 5 PF int8 peak; %s TB/s = %s of HBM; PMC traffic %s MB = %sx algorithmic (`%s`); matrix pipe busy %s of the launch (`SQ_VALU_MFMA_BUSY_CYCLES` / 1024 SIMDs / launch cycles, `%s`)"
             % (f(ALG_FLOPS / ev / 1e6 if ev else None, "%.0f"), f(ALG_FLOPS / ev / 1e6 / 5000 if ev else None, "%.3f"),
                f(ALG_BYTES / ev / 1e6 if ev else None, "%.2f"), f(ALG_BYTES / ev / 1e6 / 8 if ev else None, "%.3f"),
                f(tr["hbm_bytes_per_launch"] / 1e6 if tr else None), f(tr["ratio_to_algorithmic"] if tr else None, "%.2f"), tag("traffic_fast.json"),
                # SQ_BUSY_CYCLES sums 32 shader engines (8 XCDs x 4): / 32 = cycles of the launch; 1024 SIMDs on the chip
                f(100.0 * busy / (1024.0 * sqb / 32.0) if busy and sqb else None, "%.0f %%"), tag("pmc_sq1.txt")),
             "instruction issue beside the matrix pipe + the per-workgroup prologue (section 4.1)"))
if bench and "modes" in bench:
    for mode, kern, note in (("fast24", "alt_bank_kernel<12,3>", "three limbs"), ("exact", "exact_mfma_kernel", "scalar issue (operand funnel)")):
        m = bench["modes"].get(mode)
        if m:
            us2, c2 = kavg("bench_%s_kernel_stats.csv" % mode, kern.split("<")[0])
            rows.append(("`%s` (`%s`)" % (kern, "TSR_FAST|TSR_W24" if mode == "fast24" else "TSR_EXACT"), "config 2", "same",
                         "%s us hipEvents (`%s`, `modes.%s`); %s us rocprofv3 average (`%s`)"
                         % (f(m["kernel_ms_per_launch"] * 1e3, "%.1f"), tag("bench_line.json"), mode, f(us2, "%.1f"), tag("bench_%s_kernel_stats.csv" % mode)),
                         "%s Tflop/s algorithmic; %s M frames/s" % (f(m["alg_tflops"], "%.0f"), f(m["frames_per_s"] / 1e6, "%.0f")), note))
    x10 = bench.get("configs", {}).get("config2_x10_one_launch") if bench.get("config", {}).get("mode") != "mx" else None
    if x10:
        rows.append(("`tap2_bank_kernel<23,0>`, 10 000 utterances in one launch", "config 2 ten times over (prologue amortised 10x)", "311.5 Gflop, 732 MB",
                     "%s us per launch = %s us per 1000 utterances (`%s`, `configs.config2_x10_one_launch`)"
                     % (f(x10["kernel_ms_per_launch"] * 1e3, "%.1f"), f(x10["kernel_us_per_1000_utterances"], "%.2f"), tag("bench_line.json")),
                     "%s of the int8 peak; %s TB/s of algorithmic bytes" % (f(x10["frac_of_int8_peak"], "%.3f"), f(x10["hbm_gbs_alg"] / 1e3, "%.2f")),
                     "the steady-state loop (matrix pipe 81 % busy inside it)"))
# ---- config 3 -----------------------------------------------------------------------------------------------------
c3 = {k: kavg("config3_kernel_stats.csv", k) for k in ("code_parts_mfma_kernel", "code_parts_recheck_kernel", "spread_", "mx_bank_kernel",
                                                        "fast_bank_kernel<11, 2, false, 2, 1, true>", "frag_epilogue_kernel", "part_chain_setup_kernel")}
t3 = jload("config3_traffic.json")
if c3["code_parts_mfma_kernel"][0]:
    if c3["mx_bank_kernel"][0]:   # round 4: the bank pass on the block-scaled kernel, feature phases inside the workgroup
        bank = c3["mx_bank_kernel"][0]
        bank_label = "`mx_bank_kernel<3,true,PHASES>` (`TSR_FAST|TSR_MX`: six feature phases over the corpus' bit form, one launch, no epilogue)"
        bound = "instructions beside the matrix cores (coder: three per part-slot key); bank: as the config-2 kernel, per phase"
    else:
        # bank launches per pass: one since the K-phases share a launch (one epilogue per pass either way)
        nb, ne = c3["fast_bank_kernel<11, 2, false, 2, 1, true>"][1], c3["frag_epilogue_kernel"][1]
        per_pass = (nb / ne) if ne else 1
        bank = per_pass * c3["fast_bank_kernel<11, 2, false, 2, 1, true>"][0] + c3["frag_epilogue_kernel"][0]
        bank_label = "`fast_bank_kernel<11,2,bits>` (6 K-phases in %s launch%s) + epilogue" % ("one" if per_pass < 1.5 else "%d" % round(per_pass), "" if per_pass < 1.5 else "es")
        bound = "instructions beside the matrix cores (coder: three per part-slot key; bank: 22 MFMAs per skew emission)"
    setup = c3["part_chain_setup_kernel"][0] or 0.0
    tot = c3["code_parts_mfma_kernel"][0] + c3["code_parts_recheck_kernel"][0] + c3["spread_"][0] + bank + setup
    rows.append(("config-3 chain: setup + `code_parts_mfma_kernel<7>` + recheck + `spread_lane_bits_kernel` (dilation into the bit form) + " + bank_label,
                 "config 3: 8.2 M locations x 50 parts; D = 2050 bank", "164 + 177 Gflop, 73.2 MB algorithmic",
                 "%s + %s + %s + %s + %s = %s us of kernels (`%s`)"
                 % (f(setup), f(c3["code_parts_mfma_kernel"][0]), f(c3["code_parts_recheck_kernel"][0]), f(c3["spread_"][0]), f(bank), f(tot), tag("config3_kernel_stats.csv")),
                 "%s Tflop/s over the chain; PMC traffic %s MB = %sx algorithmic (`%s`; round 3: 317 MB, round 2: ~950 MB by its own sizes)"
                 % (f(341e9 / tot / 1e6, "%.0f"), f(t3["hbm_bytes_per_chain"] / 1e6 if t3 else None, "%.0f"), f(t3["ratio_to_algorithmic"] if t3 else None, "%.1f"), tag("config3_traffic.json")),
                 bound))
# ---- config 4 share -----------------------------------------------------------------------------------------------
us4, c4 = kavg("c45_kernel_stats.csv", "tap2_bank_kernel")
if us4:
    rows.append(("`tap2_bank_kernel<23,3>` (10 column groups, XCD-aware) + key decode", "config 4 share: 1250 utterances, M = 64", "338 Gflop (valid windows)",
                 "%s us rocprofv3 average (`%s`); per-pass hipEvent times of all five variants: `%s`" % (f(us4), tag("c45_kernel_stats.csv"), tag("config45_timing.txt")),
                 "%s Tflop/s = %s of the int8 peak" % (f(338e9 / us4 / 1e6, "%.0f"), f(338e9 / us4 / 1e6 / 5000, "%.3f")), "as config 2"))
# ---- next rows ----------------------------------------------------------------------------------------------------
for csvn, needle, label, work, alg, bound in (
        ("classify_kernel_stats.csv", "iso_mfma_kernel", "`iso_mfma_kernel` (isolated windows on the f32 matrix cores)", "133 k positions x 8 templates (N1)", "24.5 Gflop in `detect_float`'s order", "matrix pipe ~50 % busy: half-empty tiles of rare padding patterns, per-lane operand gathers"),
        ("classify_kernel_stats.csv", "isolated_windows_kernel", "`isolated_windows_kernel` (vector kernel: `TSR_EXACT_KERNEL_VALU`, banks without the matrix-core form)", "133 k positions x 8 templates (N1)", "24.5 Gflop in `detect_float`'s order", "vector instructions (f32 chains in the reference's order)"),
        ("edges_kernel_stats.csv", "edge_block_bin_kernel", "`edge_block_bin_kernel<40>`", "1000 spectrograms -> edge corpus (N3)", "72 MB in, 74 MB out", "instruction issue (stencil + two selects per wave); round 2: 445 us"),
        ("edges_kernel_stats.csv", "edge_block_wave_kernel", "`edge_block_wave_kernel<32>`", "1000 spectrograms -> edge corpus (N3)", "72 MB in, 74 MB out", "the per-block radix select")):
    u, c = kavg(csvn, needle)
    if u:
        extra = ("%s Tflop/s" % f(24.5e9 / u / 1e6, "%.1f")) if ("isolated" in needle or "iso_" in needle) else ("%s TB/s" % f(146e6 / u / 1e6, "%.2f"))
        rows.append((label, work, alg, "%s us rocprofv3 average of %d (`%s`)" % (f(u), c, tag(csvn)), extra, bound))
out = ["| kernel | workload | algorithmic work per launch | time (source) | achieved | bound by |", "|---|---|---|---|---|---|"]
out += ["| " + " | ".join(r) + " |" for r in rows]
table = "\n".join(out)
if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, b = s.index("<!-- table:begin -->"), s.index("<!-- table:end -->")
    open(path, "w").write(s[:a] + "<!-- table:begin -->\n" + table + "\n" + s[b:])
    print("DESIGN.md table rewritten from profiles/ (newest file per row, round %s and before)" % R)
else:
    print(table)
