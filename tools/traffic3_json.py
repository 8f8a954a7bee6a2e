"""HBM-side bytes of the config-3 chain per kernel from the two PMC passes of tools/traffic3.sh (pmc_summary.py text)
-> JSON: per kernel calls, mean FETCH_SIZE / WRITE_SIZE (KiB), corrected bytes per chain (gfx950: FETCH_SIZE counts
64 B per 128-B request of a wide coalesced stream -> read side x2, MI355X_MICROARCH.md), and the total against the
algorithmic 366 B per frame."""
import json, re, sys
txt, out = sys.argv[1:3]
sec, name, data = None, None, {}
for line in open(txt):
    m = re.match(r"== (\w+) ", line)
    if m:
        sec = m.group(1)
        continue
    if not line.startswith(" "):
        name = line.strip().replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+mean\s+([0-9.e+]+)\s+sum\s+([0-9.e+]+)", line)
    if m and name and sec == m.group(1):
        data.setdefault(name, {})[sec] = {"calls": int(m.group(2)), "mean_KiB": float(m.group(3)), "sum_KiB": float(m.group(4))}
# the kernels of the chain, by the stem of their names (the dilation kernel of the bit form changed during round 3)
stems = ["part_chain_setup_kernel", "code_parts_mfma_kernel", "code_parts_recheck_kernel", "spread_", "fast_bank_kernel", "frag_epilogue_kernel", "mx_bank_kernel",
         "out_geometry_kernel"]
chain = [k for st in stems for k in data if st in k]
front = next(v for k, v in data.items() if "code_parts_mfma_kernel" in k)["FETCH_SIZE"]["calls"]   # front-end calls in the run
per = {}
total = 0.0
for k in chain:
    if k not in data:
        continue
    f, w = data[k].get("FETCH_SIZE", {"sum_KiB": 0, "calls": 0}), data[k].get("WRITE_SIZE", {"sum_KiB": 0, "calls": 0})
    if "mx_bank" in k:                                # round 4, TSR_FAST | TSR_MX: one launch per bank pass (feature phases inside the workgroup)
        passes = float(f["calls"])
        rd, wr = 2 * f["sum_KiB"] * 1024 / passes, w["sum_KiB"] * 1024 / passes
    elif "fast_bank" in k or "frag_epilogue" in k:    # bank passes (several per front-end call): one epilogue each; the phases share a launch since the middle of round 3
        passes = float(next(v for kk, v in data.items() if "frag_epilogue" in kk)["FETCH_SIZE"]["calls"])
        rd, wr = 2 * f["sum_KiB"] * 1024 / passes, w["sum_KiB"] * 1024 / passes
    else:
        rd, wr = 2 * f["sum_KiB"] * 1024 / front, w["sum_KiB"] * 1024 / front
    per[k] = {"read_MB": rd / 1e6, "write_MB": wr / 1e6}
    total += rd + wr
alg = 366 * 200000
j = {"workload": "BASELINE config 3: 1000 x 200 frames x 45 x 8 edges -> code_parts(K=50) -> argmax -> spread(2,2) -> D=2050 bank, bit form",
     "per_kernel_per_chain": per, "hbm_bytes_per_chain": total, "algorithmic_bytes": alg, "ratio_to_algorithmic": total / alg,
     "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); both counters in KiB",
     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/time_config3.py 1000 bits 3 %s (two passes; tools/traffic3.sh)"
                % (sys.argv[3] if len(sys.argv) > 3 else "fast")}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j)[:600])
