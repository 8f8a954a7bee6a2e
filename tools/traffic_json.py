"""HBM bytes per launch of the dominant bank kernel from the two PMC passes of tools/profile_r03.sh traffic
(pmc_summary.py texts of FETCH_SIZE and WRITE_SIZE) -> profiles/r02_traffic_<mode>.json, the file bench.py quotes in
roofline.traffic.  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, read side x2
(MI355X_MICROARCH.md, HBM section); both counters are in KiB."""
import json, re, sys
fetch_txt, write_txt, mode, out = sys.argv[1:5]
working_set = sys.argv[5] if len(sys.argv) > 5 else None   # (round 4: given by the caller)
how = sys.argv[6] if len(sys.argv) > 6 else "--replicas 16 (tools/profile_r03.sh traffic)"


def dominant(path, counter):
    best = None
    name = None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
            continue
        m = re.match(r"\s+%s\s+n=(\d+)\s+mean\s+([0-9.e+]+)" % counter, line)
        if m and name and "bank_kernel" in name:
            n, mean = int(m.group(1)), float(m.group(2))
            if best is None or n * mean > best[1] * best[2]:
                best = (name, n, mean)
    return best


kf, nf, fetch = dominant(fetch_txt, "FETCH_SIZE")
kw, nw, write = dominant(write_txt, "WRITE_SIZE")
assert kf == kw, (kf, kw)
alg = 366 * 200000
kname = re.search(r"(\w+_bank_kernel(<[^>]*>)?)", kf)
j = {"mode": mode, "kernel": kname.group(1) if kname else kf,
     "working_set": working_set or "16 device-resident copies of the config-2 corpus (1.2 GB, 4.7x the 256 MiB Infinity Cache), each launch "
                                   "streams its 73.6 MB of frames from HBM",
     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 "
                "--no-cpu-baseline --no-modes --mode %s %s (two separate passes)" % (mode, how),
     "counters": {"FETCH_SIZE": {"launches": nf, "mean_KiB": fetch}, "WRITE_SIZE": {"launches": nw, "mean_KiB": write}},
     "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> read side x2 "
                   "(MI355X_MICROARCH.md, HBM)",
     "read_bytes_per_launch": 2 * fetch * 1024, "write_bytes_per_launch": write * 1024,
     "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "algorithmic_bytes_per_launch": alg,
     "ratio_to_algorithmic": (2 * fetch + write) * 1024 / alg}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j)[:400])
