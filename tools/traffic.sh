# HBM traffic of the dominant kernel from the TCC memory-side counters (GPU box).
#   MODE=fast bash tools/traffic.sh
# Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass); on gfx950 FETCH_SIZE
# reports half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), so the
# read side is doubled; both counters are in KiB.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
MODE=${MODE:-fast}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/tr_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/tr_$C -o tr -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $MODE > gpurun_out/tr_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for fn in glob.glob("gpurun_out/tr_%s/**/*counter_collection.csv" % C, recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == C:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    cand = [k for k in acc if "bank_kernel" in k] or list(acc)  # the scoring kernel, not the upload
    name = max(cand, key=lambda k: sum(acc[k]))
    vals = acc[name]
    out[C] = {"kernel": name[:80], "launches": len(vals), "mean_KiB": sum(vals) / len(vals)}
fetch, write = out["FETCH_SIZE"]["mean_KiB"], out["WRITE_SIZE"]["mean_KiB"]
res = {"mode": "$MODE", "counters": out,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> read side x2",
       "read_bytes_per_launch": 2 * fetch * 1024, "write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024}
json.dump(res, open("gpurun_out/traffic_$MODE.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
