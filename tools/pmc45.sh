cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf gpurun_out/p45_$C
timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/p45_$C -o p45 -- python3 tools/time_config45.py > gpurun_out/p45_$C.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/p45_$C/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if row["Counter_Name"] == "$C":
            acc[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("$C %-72s n=%d mean %.1f KiB" % (k, len(v), sum(v) / len(v)))
PY
done
