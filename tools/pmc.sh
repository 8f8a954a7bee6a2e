# PMC pass over tools/time_fast.py (GPU box).  usage: bash tools/pmc.sh "<counters>" [tag]
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
TAG=${2:-pmc}
rm -rf gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/$TAG -o $TAG -- python3 tools/time_fast.py ${3:-} > gpurun_out/$TAG.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/$TAG.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$TAG/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for row in csv.DictReader(open(fn)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
