"""Per kernel and counter: number of dispatches, mean and sum of a rocprofv3 --pmc run (counter_collection csv)."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    print(k[:150])
    for c, v in sorted(d.items()):
        print("   %-28s n=%-6d mean %.6g   sum %.6g" % (c, len(v), sum(v) / len(v), sum(v)))
