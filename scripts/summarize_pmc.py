"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel means (KB per launch)."""
import collections, csv, sys
agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            agg[(row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
print("kernel,counter,launches,mean_per_launch")
for (k, cn), v in sorted(agg.items()):
    if k.startswith("k_") or "fillBuffer" in k or "geo" in k:
        print(f"{k},{cn},{len(v)},{sum(v) / len(v):.3f}")
