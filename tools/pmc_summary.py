"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, sys, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            out[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, ctrs in sorted(out.items()):
    if not k.startswith("awsm::"):
        continue
    print(k)
    for c, v in sorted(ctrs.items()):
        print(f"   {c:22s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
