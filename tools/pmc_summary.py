"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of every counter over dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-40:]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if "at::" in k or "rocclr" in k: continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "n=%d" % len(next(iter(cs.values()))))
