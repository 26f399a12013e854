"""Summarise rocprofv3 --pmc counter_collection csv files: mean counter value per kernel."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "s360" not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
