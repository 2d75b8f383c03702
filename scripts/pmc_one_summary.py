"""Condense the rocprofv3 outputs of scripts/pmc_one.sh: python3 pmc_one_summary.py <dir> <kernel substring>"""
import csv, glob, os, sys, json
from collections import defaultdict
d, key = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Name"]:
            out.setdefault(r["Name"][:90], {})["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2); out[r["Name"][:90]]["calls"] = int(r["Calls"])
for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
