"""Condense gpurun_out/pmc_rows_<tag>_* (scripts/pmc_rows.sh) into per-kernel counter means."""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_rows_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        mm = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        agg[mm.group(1) if mm else r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(x) / len(x), 1) for c, x in v.items()} for k, v in agg.items()}
json.dump(out, open(f"{root}/profiles/{tag}_row_kernel_counters.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, json.dumps(v))
