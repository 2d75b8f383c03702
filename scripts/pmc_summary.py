"""Mean of every counter per kernel over the passes scripts/pmc_gemm.sh wrote: pmc_summary.py gpurun_out/<tag>"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        mm = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        if mm:
            agg[mm.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v):
        print(f"   {c:32s} {sum(v[c]) / len(v[c]):16.1f}")
