"""gpurun_out/prof_<tag>_cfg<N>/ (scripts/profile_configs.sh) -> profiles/<tag>_kernel_stats_cfg<N>.csv + profiles/<tag>_configs.jsonl"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {1: "deit_tiny_b1", 2: "deit_small_b64", 3: "deit_base_b256", 4: "vit_base_b128", 5: "swin_tiny_b128"}
for c, nm in names.items():
    fs = glob.glob(f"{root}/gpurun_out/prof_{tag}_cfg{c}/*/*kernel_stats.csv")
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    keys = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(f"{root}/profiles/{tag}_kernel_stats_cfg{c}_{nm}.csv", "w") as f:
        w = csv.writer(f); w.writerow(keys)
        for r in rows:
            w.writerow([r[k] for k in keys])
    print(nm, [(r["Name"][:50], r["Calls"], r["AverageNs"], r["Percentage"]) for r in rows[:6]])
src = f"{root}/gpurun_out/{tag}_configs.jsonl"
if os.path.exists(src):
    shutil.copy(src, f"{root}/profiles/{tag}_configs.jsonl")
    print(open(src).read())

# counter passes (PMC_CFGS of profile_configs.sh): per kernel and launch, FETCH_SIZE / WRITE_SIZE in KiB and the corrected fabric
# bytes (2 x FETCH + WRITE) x 1024 as MI355X_MICROARCH.md prescribes
import collections, re
for c, nm in names.items():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{root}/gpurun_out/pmc_{tag}_cfg{c}_{k}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                mm = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
                agg[mm.group(1) if mm else r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        continue
    out = {}
    for kn, v in agg.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fm, wm = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            out[kn] = {"launches_profiled": len(v["FETCH_SIZE"]), "FETCH_SIZE_KiB_mean": round(fm, 1), "WRITE_SIZE_KiB_mean": round(wm, 1),
                       "bytes_per_launch_corrected": round((2.0 * fm + wm) * 1024.0)}
    json.dump({"tag": tag, "config": c, "name": nm, "pmc": out}, open(f"{root}/profiles/{tag}_pmc_cfg{c}_{nm}.json", "w"), indent=1)
    tot = sum(o["bytes_per_launch_corrected"] * o["launches_profiled"] for o in out.values())
    print(nm, "counter traffic over the profiled run: %.1f MB" % (tot / 1e6))
