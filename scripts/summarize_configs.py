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
