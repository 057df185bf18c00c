"""Developer helper: sum rocprofv3 --pmc counter CSVs (gpurun_out/pmc_*/**/*counter_collection.csv) per kernel."""
import csv, glob, collections, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{root}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = next((n for n in ("fd_residual", "spec_xpass", "spec_ypass", "spec_rowmarch") if n in k), None)
        if not short: continue
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[short][r["Counter_Name"]] += 1
for k in sorted(acc):
    n = max(cnt[k].values())
    print(k, f"({n} dispatch-rows)")
    for c in sorted(acc[k]): print(f"   {c:28s} {acc[k][c] / cnt[k][c] * 1.0:14.4g}  per dispatch-row (x{cnt[k][c]})")
