"""Summarise rocprofv3 CSV output: per-kernel counter sums / dispatch counts and the kernel stats table."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
show_all = len(sys.argv) > 2
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "k_join" in k or show_all:
            n = max(1, len(disp[k]))
            print(f"{f.split('/')[-3]} :: {k}  dispatches={n}")
            for c, x in sorted(v.items()):
                print(f"   {c:32s} total {x:.6g}   per-dispatch {x / n:.6g}")
for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
    print(f.split('/')[-3], "kernel_stats")
    for r in csv.DictReader(open(f)):
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):12.0f} total_ns {r['TotalDurationNs']:>12s} {r['Percentage']:>6s}%")
