import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp = collections.Counter()
    for r in csv.DictReader(open(f)):
        disp[r["Kernel_Name"][:40]] = max(disp[r["Kernel_Name"][:40]], 0)
    for k, v in acc.items():
        if "k_join" in k or len(sys.argv) > 2:
            print(f, k)
            for c, x in sorted(v.items()): print(f"   {c:32s} {x:.4g}")
for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
    print(f); print(open(f).read()[:3000])
