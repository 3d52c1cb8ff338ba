"""Idle time between the dispatches of one bench step, from a rocprofv3 --kernel-trace CSV:
    python tools/gaps.py <dir with *_kernel_trace.csv>
Prints, for the last full step (k_zero_regions ... k_join), every kernel with its duration and the gap in front of it."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i - 1 for i, r in enumerate(rows) if "k_prep_sources" in r["Kernel_Name"] and i > 0]   # (the build's zeroing launch precedes it)
if len(starts) < 3:
    raise SystemExit("no steps found")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = 0
prev_end = t0
print(f"{'kernel':60s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-58:]
    print(f"{name:60s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}")
    busy += e - s
    prev_end = max(prev_end, e)
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"step span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {len(step)} dispatches")
