"""Kernels of a rocprofv3 --kernel-trace run whose single dispatches take >= MIN_US: per name the count and time per step of those
dispatches (finds the few large ATen passes hidden in per-name averages).  usage: heavy_kernels.py <kernel_trace.csv> <steps> [min_us]"""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us >= min_us:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += us
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{us / steps / 1e3:7.3f} ms/step  {n / steps:6.1f} x {us / n:7.1f} us  {name[:150]}")
