"""Summarise a rocprofv3 *_kernel_stats.csv (dev tool): python tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total GPU time {tot/1e6:.2f} ms, {calls} launches, {len(rows)} distinct kernels")
for r in rows[:n]:
    print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms {100*int(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):6d} calls "
          f"{float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
