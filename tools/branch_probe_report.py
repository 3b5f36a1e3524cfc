"""Queue of each chain's kernels in the LAST replay of tools/branch_probe.py (rocprofv3 --kernel-trace CSV)."""
import csv
import sys
from collections import Counter

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1])))
names = {"M": ("cumsum", "scan", "Scan"), "A": ("CUDAFunctorOnSelf_add", "add"), "B": ("MulFunctor", "mul"), "C": ("sub",)}


def chain(n):
    for k, pats in names.items():
        if any(p in n for p in pats):
            return k
    return "?"


# last replay = last third of the chain kernels
ks = [(s, e, chain(n), q) for s, e, n, q in rows if chain(n) != "?"]
last = ks[len(ks) * 2 // 3:]
t0 = last[0][0]
cnt = Counter((c, q) for _, _, c, q in last)
print("kernels per (chain, queue):", dict(sorted(cnt.items())))
# order of segments in time
seg = []
for s, e, c, q in last:
    if seg and seg[-1][0] == (c, q):
        seg[-1][2] = e
        seg[-1][3] += 1
    else:
        seg.append([(c, q), s, e, 1])
print("timeline:", "  ".join(f"{c}@q{q}x{n}[{(s - t0) / 1e3:.0f}-{(e - t0) / 1e3:.0f}us]" for (c, q), s, e, n in seg[:40]))
