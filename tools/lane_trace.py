"""Per hardware queue: kernels, busy time and pairwise concurrency over the LAST replayed step of a rocprofv3 --kernel-trace CSV (dev tool)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
# two adamw launches per step (G, D): step boundaries = every second one
ends = adam[1::2]
lo, hi = rows[ends[-2]][1], rows[ends[-1]][1]
step = [r for r in rows if lo <= r[0] < hi]
print(f"last step {(hi - lo) / 1e6:.2f} ms, {len(step)} kernels")
byq = defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s - lo, e - lo, n))


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def inter(a, b):
    i = j = 0
    t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            t += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


U = {}
for q, ks in sorted(byq.items()):
    U[q] = union([(s, e) for s, e, _ in ks])
    busy = sum(e - s for s, e in U[q])
    mpd = sum(1 for _, _, n in ks if any(t in n for t in ("conv_mfma", "conv_first", "conv_post", "gelu_bwd", "masked_l1", "im2col", "col2im")))
    print(f"queue {q}: {len(ks):5d} kernels ({mpd} discriminator-side), busy {busy / 1e6:6.2f} ms, first {ks[0][0] / 1e6:6.2f} last {ks[-1][1] / 1e6:6.2f} ms, "
          f"sum of kernel durations {sum(e - s for s, e, _ in ks) / 1e6:6.2f} ms")
# idle structure of each queue: gaps between consecutive kernels (launch gaps are a few us each; a wait for another stream is one long gap)
for q, ks in sorted(byq.items()):
    ks = sorted(ks)
    gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
    if not gaps:
        continue
    g = sorted(gaps)
    small = [x for x in gaps if x <= 10e3]
    big = [x for x in gaps if x > 50e3]
    top = sorted(((ks[i + 1][0] - ks[i][1], ks[i][2][:40], ks[i + 1][2][:40], ks[i][1] / 1e6) for i in range(len(ks) - 1)), reverse=True)[:5]
    print(f"queue {q}: gaps total {sum(gaps) / 1e6:6.2f} ms; <= 10 us: {len(small)} gaps, {sum(small) / 1e6:5.2f} ms (median {g[len(g) // 2] / 1e3:.1f} us); "
          f"> 50 us: {len(big)} gaps, {sum(big) / 1e6:5.2f} ms")
    for d, a_, b_, at in top:
        print(f"      {d / 1e3:8.1f} us at {at:6.2f} ms  after {a_}  before {b_}")
qs = sorted(U)
for i, a in enumerate(qs):
    for b in qs[i + 1:]:
        print(f"  queues {a} & {b} busy at the same time: {inter(U[a], U[b]) / 1e6:6.2f} ms")
# per 2-ms bin: busy fraction per queue
print("bin(ms) " + " ".join(f"q{q:>3s}" for q in qs))
nb = int((hi - lo) / 2e6) + 1
for k in range(nb):
    a, b = k * 2e6, (k + 1) * 2e6
    fr = []
    for q in qs:
        t = sum(max(0, min(e, b) - max(s, a)) for s, e in U[q])
        fr.append(t / 2e6)
    print(f"{k * 2:6d}  " + " ".join(f"{x:4.2f}" for x in fr))
# generator-side kernel durations (sum) for comparison between runs
gen = [(e - s, n) for s, e, n, q in step if not any(t in n for t in ("conv_mfma", "conv_first", "conv_post", "gelu_bwd", "masked_l1", "im2col", "col2im", "adamw"))]
print(f"generator-side (by name): {len(gen)} kernels, sum of durations {sum(d for d, _ in gen) / 1e6:.2f} ms")
