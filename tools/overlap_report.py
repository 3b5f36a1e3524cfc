"""How much do the two streams of the train step overlap?  Reads a rocprofv3 --kernel-trace CSV (dev tool).
usage: overlap_report.py <kernel_trace.csv> [steps=20]
Splits the kernels into the discriminator side (conv_mfma / conv_first / conv_post / gelu_bwd_split / im2col / col2im / featloss ...)
and the rest by NAME (graph replay does not keep the capture's stream ids), takes the last `steps`-th of the trace as one step and prints,
per 1-ms bin, the busy fraction of either side and of both at once."""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MPD = ("conv_mfma", "conv_first", "conv_post", "gelu_bwd", "im2col", "col2im", "masked_l1", "bias_gelu", "gemv_", "spectral", "power_it",
       "split_bf16", "sum_parts", "sigma", "fold", "sq_err", "hinge")
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
# the last step: find the last occurrences of the optimiser kernel
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ends = adam[1::2] if len(adam) >= 4 else adam
lo, hi = rows[ends[-2]][1], rows[ends[-1]][1]
step = [r for r in rows if lo <= r[0] < hi]
print(f"last step: {(hi - lo) / 1e6:.2f} ms, {len(step)} kernels, queues {sorted(set(r[3] for r in step))}")


def busy(iv, a, b):
    return sum(max(0, min(e, b) - max(s, a)) for s, e in iv)


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


side = union([(s, e) for s, e, n, q in step if any(k in n for k in MPD)])
main = union([(s, e) for s, e, n, q in step if not any(k in n for k in MPD)])
both = []
i = j = 0
while i < len(side) and j < len(main):
    s, e = max(side[i][0], main[j][0]), min(side[i][1], main[j][1])
    if s < e:
        both.append((s, e))
    if side[i][1] < main[j][1]:
        i += 1
    else:
        j += 1
tot = hi - lo
print(f"discriminator-side kernels busy {busy(side, lo, hi) / 1e6:.2f} ms, other kernels busy {busy(main, lo, hi) / 1e6:.2f} ms, "
      f"both at once {busy(both, lo, hi) / 1e6:.2f} ms, neither {(tot - busy(union(side + main), lo, hi)) / 1e6:.2f} ms")
ms = 1_000_000
t = lo
while t < hi:
    b = min(hi, t + ms)
    print(f"  {(t - lo) / 1e6:5.1f} ms  side {busy(side, t, b) / (b - t):4.2f}  main {busy(main, t, b) / (b - t):4.2f}  both {busy(both, t, b) / (b - t):4.2f}")
    t = b
if len(sys.argv) > 3:      # compressed run listing: consecutive kernels of one queue
    cur = None
    for s, e, n, q in step:
        short = n.split("(")[0][-40:]
        if cur and cur[0] == q:
            cur[2] = e; cur[3] += 1; cur[5] = short
        else:
            if cur: print(f"  q{cur[0]} {(cur[1]-lo)/1e6:7.3f} -> {(cur[2]-lo)/1e6:7.3f} ms  {cur[3]:4d} kernels  {cur[4]} ... {cur[5]}")
            cur = [q, s, e, 1, short, short]
    print(f"  q{cur[0]} {(cur[1]-lo)/1e6:7.3f} -> {(cur[2]-lo)/1e6:7.3f} ms  {cur[3]:4d} kernels  {cur[4]} ... {cur[5]}")
