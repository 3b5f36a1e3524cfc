"""Join FETCH_SIZE / WRITE_SIZE PMC passes of tools/pmc_scan.py into per-launch HBM traffic (dev tool).
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, sys, collections, re
def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if "vmasr" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows
f, w = load(sys.argv[1]), load(sys.argv[2])
assert len(f) == len(w), (len(f), len(w))
def short(n):
    m = re.search(r"(sscan_\w+?_kernel|sscan_carry_kernel|sscan_bwd_reduce_kernel)<([^>]*)>", n)
    return f"{m.group(1)}<{m.group(2)}>" if m else n[:60]
SHAPES = [(8, 262144), (64, 65536), (128, 16384), (256, 4096), (512, 1024), (1024, 256)]
B = 4
# launches come in shape order, 3 repetitions of (fwd kernels, bwd kernels) per shape
agg = collections.OrderedDict()
for (df, nf, gf, vf), (dw, nw, gw, vw) in zip(f, w):
    key = (short(nf), gf)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += vf * 2 * 1024; a[2] += vw * 1024
print(f"{'kernel':70s} {'grid':>9s} {'n':>3s} {'fetch MB':>9s} {'write MB':>9s} {'total MB':>9s}")
for (k, g), (n, fb, wb) in agg.items():
    print(f"{k:70s} {g:9d} {n:3d} {fb/n/1e6:9.2f} {wb/n/1e6:9.2f} {(fb+wb)/n/1e6:9.2f}")
print("\nalgorithmic MB per call (B=4, fp32): ")
for KD, L in SHAPES:
    print(f"  KD={KD:5d} L={L:7d} fwd {(3*KD+8)*L*4*B/1e6:8.2f}  bwd {(5*KD+16)*L*4*B/1e6:8.2f}")
