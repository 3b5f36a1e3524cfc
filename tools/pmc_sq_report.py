"""Per-kernel SQ counter summary from a rocprofv3 --pmc counter_collection.csv (dev tool)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if "vmasr" not in r["Kernel_Name"]:
        continue
    m = re.search(r"(sscan_\w+?_kernel)<([^>]*)>", r["Kernel_Name"])
    k = (f"{m.group(1)}<{m.group(2)}>" if m else r["Kernel_Name"][:50], int(r["Grid_Size"]), int(r.get("VGPR_Count", 0) or 0))
    d = agg.setdefault(k, collections.Counter())
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    d["_n_" + r["Counter_Name"]] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("counters:", names)
for k, d in agg.items():
    n = d["_n_" + names[0]]
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k[0]:55s} grid {k[1]:8d} vgpr {k[2]:3d} n={n:2d} waves {d.get('SQ_WAVES',0)/n:9.0f} "
          f"wait_any {d.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active_any {d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
          f"valu {d.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} insts_valu/wave {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_WAVES',1)):7.0f} "
          f"salu/wave {d.get('SQ_INSTS_SALU',0)/max(1,d.get('SQ_WAVES',1)):7.0f} wavecyc/wave {4*wc/max(1,d.get('SQ_WAVES',1)):9.0f}"
          + "".join(f" {c[3:].lower()}/wave {d[c]/max(1,d.get('SQ_WAVES',1)):9.0f}" for c in names if c not in
                    ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "SQ_WAVE_CYCLES")))
