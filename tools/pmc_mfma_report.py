"""MFMA utilisation of the GEMM kernels of one eager train step from a rocprofv3 --pmc pass (dev tool).
usage: pmc_mfma_report.py <counter_collection.csv> <out.json>
Counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy, summed over SIMDs), GRBM_GUI_ACTIVE (GPU-active
cycles, summed over the 8 XCDs).  util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)."""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (did, name), c in per.items():
    k = ("hipblaslt/tensile GEMM (Cijk_*)" if name.startswith("Cijk") else
         "vmasr conv_mfma kernels (discriminator implicit GEMMs)" if ("vmasr" in name and "conv_mfma" in name) else
         "vmasr MFMA kernels (mlp_fwd / mlp_bwd)" if ("vmasr" in name and "mlp_" in name) else
         "vmasr library kernels (no MFMA)" if "vmasr" in name else "ATen / other")
    f = fam[k]
    f[0] += 1; f[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); f[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
top = sorted(((n, c) for (d, n), c in per.items() if n.startswith("Cijk")), key=lambda x: -x[1].get("GRBM_GUI_ACTIVE", 0))[:12]
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `python bench.py --steps 1 --warmup 1 --no-graphs "
               "--no-cpu-baseline --no-kernel-timing`; util = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
       "families": {k: {"dispatches": v[0], "mfma_busy_cycles": v[1], "gui_active_cycles_sum_xcd": v[2],
                        "mfma_util": (v[1] / (v[2] / 8 * 1024)) if v[2] else 0.0} for k, v in fam.items()},
       "largest_gemm_dispatches": [{"kernel": n[:90], "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else 0.0,
                                    "gui_active_cycles_per_xcd": c.get("GRBM_GUI_ACTIVE", 0) / 8} for n, c in top]}
own = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (did, name), c in per.items():
    if "vmasr" in name and ("mlp_" in name or "conv_mfma" in name):
        import re
        m = re.search(r"(mlp_(?:fwd|bwd)_kernel)<?(?:ILi)?(\d+)", name)
        key = f"{m.group(1)} d={m.group(2)}" if m else ("conv_mfma_wgrad" if "wgrad" in name else "conv_mfma_nt (fwd / dgrad)") if "conv_mfma" in name else name[:60]
        o = own[key]
        o[0] += 1; o[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); o[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
out["vmasr_mfma_kernels"] = {k: {"dispatches": v[0], "mfma_util": (v[1] / (v[2] / 8 * 1024)) if v[2] else 0.0,
                                 "gui_active_cycles_per_xcd_avg": v[2] / 8 / max(1, v[0])} for k, v in sorted(own.items())}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["vmasr_mfma_kernels"].items():
    print(f"   own MFMA kernel {k:28s} dispatches {v['dispatches']:4d}  mfma_util {v['mfma_util']:.3f}  {v['gui_active_cycles_per_xcd_avg']:9.0f} cyc/dispatch")
for k, v in out["families"].items():
    print(f"{k:36s} dispatches {v['dispatches']:6d}  mfma_util {v['mfma_util']:.3f}")
for t in out["largest_gemm_dispatches"][:6]:
    print(f"   {t['mfma_util']:.3f}  {t['gui_active_cycles_per_xcd']:10.0f} cyc  {t['kernel'][:70]}")
