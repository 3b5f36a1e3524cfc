"""Print the numbers the docs quote from an evidence directory (default gpurun_out/r03).  dev tool."""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03"


def line(name):
    try:
        return json.loads(open(os.path.join(d, name)).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}


b = line("bench.json")
if "roofline" in b:
    r = b["roofline"]
    print(f"headline {b['value']:.1f} clips/s {b['ms_per_step']:.2f} ms | dominant {r['kernel']} {r['achieved']:.0f} GB/s frac {r['frac']:.3f} "
          f"avg {r['avg_launch_us']:.1f} us x{r['launches']} traffic {r.get('traffic')} alg/launch {r.get('alg_bytes_per_launch')}")
    o = r["selective_scan_op"]
    print(f"  scan op {o['achieved']:.0f} GB/s frac {o['frac']:.3f} {o['ms_per_step']:.3f} ms/step {o['alg_bytes_per_step'] / 1e9:.2f} GB")
    ks = r["kernels"]
    for k in sorted(ks, key=lambda k: -ks[k]["ms_per_step"])[:45]:
        print(f"    {k:22s} {ks[k]['launches']:5d} x {ks[k]['avg_us']:7.1f} us = {ks[k]['ms_per_step']:.3f} ms/step  {ks[k]['GB/s']:7.0f} GB/s")
    for k, v in b.get("operating_points", {}).items():
        rr = v.get("roofline", {})
        print(f"  point {k}: {v['value']:.1f} clips/s {v['ms_per_step']:.2f} ms dominant {rr.get('kernel')} {rr.get('frac', 0):.3f} op {rr.get('selective_scan_op')}")
    print("  cpu", b.get("cpu_baseline"))
for f in ("bench_b8.json", "bench_gonly_b35.json", "bench_gonly_b4.json", "bench_amp_step.json"):
    x = line(f)
    if "roofline" in x:
        r = x["roofline"]
        print(f"{f}: {x['value']:.1f} clips/s {x['ms_per_step']:.2f} ms dominant {r['kernel']} {r['frac']:.3f} scan op {r['selective_scan_op']['frac']:.3f}")
    else:
        print(f, x)
