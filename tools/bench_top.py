"""Top kernels of the last bench.py run (bench_detail.json): dev tool.   python tools/bench_top.py [n] [path]"""
import json
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "bench_detail.json"))
k = d["roofline"]["kernels"]
tot = sum(v["ms_per_step"] for v in k.values())
print(f"{d['value']:.1f} {d['unit']}, {d['ms_per_step']:.2f} ms/step; library kernels {tot:.2f} ms/step in {sum(v['launches'] for v in k.values()) // d['steps']} launches")
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:n]:
    print(f"{name:24s} {v['launches'] // d['steps']:5d}/step {v['avg_us']:9.1f} us {v['ms_per_step']:8.3f} ms/step")
