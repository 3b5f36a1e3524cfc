"""Instruction mix per basic block of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): dev tool.
   python tools/isa_mix.py file.s <substring of the mangled kernel name> [min instructions per block]"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
start = [i for i, l in enumerate(s) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0]][0]
end = [i for i in range(start, len(s)) if "s_endpgm" in s[i]][0]
cur, counts, order = "entry", {}, ["entry"]
for l in s[start + 1:end]:
    l = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = l.split(":")[0]
        order.append(cur)
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
           else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    if op.startswith("s_waitcnt"): cls = "wait"
    if op.startswith("s_barrier"): cls = "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): cls = "branch"
    if op.startswith("v_pk_"): cls = "valu_pk"
    if "dpp" in l: cls = "valu_dpp"
    counts.setdefault(cur, {})
    counts[cur][cls] = counts[cur].get(cls, 0) + 1
for k in order:
    if k in counts and sum(counts[k].values()) >= mn:
        print(k, dict(sorted(counts[k].items())))
