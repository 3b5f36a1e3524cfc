"""Per-kernel-id HBM traffic of one eager train step from two rocprofv3 PMC passes (dev tool).
usage: pmc_bench_report.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import csv, json, re, sys, collections

def kid(name):
    m = re.search(r"sscan_(fwd|bwd)_kernel<[^>]*?(\d)>", name)
    if m:
        d, mode = m.group(1), int(m.group(2))
        return {("fwd", 0): "sscan_fwd", ("fwd", 3): "sscan_fwd", ("fwd", 2): "sscan_fwd_agg", ("fwd", 1): "sscan_fwd_apply",
                ("bwd", 0): "sscan_bwd", ("bwd", 2): "sscan_bwd_agg", ("bwd", 1): "sscan_bwd_apply"}[(d, mode)]
    m = re.search(r"ss2d_(fwd|bwd)_kernel<[^>]*?(\d)>", name) or re.search(r"ss2d_(fwd|bwd)_kernelI\w*?Li(\d)EEE", name)
    if m:   # (demangled or mangled: rocprofv3 leaves some instantiations mangled)
        return f"ss2d_{m.group(1)}_{'apply' if m.group(2) == '1' else 'agg'}"
    for k, v in (("ss2d_carry_kernel", "ss2d_carry"), ("ss2d_bwd_reduce_kernel", "ss2d_carry"), ("transpose_hw_kernel", "ss2d_transpose"),
                 ("merge_pairs_kernel", "ss2d_merge"), ("split_bf16_kernel", "split_bf16"), ("gelu_bwd_split_kernel", "gelu_bwd_split"),
                 ("bias_gelu_fwd_kernel", "bias_gelu_fwd"), ("im2col_split_kernel", "im2col_split"), ("im2col_kernel", "im2col_kx1"),
                 ("col2im_kernel", "col2im_kx1"), ("deep_bwd_kernel", "ss2d_deep_bwd"), ("deep_fwd_kernel", "ss2d_deep_fwd"),
                 ("deep_xproj_kernel", "ss2d_deep_xproj"), ("deep_xg_kernel", "ss2d_deep_xbwd"), ("deep_dx_kernel", "ss2d_deep_xbwd"),
                 ("mlp_fwd_kernel", "mlp_fwd"), ("mlp_bwd_kernel", "mlp_bwd"), ("inproj_kernel", "inproj"), ("outproj_kernel", "outproj"),
                 ("ln_gate_pair", "ln_gate_pair"), ("ln_gate", "ln_gate"), ("conv_mfma_nt_kernel", "conv_mfma_nt"),
                 ("conv_mfma_wgrad_kernel", "conv_mfma_wgrad")):
        if k in name:
            return v
    for k in ("sscan_carry_kernel<false>", "sscan_carry_kernel<true>", "sscan_bwd_reduce_kernel", "cross_scan_kernel",
              "cross_merge_kernel", "dwconv_silu", "stft_like_kernel", "istft_frames_kernel", "istft_ola_kernel",
              "ln_fwd_kernel", "ln_bwd_kernel", "ln_bwd_reduce_kernel", "small_linear_fwd_kernel", "small_linear_bwd_kernel"):
        if k in name:
            return k
    return None

def load(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "vmasr" not in r["Kernel_Name"]:
            continue
        k = kid(r["Kernel_Name"])
        if k:
            a = out.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return out

f, w = load(sys.argv[1]), load(sys.argv[2])
res = {}
for k in f:
    n = f[k][0]
    fb, wb = f[k][1] * 2 * 1024 / n, (w[k][1] * 1024 / w[k][0]) if k in w else 0.0
    res[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    print(f"{k:28s} n={n:5d} fetch {fb/1e6:9.2f} MB  write {wb/1e6:9.2f} MB  total {(fb+wb)/1e6:9.2f} MB per launch")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest   # the digest bench.py compares against before quoting `traffic`
json.dump({"csrc_digest": csrc_digest(), "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                   "`python bench.py --steps 1 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-timing`; "
                   "KiB units, FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B)", "kernels": res},
          open(sys.argv[3], "w"), indent=1)
