"""Which operator makes the HIP fp32 forward noisier than the reference's own fp32 run?  (dev tool, GPU box)

Full-size forward (tests/golden/fullsize2.npz cases, B = 1) through the product modules, adjudicated by the float64
evaluation of the reference (y64) exactly as tests/test_fullsize.py does.  One operator FAMILY at a time is replaced
by a float64 evaluation on the GPU (plain torch, result rounded once to the operator's output dtype): the drop of
the final error says how much of it that family's fp32 arithmetic contributes.  `all` replaces every family (sanity:
what remains is the rounding of the operator outputs alone).

    python tools/accuracy_probe.py [--cases 16k,n2048] [--families ...] [--core-shapes]

VMASR_LIB=<variant .so> selects an A/B build of the library (csrc/Makefile VARIANT=...).
"""
import argparse
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import test_fullsize as tf


from f64ref import FAMILIES, Patch, core64, forward64, rebind, rms  # noqa: E402  (tests/f64ref.py)


def seeds(tag, n):
    """Distribution of the statistic of tests/test_fullsize.py over n fresh input clips: error of the HIP fp32 forward
    and of the CPU oracle's fp32 forward (the sequential recurrence of selective_scan_ref on the same modules) against
    float64, and their ratio — one clip is one draw of a heavy-tailed statistic (a handful of amplified positions)."""
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    (dims, n_fft, win, hop), wave0, target, hf, y32, y64, lsd_ref = tf._case(tag)
    m = rebind(tf._model(hop, dims, n_fft, win).to("cuda:0"))
    m_cpu = use_oracle(tf._model(hop, dims, n_fft, win))
    chk = np.abs(forward64(m, wave0, hf) - y64).max() / np.abs(y64).max()
    print(f"[{tag}] GPU float64 model vs the golden float64 reference: {chk:.2e} of peak", flush=True)
    rows = []
    for s in range(n):
        wave = wave0 if s == 0 else 0.1 * torch.randn(wave0.shape, generator=torch.Generator().manual_seed(9000 + s))
        y64s = forward64(m, wave, hf)
        with torch.no_grad():
            y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
        with oracle_stft_patch(), torch.no_grad():
            yc = m_cpu(wave, hf).float().numpy()
        peak = np.abs(y64s).max()
        e, ec = np.abs(y - y64s), np.abs(yc - y64s)
        rows.append(dict(seed=s, hip_rms=rms(e) / peak, cpu_rms=rms(ec) / peak, hip_max=e.max() / peak, cpu_max=ec.max() / peak))
        print(f"[{tag}] clip {s}: hip rms {rms(e) / peak:.2e} max {e.max() / peak:.2e} | cpu-oracle fp32 rms {rms(ec) / peak:.2e} max "
              f"{ec.max() / peak:.2e} | ratio rms {rms(e) / rms(ec):.2f} max {e.max() / ec.max():.2f}", flush=True)
    gm = lambda k: float(np.exp(np.mean([np.log(r["hip_" + k] / r["cpu_" + k]) for r in rows])))      # noqa: E731
    print(f"[{tag}] geometric mean of the ratio over {n} clips: rms {gm('rms'):.2f}  max {gm('max'):.2f}", flush=True)
    return dict(rows=rows, gm_rms=gm("rms"), gm_max=gm("max"), f64_check=chk)


def run_case(tag, fam_sets):
    (dims, n_fft, win, hop), wave, target, hf, y32, y64, lsd_ref = tf._case(tag)
    m = tf._model(hop, dims, n_fft, win).to("cuda:0")
    rebind(m)
    e_ref = np.abs(y32.astype(np.float64) - y64)
    peak = np.abs(y64).max()
    out = {"reference_fp32": {"rms": rms(e_ref) / peak, "max": e_ref.max() / peak}}
    print(f"[{tag}] reference fp32 vs f64: rms {rms(e_ref) / peak:.3e}  max {e_ref.max() / peak:.3e}", flush=True)
    for name, fams in fam_sets:
        with Patch(fams), torch.no_grad():
            y = m(wave.cuda(), hf.cuda()).float().cpu().numpy()
        e = np.abs(y.astype(np.float64) - y64)
        out[name] = {"rms": rms(e) / peak, "max": e.max() / peak, "rms_ratio": rms(e) / rms(e_ref), "max_ratio": e.max() / e_ref.max()}
        print(f"[{tag}] f64 for {name:12s}: rms {rms(e) / peak:.3e} ({rms(e) / rms(e_ref):.2f}x ref)  max {e.max() / peak:.3e} "
              f"({e.max() / e_ref.max():.2f}x ref)", flush=True)
    return out


def core_shapes():
    """The fused core alone at the benchmark shapes (B = 1) and the small test shapes: distance from float64 (GPU core64)
    of the HIP operator and of the CPU oracle's fp32 chain (the sequential recurrence of selective_scan_ref)."""
    import test_ss2d_fused as t
    from vm_asr_amd.ss2d_core import ss2d_core
    res = {}
    for shape in [(1, 32, 128, 128), (1, 16, 256, 256), (1, 2, 512, 512), (2, 2, 64, 64), (1, 16, 128, 64)]:
        B, D, H, W = shape
        g = torch.Generator().manual_seed(D * 7 + H)
        x = torch.randn(B, D, H, W, generator=g)
        gy = torch.randn(B, D, H * W, generator=g)
        params = t._params(D, D + 1)

        def f64fn(x_, *ps):
            return core64(x_, ps[0], ps[1], ps[2], ps[3], ps[4])
        got = t._run(ss2d_core, x.cuda(), [p.cuda() for p in params], gy)
        r64 = t._run(f64fn, x.double().cuda(), [p.cuda() for p in t._params(D, D + 1, torch.float64)], gy.double())
        ref = t._run(t._oracle_core, x, params, gy)
        row = {}
        for n, a, b, c in zip(t.NAMES, got, ref, r64):
            a, b, c = a.double().cpu(), b.double(), c.double().cpu()
            scale = max(c.abs().max().item(), 1e-300)
            row[n] = {"hip_max": (a - c).abs().max().item() / scale, "cpu_max": (b - c).abs().max().item() / scale,
                      "hip_rms": rms((a - c).numpy()) / scale, "cpu_rms": rms((b - c).numpy()) / scale}
        res[str(shape)] = row
        print(shape, " ".join(f"{n}: hip {v['hip_rms']:.2e}/{v['hip_max']:.2e} cpu {v['cpu_rms']:.2e}/{v['cpu_max']:.2e} |" for n, v in row.items()),
              flush=True)
    return res


def ops():
    """Every HIP-backed operator family alone at model shapes: distance from float64 of the HIP kernel and of the plain
    torch CPU fp32 evaluation of the same reference lines (what the reference's fp32 run executes)."""
    import torch.nn.functional as F
    import f64ref as r
    from vm_asr_amd import ss2d_glue as glue
    from vm_asr_amd.dwconv import dwconv3x3_silu
    from vm_asr_amd.layernorm import layer_norm
    from vm_asr_amd.stft import spectro2wav, wav2spectro
    g = torch.Generator().manual_seed(5)
    res = {}

    def report(name, hip, cpu, f64):
        hip, cpu, f64 = hip.double().cpu(), cpu.double().cpu(), f64.double().cpu()
        sc = f64.abs().max().item()
        row = dict(hip_rms=rms((hip - f64).numpy()) / sc, cpu_rms=rms((cpu - f64).numpy()) / sc,
                   hip_max=(hip - f64).abs().max().item() / sc, cpu_max=(cpu - f64).abs().max().item() / sc)
        res[name] = row
        print(f"{name:34s} hip rms {row['hip_rms']:.2e} max {row['hip_max']:.2e} | torch-cpu fp32 rms {row['cpu_rms']:.2e} max {row['cpu_max']:.2e} "
              f"| ratio rms {row['hip_rms'] / max(row['cpu_rms'], 1e-300):.2f}", flush=True)
    with torch.no_grad():
        for D, H in ((32, 128), (2, 512), (128, 32)):
            x = torch.randn(1, D, H, H, generator=g); w = torch.randn(D, 1, 3, 3, generator=g) / 3; b = 0.1 * torch.randn(D, generator=g)
            report(f"dwconv_silu D={D} {H}x{H}", dwconv3x3_silu(x.cuda(), w.cuda(), b.cuda()), F.silu(F.conv2d(x, w, b, padding=1, groups=D)),
                   F.silu(F.conv2d(x.double(), w.double(), b.double(), padding=1, groups=D)))
            xz = torch.randn(1, H, H, 2 * D, generator=g)
            xT, sz = glue.ss2d_pre(xz.cuda())
            report(f"ss2d_pre silu(z) D={D}", sz, F.silu(xz[..., D:]), F.silu(xz[..., D:].double()))
            y = torch.randn(1, D, H * H, generator=g) * (0.01 if D == 2 else 1.0) + (0.5 if D == 2 else 0.0); szc = torch.randn(1, H, H, D, generator=g)
            gm, bt = 1 + 0.1 * torch.randn(D, generator=g), 0.05 * torch.randn(D, generator=g)
            f64 = r.ln_gate64(y.double(), szc.double(), gm, bt, 1e-5)
            cpu = F.layer_norm(y.transpose(1, 2), (D,), gm, bt, 1e-5).view(1, H, H, D) * szc
            report(f"ln_gate D={D}", glue.ln_gate(y.cuda(), szc.cuda(), gm.cuda(), bt.cuda(), 1e-5), cpu, f64)
        for C, rows in ((16, 16384), (64, 4096), (512, 256), (4, 65536), (8, 65536)):
            x = torch.randn(rows, C, generator=g) + 0.3; gm, bt = 1 + 0.1 * torch.randn(C, generator=g), 0.05 * torch.randn(C, generator=g)
            report(f"layer_norm C={C}", layer_norm(x.cuda(), gm.cuda(), bt.cuda(), 1e-5), F.layer_norm(x, (C,), gm, bt, 1e-5),
                   F.layer_norm(x.double(), (C,), gm.double(), bt.double(), 1e-5))
        for n_fft, hop, win, T in ((1024, 80, 1024, 40880), (2048, 80, 1024, 40880)):
            wv = 0.1 * torch.randn(1, 1, T, generator=g)
            mh, ph = wav2spectro(wv.cuda(), n_fft, hop, win, "log2")
            S = torch.stft(wv[0], n_fft, hop, win, torch.hann_window(win), center=True, pad_mode="reflect", normalized=True, onesided=True, return_complex=True)
            m64, p64 = r.wav2spectro64(wv.double(), n_fft, hop, win, "log2")
            report(f"stft mag n_fft={n_fft}", mh[0], torch.log2(S.abs() + 1e-8), m64[0])
            ex = lambda p: torch.stack([torch.cos(p), torch.sin(p)])         # noqa: E731  (phase compared on the circle)
            report(f"stft phase n_fft={n_fft}", ex(ph[0].double()), ex(torch.angle(S).double()), ex(p64[0]))
            mag, phs = m64.float(), p64.float()
            wc = torch.istft(torch.polar(torch.exp2(mag[0]), phs[0]), n_fft, hop, win, torch.hann_window(win), center=True, normalized=True)
            report(f"istft n_fft={n_fft}", spectro2wav(mag.cuda(), phs.cuda(), n_fft, hop, win, "log2")[0], wc, r.spectro2wav64(mag.double(), phs.double(), n_fft, hop, win, "log2")[0])
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="16k,n2048")
    ap.add_argument("--families", default="none," + ",".join(FAMILIES) + ",all")
    ap.add_argument("--core-shapes", action="store_true")
    ap.add_argument("--ops", action="store_true")
    ap.add_argument("--seeds", type=int, default=0, help="clips per case for the ratio distribution (0: skip)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    sets = []
    for n in [f for f in a.families.split(",") if f]:
        if n.startswith("all-"):      # isolation: everything in float64 EXCEPT the named families (their own contribution)
            sets.append((n, [f for f in FAMILIES if f not in n[4:].split("+")]))
        else:
            sets.append((n, [] if n == "none" else FAMILIES if n == "all" else n.split("+")))
    result = {"lib": os.environ.get("VMASR_LIB", "default")}
    print("library:", result["lib"], flush=True)
    if a.core_shapes:
        result["core"] = core_shapes()
    if a.ops:
        result["ops"] = ops()
    for tag in [c for c in a.cases.split(",") if c]:
        result[tag] = run_case(tag, sets) if sets else {}
        if a.seeds:
            result[tag]["seeds"] = seeds(tag, a.seeds)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(result, open(a.out, "w"), indent=1)
