"""How far apart are the gradients of ONE train step under the step's stream layouts?  (sets the tolerances of tests/test_determinism.py)

reference R: one-stream eager step, deterministic mode (bit-reproducible).  Compared with it, each evaluated N times from one snapshot:
  one-stream eager, default (atomic) mode        -> the floor: order of the fp32 atomic additions
  two-stream eager, deterministic mode (forced)  -> bit-equal among themselves?  distance to R = the cut at the waveform (_backward_two)
  two-stream eager, default mode
  captured step, each variant of enable_graphs() (generator on one stream / phase lane), default mode
metric per tensor: max|g - R| / max|R|; printed: the worst tensor of each run, and the worst over runs."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch  # noqa: E402


def named_grads(tr):
    out = {}
    for key, m in tr.models.items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                out[f"{key}.{n}"] = p.grad.detach().clone()
    return out


def eager(tr, batch, snap):
    tr._restore_training_state(snap)
    for m in tr.models.values():
        for p in m.parameters():
            p.grad = None
    torch.manual_seed(77)
    torch.cuda.manual_seed_all(77)
    tr._forward_backward(*batch)
    torch.cuda.synchronize()
    return named_grads(tr)


TOP = {}


def dist(a, ref):
    worst, name = 0.0, None
    for k, r in ref.items():
        d = float((a[k].double() - r.double()).abs().max()) / max(float(r.double().abs().max()), 1e-30)
        TOP[k] = max(TOP.get(k, 0.0), d)
        if d > worst:
            worst, name = d, k
    return worst, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-amp", action="store_true")
    ap.add_argument("--no-droppath", action="store_true")
    a = ap.parse_args()
    import bench
    from vm_asr_amd import _lib
    from vm_asr_amd.trainer import Trainer
    lib = _lib.lib()
    cfg = bench.make_config("vm_asr_48k_MPD", a.batch)
    if a.no_droppath:
        cfg.defrost()
        cfg.MODEL.VSSM.DROP_PATH_RATE = 0.0
        cfg.freeze()
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(cfg, dev, amp=not a.no_amp, capturable=True)
    for m in tr.models.values():
        m.train()
    batch = bench.synth_batch(cfg, dev, 0)
    tr._forward_backward(*batch)
    snap = tr._snapshot_training_state()

    def runs(tag, n, fn, ref):
        ds = []
        first = None
        same = True
        for _ in range(n):
            g = fn()
            if first is None:
                first = g
            else:
                same = same and all(torch.equal(first[k], g[k]) for k in first)
            ds.append(dist(g, ref) if ref is not None else (0.0, None))
        w = max(ds, key=lambda t: t[0])
        if ref is not None:
            top = sorted(TOP.items(), key=lambda kv: -kv[1])
            print("      top:", [(k.split("generator.")[-1], f"{v:.2e}", f"|R|max {float(ref[k].abs().max()):.2e}") for k, v in top[:4]],
                  " tensors above 1e-3:", sum(v > 1e-3 for v in TOP.values()), "above 1e-5:", sum(v > 1e-5 for v in TOP.values()), "of", len(TOP), flush=True)
            TOP.clear()
        print(f"{tag:52s} n={n} bit-equal among themselves={same}  worst distance to R {w[0]:.3e} ({w[1]})  median {sorted(d for d, _ in ds)[len(ds) // 2]:.3e}", flush=True)
        return first

    os.environ["VMASR_TWO_STREAM"] = "0"
    lib.vmasr_set_deterministic(1)
    R = runs("one stream, deterministic (R)", 3, lambda: eager(tr, batch, snap), None)
    runs("one stream, deterministic again", 3, lambda: eager(tr, batch, snap), R)
    lib.vmasr_set_deterministic(0)
    runs("one stream, atomics", a.n, lambda: eager(tr, batch, snap), R)
    os.environ["VMASR_TWO_STREAM"] = "force"
    lib.vmasr_set_deterministic(1)
    runs("two streams (forced), deterministic", a.n, lambda: eager(tr, batch, snap), R)
    print("   det timeouts:", lib.vmasr_det_timeouts(), flush=True)
    lib.vmasr_set_deterministic(0)
    os.environ["VMASR_TWO_STREAM"] = "1"
    runs("two streams, atomics", a.n, lambda: eager(tr, batch, snap), R)

    # captured variants
    for pin in ("one", "lane:0.625", "lane:0.75"):
        os.environ["VMASR_STEP_VARIANT"] = pin
        tr2 = bench.build_trainer(cfg, dev, amp=not a.no_amp, capturable=True)
        for m in tr2.models.values():
            m.train()
        tr2._restore_training_state  # noqa: B018
        tr2.train_step(*batch)
        ok = tr2.enable_graphs(batch, warmup=2)
        # same weights as the reference trainer
        with torch.no_grad():
            for (ka, ma), (kb, mb) in zip(tr.models.items(), tr2.models.items()):
                tr._restore_training_state(snap)
                mb.load_state_dict(ma.state_dict())
        tr2._refresh_shadows()
        snap2 = tr2._snapshot_training_state()
        names = {key: [f"{key}.{n}" for n, p in m.named_parameters() if any(p is q for q in tr2._flat_params[key])] for key, m in tr2.models.items()}

        def replay():
            tr2._restore_training_state(snap2)
            torch.manual_seed(77)
            torch.cuda.manual_seed_all(77)
            tr2._graphed(*batch)
            torch.cuda.synchronize()
            out = {}
            for key in tr2._flat_params:
                for nme, v in zip(names[key], tr2._flat_views[key]):
                    out[nme] = v.detach().clone()
            return out
        from vm_asr_amd.trainer import unwrap
        runs(f"captured, variant {pin} (graphs={ok}, lane={unwrap(tr2.models['generator']).phase_lane})", a.n, replay, R)
        del tr2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
