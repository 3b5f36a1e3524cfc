"""Hunt for the cross-stream hazard behind GPUTEST_r05's red test (tests/test_determinism.py): evaluate the SAME full-size train-step
backward N times from one snapshot and report which gradients differ from the first evaluation.

    python tools/det_hunt.py --iters 30 [--det 0|1] [--watch] [--batch 1]
    env: VMASR_TWO_STREAM=0/1, VMASR_GEN_STREAMS, PYTORCH_NO_CUDA_MEMORY_CACHING=1, AMD_SERIALIZE_KERNEL=3 ... (A/B knobs)

--watch: every tensor saved for backward is cloned when it is saved and compared (on the device, no host sync) when it is unpacked:
a saved activation that changes between forward and backward is a buffer-lifetime bug and is reported by save index / shape / dtype.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")

import torch  # noqa: E402


class Watch:
    """saved_tensors_hooks: clone at save, compare at unpack (device-side mismatch counts, read at the end)."""

    def __init__(self):
        self.results = []
        self.n = 0

    def pack(self, t):
        if not (torch.is_tensor(t) and t.is_cuda and t.numel() > 0):
            return (None, t, None)
        idx = self.n
        self.n += 1
        return (idx, t, t.detach().clone())

    def unpack(self, item):
        idx, t, c = item
        if idx is None:
            return t
        a, b = t.detach(), c
        if a.is_floating_point():
            bad = ~((a == b) | (a.isnan() & b.isnan()))
        else:
            bad = a != b
        self.results.append((idx, tuple(t.shape), str(t.dtype), t.data_ptr(), bad.sum()))
        return t

    def report(self, tag):
        torch.cuda.synchronize()
        hits = [(i, s, d, hex(p), int(n)) for i, s, d, p, n in self.results if int(n) != 0]
        print(f"[watch {tag}] saved={self.n} unpacked={len(self.results)} modified={len(hits)}", flush=True)
        for h in hits[:20]:
            print("   MODIFIED", h, flush=True)
        self.results, self.n = [], 0
        return hits


def grads(tr, batch, snap, watch=None):
    tr._restore_training_state(snap)
    for m in tr.models.values():
        for p in m.parameters():
            p.grad = None
    torch.manual_seed(77)
    torch.cuda.manual_seed_all(77)
    if watch is not None:
        with torch.autograd.graph.saved_tensors_hooks(watch.pack, watch.unpack):
            tr._forward_backward(*batch)
    else:
        tr._forward_backward(*batch)
    torch.cuda.synchronize()
    out = {}
    for key, m in tr.models.items():
        for n, p in m.named_parameters():
            if p.grad is not None:
                out[f"{key}.{n}"] = p.grad.detach().clone()
    return out


TRACE = []


def trace_small_linear():
    """record (x2, gy2, dx, dw, db) of every _SmallLinearFn.backward call of a pass (clones, in call order)"""
    from vm_asr_amd import linear
    orig = linear._SmallLinearFn.backward

    def traced(ctx, gy):
        # the body of linear._SmallLinearFn.backward with the workspace kept: its partial rows are what colsum adds up
        import ctypes
        from vm_asr_amd import _lib
        _p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
        x2, w32 = ctx.saved_tensors
        shape, wdt, bdt, need_dx = ctx.meta
        out_f, in_f = w32.shape
        rows = x2.shape[0]
        gy2 = gy.reshape(rows, out_f)
        if gy2.dtype != x2.dtype and not (x2.dtype == torch.float32 and gy2.dtype in (torch.float16, torch.bfloat16)):
            gy2 = gy2.to(x2.dtype)
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        lib = _lib.lib()
        with torch.cuda.device(x2.device):
            dx = torch.empty_like(x2) if need_dx else None
            dw = torch.empty((out_f, in_f), dtype=torch.float32, device=x2.device)
            db = torch.empty(out_f, dtype=torch.float32, device=x2.device) if bdt is not None else None
            ws = torch.empty(lib.vmasr_small_linear_bwd_workspace(rows, in_f, out_f) // 4, dtype=torch.float32, device=x2.device)
            _lib.check(lib.vmasr_small_linear_bwd(_p(x2), _p(w32), _p(gy2), _p(dx), _p(dw), _p(db), _p(ws), rows, in_f,
                                                  out_f, _lib.torch_dtype_code(x2.dtype), _lib.torch_dtype_code(gy2.dtype),
                                                  _lib.current_stream(x2.device)), "small_linear_bwd")
        out = (dx.view(shape) if need_dx else None, dw.to(wdt), db.to(bdt) if bdt is not None else None, None)
        TRACE.append(dict(shape=(tuple(x2.shape), tuple(w32.shape)), x2=x2.detach().clone(), gy=gy.detach().clone(),
                          dx=None if out[0] is None else out[0].detach().clone(), dw=out[1].detach().clone(),
                          db=None if out[2] is None else out[2].detach().clone(), xptr=x2.data_ptr(), gptr=gy.data_ptr(),
                          ws=ws.clone().view(-1, in_f * out_f + out_f), wsptr=ws.data_ptr(), xdt=str(x2.dtype), gdt=str(gy2.dtype)))
        return out
    linear._SmallLinearFn.backward = staticmethod(traced)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--det", type=int, default=1)
    ap.add_argument("--watch", action="store_true")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--workload", default="vm_asr_48k_MPD")
    a = ap.parse_args()
    import bench
    from vm_asr_amd import _lib
    lib = _lib.lib()
    lib.vmasr_set_deterministic(a.det)
    cfg = bench.make_config(a.workload, a.batch)
    dev = torch.device("cuda:0")
    tr = bench.build_trainer(cfg, dev, amp=True, capturable=False)
    for m in tr.models.values():
        m.train()
    batch = bench.synth_batch(cfg, dev, 0)
    tr._forward_backward(*batch)
    snap = tr._snapshot_training_state()
    print(f"two_streams={tr._two_streams()} det={lib.vmasr_get_deterministic()} env TWO={os.environ.get('VMASR_TWO_STREAM')} "
          f"NOCACHE={os.environ.get('PYTORCH_NO_CUDA_MEMORY_CACHING')} SERIAL={os.environ.get('AMD_SERIALIZE_KERNEL')}", flush=True)
    watch = Watch() if a.watch else None
    if a.trace:
        trace_small_linear()
    ref = grads(tr, batch, snap, watch)
    ref_trace = list(TRACE)
    TRACE.clear()
    if watch:
        watch.report("ref")
    nbad = 0
    seen = {}
    for it in range(a.iters):
        g = grads(tr, batch, snap, watch)
        diff = [k for k in ref if not torch.equal(ref[k], g[k])]
        if watch:
            watch.report(f"it{it}")
        if a.trace:
            if diff:
                for i, (r, t) in enumerate(zip(ref_trace, TRACE)):
                    eq = {k: (r[k] is None or torch.equal(r[k], t[k])) for k in ("x2", "gy", "dx", "dw", "db")}
                    if not all(eq.values()):
                        print(f"   call {i} {t['shape']} equal={eq} xptr {hex(r['xptr'])}/{hex(t['xptr'])} gptr {hex(r['gptr'])}/{hex(t['gptr'])}", flush=True)
                        dws = (r["ws"] != t["ws"])
                        rows_bad = dws.any(1).nonzero().flatten()
                        print(f"      ws {tuple(t['ws'].shape)} x {t['xdt']} gy {t['gdt']} wsptr {hex(r['wsptr'])}/{hex(t['wsptr'])}: partial rows that differ {rows_bad.tolist()[:40]} (n={rows_bad.numel()}), "
                              f"columns {dws.any(0).nonzero().flatten().tolist()}", flush=True)
                        for rb in rows_bad[:3].tolist():
                            print(f"         row {rb}: ref {r['ws'][rb].tolist()}\n                  got {t['ws'][rb].tolist()}", flush=True)
                        for k in ("x2", "gy"):
                            if not eq[k]:
                                d = (r[k].float() - t[k].float()).abs().flatten()
                                nz = d.nonzero().flatten()
                                print(f"      {k}: {nz.numel()} of {d.numel()} elements differ, first idx {nz[:8].tolist()}, last {nz[-4:].tolist()}, max {float(d.max()):.3e}", flush=True)
            TRACE.clear()
        if diff:
            nbad += 1
            for k in diff:
                seen[k] = seen.get(k, 0) + 1
            worst = [(k, float((ref[k] - g[k]).abs().max()), float(ref[k].abs().max())) for k in diff[:6]]
            print(f"it {it}: {len(diff)} tensors differ: {worst}", flush=True)
    print(f"RESULT iters={a.iters} bad_iters={nbad} det_timeouts={lib.vmasr_det_timeouts()} tensors={sorted(seen.items(), key=lambda kv: -kv[1])[:12]}", flush=True)


if __name__ == "__main__":
    main()
