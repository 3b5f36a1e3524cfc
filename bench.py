"""bench.py — headline benchmark: audio clips/sec for one training step at 48 kHz, n_fft 1024.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic VCTK-shaped clips resident in
HBM: generator forward (HIP STFT -> 28 SS2D calls -> HIP iSTFT) under bf16 autocast
(scan fp32, as the reference's forward type v5 forces), MR-STFT + LSGAN + feature losses against
the 41 M-parameter MPD, backward, AdamW for G and D — `configs/vm_asr_48k_MPD.yaml` of the
reference as written (per-GPU batch 4, DIMS 16).  N GPUs = N processes, batch sharded by clip,
DDP all-reduce over RCCL; value = N*B*K / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      dominant HIP kernel, timed live with HIP events on its launch stream
  cpu_baseline  the same train step on the host cores with the CPU oracle kernels (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # vm_asr_amd/hip_env.py: before the GPU is initialised
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # vm_asr_amd/hip_env.py: stream-K GEMMs of two streams can stall the device

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PMC_FILE = "r06_pmc_traffic.json"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def make_config(workload, batch):
    from vm_asr_amd.config import get_default_config, update_config
    c = get_default_config()
    c.MODEL.NAME = "DualStreamInteractiveMambaUNet"
    c.TRAIN.LOW_FREQ_REPLACEMENT = True
    c.DATA.TARGET_SR = 48000
    c.DATA.LPF.MULTIFILTER = True
    if workload == "vm_asr_48k_MPD":
        c.TRAIN.ADVERSARIAL.ENABLE = True
        c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
        c.DATA.BATCH_SIZE = 4
    elif workload == "vm_asr_48k_16k_MPD_VSSM32":     # configs[4]: the yaml as written sets DIMS 32, batch 8
        c.MODEL.VSSM.DIMS = 32
        c.TRAIN.ADVERSARIAL.ENABLE = True
        c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
        c.TRAIN.ADVERSARIAL.STFT_LOSS.EMPHASIZE_HIGH_FREQ = False
        c.DATA.BATCH_SIZE = 8
    elif workload == "vm_asr_48k_16k_nfft2048":       # configs/vm_asr_48k_16k_nfft2048.yaml:17-19 as written: n_fft 2048 (win stays 1024), batch 8
        c.DATA.STFT.N_FFT = 2048
        c.TRAIN.ADVERSARIAL.ENABLE = True
        c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
        c.TRAIN.ADVERSARIAL.STFT_LOSS.EMPHASIZE_HIGH_FREQ = False
        c.DATA.BATCH_SIZE = 8
    elif workload == "vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048":
        # configs[4] as BASELINE.json words it: the DIMS-32 yaml + `--opts MODEL.VSSM.SSM_D_STATE 32 DATA.STFT.N_FFT 2048`
        # (main.py:39-44, config.py:100,55) — the long-sequence, general-N stress point
        c.MODEL.VSSM.DIMS = 32
        c.MODEL.VSSM.SSM_D_STATE = 32
        c.DATA.STFT.N_FFT = 2048
        c.TRAIN.ADVERSARIAL.ENABLE = True
        c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
        c.TRAIN.ADVERSARIAL.STFT_LOSS.EMPHASIZE_HIGH_FREQ = False
        c.DATA.BATCH_SIZE = 8
    elif workload == "vm_asr_48k":
        c.TRAIN.ADVERSARIAL.ENABLE = False
        c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
        c.DATA.BATCH_SIZE = 35
    else:
        raise ValueError(workload)
    if batch:
        c.DATA.BATCH_SIZE = batch
    return update_config(c)


def build_trainer(config, device, amp=True, capturable=False, amp_scope="generator"):
    import vm_asr_amd
    from vm_asr_amd.trainer import Trainer, build_optimizer
    torch.manual_seed(config.SEED)
    models = vm_asr_amd.get_model(config)
    gan = config.TRAIN.ADVERSARIAL.ENABLE
    for m in models.values():
        m.to(device)
    opts = {"generator": build_optimizer(config, models["generator"], capturable)}
    if gan:
        opts["discriminator"] = build_optimizer(config, [models["mpd"]], capturable)
    return Trainer(models, [], opts, config, device, None, None, {}, amp=amp, gan=gan, len_epoch=0, dp_mode="flat",
                   amp_scope=amp_scope)


def synth_batch(config, device, rank):
    """One batch of the trainer's own synthetic VCTK-shaped dataset (vm_asr_amd.trainer.SyntheticVCTK: the batch contract of
    data_loader/data_loaders.py:482-513 — (wave_in, wave_target, highcut int64, name, pad)), default-collated, rank-seeded."""
    from vm_asr_amd.trainer import SyntheticVCTK
    ds = SyntheticVCTK(config, length=config.DATA.BATCH_SIZE, sr_in=16000, seed=123 + 1000 * rank)
    inp, tgt, hc = next(iter(torch.utils.data.DataLoader(ds, batch_size=config.DATA.BATCH_SIZE, shuffle=False)))[:3]
    return inp.to(device), tgt.to(device), hc.to(device)


def cpu_baseline(config, budget_s=40.0):
    """Same train step on the host: torch-CPU modules with the oracle's C kernels in the operator
    hooks (kind 'port').  Sample: 1 clip, two full steps back to back — the first is cold (thread pools spin up, pages fault in:
    it varied 0.037-0.094 clips/s between boxes in rounds 4-5), the second is what is reported; a second step that would not fit the
    budget is skipped and the cold one reported, saying so."""
    import oracle
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    from vm_asr_amd.config import update_config
    cfg = config.clone()
    cfg.defrost()
    cfg.DATA.BATCH_SIZE = 1
    cfg.freeze()
    oracle.lib()
    tr = build_trainer(cfg, torch.device("cpu"), amp=False)
    use_oracle(tr.models["generator"])
    inp, tgt, hc = synth_batch(cfg, torch.device("cpu"), 0)
    dts = []
    with oracle_stft_patch():
        for _ in range(2):
            t0 = time.time()
            tr.train_step(inp, tgt, hc)
            dts.append(time.time() - t0)
            if dts[0] * 2 > budget_s:
                break
    dt = dts[-1]
    what = "G+MPD" if cfg.TRAIN.ADVERSARIAL.ENABLE else "G"
    return {"value": 1.0 / dt, "unit": "clips/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": (f"1 clip x 2 full train steps ({what}, fp32), the second timed: {dt:.1f} s (the cold first: {dts[0]:.1f} s)" if len(dts) == 2 else
                       f"1 clip x 1 full train step ({what}, fp32), cold: {dt:.1f} s (a second would not fit the {budget_s:.0f} s budget)")
                      + ": torch-CPU modules + oracle C kernels (OpenMP)"}


RUN_INFO = {}       # of the last run_point(): the captured step's generator stream count and, if both variants were captured, their times


def run_point(config, args, device, rank, world, steps, warmup, timing, with_metrics=False):
    """Build the trainer for `config`, warm up, capture, time `steps` steps (barrier + synchronize on both sides, max over
    ranks) and — `timing` — run the same steps once more eagerly with the library's HIP-event timer on.
    -> (seconds for `steps` steps, graphed?, per-rank dict or None, {kernel: dict(launches, ms, alg_bytes)} or None)."""
    from vm_asr_amd import _lib
    trainer = build_trainer(config, device, amp=not args.no_amp, capturable=not args.no_graphs, amp_scope=args.amp_scope)
    for m in trainer.models.values():
        m.train()
    torch.manual_seed(config.SEED + 1 + rank)  # per-rank DropPath streams
    batch = synth_batch(config, device, rank)

    graphed = False
    RUN_INFO.clear()
    if not args.no_graphs:
        graphed = trainer.enable_graphs(batch, warmup=max(2, min(3, warmup)))
        from vm_asr_amd.trainer import unwrap
        RUN_INFO.update(generator_streams=2 if getattr(unwrap(trainer.models["generator"]), "phase_lane", False) else 1,
                        graph_variants_ms=getattr(trainer, "graph_variants", None))
        if not graphed:
            # never report the host-bound eager step (2x slower) as if it were the product's number: fail loudly
            print(json.dumps({"error": "HIP graph capture / replay self-test failed; rerun with --no-graphs to measure the eager step",
                              "detail": getattr(trainer, "graph_error", None), "rank": rank}), flush=True)
            sys.exit(3)
    mets = []
    if with_metrics:        # trainer/trainer.py:179-182: SNR / LSD / LSD-HF / LSD-LF of every step's output, read on the host
        from vm_asr_amd.trainer import default_metric_ftns
        mets = default_metric_ftns(config)

    def one_step():
        out, _ = trainer.train_step(*batch)
        for m in mets:
            float(m(out.float().squeeze(1), batch[1].squeeze(1), hf=batch[2]))
    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    trainer.time_reduces = world > 1
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own time to finish its K steps
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        trainer.time_reduces = False
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        in_graph = bool(getattr(getattr(trainer, "_graphed", None), "collectives_in_graph", False))
        mine = torch.tensor([dt_own / steps * 1e3, trainer.reduce_exposed_ms() / steps], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(v[0]), 3) for v in allr],
                    # in-graph collectives cannot be bracketed by events (no timing inside a capture): null there — compare
                    # ms_per_step with the N = 1 line instead (weak scaling: same per-GPU work)
                    "allreduce_exposed_ms_per_step": [None if in_graph else round(float(v[1]), 3) for v in allr],
                    "collectives": "branches of the step's graph (RCCL captured; MPD gradient as bf16)" if in_graph else "between the graphs",
                    "note": "exposed = time the compute stream waits at the join of the two asynchronous gradient all-reduces "
                            "(MPD 82 MB bf16 / 164 MB fp32 issued behind the D loss' backward, generator 9 MB after its pack)"}

    # per-kernel device time: HIP events recorded by the library around each of its launches, on the
    # launch stream.  Events cannot be read inside a replayed graph, so this is a second pass of the
    # same K steps, executed eagerly right after the timed region (rank 0's numbers are reported).
    prof = None
    if timing:
        g_saved, trainer._graphed = trainer._graphed, None
        _lib.prof_reset()
        if not (getattr(trainer, "_two_streams", lambda: False)() and getattr(args, "timing_pass", "both") == "unshared"):
            _lib.prof_enable(True)
            for _ in range(steps):
                trainer.train_step(*batch)
            torch.cuda.synchronize()
            _lib.prof_enable(False)
        prof = _lib.prof_collect() if rank == 0 else {}
        if rank == 0:
            prof["__shapes__"] = {k: _lib.prof_collect_shapes(k) for k in prof if k in SCAN_KERNELS or k.startswith("sscan") or k.startswith("conv_mfma")}
        if getattr(trainer, "_two_streams", lambda: False)() and getattr(args, "timing_pass", "both") != "shared":
            # the step runs the discriminator's kernels BESIDE the generator's (two streams): the durations above are those of
            # kernels sharing the chip.  Third pass, one stream: every kernel alone on the chip — its own roofline figure.
            prev = os.environ.get("VMASR_TWO_STREAM")
            os.environ["VMASR_TWO_STREAM"] = "0"
            try:
                _lib.prof_reset()
                _lib.prof_enable(True)
                for _ in range(steps):
                    trainer.train_step(*batch)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
                if rank == 0:
                    alone = _lib.prof_collect()
                    alone["__shapes__"] = {k: _lib.prof_collect_shapes(k) for k in alone if k in SCAN_KERNELS or k.startswith("sscan")}
                    prof["__unshared__"] = alone
            finally:
                if prev is None:
                    del os.environ["VMASR_TWO_STREAM"]
                else:
                    os.environ["VMASR_TWO_STREAM"] = prev
    del trainer
    torch.cuda.empty_cache()
    return dt, graphed, per_rank, prof


SCAN_KERNELS = ("ss2d_fwd_agg", "ss2d_fwd_apply", "ss2d_bwd_agg", "ss2d_bwd_apply", "ss2d_carry", "ss2d_deep_fwd", "ss2d_deep_bwd")
SCAN_BYTES_IN = ("sscan_fwd", "sscan_fwd_apply", "sscan_bwd", "sscan_bwd_apply", "ss2d_fwd_apply", "ss2d_bwd_apply", "ss2d_deep_fwd",
                 "ss2d_deep_bwd")


def scan_summary(prof, steps):
    """(kernel table, dominant scan kernel name, op-level dict) from the library's event records.
    The selective-scan operator = the scan kernels of the two fused SS2D cores (ss2d.hip: aggregate / carry / apply — its
    transpose and pair-merge kernels are what is left of cross-scan / cross-merge and are listed in the table, not counted here;
    ss2d_deep.hip: the whole-row forward / backward kernels — its x_proj kernels are listed, not counted) + sscan.hip's kernels
    where a call still takes the unfused path; algorithmic bytes are counted once per op (the apply / single-pass kernels carry
    them)."""
    kern = {k: dict(v, avg_us=v["ms"] / v["launches"] * 1e3, gbs=v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0)
            for k, v in prof.items() if not k.startswith("__")}
    scan = {k: v for k, v in kern.items() if k.startswith("sscan") or k in SCAN_KERNELS}
    if not scan:
        return kern, None, None
    dom = max(scan, key=lambda k: scan[k]["ms"])
    op_bytes = sum(v["alg_bytes"] for k, v in scan.items() if k in SCAN_BYTES_IN)
    op_ms = sum(v["ms"] for v in scan.values())
    op = {"achieved": op_bytes / (op_ms * 1e-3) / 1e9, "frac": op_bytes / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "ms_per_step": op_ms / steps, "alg_bytes_per_step": op_bytes / steps}
    return kern, dom, op


def shape_table(prof, name):
    """Per call shape of kernel `name`: launches, average duration, contract-algorithmic GB/s.  A fused kernel can read > peak
    here: the contract counts the reference operator's Delta / B / C / direction streams, which never exist in HBM (e.g. the
    d_inner-2 call of ss2d_bwd_apply) — `frac_traffic` in the same block is the fraction on bytes actually moved."""
    rows = []
    for g in (prof.get("__shapes__") or {}).get(name, []):
        us = g["ms"] / g["launches"] * 1e3
        gbs = g["alg_bytes_per_launch"] / (us * 1e-6) / 1e9 if us > 0 else 0.0
        rows.append({"alg_bytes_per_launch": g["alg_bytes_per_launch"], "launches": g["launches"], "avg_us": round(us, 2),
                     "GB/s": round(gbs, 1), "frac_contract": round(gbs / HBM_PEAK_GBS, 4),
                     **({"note": "contract bytes exceed what any kernel could move: the streams counted never reach HBM (fusion)"}
                        if gbs > HBM_PEAK_GBS else {})})
    return rows


def csrc_digest():
    """sha256 over the kernel sources: a PMC file measured on other sources must not be quoted as this build's traffic."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "vm_asr_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vm_asr_amd", "csrc", "*.h"))
                    + [os.path.join(ROOT, "vm_asr_amd", "csrc", "Makefile")]):       # (the build flags are part of what was measured)
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


VALU_LANE_RATE = 256 * 4 * 16 * 2.4e9     # lane-instructions per second: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
# VALU issue slots per state-step of csrc/sscan_n.hip's forward / backward, counted on the round-5 ISA (two fp32 operations per packed
# instruction).  Since round 6 the library is built WITHOUT packed-fp32 instructions (csrc/Makefile) and the stress point runs at the
# same speed (41.2 vs 40.5 clips/s, tools/hunt13.sh): a packed instruction takes two passes, so the slot count stands as a count of passes.
VALU_SLOTS_FWD, VALU_SLOTS_BWD = 14.0, 35.0


def scan_state_steps_per_clip(cfg):
    """Sum over the 34 selective-scan calls of a clip of KD x L x d_state (per-stream block schedule, x2 streams):
    the unit of the general-N scan's VALU roofline."""
    d = cfg.MODEL.VSSM.DIMS
    d = d[0] if isinstance(d, (list, tuple)) else d
    F, T = cfg.DATA.STFT.N_FFT // 2, 512          # bins without DC x frames of one clip
    stages = [(d, 4, 3), (2 * d, 8, 4), (4 * d, 16, 4), (8 * d, 32, 4), (d // 2, 2, 1), (1, 1, 1)]   # (d_model, downscale, blocks per stream); the last output block has d_model 1
    return sum(2 * blocks * (4 * 2 * dm) * (F // ds) * (T // ds) for dm, ds, blocks in stages) * cfg.MODEL.VSSM.SSM_D_STATE


def extra_point(name, workload, batch, mpd_gemm, args, device, rank, world, steps=10, with_metrics=False):
    """A second operating point measured in the same process after the headline (N = 1 only): compact record with its own
    roofline block.  `mpd_gemm`: VMASR_MPD_GEMM for this point (read when the discriminator's layers are built into the graph).
    A point that does not fit at its yaml's batch is retried at half the batch (recorded)."""
    saved = os.environ.get("VMASR_MPD_GEMM")
    os.environ["VMASR_MPD_GEMM"] = mpd_gemm
    tried = []
    try:
        while True:
            cfg = make_config(workload, batch)
            try:
                dt, graphed, _, prof = run_point(cfg, args, device, rank, world, steps, 3, True, with_metrics=with_metrics)
                break
            except torch.OutOfMemoryError:
                tried.append(cfg.DATA.BATCH_SIZE)
                torch.cuda.empty_cache()
                if cfg.DATA.BATCH_SIZE <= 1:
                    raise
                batch = cfg.DATA.BATCH_SIZE // 2
    finally:
        os.environ["VMASR_MPD_GEMM"] = saved if saved is not None else "bf16x3"
    B = cfg.DATA.BATCH_SIZE
    rec = {"workload": f"{workload}.yaml, per-GPU batch {B}, DIMS {cfg.MODEL.VSSM.DIMS}, d_state {cfg.MODEL.VSSM.SSM_D_STATE}, "
                       f"n_fft {cfg.DATA.STFT.N_FFT}, MPD GEMMs {mpd_gemm}" + (", SNR/LSD/LSD-HF/LSD-LF evaluated and read every step" if with_metrics else ""),
           "value": B * steps / dt, "unit": "clips/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "batch": B,
           "execution": "HIP graph replay" if graphed else "eager", **RUN_INFO}
    if tried:
        rec["out_of_memory_at_batch"] = tried
    shared_prof = prof if "__unshared__" in prof and any(not k.startswith("__") for k in prof) else None
    if "__unshared__" in prof:          # (two-stream step: kernels alone on the chip, see main()'s roofline block)
        prof = prof["__unshared__"]
    kern, dom, op = scan_summary(prof, steps)
    if dom:
        d = kern[dom]
        rec["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": d["gbs"] / HBM_PEAK_GBS, "avg_launch_us": d["avg_us"], "launches": d["launches"],
                           "alg_bytes_per_launch": d["alg_bytes"] / d["launches"], "traffic": None,
                           "selective_scan_op": op, "shapes": shape_table(prof, dom)}
        if shared_prof is not None:
            kern2, _, op2 = scan_summary(shared_prof, steps)
            d2 = kern2[dom]
            rec["roofline"]["shared_chip"] = {"kernel": dom, "achieved": d2["gbs"], "frac": d2["gbs"] / HBM_PEAK_GBS, "avg_launch_us": d2["avg_us"],
                                              "selective_scan_op": op2}
        rec["scan_alg_bytes_per_clip"] = op["alg_bytes_per_step"] / B if op else None
        if cfg.MODEL.VSSM.SSM_D_STATE > 1 and op:
            # general d_state: the scan is VALU-issue bound by a factor of N, so its roofline is the vector ALU's, not HBM's
            ss = scan_state_steps_per_clip(cfg) * B
            floor_ms = ss * (VALU_SLOTS_FWD + VALU_SLOTS_BWD) / VALU_LANE_RATE * 1e3
            rec["roofline"]["valu"] = {"state_steps_per_step": ss, "slots_per_state_step": {"fwd": VALU_SLOTS_FWD, "bwd": VALU_SLOTS_BWD},
                                       "floor_ms_per_step": floor_ms, "scan_op_ms_per_step": op["ms_per_step"], "frac": floor_ms / op["ms_per_step"],
                                       "achieved_slots_per_state_step": op["ms_per_step"] * 1e-3 * VALU_LANE_RATE / ss}
    top = sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:14]
    rec["top_kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "ms_per_step": round(v["ms"] / steps, 3)} for k, v in top}
    rec["unfused_chain_kernels"] = sorted(k for k in kern if k in ("cross_scan", "cross_merge", "xproj_fwd", "xproj_bwd_a", "xproj_bwd_b")
                                          or k.startswith("sscan_"))
    mf = {k: v for k, v in kern.items() if k.startswith("mlp_")}
    if mf:
        rec["mlp_mfma_kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2)} for k, v in mf.items()}
    return rec


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def _pick(d, keys, nd=4):
    return {k: _r(d[k], nd) for k in keys if d is not None and k in d}


MAX_LINE = 3000      # the driver keeps an 8 000-character tail of stdout: the final line must fit with room to spare


def compact(out):
    """The driver's line: contract keys + `roofline`, `cpu_baseline`, `operating_points`, `distributed` reduced to their
    headline numbers (<= MAX_LINE characters for any world size).  Per-kernel tables, per-shape tables, notes and per-rank
    device records are the DETAIL record (`bench_detail.json`), never part of this line."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data") if k in out}
    c["value"], c["ms_per_step"] = _r(c["value"], 3), _r(c["ms_per_step"], 4)
    cfg = out.get("config", {})
    c["config"] = {k: (v[:120] if isinstance(v, str) else v) for k, v in cfg.items() if not k.endswith("_note")}
    c["dtype"] = str(c.get("dtype"))[:80]
    r = out.get("roofline")
    if r:
        cr = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "avg_launch_us", "launches",
                       "alg_bytes_per_launch"))
        if r.get("selective_scan_op"):
            cr["selective_scan_op"] = _pick(r["selective_scan_op"], ("achieved", "frac", "ms_per_step"))
        sc = r.get("shared_chip")
        if sc:
            cr["shared_chip"] = {**_pick(sc, ("frac", "avg_launch_us")),
                                 "op_frac": _r((sc.get("selective_scan_op") or {}).get("frac")),
                                 "op_ms_per_step": _r((sc.get("selective_scan_op") or {}).get("ms_per_step"))}
        c["roofline"] = cr
    if "cpu_baseline" in out:
        cb = dict(out["cpu_baseline"])
        cb["value"] = _r(cb.get("value"), 5)
        cb["sample"] = str(cb.get("sample", ""))[:160]
        c["cpu_baseline"] = cb
    pts = out.get("operating_points")
    if pts:
        cp = {}
        for name, p in pts.items():
            if "error" in p:
                cp[name] = {"error": str(p["error"])[:80]}
                continue
            rr = p.get("roofline") or {}
            cp[name] = {"value": _r(p.get("value"), 2), "ms_per_step": _r(p.get("ms_per_step"), 3), "batch": p.get("batch"),
                        "scan_op_frac": _r((rr.get("selective_scan_op") or {}).get("frac")),
                        **({"valu_frac": _r(rr["valu"]["frac"])} if rr.get("valu") else {})}
        c["operating_points"] = cp
    d = out.get("distributed")
    if d:
        c["distributed"] = _pick(d, ("world_size", "backend", "rccl_version", "distinct_devices"))
    pr = out.get("per_rank")
    if pr:          # max over ranks only: the per-rank lists are in the detail record
        c["distributed"] = {**c.get("distributed", {}),
                            "ms_per_step_max": max(pr["ms_per_step"]), "ms_per_step_min": min(pr["ms_per_step"]),
                            "allreduce_exposed_ms_per_step": (None if any(v is None for v in pr["allreduce_exposed_ms_per_step"])
                                                              else max(pr["allreduce_exposed_ms_per_step"])),
                            "collectives": str(pr.get("collectives", ""))[:40]}
    c["detail"] = out.get("detail_file", "bench_detail.json")
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > MAX_LINE:        # never let an unforeseen field push the line out of the driver's tail again
        for k in ("operating_points", "distributed", "cpu_baseline"):
            if len(line) <= MAX_LINE:
                break
            if k == "cpu_baseline" and k in c:
                c[k] = _pick(c[k], ("value", "unit", "cores", "kind"))
            elif k in c:
                c[k] = {"see": c["detail"]}
            line = json.dumps(c, separators=(",", ":"))
    assert len(line) <= MAX_LINE, len(line)
    return line


def emit(out, detail_path):
    """Write the full record to `detail_path` (and a pointer to stderr), then print the compact line LAST on stdout."""
    out["detail_file"] = os.path.basename(detail_path)
    try:
        with open(detail_path, "w") as f:
            json.dump(out, f, indent=1)
        print(f"[bench] full record ({len(json.dumps(out))} characters): {detail_path}", file=sys.stderr, flush=True)
    except OSError as e:        # read-only checkout: the full record goes to stderr instead
        print(f"[bench] could not write {detail_path} ({e}); full record follows on stderr", file=sys.stderr)
        print(json.dumps(out), file=sys.stderr, flush=True)
    sys.stderr.flush()
    print(compact(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record (per-kernel / per-shape tables, notes, per-rank device records) is written; the "
                         "stdout line is the compact record only")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vm_asr_48k_MPD", choices=["vm_asr_48k_MPD", "vm_asr_48k", "vm_asr_48k_16k_MPD_VSSM32", "vm_asr_48k_16k_nfft2048",
                                                                  "vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the yaml's)")
    ap.add_argument("--no-amp", action="store_true")
    ap.add_argument("--amp-scope", default="generator", choices=["generator", "step"],
                    help="what bf16 autocast covers: 'generator' = the reference's scope (trainer/trainer.py:138-139: the "
                         "generator forward; losses and discriminator in fp32), 'step' = losses and discriminator too")
    ap.add_argument("--mpd-gemm", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="the fp32 discriminator's compute-bound GEMMs: 'bf16x3' = error-compensated triple bf16 MFMA products "
                         "(fp32 operands split into hi + lo bf16, fp32 accumulation; csrc/split.hip), 'fp32' = f32-input MFMA GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-points", action="store_true",
                    help="skip the secondary operating points (fp32 MPD GEMMs; per-step metrics; DIMS 32; n_fft 2048; d_state 32 + n_fft 2048)")
    ap.add_argument("--with-metrics", action="store_true",
                    help="evaluate SNR / LSD / LSD-HF / LSD-LF on the HIP STFT and read them on the host every step, as trainer/trainer.py:179-182 does")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-pass", choices=["both", "shared", "unshared"], default="both",
                    help="two-stream step: which eager HIP-event passes follow the timed region — the step as it runs (kernels of both streams "
                         "sharing the chip), a one-stream pass (every kernel alone on the chip: the `roofline` figures), or both (default)")
    ap.add_argument("--no-graphs", action="store_true", help="run the step eagerly instead of replaying HIP graphs")
    args = ap.parse_args()
    if os.environ.get("VMASR_BENCH_WATCHDOG"):      # debugging aid: dump all stacks and exit if the run wedges
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VMASR_BENCH_WATCHDOG"]), exit=True)

    os.environ["VMASR_MPD_GEMM"] = args.mpd_gemm
    from vm_asr_amd import _lib
    from vm_asr_amd.trainer import init_distributed
    rank, local, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)

    config = make_config(args.workload, args.batch)
    from vm_asr_amd.trainer import distributed_info
    dinfo = distributed_info(device)
    dt, graphed, per_rank, prof = run_point(config, args, device, rank, world, args.steps, args.warmup, not args.no_kernel_timing,
                                            with_metrics=args.with_metrics)
    main_info = dict(RUN_INFO)
    timing = prof is not None

    B = config.DATA.BATCH_SIZE
    two_stream = (config.TRAIN.ADVERSARIAL.ENABLE and os.environ.get("VMASR_TWO_STREAM", "1") == "1"
                  and not _lib.det_mode() and os.environ.get("VMASR_SHARE_FAKE_PASS", "1") == "1")
    out = {
        "metric": "audio clips/sec (train step) 48kHz n_fft=1024", "value": world * B * args.steps / dt,
        "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32" if args.no_amp else "bf16 (autocast over G; scan/STFT f32; MPD f32" + ("/bf16x3 MFMA)" if args.mpd_gemm == "bf16x3" else ")")
                  if args.amp_scope == "generator" else "bf16 (autocast over G, losses, MPD; scan/STFT f32)"),
        "dtype_note": ("f32" if args.no_amp else
                       "bf16 autocast over the generator forward as in the reference (selective scan / STFT fp32; losses and "
                       "discriminator fp32" + ("; its three compute-bound GEMM layers as error-compensated bf16x3 MFMA products, fp32 "
                                               "accumulation)" if args.mpd_gemm == "bf16x3" else ")") if args.amp_scope == "generator" else
                       "bf16 autocast over generator, losses and discriminator (selective scan / STFT fp32)"), "data": "synthetic",
        "config": {"workload": f"{args.workload}.yaml full train step, DIMS {config.MODEL.VSSM.DIMS} d_state {config.MODEL.VSSM.SSM_D_STATE} "
                               f"n_fft {config.DATA.STFT.N_FFT} clip 122640@48k",
                   "workload_note": f"{args.workload}.yaml full train step (G fwd+bwd, "
                                    f"{'MR-STFT+LSGAN+feature losses, MPD fwd+bwd, ' if config.TRAIN.ADVERSARIAL.ENABLE else 'MR-STFT loss, '}"
                                    f"AdamW); DIMS {config.MODEL.VSSM.DIMS}, d_state {config.MODEL.VSSM.SSM_D_STATE}, clip 122640 @48 kHz, "
                                    f"n_fft {config.DATA.STFT.N_FFT} hop 240",
                   "per_gpu_batch": B, "global_batch": B * world,
                   "parallelism": f"dp{world}",
                   "parallelism_note": "clip-sharded; one RCCL all-reduce per flat gradient buffer",
                   "execution": ("hipGraph replay" + (", 2 streams" if two_stream else "")) if graphed else ("eager, 2 streams" if two_stream else "eager"),
                   "execution_note": (("HIP graph replay, node by node (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): one forward + backward graph with a fork / "
                                       "join — the period discriminator on a side stream beside the generator (trainer._two_streams), its convolution "
                                       "kernels under a CU limit while they overlap (trainer.side_cu_limits) — and an optimiser graph")
                                      if two_stream else
                                      ("HIP graph replay, node by node (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; forward + D-backward graph, G-backward graph, "
                                       "optimiser graph)")) if graphed else ("eager, two streams" if two_stream else "eager")},
    }
    out["config"]["generator_streams"] = main_info.get("generator_streams")      # (trainer.enable_graphs: the faster of the two captures)
    out["config"]["graph_variants_ms_note"] = main_info.get("graph_variants_ms")
    out["distributed"] = dinfo     # what the process group actually was: world size, backend, RCCL version, every rank's device
    out["config"]["per_step_metrics"] = bool(args.with_metrics)
    if per_rank is not None:
        out["per_rank"] = per_rank
    if rank == 0 and timing:
        # two-stream step: `roofline` is each kernel ALONE on the chip (the one-stream pass: the kernel's own roofline fraction, comparable with
        # earlier rounds and with profiles/*_onestream_kernel_stats.csv); the durations in the step as it runs, where the generator's kernels
        # share the chip with the discriminator's, are `roofline.shared_chip` (profiles/*_trainstep_kernel_stats.csv)
        shared_prof = prof if "__unshared__" in prof and any(not k.startswith("__") for k in prof) else None
        if "__unshared__" in prof:
            prof = prof["__unshared__"]
        kern, dom, op = scan_summary(prof, args.steps)
        if dom:
            d = kern[dom]
            # HBM bytes per launch from the PMC passes of this same workload (rocprofv3 cannot run inside bench.py): profiles/<PMC_FILE>,
            # made by tools/evidence_r05.sh / tools/pmc_bench_report.py, which records the digest of the kernel sources it measured;
            # a file measured on other sources is NOT quoted (traffic = null, traffic_note says why)
            traffic, traffic_note, pmc_file = None, None, PMC_FILE
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                if pj.get("csrc_digest") != csrc_digest():
                    traffic_note = f"profiles/{pmc_file} was measured on other kernel sources (digest {pj.get('csrc_digest')} != {csrc_digest()}): not quoted"
                elif B == 4 and args.workload == "vm_asr_48k_MPD":
                    traffic = pj["kernels"][dom]["hbm_bytes_per_launch"]
            except Exception as e:
                traffic_note = f"no PMC file for this build ({type(e).__name__})"
            out["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": d["gbs"] / HBM_PEAK_GBS, "traffic": traffic,
                "frac_traffic": (traffic / (d["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "traffic_source": f"profiles/{pmc_file} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                  "FETCH_SIZE x2 per the gfx950 correction; same kernel sources as this build)" if traffic else traffic_note,
                "frac_definition": "frac = contract-algorithmic bytes / time / peak (what the reference's operator would move); "
                                   "frac_traffic = PMC bytes actually moved / time / peak — the fused kernels are VALU-issue-bound, not HBM-bound",
                "shapes": shape_table(prof, dom),
                "bytes_definition": "SURVEY.md 8(d) algorithmic bytes of the selective-scan calls the launch performs (for the fused "
                                    "ss2d_* kernels: the reference contract's Delta/B/C/direction streams that never reach HBM here "
                                    "are part of it, so `traffic` << algorithmic bytes is the fusion, not over-fetch)",
                "avg_launch_us": d["avg_us"],
                "timed_in": "eager pass of the same K steps right after the timed region (HIP events on the launch stream)" +
                            (", on ONE stream (see `chip`)" if two_stream and args.timing_pass != "shared" else ""),
                "launches": d["launches"], "alg_bytes_per_launch": d["alg_bytes"] / d["launches"],
                "selective_scan_op": op,
                "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "GB/s": round(v["gbs"], 1),
                                "ms_per_step": round(v["ms"] / args.steps, 3)} for k, v in sorted(kern.items())},
            }
            out["roofline"]["chip"] = ("kernel alone on the chip: one-stream eager pass (VMASR_TWO_STREAM=0) of the same K steps" if two_stream and args.timing_pass != "shared"
                                       else "step as it runs" + (": kernels of the two streams share the chip" if two_stream else ""))
            if shared_prof is not None:
                kern2, _, op2 = scan_summary(shared_prof, args.steps)
                d2 = kern2[dom]
                out["roofline"]["shared_chip"] = {
                    "note": "the same kernels timed in the step as it runs — two streams, the period discriminator's MFMA kernels beside the generator's "
                            "(what rocprofv3 of the two-stream step sees): durations of kernels that have a share of the chip, not a statement about the kernels",
                    "kernel": dom, "achieved": d2["gbs"], "frac": d2["gbs"] / HBM_PEAK_GBS, "avg_launch_us": d2["avg_us"],
                    "selective_scan_op": op2,
                    "shapes": shape_table(shared_prof, dom),      # per call shape, beside `roofline.shapes` (alone on the chip)
                    "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "GB/s": round(v["gbs"], 1),
                                    "ms_per_step": round(v["ms"] / args.steps, 3)} for k, v in sorted(kern2.items())}}
    if rank == 0 and world == 1 and not args.no_extra_points and args.workload == "vm_asr_48k_MPD" and not args.batch and not args.no_graphs:
        # driver-visible secondary operating points: the reference-precision discriminator GEMMs, and
        # configs[4]'s yaml (DIMS 32, batch 8) with its own roofline block.  Never part of `value`.
        pts = {}
        for name, wl, bsz, gemm, met in (("mpd_gemm_fp32", "vm_asr_48k_MPD", 0, "fp32", False),
                                         ("with_metrics", "vm_asr_48k_MPD", 0, args.mpd_gemm, True),
                                         ("vm_asr_48k_16k_MPD_VSSM32", "vm_asr_48k_16k_MPD_VSSM32", 0, args.mpd_gemm, False),
                                         ("vm_asr_48k_generator_only", "vm_asr_48k", 0, args.mpd_gemm, False),
                                         ("vm_asr_48k_16k_nfft2048", "vm_asr_48k_16k_nfft2048", 0, args.mpd_gemm, False),
                                         ("vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048", "vm_asr_48k_16k_MPD_VSSM32_dstate32_nfft2048", 0,
                                          args.mpd_gemm, False)):
            try:
                pts[name] = extra_point(name, wl, bsz, gemm, args, device, rank, world, steps=(5 if "dstate32" in name else 10), with_metrics=met)
            except Exception as e:   # informative only
                pts[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        out["operating_points"] = pts
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(config)
        except Exception as e:  # the baseline is informative; never lose the GPU number over it
            out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        emit(out, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
