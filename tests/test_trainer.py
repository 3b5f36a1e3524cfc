"""Host-logic tests of the training harness on CPU (oracle kernels in the operator hooks):
one train step of G+MPD runs, updates parameters, leaves the 129 never-used tensors untouched;
world_size-2 gloo DDP gives both ranks identical parameters equal to a single-process step on
the concatenated batch."""
import numpy as np
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_config(gan=True, batch=2):
    from vm_asr_amd.config import get_default_config, update_config
    c = get_default_config()
    c.MODEL.NAME = "DualStreamInteractiveMambaUNet"
    c.MODEL.VSSM.DIMS = 8
    c.MODEL.VSSM.DROP_PATH_RATE = 0.0
    c.DATA.STFT.N_FFT = 128
    c.DATA.STFT.WIN_LENGTH = 128
    c.DATA.TARGET_SR = 16000           # -> hop 80
    c.DATA.SEGMENT = 80 * 63 / 16000   # 64 frames
    c.DATA.BATCH_SIZE = batch
    c.TRAIN.LOW_FREQ_REPLACEMENT = True
    c.TRAIN.ADVERSARIAL.ENABLE = gan
    c.TRAIN.ADVERSARIAL.DISCRIMINATORS = ["mpd"]
    c.TRAIN.ADVERSARIAL.MPD_HIDDEN = 2
    return update_config(c)


def _make_trainer(cfg, dp_mode="flat"):
    import vm_asr_amd
    from oracle.torch_backend import use_oracle
    from vm_asr_amd.trainer import Trainer, build_optimizer
    torch.manual_seed(cfg.SEED)
    models = vm_asr_amd.get_model(cfg)
    use_oracle(models["generator"])
    opts = {"generator": build_optimizer(cfg, models["generator"])}
    if cfg.TRAIN.ADVERSARIAL.ENABLE:
        opts["discriminator"] = build_optimizer(cfg, [models["mpd"]])
    return Trainer(models, [], opts, cfg, torch.device("cpu"), None, None, {}, amp=False,
                   gan=cfg.TRAIN.ADVERSARIAL.ENABLE, len_epoch=0, dp_mode=dp_mode)


def _batch(cfg, n, seed=0):
    T = int(cfg.DATA.SEGMENT * cfg.DATA.TARGET_SR)
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(n, 1, T, generator=g), 0.1 * torch.randn(n, 1, T, generator=g),
            torch.full((n,), 21, dtype=torch.int64))


def test_train_step_cpu_oracle_backend():
    from oracle.torch_backend import oracle_stft_patch
    from vm_asr_amd.trainer import unwrap
    cfg = _tiny_config()
    tr = _make_trainer(cfg)
    for m in tr.models.values():
        m.train()
    before = {k: v.clone() for k, v in tr.models["generator"].state_dict().items()}
    d_before = {k: v.clone() for k, v in tr.models["mpd"].state_dict().items()}
    with oracle_stft_patch():
        out, logs = tr.train_step(*_batch(cfg, 2))
    assert out.shape == (2, 1, 80 * 63)
    assert all(torch.isfinite(v) for v in logs.values())
    assert {"generator/multi_resolution_stft", "generator/adversarial_mpd", "generator/features_mpd",
            "total_disc_loss"} <= set(logs)
    after = tr.models["generator"].state_dict()
    changed = [k for k in before if not torch.equal(before[k], after[k])]
    unused = [k for k in before if k.startswith("layers_decoder_phase.") and not k.startswith("layers_decoder_phase.0.")]
    assert len(unused) == 129 and not set(unused) & set(changed)
    # the last output block has d_model 1 -> d_inner 2: its out_norm is a LayerNorm over TWO elements
    # (+-1 up to eps), which passes (numerically) zero gradient to the dt projection inside it
    assert len(before) - 129 - 4 <= len(changed) <= len(before) - 129
    assert any(not torch.equal(d_before[k], v) for k, v in unwrap(tr.models["mpd"]).state_dict().items())


def _ddp_worker(rank, world, port, ret, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle.torch_backend import oracle_stft_patch
    from vm_asr_amd.trainer import init_distributed, unwrap
    init_distributed()
    cfg = _tiny_config(gan=True, batch=1)
    tr = _make_trainer(cfg, dp_mode=mode)
    for m in tr.models.values():
        m.train()
    full = _batch(cfg, 2, seed=5)
    mine = tuple(t[rank:rank + 1] for t in full)
    with oracle_stft_patch():
        tr.train_step(*mine)
    sd = {k: v.detach().clone() for k, v in unwrap(tr.models["generator"]).state_dict().items()}
    ret[rank] = sd
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["flat", "ddp"])
def test_ddp_gloo_world2_matches_single_process(mode):
    from oracle.torch_backend import oracle_stft_patch
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + (7 if mode == "ddp" else 0)) % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, ret, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    sd0, sd1 = ret[0], ret[1]
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), f"ranks diverged on {k}"
    if mode == "ddp":
        return
    # single process, batch of both clips: DDP averages gradients of per-rank mean losses, which equals
    # the gradient of the mean over the global batch for the losses that are batch means; the
    # spectral-convergence term is a ratio of norms over the batch (not a mean), so compare loosely.
    cfg = _tiny_config(gan=True, batch=2)
    tr = _make_trainer(cfg)
    for m in tr.models.values():
        m.train()
    ref0 = {k: v.clone() for k, v in tr.models["generator"].state_dict().items()}
    with oracle_stft_patch():
        tr.train_step(*_batch(cfg, 2, seed=5))
    sd = tr.models["generator"].state_dict()
    moved = same_dir = 0
    for k in sd0:
        d_ddp, d_one = (sd0[k] - ref0[k]).flatten().double(), (sd[k] - ref0[k]).flatten().double()
        if d_one.abs().sum() > 0:
            moved += 1
            same_dir += int(torch.dot(d_ddp, d_one) > 0)
    assert moved > 400 and same_dir / moved > 0.9


def _gpu_trainer(cfg, amp, capturable=False):
    import vm_asr_amd
    from vm_asr_amd.trainer import Trainer, build_optimizer
    torch.manual_seed(cfg.SEED)
    models = vm_asr_amd.get_model(cfg)
    opts = {"generator": build_optimizer(cfg, models["generator"], capturable=capturable),
            "discriminator": build_optimizer(cfg, [models["mpd"]], capturable=capturable)}
    return Trainer(models, [], opts, cfg, torch.device("cuda", 0), None, None, {}, amp=amp, gan=True, len_epoch=0)


@pytest.mark.gpu
def test_train_step_gpu_lp_shadows(monkeypatch):
    """bf16 autocast step on the GPU (HIP kernels): the trainer's bf16 shadow weights give the same update
    as casting every weight in every forward, stay equal to the fp32 weights' bf16 rounding, and the
    HIP-graph replay of the step matches the eager step."""
    from vm_asr_amd.linear import LP_ATTR
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    results = {}
    for shadows in ("1", "0"):
        monkeypatch.setenv("VMASR_LP_SHADOWS", shadows)
        tr = _gpu_trainer(cfg, amp=True)
        for m in tr.models.values():
            m.train()
        assert bool(tr._shadow_dst) == (shadows == "1")
        for _ in range(2):
            out, logs = tr.train_step(*batch)
        assert all(torch.isfinite(v) for v in logs.values())
        results[shadows] = {k: v.detach().float().clone() for k, v in tr.models["generator"].state_dict().items()}
        if shadows == "1":
            for p in tr.models["generator"].parameters():
                assert torch.equal(getattr(p, LP_ATTR), p.detach().to(torch.bfloat16))
    worst = max((results["1"][k] - results["0"][k]).abs().max().item() for k in results["1"])
    assert worst <= 2e-3, worst       # AdamW steps are lr-sized (1e-4..1e-3): same update direction everywhere


@pytest.mark.gpu
def test_train_step_gpu_graph_matches_eager():
    """Replaying the captured HIP graphs trains like the eager step: same loss trajectory over six steps
    (the weights themselves drift apart chaotically - AdamW amplifies rounding-level gradient differences
    to +-lr, two eager runs differ the same way, tools/graph_vs_eager.py - so the losses are the check)."""
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    hist = []
    for graphs in (False, True):
        tr = _gpu_trainer(cfg, amp=False, capturable=True)
        for m in tr.models.values():
            m.train()
        h = []
        if graphs:
            assert tr.enable_graphs(batch, warmup=3)     # runs 3 real steps on `batch` before capturing
        for _ in range(3 if graphs else 6):
            out, logs = tr.train_step(*batch)
            h.append({k: float(v) for k, v in logs.items()})
        torch.cuda.synchronize()
        hist.append(h[-3:])
    for a, b in zip(*hist):
        for k, v in a.items():
            assert abs(v - b[k]) <= 0.02 * abs(v) + 1e-4, (k, v, b[k])
    assert hist[0][-1]["total_loss"] < hist[0][0]["total_loss"]     # and it is learning


@pytest.mark.gpu
def test_shared_fake_pass_gives_the_same_gradients(monkeypatch):
    """One discriminator pass over the generated signal serving both losses (phase-restricted backwards) ==
    the reference's two passes (detached for the D loss, attached for the G loss): same losses, same gradients
    of every generator and discriminator parameter."""
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VMASR_SHARE_FAKE_PASS", flag)
        tr = _gpu_trainer(cfg, amp=False)
        for m in tr.models.values():
            m.train()
        assert tr._share_fake_pass() == (flag == "1")
        _, logs = tr._forward_backward(*batch)
        grads = {f"{k}.{n}": p.grad.detach().clone() for k in ("generator", "mpd")
                 for n, p in tr.models[k].named_parameters() if p.grad is not None}
        out[flag] = ({k: float(v) for k, v in logs.items()}, grads)
    (la, ga), (lb, gb) = out["1"], out["0"]
    for k in lb:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]) + 1e-7, (k, la[k], lb[k])
    assert ga.keys() == gb.keys()
    for k in gb:
        scale = max(gb[k].abs().max().item(), 1e-8)
        assert (ga[k] - gb[k]).abs().max().item() <= 2e-3 * scale + 1e-7, (k, (ga[k] - gb[k]).abs().max().item(), scale)


# ---- round 2: batch contract, checkpoints, schedule, accumulation, multi-process decisions -------------------
def test_synthetic_vctk_batch_contract():
    """H0: `(wave_in (1,T), wave_tgt (1,T), highcut int64, name, pad)` with T = int(SEGMENT * TARGET_SR) and
    highcut = int((n_fft/2+1) * sr_in / sr_tgt) — data_loader/data_loaders.py:490-513, :482-486, :138-140 —
    collated by the default DataLoader into what Trainer._to_dev consumes."""
    from vm_asr_amd.config import get_config
    from vm_asr_amd.trainer import SyntheticVCTK
    cfg = get_config(opts=["DATA.TARGET_SR", 48000])
    ds = SyntheticVCTK(cfg, length=6, sr_in=16000)
    assert len(ds) == 6
    inp, tgt, hc, name, pad = ds[3]
    T = int(2.555 * 48000)
    assert T == 122640 and inp.shape == tgt.shape == (1, T) and inp.dtype == tgt.dtype == torch.float32
    assert hc.dtype == torch.int64 and int(hc) == int(513 * 16000 / 48000) == 171 and pad == 0 and isinstance(name, str)
    assert 0.05 < inp.std() < 0.2 and inp.abs().max() < 1.0 and not torch.equal(inp, tgt)
    assert torch.equal(ds[3][0], inp) and not torch.equal(ds[4][0], inp)      # deterministic per index
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))
    assert batch[0].shape == (4, 1, T) and batch[1].shape == (4, 1, T) and batch[2].shape == (4,) and batch[2].dtype == torch.int64
    assert len(batch[3]) == 4 and batch[4].shape == (4,)
    c16 = get_config(opts=["DATA.TARGET_SR", 16000])
    assert SyntheticVCTK(c16, 1, sr_in=8000)[0][0].shape == (1, 40880) and c16.DATA.STFT.HOP_LENGTH == 80


def _resumable(cfg, tmp, device="cpu", capturable=False):
    import vm_asr_amd
    from oracle.torch_backend import use_oracle
    from vm_asr_amd.trainer import CosineWarmupScheduler, Trainer, build_optimizer
    torch.manual_seed(cfg.SEED)
    models = vm_asr_amd.get_model(cfg)          # built on the CPU, as main.py does before the trainer moves them
    if device == "cpu":
        use_oracle(models["generator"])
    opts = {"generator": build_optimizer(cfg, models["generator"], capturable),
            "discriminator": build_optimizer(cfg, [models["mpd"]], capturable)}
    sched = {k: CosineWarmupScheduler(o, 100, 10, cfg.TRAIN.BASE_LR, cfg.TRAIN.MIN_LR) for k, o in opts.items()}
    tr = Trainer(models, [], opts, cfg, torch.device(device), None, None, sched, amp=False, gan=True, len_epoch=0)
    for m in tr.models.values():
        m.train()
    return tr


def _ckpt_roundtrip(device, tmp_path, graphs=False):
    import contextlib
    from vm_asr_amd.config import yacs_pickle_compat
    from vm_asr_amd.trainer import unwrap
    if device == "cpu":
        from oracle.torch_backend import oracle_stft_patch as patch
    else:
        patch = contextlib.nullcontext
    cfg = _tiny_config()
    cfg.defrost()
    cfg.OUTPUT = str(tmp_path)
    cfg.freeze()
    batch = [t.to(device) for t in _batch(cfg, 2)]
    with patch():
        a = _resumable(cfg, tmp_path, device, capturable=graphs)
        a.train_step(*batch)
        a.mnt_best = 0.75
        a._save_checkpoint(3, save_best=True)
        files = sorted(os.listdir(tmp_path))
        assert files == ["checkpoint-best-G.pth", "checkpoint-best-mpd.pth", "checkpoint-latest-G.pth", "checkpoint-latest-mpd.pth"]
        # dict layout of base/base_trainer.py:146-153; `config` is an object with defrost()/freeze() that pickles as
        # yacs.config.CfgNode, which is what utils/utils.py:141-145 needs when the REFERENCE resumes from this file
        with yacs_pickle_compat():
            ck = torch.load(os.path.join(tmp_path, "checkpoint-best-G.pth"), map_location="cpu", weights_only=False)
        assert set(ck) == {"name", "epoch", "state_dict", "optimizer", "monitor_best", "config"}
        assert ck["name"] == "G" and ck["epoch"] == 3 and ck["monitor_best"] == 0.75
        assert (type(ck["config"]).__module__, type(ck["config"]).__name__) == ("yacs.config", "CfgNode")
        ck["config"].defrost(); ck["config"].MODEL.RESUME_PATH = "x"; ck["config"].freeze()
        assert ck["config"].DATA.STFT.N_FFT == 128 and ck["config"].is_frozen()
        raw = open(os.path.join(tmp_path, "checkpoint-best-G.pth"), "rb").read()
        assert b"vm_asr_amd" not in raw          # nothing in the file needs this package to unpickle
        a.train_step(*batch)                      # the step the resumed trainer has to reproduce
        want = {k: v.detach().cpu().clone() for m in ("generator", "mpd") for k, v in unwrap(a.models[m]).state_dict().items()}

        cfg2 = cfg.clone()
        cfg2.MODEL.RESUME_PATH = str(tmp_path)
        cfg2.freeze()
        b = _resumable(cfg2, tmp_path, device, capturable=graphs)      # resumes inside the constructor
        assert b.start_epoch == 4 and b.mnt_best == 0.75 and b.config.MODEL.RESUME_PATH == str(tmp_path)
        for opt in (b.optimizer_G, b.optimizer_D):
            for st in opt.state.values():
                assert all(v.device.type == torch.device(device).type for v in st.values() if torch.is_tensor(v) and v.ndim > 0)
        if graphs:
            assert all(torch.is_tensor(g["lr"]) and g["lr"].is_cuda for o in (b.optimizer_G, b.optimizer_D) for g in o.param_groups)
        b.train_step(*batch)
        got = {k: v.detach().cpu() for m in ("generator", "mpd") for k, v in unwrap(b.models[m]).state_dict().items()}
        if graphs:      # and the resumed trainer can be captured and replayed
            assert b.enable_graphs(batch, warmup=2)
            _, logs = b.train_step(*batch)
            assert all(torch.isfinite(v) for v in logs.values())
    return want, got


def test_checkpoint_save_resume_step_roundtrip_cpu(tmp_path):
    """save -> new trainer with MODEL.RESUME_PATH (models built on the CPU) -> one step == the step the saving
    trainer takes next (base/base_trainer.py:130-179, utils/utils.py:112-178)."""
    want, got = _ckpt_roundtrip("cpu", tmp_path)
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=1e-5, atol=1e-7), k


def test_accumulation_steps_and_epoch_schedule_cadence():
    """TRAIN.ACCUMULATION_STEPS = 2: the optimisers run on every second micro-batch; the gradient they see is the
    sum of both micro-batches' (loss / 2) gradients.  The LR schedule advances once per epoch with the reference's
    update index (trainer/trainer.py:196-218)."""
    from oracle.torch_backend import oracle_stft_patch
    from vm_asr_amd.trainer import CosineWarmupScheduler
    cfg = _tiny_config(gan=False, batch=1)    # (G only: with the MPD in train mode its u, v advance per forward)
    cfg.defrost()
    cfg.TRAIN.ACCUMULATION_STEPS = 2
    cfg.freeze()
    tr = _make_trainer(cfg)
    for m in tr.models.values():
        m.train()
    p = tr.models["generator"].patch_embed_mag[0].weight
    p0 = p.detach().clone()
    b1, b2 = _batch(cfg, 1, seed=1), _batch(cfg, 1, seed=2)
    with oracle_stft_patch():
        tr.train_step(*b1)
        assert torch.equal(p.detach(), p0) and tr.global_step == 0
        g1 = p.grad.detach().clone()
        tr.train_step(*b2)
        assert not torch.equal(p.detach(), p0) and tr.global_step == 1
        g12 = p.grad.detach().clone()
        # second micro-batch alone (fresh accumulation cycle, same weights would be needed for equality -> use a twin)
        tw = _make_trainer(cfg)
        for m in tw.models.values():
            m.train()
        tw._micro = 1                        # zero=False semantics are exercised above; here: b2's own gradient
        st = tw._forward_losses(*b2)
        tw._backward_d(st, zero=True)
        tw._backward_g(st, zero=True)
        g2 = tw.models["generator"].patch_embed_mag[0].weight.grad
    assert torch.allclose(g12, g1 + g2, rtol=1e-4, atol=1e-6 * g12.abs().max().item())
    # schedule cadence
    sched = CosineWarmupScheduler(tr.optimizer_G, total_steps=40, warmup_steps=4, base_lr=1e-3, min_lr=1e-5)
    tr.lr_scheduler_G, tr.len_epoch = sched, 4
    tr.data_loader = [(b1[0], b1[1], b1[2], "n", 0)] * 4
    lr0 = float(tr.optimizer_G.param_groups[0]["lr"])
    with oracle_stft_patch():
        tr._train_epoch(1)
    # 4 batches, acc 2 -> num_steps 2, update index (1*2 + 3) // 2 = 2 -> lr = min + (base-min) * 2/4
    assert abs(lr0 - 1e-5) < 1e-12 and abs(float(tr.optimizer_G.param_groups[0]["lr"]) - (1e-5 + (1e-3 - 1e-5) * 0.5)) < 1e-9


def test_wgan_gp_penalty_reaches_discriminator_weights_cpu():
    from vm_asr_amd.discriminator import MultiPeriodDiscriminator
    from vm_asr_amd.loss import HiFiGANLoss
    torch.manual_seed(0)
    D = MultiPeriodDiscriminator(hidden=2).train()
    y, yh = 0.3 * torch.randn(2, 1, 700), 0.3 * torch.randn(2, 1, 700)
    torch.manual_seed(1)
    gp = HiFiGANLoss("wgan-gp").gradient_penalty(y, yh, D)
    gp.backward()
    n = sum(float(p.grad.abs().sum()) > 0 for p in D.parameters() if p.grad is not None)
    assert gp.item() > 0 and n >= 30, n


def _sync_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vm_asr_amd.config import get_config
    from vm_asr_amd.trainer import BaseTrainer, init_distributed
    init_distributed()

    class _T(BaseTrainer):
        def __init__(self, cfg):
            super().__init__({}, [], {}, cfg)
            self.n = 0

        def _train_epoch(self, epoch):
            self.n += 1
            # rank-local values that disagree: rank 1 keeps "improving", rank 0 never does; NaN on rank 1 in epoch 4
            self.epoch_log = {"lsd": (5.0 - epoch) if self.rank == 1 else 5.0, "total_loss": float("nan") if (self.rank == 1 and epoch == 4) else 1.0}

        def _save_checkpoint(self, epoch, save_best=False):
            ret[(self.rank, epoch)] = (save_best, self.mnt_best)

    cfg = get_config(opts=["TRAIN.EPOCHS", 6, "TRAIN.EARLY_STOPPING", 1])
    t = _T(cfg)
    try:
        t.train()
        ret[(rank, "exit")] = "done"
    except SystemExit:
        ret[(rank, "exit")] = "nan-abort"
    ret[(rank, "epochs")] = t.n
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_take_the_same_epoch_decisions_gloo_world2():
    """Rank-local epoch metrics differ, yet both ranks see the mean, mark the same epochs as best and leave train()
    in the same epoch (here: the NaN abort of epoch 4 raised on BOTH ranks although only rank 1 produced the NaN)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() + 13) % 2000
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret[(0, "exit")] == ret[(1, "exit")] == "nan-abort"
    assert ret[(0, "epochs")] == ret[(1, "epochs")] == 4
    for e in (1, 2, 3):
        assert ret[(0, e)] == ret[(1, e)] == (True, 5.0 - e / 2)       # mean of 5 and 5-e, improving every epoch


@pytest.mark.gpu
def test_checkpoint_save_resume_step_roundtrip_gpu(tmp_path):
    """Same on the GPU with capturable optimisers (device lr tensors): the resumed optimiser state lives on the
    device although the models are built on the CPU, the step it takes equals the one the saving trainer takes
    (two GPU runs differ by atomics-order rounding, which AdamW turns into at most +-lr where |g| is at rounding
    level: atol 2e-4), and the resumed trainer captures and replays as HIP graphs."""
    want, got = _ckpt_roundtrip("cuda:0", tmp_path, graphs=True)
    bad = [k for k in want if not torch.allclose(got[k].float(), want[k].float(), rtol=1e-3, atol=2e-4)]
    assert len(bad) <= 0.01 * len(want), (len(bad), len(want), bad[:5])


@pytest.mark.gpu
def test_lr_schedule_takes_effect_under_graph_replay():
    """with HIP-graph replay the learning rate must be read from device memory at replay time.
    lr = 0 -> a replayed step changes no parameter; lr back to 1e-3 -> it does."""
    from vm_asr_amd.trainer import CosineWarmupScheduler
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    tr = _gpu_trainer(cfg, amp=False, capturable=True)
    for m in tr.models.values():
        m.train()
    scheds = [CosineWarmupScheduler(o, 1000, 0, 1e-3, 0.0) for o in (tr.optimizer_G, tr.optimizer_D)]
    assert all(torch.is_tensor(g["lr"]) and g["lr"].is_cuda for o in (tr.optimizer_G, tr.optimizer_D) for g in o.param_groups)
    assert tr.enable_graphs(batch, warmup=2)
    snap = lambda: {k: v.detach().clone() for m in ("generator", "mpd") for k, v in tr.models[m].state_dict().items()   # noqa: E731
                    if v.is_floating_point() and not k.endswith(("._u", "._v"))}
    for s in scheds:
        s.base_lr = 0.0
        s.step_update(0)               # lr := 0 through the scheduler's own path
    before = snap()
    tr.train_step(*batch)
    torch.cuda.synchronize()
    after = snap()
    assert all(torch.equal(before[k], after[k]) for k in before), "lr = 0 must freeze the weights under replay"
    for s in scheds:
        s.base_lr = 1e-3
        s.step_update(0)
    tr.train_step(*batch)
    torch.cuda.synchronize()
    moved = sum(not torch.equal(after[k], v) for k, v in snap().items())
    assert moved > 0.5 * len(after), moved


@pytest.mark.gpu
def test_wgan_gp_penalty_on_gpu_matches_cpu():
    """the gradient penalty needs double backward; on the GPU it runs the discriminator on plain torch
    operators (discriminator.plain_torch_ops) and must deliver the same D-weight gradients as the CPU run."""
    import copy
    from vm_asr_amd.discriminator import MultiPeriodDiscriminator
    from vm_asr_amd.loss import HiFiGANLoss
    torch.manual_seed(0)
    D = MultiPeriodDiscriminator(hidden=2).eval()       # eval: u, v fixed -> same sigma on both devices
    E = copy.deepcopy(D).cuda()
    y, yh = 0.3 * torch.randn(2, 1, 700), 0.3 * torch.randn(2, 1, 700)
    L = HiFiGANLoss("wgan-gp")
    torch.manual_seed(1); a = L.gradient_penalty(y, yh, D)
    a.backward()
    alpha = torch.rand(2, 1, 1, generator=torch.Generator().manual_seed(1))    # noqa: F841  (documentation: same alpha needed)
    # same interpolation points on the GPU: draw alpha on the CPU generator state, then move
    torch.manual_seed(1)
    al = torch.rand(2, 1, 1)
    orig = torch.rand
    try:
        torch.rand = lambda *a_, **k_: al.to(k_.get("device", "cpu"))
        b = L.gradient_penalty(y.cuda(), yh.cuda(), E)
    finally:
        torch.rand = orig
    b.backward()
    assert abs(a.item() - b.item()) <= 1e-3 * abs(a.item())
    for (n, p), (_, q) in zip(D.named_parameters(), E.named_parameters()):
        if p.grad is None or q.grad is None:       # e.g. conv_post.bias: the input gradient does not depend on it
            assert (p.grad is None or float(p.grad.abs().max()) == 0) and (q.grad is None or float(q.grad.abs().max()) == 0), n
            continue
        assert torch.allclose(q.grad.cpu(), p.grad, rtol=5e-3, atol=1e-4 * p.grad.abs().max().item() + 1e-8), n
    assert sum(q.grad is not None and float(q.grad.abs().max()) > 0 for q in E.parameters()) >= 30


@pytest.mark.gpu
def test_hip_adamw_step_matches_torch_fused_adamw():
    """fused_adamw.HipAdamWStep (csrc/adamw.hip: one launch for all tensors, bf16 shadows in the same pass) ==
    torch.optim.AdamW(fused, capturable) on the same state, over several steps with a changing device learning rate:
    parameters, exp_avg, exp_avg_sq, step counters; odd sizes, unaligned gradient views, two weight-decay groups."""
    from vm_asr_amd.fused_adamw import HipAdamWStep
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    shapes = [(1024, 33), (7,), (4097,), (64, 5, 3, 1), (1,), (8192,), (3, 5)]
    flat = torch.zeros(sum(int(np.prod(s)) for s in shapes) + 3, device=dev)

    def make():
        ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
        off = 3                                                      # gradient views start at an odd offset: unaligned
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        lr = torch.tensor(1e-3, device=dev)
        groups = [{"params": [p for p in ps if p.ndim > 1]}, {"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0}]
        return ps, lr, torch.optim.AdamW(groups, lr=lr, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.05, capturable=True, fused=True)
    torch.manual_seed(4)
    pa, lra, oa = make()
    torch.manual_seed(4)
    pb, lrb, ob = make()
    flat.normal_()
    oa.step(); ob.step()                                             # torch creates the state
    with pytest.raises(ValueError):
        HipAdamWStep(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(3, device=dev))], lr=1e-3))   # not capturable / host lr
    shadows = {id(p): p.detach().to(torch.bfloat16) for p in pb if p.ndim > 1}
    hip = HipAdamWStep(ob, shadows)
    for it in range(4):
        flat.normal_()
        lra.fill_(1e-3 / (it + 1)); lrb.fill_(1e-3 / (it + 1))
        oa.step()
        hip.step()
    assert hip.still_valid()
    for x, y in zip(pa, pb):
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"]) == 5.0
        for name, u, v in (("p", x, y), ("m", sa["exp_avg"], sb["exp_avg"]), ("v", sa["exp_avg_sq"], sb["exp_avg_sq"])):
            err = (u - v).abs().max().item()
            assert err <= 2e-6 * max(1e-3, u.abs().max().item()), (name, tuple(x.shape), err)
        if id(y) in shadows:
            assert torch.equal(shadows[id(y)], y.detach().to(torch.bfloat16))
    import copy
    ob.load_state_dict(copy.deepcopy(ob.state_dict()))               # a resume: new state tensors, the pointer table is stale
    assert not hip.still_valid()
    hip = HipAdamWStep(ob, shadows)
    assert hip.still_valid()
    pb[0].grad = torch.zeros_like(pb[0])
    assert not hip.still_valid()


@pytest.mark.gpu
def test_graph_replay_selftest_passes_in_this_process():
    """The runtime setting of vm_asr_amd/hip_env.py is in force (tests/conftest.py sets it before the GPU is initialised):
    a captured graph of ten multi-block reductions replays faithfully on new data.  Without it (ROCm 7.2 AQL-packet replay)
    memset nodes lose their order and reductions return stale results from the second replay on."""
    import os
    from vm_asr_amd.graph_step import replay_selftest
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
    assert replay_selftest(torch.device("cuda:0"))


@pytest.mark.gpu
def test_graph_replay_on_new_batches_matches_eager_training():
    """Three optimisation steps on three DIFFERENT batches: the replayed graphs give the same losses as the eager step from
    the same initial state (a graph replayed on the batch it was captured with cannot expose a replay that returns stale
    intermediate results — new data does).  _tiny_config has no stochastic depth, _gpu_trainer seeds the weights."""
    cfg = _tiny_config()
    batches = [[t.cuda() for t in _batch(cfg, 2, seed=s)] for s in range(4)]
    logs = {}
    for mode in ("eager", "graph"):
        tr = _gpu_trainer(cfg, amp=False, capturable=True)
        for m in tr.models.values():
            m.train()
        tr.train_step(*batches[0])
        if mode == "graph":
            assert tr.enable_graphs(batches[0], warmup=2)
        else:
            for _ in range(2):
                tr.train_step(*batches[0])                  # the eager warm-up steps graph capture runs
        out = []
        for b in batches[1:]:
            _, lg = tr.train_step(*b)
            out.append({k: float(v) for k, v in lg.items()})
        logs[mode] = out
    # first new batch: same weights on both sides (up to atomics order in the warm-up steps) -> tight; afterwards AdamW's
    # normalised updates amplify rounding-level gradient differences, the trajectories drift by ~1 % — a stale-result replay
    # is off by orders of magnitude (the MR-STFT term read 3 000 instead of 0.9)
    for i, (a, b) in enumerate(zip(logs["eager"], logs["graph"])):
        for k in a:
            assert abs(a[k] - b[k]) <= (2e-3 if i == 0 else 5e-2) * max(1.0, abs(a[k])), (i, k, a[k], b[k])


@pytest.mark.gpu
def test_fullsize_graph_replay_matches_eager_on_new_batches():
    """The headline workload (vm_asr_48k_MPD: dims 16, 48 kHz clips, MPD hidden 32, bf16 autocast over the generator) without
    stochastic depth: two optimisation steps on NEW batches through the replayed graphs give the losses of the eager step from
    the same state.  Exercises every fused discriminator path (stacked spectral-norm weights, direct first / last convolutions,
    bf16x3 triples, HIP AdamW) under replay, where a stale-result replay shows at once."""
    import bench
    cfg = bench.make_config("vm_asr_48k_MPD", 2)
    cfg.defrost()
    cfg.MODEL.VSSM.DROP_PATH_RATE = 0.0
    cfg.freeze()
    dev = torch.device("cuda", 0)
    batches = [bench.synth_batch(cfg, dev, s) for s in range(3)]
    logs = {}
    for mode in ("eager", "graph"):
        tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
        for m in tr.models.values():
            m.train()
        tr.train_step(*batches[0])
        if mode == "graph":
            assert tr.enable_graphs(batches[0], warmup=2)
        else:
            for _ in range(2):
                tr.train_step(*batches[0])
        out = []
        for b in batches[1:]:
            _, lg = tr.train_step(*b)
            out.append({k: float(v) for k, v in lg.items()})
        logs[mode] = out
        del tr
        torch.cuda.empty_cache()
    for i, (a, b) in enumerate(zip(logs["eager"], logs["graph"])):
        for k in a:
            assert abs(a[k] - b[k]) <= (5e-3 if i == 0 else 5e-2) * max(1.0, abs(a[k])), (i, k, a[k], b[k])
