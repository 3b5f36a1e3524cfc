"""Host-logic tests of the training harness on CPU (oracle kernels in the operator hooks):
one train step of G+MPD runs, updates parameters, leaves the 129 never-used tensors untouched;
world_size-2 gloo DDP gives both ranks identical parameters equal to a single-process step on
the concatenated batch."""
import numpy as np
import os
import sys
import time

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
    from vm_asr_amd.linear import LP_ATTR, LPT_ATTR
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
                if p.dim() == 2:     # the transposed shadows (operands of the fused kernels' backward) follow too
                    assert torch.equal(getattr(p, LPT_ATTR), p.detach().to(torch.bfloat16).t())
    worst = max((results["1"][k] - results["0"][k]).abs().max().item() for k in results["1"])
    assert worst <= 2e-3, worst       # AdamW steps are lr-sized (1e-4..1e-3): same update direction everywhere


@pytest.mark.gpu
def test_shadows_follow_the_hip_adamw_kernel():
    """Capturable optimisers + flat gradient buffers: the one-launch HIP AdamW (csrc/adamw.hip) writes the bf16 shadow AND the
    transposed bf16 shadow of every weight it updates; after three steps both equal the bf16 rounding of the fp32 weights."""
    from vm_asr_amd.linear import LP_ATTR, LPT_ATTR
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    tr = _gpu_trainer(cfg, amp=True, capturable=True)
    for m in tr.models.values():
        m.train()
    assert tr.enable_graphs(batch, warmup=3)          # flat gradient buffers + the HIP AdamW table live in the graphed step
    before = {n: p.detach().clone() for n, p in tr.models["generator"].named_parameters()}
    for p in tr.models["generator"].parameters():     # (after the warm-up's restore the shadows match the restored weights)
        assert torch.equal(getattr(p, LP_ATTR), p.detach().to(torch.bfloat16))
    for _ in range(3):
        tr.train_step(*batch)
    torch.cuda.synchronize()
    assert getattr(tr, "_hip_adamw", None), "the HIP AdamW path did not engage"
    moved = n2d = 0
    for n, p in tr.models["generator"].named_parameters():
        if p.grad is None:
            continue
        moved += int(not torch.equal(p.detach(), before[n]))
        assert torch.equal(getattr(p, LP_ATTR), p.detach().to(torch.bfloat16)), n
        if p.dim() == 2:
            n2d += 1
            assert torch.equal(getattr(p, LPT_ATTR), p.detach().to(torch.bfloat16).t()), n
    assert moved > 0 and n2d > 0


@pytest.mark.gpu
def test_train_step_gpu_graph_matches_eager():
    """Replaying the captured HIP graphs trains like the eager step: same loss trajectory over four steps
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
            assert tr.enable_graphs(batch, warmup=3)     # warm-up steps are undone afterwards: both runs start from the same state
        for _ in range(4):
            out, logs = tr.train_step(*batch)
            h.append({k: float(v) for k, v in logs.items()})
        torch.cuda.synchronize()
        hist.append(h)
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


@pytest.mark.gpu
def test_two_stream_step_gives_the_same_gradients(monkeypatch):
    """The step on two streams (discriminator on a side stream beside the generator: trainer._two_streams) == the same step on
    one stream: same losses, same gradients of every parameter — eagerly and as one captured graph with a fork / join
    (replayed three times on changing inputs: a missing dependency between the branches would show as stale values)."""
    cfg = _tiny_config()
    batches = [[t.cuda() for t in _batch(cfg, 2, seed=s)] for s in (0, 1, 2)]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VMASR_TWO_STREAM", flag)
        tr = _gpu_trainer(cfg, amp=False)
        for m in tr.models.values():
            m.train()
        assert tr._two_streams() == (flag == "1")
        if flag == "1":      # not without data-parallel library GEMMs (stream-K GEMMs of two streams can stall the device: hip_env.py)
            with monkeypatch.context() as mp, pytest.warns(UserWarning, match="TENSILE_STREAMK_DATA_PARALLEL"):
                mp.delenv("TENSILE_STREAMK_DATA_PARALLEL")
                assert not tr._two_streams()
            assert tr._two_streams()
        res = []
        for b in batches:
            _, logs = tr._forward_backward(*b)
            torch.cuda.synchronize()
            grads = {f"{k}.{n}": p.grad.detach().clone() for k in ("generator", "mpd")
                     for n, p in tr.models[k].named_parameters() if p.grad is not None}
            res.append(({k: float(v) for k, v in logs.items()}, grads))
        out[flag] = res
    for (la, ga), (lb, gb) in zip(out["1"], out["0"]):
        for k in lb:
            assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]) + 1e-7, (k, la[k], lb[k])
        assert ga.keys() == gb.keys()
        for k in gb:
            scale = max(gb[k].abs().max().item(), 1e-8)
            assert (ga[k] - gb[k]).abs().max().item() <= 1e-4 * scale + 1e-7, (k, (ga[k] - gb[k]).abs().max().item(), scale)
    # captured: loss trajectories of the graphed two-stream step against the graphed one-stream step
    hist = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VMASR_TWO_STREAM", flag)
        tr = _gpu_trainer(cfg, amp=False, capturable=True)
        for m in tr.models.values():
            m.train()
        assert tr.enable_graphs(batches[0], warmup=3)
        assert (tr._graphed.graph_g is None) == (flag == "1")
        h = []
        for i in range(4):
            _, logs = tr.train_step(*batches[i % 3])
            h.append({k: float(v) for k, v in logs.items()})
        torch.cuda.synchronize()
        hist[flag] = h
    for a, b in zip(hist["1"], hist["0"]):
        for k, v in a.items():
            assert abs(v - b[k]) <= 0.02 * abs(v) + 1e-4, (k, v, b[k])


def test_generator_losses_in_parts_equal_the_whole():
    """_generator_losses(parts=...) — the two-stream step evaluates the waveform losses and the ones through the discriminator on
    different streams — returns the same terms, in the reference's order, as the one-call form (CPU, oracle STFT)."""
    from oracle.torch_backend import oracle_stft_patch
    cfg = _tiny_config()
    tr = _make_trainer(cfg)
    for m in tr.models.values():
        m.train()
    wave_in, wave_tgt, hf = _batch(cfg, 1)
    with oracle_stft_patch(), tr._mpd_weights_once():      # (one set of spectral-norm weights for all three calls, as within a step)
        torch.manual_seed(0)
        wave_out = tr.models["generator"](wave_in, hf)
        whole = tr._generator_losses(wave_out, wave_tgt)
        sig = tr._generator_losses(wave_out, wave_tgt, parts=("signal",))
        mpd = tr._generator_losses(wave_out, wave_tgt, parts=("mpd",))
    assert list(whole) == list(sig) + list(mpd) and set(sig).isdisjoint(mpd) and sig and mpd
    for k, v in whole.items():
        assert torch.allclose(v, {**sig, **mpd}[k], rtol=1e-6, atol=0), k
    assert not tr._two_streams()          # CPU: one stream


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
            assert tr.enable_graphs(batches[0], warmup=2)   # its warm-up steps are undone: same state as the eager trainer
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
            assert tr.enable_graphs(batches[0], warmup=2)   # (warm-up undone afterwards)
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


@pytest.mark.gpu
def test_generator_only_graph_with_the_phase_lane_matches_eager():
    """Generator-only training (vm_asr_48k: no discriminator): the captured step runs the generator's phase branch on a second
    HIP stream (model._lanes, trainer.enable_graphs sets phase_lane).  Two optimisation steps on NEW batches through the replayed
    graph give the eager step's losses from the same state, and the lane is really in the graph (VMASR_GEN_STREAMS=1 gives a graph
    with fewer concurrent branches: checked through the model's flag and the stream pool)."""
    import bench
    from vm_asr_amd import model as gen_model
    from vm_asr_amd.trainer import unwrap
    cfg = bench.make_config("vm_asr_48k", 3)
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
            gen_model._PHASE_STREAMS.clear()
            assert tr.enable_graphs(batches[0], warmup=2)
            assert unwrap(tr.models["generator"]).phase_lane and dev in gen_model._PHASE_STREAMS     # the lane was used in the capture
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


@pytest.mark.gpu
@pytest.mark.parametrize("lane", [False, True])
def test_gan_step_capture_keeps_the_faster_generator_layout(lane, monkeypatch):
    """GAN step on two streams: enable_graphs() captures the step with the generator on one stream AND with its phase branch on a second
    one, times both, and keeps the faster — always as the LAST capture (an earlier capture's graphs point into buffers the next capture's
    warm-up re-creates: replaying it faulted).  Both outcomes forced here: the kept step replays without fault and trains like the eager
    step from the same state (first new batch tight, then drifting by AdamW's normalised updates as in the tests above)."""
    import bench
    from vm_asr_amd.trainer import Trainer, unwrap
    cfg = bench.make_config("vm_asr_48k_MPD", 2)
    cfg.defrost()
    cfg.MODEL.VSSM.DROP_PATH_RATE = 0.0
    cfg.freeze()
    dev = torch.device("cuda", 0)
    batches = [bench.synth_batch(cfg, dev, s) for s in range(3)]
    monkeypatch.setattr(Trainer, "_pick_variant", staticmethod(lambda vs, ms: next(v for v in vs if v[0] == lane)))
    logs = {}
    for mode in ("eager", "graph"):
        tr = bench.build_trainer(cfg, dev, amp=True, capturable=True)
        for m in tr.models.values():
            m.train()
        tr.train_step(*batches[0])
        if mode == "graph":
            assert tr.enable_graphs(batches[0], warmup=2)
            assert "one_generator_stream_ms" in tr.graph_variants and len(tr.graph_variants) == 3      # + the lane at 5/8 and at 3/4 of the CUs
            assert unwrap(tr.models["generator"]).phase_lane == lane
        out = []
        for b in batches[1:]:
            _, lg = tr.train_step(*b)
            out.append({k: float(v) for k, v in lg.items()})
        torch.cuda.synchronize()
        logs[mode] = out
        del tr
        torch.cuda.empty_cache()
    for i, (a, b) in enumerate(zip(logs["eager"], logs["graph"])):
        for k in a:
            assert abs(a[k] - b[k]) <= (5e-3 if i == 0 else 5e-2) * max(1.0, abs(a[k])), (i, k, a[k], b[k])


@pytest.mark.gpu
def test_two_stream_step_at_batch_35_finishes():
    """Regression: with hipBLASLt's stream-K GEMMs (every gfx950 kernel of this stack is one) the two-stream step stopped the device
    for good at batch 35 — two concurrent GEMMs waiting for each other's partial tiles (vm_asr_amd/hip_env.py,
    profiles/r05_streamk_stall.md).  TENSILE_STREAMK_DATA_PARALLEL=1 (set by every entry point) removes the cross-workgroup
    wait.  Run in a child process under a timeout, so that a regression fails this test instead of hanging the suite."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1"
    # (GAN step: discriminator side stream + ONE generator stream — the layout that stalled, and one capture instead of three;
    #  generator only: its phase lane, the other pair of streams that stalled)
    for extra, streams in ((["--batch", "35"], "1"), (["--workload", "vm_asr_48k", "--batch", "35"], "auto")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                            "--no-extra-points", "--no-kernel-timing"] + extra, capture_output=True, text=True, timeout=240, cwd=root,
                           env={**os.environ, "VMASR_GEN_STREAMS": streams})
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line.get("value", 0) > 50, line


@pytest.mark.gpu
def test_fullsize_b4_train_step_pinned_to_cpu_oracle():
    """BASELINE configs[2] at FULL size and the yaml's batch (vm_asr_48k_MPD: B = 4, dims 16, 48 kHz clips, MPD hidden 32):
    one G+MPD train step — forward, MR-STFT + LSGAN + feature-matching losses, both backwards — on the GPU against the
    SAME trainer on the host with the oracle's C kernels in the generator's operator hooks (torch-CPU modules elsewhere),
    i.e. against the CPU restatement of the reference, not against another HIP run.  No stochastic depth (different RNG
    streams on the two devices).  Checked: the five loss values; the whole gradient vector of each model (relative L2 and
    cosine) and, as checksums, sum and L1 norm.
      fp32 GPU step (MPD GEMMs as bf16x3 triples, the default): losses to 1e-4 relative (north_star, fp32); whole gradient
      vector within 1e-3 (generator; measured 2.4e-5) / 1e-4 (MPD; measured 1.4e-6) relative L2 of the oracle run's;
      bf16-autocast GPU step (the benchmark's dtype): losses to 1e-2 relative (north_star, bf16), gradient cosine
      >= 0.995 (generator; measured 0.9993, relative L2 3.8e-2) / 0.9999 (MPD, which runs outside autocast)."""
    import bench
    from oracle.torch_backend import oracle_stft_patch, use_oracle
    from vm_asr_amd.trainer import unwrap
    cfg = bench.make_config("vm_asr_48k_MPD", 4)
    cfg.defrost()
    cfg.MODEL.VSSM.DROP_PATH_RATE = 0.0
    cfg.freeze()

    def grads(tr):
        out = {}
        for key in ("generator", "mpd"):
            gs = [p.grad.detach().double().flatten().cpu() for p in unwrap(tr.models[key]).parameters() if p.grad is not None]
            out[key] = torch.cat(gs)
        return out

    def step(device, amp):
        tr = bench.build_trainer(cfg, device, amp=amp)
        if device.type == "cpu":
            use_oracle(tr.models["generator"])
        for m in tr.models.values():
            m.train()
        batch = bench.synth_batch(cfg, device, 0)
        _, logs = tr._forward_backward(*batch)
        if device.type == "cuda":
            torch.cuda.synchronize()
        return {k: float(v) for k, v in logs.items()}, grads(tr)

    with oracle_stft_patch():
        l_cpu, g_cpu = step(torch.device("cpu"), False)
    assert g_cpu["generator"].numel() == 3_010_352 - 764_288 and g_cpu["mpd"].numel() == 41_092_165
    for amp, tol_loss, tol_g, tol_d, cos_g, cos_d in ((False, 1e-4, 1e-3, 1e-4, 0.999999, 0.999999), (True, 1e-2, None, None, 0.995, 0.9999)):
        l_gpu, g_gpu = step(torch.device("cuda", 0), amp)
        torch.cuda.empty_cache()
        assert set(l_gpu) == set(l_cpu)
        for k in l_cpu:
            assert abs(l_gpu[k] - l_cpu[k]) <= tol_loss * max(abs(l_cpu[k]), 1e-3), (amp, k, l_gpu[k], l_cpu[k])
        for key, tol, cmin in (("generator", tol_g, cos_g), ("mpd", tol_d, cos_d)):
            a, b = g_gpu[key], g_cpu[key]
            assert a.shape == b.shape and torch.isfinite(a).all()
            rel = ((a - b).norm() / b.norm()).item()
            cos = (torch.dot(a, b) / (a.norm() * b.norm())).item()
            print(f"amp={amp} {key}: losses ok; gradient rel L2 {rel:.2e} cosine {cos:.6f}; checksums sum {a.sum().item():.6e} vs {b.sum().item():.6e}, "
                  f"L1 {a.abs().sum().item():.6e} vs {b.abs().sum().item():.6e}")
            assert cos >= cmin, (amp, key, cos)
            if tol is not None:
                assert rel <= tol, (amp, key, rel)
                assert abs(a.abs().sum() - b.abs().sum()).item() <= tol * b.abs().sum().item(), (amp, key)


def _flat8_worker(rank, world, port, ret):
    """World-size-8 leg of the flat-buffer path: rank r trains on clip r of an 8-clip batch (torch.set_num_threads(1):
    eight processes share this container's cores)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)          # (the parent exports OMP_NUM_THREADS=1 for the children: libgomp reads it when it loads)
    from oracle.torch_backend import oracle_stft_patch
    from vm_asr_amd.trainer import init_distributed, unwrap
    init_distributed()
    cfg = _tiny_config(gan=True, batch=1)
    tr = _make_trainer(cfg, dp_mode="flat")
    for m in tr.models.values():
        m.train()
    full = _batch(cfg, world, seed=11)
    with oracle_stft_patch():
        _, logs = tr._forward_backward(*(t[rank:rank + 1] for t in full))
        local = {k: tr._flat[k].clone() if k in tr._flat else None for k in ("generator", "mpd")}
        if local["generator"] is None:      # flat buffers are set up lazily by the first reduce
            tr._setup_flat("generator", tr.optimizer_G); tr._setup_flat("mpd", tr.optimizer_D)
            local = {k: tr._flat[k].clone() for k in ("generator", "mpd")}
        tr._reduce_and_step()
    ret[rank] = dict(local={k: v for k, v in local.items()}, reduced={k: tr._flat[k].clone() for k in local},
                     sd={k: v.detach().clone() for k, v in unwrap(tr.models["generator"]).state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_gloo_world8():
    """configs[3]'s process layout (8 ranks, one clip shard each) on the CPU: after the step every rank holds the MEAN of
    the eight local flat gradient buffers (generator and MPD: one all-reduce each) and identical weights."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() + 29) % 2000
    procs = [ctx.Process(target=_flat8_worker, args=(r, 8, port, ret)) for r in range(8)]
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "GOMP_SPINCOUNT")}
    os.environ.update(OMP_NUM_THREADS="1", GOMP_SPINCOUNT="0")     # eight processes share this machine's cores
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = {r: ret[r] for r in range(8)}          # ONE transfer per rank (every access to the manager's dict pickles the whole entry)
    for key in ("generator", "mpd"):
        mean = torch.stack([res[r]["local"][key].double() for r in range(8)]).mean(0)
        assert float(mean.abs().max()) > 0
        for r in range(8):
            got = res[r]["reduced"][key].double()
            assert torch.allclose(got, mean, rtol=1e-5, atol=1e-7 * float(mean.abs().max())), (key, r)
    for r in range(1, 8):
        for k, v in res[0]["sd"].items():
            assert torch.equal(v, res[r]["sd"][k]), (r, k)


def test_epoch_decisions_gloo_world8():
    """The epoch-decision logic (mean of the ranks' epoch scalars -> best / early stop / NaN abort) with 8 ranks."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() + 41) % 2000
    procs = [ctx.Process(target=_sync_worker, args=(r, 8, port, ret)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret[(r, "exit")] == "nan-abort" and ret[(r, "epochs")] == 4 for r in range(8))
    for e in (1, 2, 3):
        assert all(ret[(r, e)] == ret[(0, e)] for r in range(8))
        assert ret[(0, e)][0] is True and abs(ret[(0, e)][1] - (5.0 - e / 8)) < 1e-12      # one of eight ranks improves by e


def test_checkpoint_optimizer_state_loads_into_a_plain_adamw(tmp_path):
    """The direction the reference needs: a torch.optim.AdamW built the way the reference builds it (main.py:168-201:
    float lr, not fused, not capturable) loads the optimiser state of a checkpoint written by this trainer — whose own
    optimisers may be capturable / fused with a device lr tensor — and steps.  The saved param_groups carry no runtime flag."""
    from oracle.torch_backend import oracle_stft_patch
    from vm_asr_amd.config import yacs_pickle_compat
    from vm_asr_amd.trainer import load_optimizer_state, portable_optimizer_state, set_weight_decay
    cfg = _tiny_config()
    cfg.defrost(); cfg.OUTPUT = str(tmp_path); cfg.freeze()
    with oracle_stft_patch():
        a = _resumable(cfg, tmp_path, "cpu")
        a.train_step(*_batch(cfg, 2))
        for o in (a.optimizer_G, a.optimizer_D):     # what a GPU run's optimisers look like (capturable needs a GPU to STEP)
            for g in o.param_groups:
                g.update(lr=torch.tensor(float(g["lr"])), capturable=True, fused=True, foreach=None)
        a._save_checkpoint(1, save_best=True)
        for o in (a.optimizer_G, a.optimizer_D):
            for g in o.param_groups:
                g.update(lr=float(g["lr"]), capturable=False, fused=None)
    with yacs_pickle_compat():
        ck = torch.load(os.path.join(tmp_path, "checkpoint-best-G.pth"), map_location="cpu", weights_only=False)
    for g in ck["optimizer"]["param_groups"]:
        assert isinstance(g["lr"], float) and g.get("capturable", False) is False and g.get("fused") is None and g.get("foreach") is None
    assert all(st["step"].device.type == "cpu" for st in ck["optimizer"]["state"].values())
    gen = a.models["generator"]
    plain = torch.optim.AdamW(set_weight_decay([gen]), lr=cfg.TRAIN.BASE_LR, eps=cfg.TRAIN.OPTIMIZER.EPS,
                              betas=tuple(cfg.TRAIN.OPTIMIZER.BETAS), weight_decay=cfg.TRAIN.WEIGHT_DECAY)
    plain.load_state_dict(ck["optimizer"])
    assert all(isinstance(g["lr"], float) and not g["capturable"] for g in plain.param_groups)
    before = [p.detach().clone() for p in gen.parameters()]
    plain.step()                                                       # gradients of the last train_step are still there
    assert any(not torch.equal(b, p) for b, p in zip(before, gen.parameters()))
    # and back: an optimiser of this package keeps ITS runtime flags when it loads a state written by another one
    for g in a.optimizer_G.param_groups:
        g["foreach"] = True
    sd = portable_optimizer_state(plain)
    assert all(g["foreach"] is None for g in sd["param_groups"])
    load_optimizer_state(a.optimizer_G, sd, torch.device("cpu"))
    assert all(g["foreach"] is True and g["capturable"] is False for g in a.optimizer_G.param_groups)


def test_resume_builds_from_the_cli_config_then_adopts_the_stored_one(tmp_path):
    """base/base_trainer.py:24-56,181-192 + utils/utils.py:141-145: everything is BUILT from the CLI config (so
    `--resume DIR --epochs N`, `--output`, `--batch-size` ... take effect: extending a finished run trains the extra epochs);
    only `self.config` is replaced by the generator checkpoint's afterwards, with RESUME_PATH re-pointed."""
    from oracle.torch_backend import oracle_stft_patch
    cfg = _tiny_config()
    cfg.defrost(); cfg.OUTPUT = str(tmp_path); cfg.TRAIN.EPOCHS = 7; cfg.PRINT_FREQ = 3; cfg.freeze()
    with oracle_stft_patch():
        a = _resumable(cfg, tmp_path, "cpu")
        a._save_checkpoint(7, save_best=True)          # a finished 7-epoch run
    out2 = tmp_path / "continued"
    cli = _tiny_config()
    cli.defrost(); cli.MODEL.RESUME_PATH = str(tmp_path); cli.TRAIN.EPOCHS = 9; cli.OUTPUT = str(out2); cli.PRINT_FREQ = 11; cli.freeze()
    b = _resumable(cli, tmp_path, "cpu")
    assert b.epochs == 9 and b.start_epoch == 8                      # two more epochs to train, not zero
    assert b.log_dir == str(out2)                                    # built from the CLI config
    assert b.config is not cli and b.config.TRAIN.EPOCHS == 7 and b.config.PRINT_FREQ == 3 and b.config.is_frozen()
    assert b.config.MODEL.RESUME_PATH == str(tmp_path) and b.checkpoint_config.TRAIN.EPOCHS == 7
    # evaluation keeps the CLI config (utils/utils.py:154-176 never touches it)
    ev = _tiny_config()
    ev.defrost(); ev.MODEL.RESUME_PATH = str(tmp_path); ev.EVAL_MODE = True; ev.TRAIN.EPOCHS = 9; ev.freeze()
    c = _resumable(ev, tmp_path, "cpu")
    assert c.config is ev


def test_resume_refuses_a_checkpoint_whose_step_config_differs(tmp_path, monkeypatch):
    """the step reads TRAIN.LOSSES / TRAIN.ADVERSARIAL / accumulation from the adopted checkpoint config while the loss
    modules, discriminators and buffers were built from the CLI config: a mismatch is refused (or, on request, logged and the
    built values kept) instead of silently training another loss set."""
    from oracle.torch_backend import oracle_stft_patch
    cfg = _tiny_config()
    cfg.defrost(); cfg.OUTPUT = str(tmp_path); cfg.freeze()
    with oracle_stft_patch():
        a = _resumable(cfg, tmp_path, "cpu")
        a._save_checkpoint(1, save_best=True)
    cli = _tiny_config()
    cli.defrost(); cli.MODEL.RESUME_PATH = str(tmp_path); cli.OUTPUT = str(tmp_path / "c")
    cli.TRAIN.ADVERSARIAL.FEATURE_LOSS_LAMBDA = cfg.TRAIN.ADVERSARIAL.FEATURE_LOSS_LAMBDA * 2 + 1; cli.freeze()
    with pytest.raises(ValueError, match="FEATURE_LOSS_LAMBDA"):
        _resumable(cli, tmp_path, "cpu")
    monkeypatch.setenv("VMASR_RESUME_CONFIG_MISMATCH", "warn")
    b = _resumable(cli, tmp_path, "cpu")
    assert b.config is not cli and b.config.TRAIN.ADVERSARIAL.FEATURE_LOSS_LAMBDA == cli.TRAIN.ADVERSARIAL.FEATURE_LOSS_LAMBDA
    assert b.config.is_frozen() and b.start_epoch == 2


@pytest.mark.gpu
def test_graph_warmup_leaves_the_training_state_untouched():
    """Trainer.enable_graphs runs real optimiser steps while warming up and capturing; with preserve_state (default) the
    weights, spectral-norm buffers, Adam moments and step counters afterwards equal what they were before — for a fresh
    trainer (no optimiser state yet: moments and steps are zero afterwards) and for one that has already stepped — and the
    first replayed step equals the eager step from the same state."""
    from vm_asr_amd.trainer import unwrap
    cfg = _tiny_config()
    batch = [t.cuda() for t in _batch(cfg, 2)]
    other = [t.cuda() for t in _batch(cfg, 2, seed=3)]

    def state(tr):
        out = {f"{k}.{n}": v.detach().clone() for k, m in tr.models.items() for n, v in unwrap(m).state_dict().items()}
        for name, o in (("G", tr.optimizer_G), ("D", tr.optimizer_D)):
            for i, p in enumerate(q for g in o.param_groups for q in g["params"]):
                for kk, v in o.state.get(p, {}).items():
                    out[f"opt{name}.{i}.{kk}"] = v.detach().clone().float()
        return out
    for presteps in (0, 2):
        tr = _gpu_trainer(cfg, amp=True, capturable=True)
        ref = _gpu_trainer(cfg, amp=True, capturable=True)
        for t in (tr, ref):
            for m in t.models.values():
                m.train()
            for _ in range(presteps):
                t.train_step(*batch)
        before = state(tr)
        assert tr.enable_graphs(other, warmup=2) and tr.global_step == presteps
        after = state(tr)
        for k, v in after.items():
            if k in before:
                assert torch.equal(v, before[k]), (presteps, k)
            else:
                assert float(v.abs().max()) == 0.0, (presteps, k)      # state created by the warm-up, reset to "fresh"
        _, lg = tr.train_step(*batch)
        _, le = ref.train_step(*batch)
        for k in le:
            assert abs(float(lg[k]) - float(le[k])) <= 2e-3 * max(1.0, abs(float(le[k]))), (presteps, k, float(lg[k]), float(le[k]))


def test_pick_variant_takes_the_fastest_capture():
    """host logic of Trainer.enable_graphs(): of the captured step layouts (generator on one stream; phase lane with 5/8 or 3/4 of the CUs
    for the discriminator's backward) the one with the smallest replay time stays."""
    from vm_asr_amd.trainer import Trainer
    vs = [(False, None), (True, 5 / 8), (True, 3 / 4)]
    assert Trainer._pick_variant(vs, [23.0, 21.7, 20.9]) == (True, 3 / 4)
    assert Trainer._pick_variant(vs, [39.6, 40.2, 39.9]) == (False, None)
    assert Trainer._pick_variant(vs[:1], [5.0]) == (False, None)


# ---- multi-GPU hardening without hardware: the failure paths of the in-graph RCCL route, on one rank ----------

def _one_rank_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
        return True
    return False


def test_rccl_comm_setup_failure_is_a_clean_error_on_every_rank(monkeypatch):
    """ncclCommInitRank failing (bad unique id, no fabric) must not leave one rank with a communicator and the others without: the ranks
    agree by all-reduce(MIN) and ALL raise RuntimeError — which GraphedTrainStep turns into 'collectives between the graphs'."""
    import ctypes
    import torch.distributed as dist
    from vm_asr_amd import rccl

    class FakeRccl:
        def __init__(self, init_rc):
            self.init_rc, self.destroyed = init_rc, 0

        def ncclGetUniqueId(self, p):
            return 0

        def ncclCommInitRank(self, comm_p, world, uid, rank):
            if self.init_rc == 0:
                ctypes.cast(comm_p, ctypes.POINTER(ctypes.c_void_p))[0] = 0x1234
            return self.init_rc

        def ncclCommDestroy(self, comm):
            self.destroyed += 1
            return 0

        def ncclGetErrorString(self, code):
            return b"unhandled system error"

    made = _one_rank_group()
    try:
        bad = FakeRccl(2)
        monkeypatch.setattr(rccl, "_rccl", lambda: bad)
        with pytest.raises(RuntimeError, match="communicator setup failed on this rank.*RCCL error 2"):
            rccl.RcclComm(torch.device("cpu"))
        good = FakeRccl(0)
        monkeypatch.setattr(rccl, "_rccl", lambda: good)
        comm = rccl.RcclComm(torch.device("cpu"))
        assert comm._comm.value == 0x1234
        comm.close()
        assert good.destroyed == 1 and comm._comm is None
        comm.close()                                   # idempotent
        assert good.destroyed == 1
    finally:
        if made:
            dist.destroy_process_group()


def test_collective_watchdog_fires_on_a_hung_event_and_not_on_a_finished_one():
    """rccl.CollectiveWatchdog: an armed event that never completes -> the timeout action (default: exit code 3); completed events -> nothing."""
    import threading
    from vm_asr_amd.rccl import CollectiveWatchdog

    class Ev:
        def __init__(self, done_after):
            self.n, self.done_after = 0, done_after

        def query(self):
            self.n += 1
            return self.n > self.done_after

    fired = []
    hit = threading.Event()

    def on_timeout(what, waited):
        fired.append((what, waited))
        hit.set()
    wd = CollectiveWatchdog(timeout_s=0.3, poll_s=0.01, on_timeout=on_timeout)
    wd.arm(Ev(3), "finishes")
    time.sleep(0.6)
    assert not fired
    wd.arm(Ev(10 ** 9), "hangs")
    assert hit.wait(5.0) and fired[0][0] == "hangs" and fired[0][1] >= 0.3


def test_collective_watchdog_default_action_exits_the_process_non_zero():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import time\nfrom vm_asr_amd.rccl import CollectiveWatchdog\n"
            "class Ev:\n    def query(self):\n        return False\n"
            "wd = CollectiveWatchdog(timeout_s=0.2, poll_s=0.01)\nwd.arm(Ev())\ntime.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "exiting with code 3" in r.stderr


def test_grad_wire_and_graph_collectives_defaults_are_the_conservative_ones(monkeypatch):
    """until a multi-GPU run has shown parity, gradients travel as fp32 on torch.distributed's communicator between the graphs;
    the bf16 wire and the in-graph RCCL route are opt-in."""
    from vm_asr_amd.trainer import Trainer
    for k in ("VMASR_GRAD_COMM", "VMASR_GRAPH_COLLECTIVES"):
        monkeypatch.delenv(k, raising=False)
    t = Trainer.__new__(Trainer)
    assert t.graph_collectives() is False
    monkeypatch.setenv("VMASR_GRAPH_COLLECTIVES", "1")
    assert t.graph_collectives() is True
    t._graph_collectives = False                       # a failed capture's retry: this trainer only, the environment untouched
    assert t.graph_collectives() is False and os.environ["VMASR_GRAPH_COLLECTIVES"] == "1"
