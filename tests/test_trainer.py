"""Host-logic tests of the training harness on CPU (oracle kernels in the operator hooks):
one train step of G+MPD runs, updates parameters, leaves the 129 never-used tensors untouched;
world_size-2 gloo DDP gives both ranks identical parameters equal to a single-process step on
the concatenated batch."""
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
