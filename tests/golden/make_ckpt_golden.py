"""f3 fixture: a checkpoint directory written by the REFERENCE's own `_save_checkpoint` and the evaluation of three clips by
the reference's own modules — data only (tensors, optimiser state, config values, expected waves / metrics).

Run in the build container only:   python tests/golden/make_ckpt_golden.py          (needs /root/reference, see _refload.py)

Writes
  tests/golden/ckpt_ref/checkpoint-best-{G,mpd}.pth   by `BaseTrainer._save_checkpoint` (base/base_trainer.py:130-179),
        taken from the reference by AST and run on the reference's models + optimisers (`utils/optimizer.py:get_optimizer`)
        after one real optimiser step (so that exp_avg / exp_avg_sq / step are non-trivial); dict layout
        {name, epoch, state_dict, optimizer, monitor_best, config}; `config` pickles as yacs.config.CfgNode
  tests/golden/ckpt_eval.npz      three evaluation clips — one of segment length, one long (three overlapping segments), one long
        and ragged (its tail is covered by no segment and stays zero) — with the reference's outputs: the generator (reference
        modules on the CPU scan of kernels/selective_scan/test_selective_scan.py:287-367) applied as trainer/tester.py:89-131 does
        (`unfold_audio` / `fold_audio` of utils/post_processing.py:4-34), and the metrics of `_evaluate_batch` (:193-199) per clip.
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _refload import REF, load_reference, patch_ss2d_to_cpu  # noqa: E402

torch.set_num_threads(8)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _reference_save_checkpoint():
    """BaseTrainer._save_checkpoint as a plain function (the class itself needs tensorboard / logger packages)."""
    src = open(os.path.join(REF, "base/base_trainer.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "BaseTrainer"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_save_checkpoint"]
    assert len(fn) == 1
    g = dict(os=os, torch=torch)
    exec(compile(ast.Module(body=fn, type_ignores=[]), "base_trainer._save_checkpoint", "exec"), g)
    return g["_save_checkpoint"]


class _Log:
    def info(self, *a):
        pass


def eval_config():
    """The tiny evaluation config of tests/test_ckpt_fixture.py (kept in ONE place: imported from there)."""
    from test_ckpt_fixture import fixture_config
    return fixture_config()


def main():
    from vm_asr_amd.config import to_yacs, yacs_pickle_compat
    ns = load_reference()
    cfg = eval_config()
    v = cfg.MODEL.VSSM
    torch.manual_seed(cfg.SEED)
    gen = ns.model.DualStreamInteractiveMambaUNet(
        in_chans=v.IN_CHANS, patch_size=v.PATCH_SIZE, depths=list(v.DEPTHS), dims=v.DIMS, ssm_d_state=v.SSM_D_STATE,
        ssm_ratio=v.SSM_RATIO, ssm_dt_rank=("auto" if v.SSM_DT_RANK == "auto" else int(v.SSM_DT_RANK)),
        ssm_act_layer=v.SSM_ACT_LAYER, ssm_conv=v.SSM_CONV, ssm_conv_bias=v.SSM_CONV_BIAS, ssm_drop_rate=v.SSM_DROP_RATE,
        ssm_init=v.SSM_INIT, forward_type=v.SSM_FORWARDTYPE, mlp_ratio=v.MLP_RATIO, mlp_act_layer=v.MLP_ACT_LAYER,
        mlp_drop_rate=v.MLP_DROP_RATE, gmlp=v.GMLP, drop_path_rate=v.DROP_PATH_RATE, patch_norm=v.PATCH_NORM,
        norm_layer=v.NORM_LAYER, patchembed_version=v.PATCHEMBED, downsample_version=v.DOWNSAMPLE, upsample_version=v.UPSAMPLE,
        output_version=v.OUTPUT, concat_skip=v.CONCAT_SKIP, interact=v.INTERACT, n_fft=cfg.DATA.STFT.N_FFT,
        hop_length=cfg.DATA.STFT.HOP_LENGTH, win_length=cfg.DATA.STFT.WIN_LENGTH, spectro_scale=cfg.DATA.STFT.SCALE,
        low_freq_replacement=cfg.TRAIN.LOW_FREQ_REPLACEMENT)
    patch_ss2d_to_cpu(ns, gen)
    mpd = ns.discriminator.MultiPeriodDiscriminator(hidden=cfg.TRAIN.ADVERSARIAL.MPD_HIDDEN)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():          # de-symmetrise the streams, make zero / one initialised tensors non-trivial
        for p in gen.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=g))

    ropt = _load("ref_optimizer", os.path.join(REF, "utils/optimizer.py"))
    opts = {"generator": ropt.get_optimizer(cfg, gen, _Log()), "discriminator": ropt.get_optimizer(cfg, [mpd], _Log())}

    # one real training-shaped step so that the optimiser states exist: L1 to the target for G, LSGAN-like sum for D
    seg = int(int(cfg.DATA.SEGMENT * cfg.DATA.FLAC2WAV.SRC_SR) * cfg.DATA.TARGET_SR / cfg.DATA.FLAC2WAV.SRC_SR)
    hf = torch.full((1,), int((cfg.DATA.STFT.N_FFT // 2 + 1) * 8000 / cfg.DATA.TARGET_SR), dtype=torch.int64)
    wave = 0.1 * torch.randn(1, 1, seg, generator=g)
    tgt = 0.1 * torch.randn(1, 1, seg, generator=g)
    gen.train(); mpd.train()
    y = gen(wave, hf)
    (y - tgt).abs().mean().backward()
    opts["generator"].step()
    yr, yg, _, _ = mpd(tgt, y.detach())
    sum(((1 - a) ** 2).mean() + (b ** 2).mean() for a, b in zip(yr, yg)).backward()
    opts["discriminator"].step()

    out_dir = os.path.join(HERE, "ckpt_ref")
    os.makedirs(out_dir, exist_ok=True)
    save = _reference_save_checkpoint()
    with yacs_pickle_compat():
        fake_self = types.SimpleNamespace(models={"generator": gen, "mpd": mpd}, optimizer=opts, mnt_best=0.4321,
                                          config=to_yacs(cfg), log_dir=out_dir, logger=_Log())
        # the reference reads self.config.SAVE_EPOCH_FREQ as an attribute: the yacs node resolves it
        save(fake_self, 5, save_best=True)
    written = sorted(os.listdir(out_dir))
    assert written == ["checkpoint-best-G.pth", "checkpoint-best-mpd.pth", "checkpoint-latest-G.pth", "checkpoint-latest-mpd.pth"], written
    for f in written:            # the `latest` files hold the same dicts: not kept in the fixture (size)
        if "latest" in f:
            os.remove(os.path.join(out_dir, f))
    for f in sorted(os.listdir(out_dir)):
        print(f"  wrote ckpt_ref/{f}: {os.path.getsize(os.path.join(out_dir, f)) / 1024:.1f} KiB")

    # ---- evaluation of three clips as trainer/tester.py:89-131 does ------------------------------------------------
    post = _load("ref_post", os.path.join(REF, "utils/post_processing.py"))
    gen.eval()
    # Frame 0 of a reflect-padded STFT is an exactly real spectrum, so its phase is +-pi by FFT rounding noise — in the reference
    # too — and an untrained network spreads that coin flip over the whole clip.  The spectrogram the reference fed
    # to the network is therefore part of the fixture (per generator call, keyed by the call's first input samples); the tests
    # inject it and pin the STFT itself separately (tests/test_gpu_kernels.py, tests/test_oracle.py).
    spectra = []
    ref_mag_phase = gen._mag_phase

    def recording_mag_phase(x, *a, **k):
        mag, phase = ref_mag_phase(x, *a, **k)
        spectra.append((x[0, 0, :8].clone(), mag.clone(), phase.clone()))
        return mag, phase
    gen._mag_phase = recording_mag_phase
    overlap = cfg.TEST.OVERLAP
    step = seg - overlap
    lengths = [seg, seg + 2 * step, seg + 2 * step + 3 * cfg.DATA.STFT.HOP_LENGTH + 17]
    arrs = {"seg": np.array(seg), "overlap": np.array(overlap), "hf": hf.numpy()}
    metrics = (ns.metric.snr, ns.metric.lsd, ns.metric.lsd_hf, ns.metric.lsd_lf)
    sums = {m.__name__: 0.0 for m in metrics}
    with torch.no_grad():
        for i, T in enumerate(lengths):
            wi = 0.1 * torch.randn(1, 1, T, generator=g)
            wt = wi + 0.02 * torch.randn(1, 1, T, generator=g)
            if wi.size(2) <= seg:
                wo = gen(wi, hf)
            else:
                segments = post.unfold_audio(audio=wi, segment_length=seg, overlap=overlap)
                processed = torch.zeros_like(segments)
                for j in range(segments.size(2)):
                    processed[:, :, j] = gen(segments[:, :, j], hf)
                wo = post.fold_audio(processed, total_length=wi.size(2), segment_length=seg, overlap=overlap)
            arrs[f"in{i}"], arrs[f"tgt{i}"], arrs[f"out{i}"] = wi.numpy(), wt.numpy(), wo.numpy()
            for m in metrics:
                val = float(m(wo.squeeze(1), wt.squeeze(1), hf=hf))
                arrs[f"{m.__name__}{i}"] = np.array(val)
                sums[m.__name__] += val
            print(f"  clip {i}: T={T} " + " ".join(f"{m.__name__}={float(arrs[m.__name__ + str(i)]):.5f}" for m in metrics))
    for k, s in sums.items():
        arrs[f"mean_{k}"] = np.array(s / len(lengths))
    uniq = {}
    for key, mag, phase in spectra:          # (two calls per forward on the same input: model/model.py:1105,1217-1221)
        uniq.setdefault(tuple(key.tolist()), (key, mag, phase))
    for j, (key, mag, phase) in enumerate(uniq.values()):
        arrs[f"spec_key{j}"], arrs[f"spec_mag{j}"], arrs[f"spec_phase{j}"] = key.numpy(), mag.numpy(), phase.numpy()
    arrs["n_spec"] = np.array(len(uniq))
    np.savez_compressed(os.path.join(HERE, "ckpt_eval.npz"), **arrs)
    print(f"  wrote ckpt_eval.npz: {os.path.getsize(os.path.join(HERE, 'ckpt_eval.npz')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
