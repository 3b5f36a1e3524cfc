"""Thin command line over vm_asr_amd with the reference's flags (main.py:28-318): train / --eval on the HIP path.

    python main.py --cfg configs/vm_asr_48k_MPD.yaml --synthetic 64                 # train (one process per GPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --cfg ... --synthetic 512
    python main.py --cfg ... --eval --resume logs/.../ --tag 16000_48000 --synthetic 8

`--cfg` takes the reference's yaml files unchanged.  The VCTK pipeline (download, resampling, low-pass filters:
data_loader/data_loaders.py) is out of scope: clips come from `--synthetic N` (trainer.SyntheticVCTK, the
reference's batch contract) — a real dataset plugs in as any DataLoader yielding `(wave_in, wave_tgt, highcut, name,
pad)`.  `--inference` (file I/O, wav decoding) is not built; `--throughput` runs bench.py's measurement.
"""
import argparse
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # vm_asr_amd/hip_env.py: before the GPU is initialised
os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")   # vm_asr_amd/hip_env.py: stream-K GEMMs of two streams can stall the device
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_option(argv=None):
    p = argparse.ArgumentParser("VM-ASR (MI355X) training and evaluation")
    p.add_argument("--cfg", type=str, required=True, metavar="FILE")
    p.add_argument("--opts", default=None, nargs="+", help="KEY VALUE pairs")
    p.add_argument("--batch-size", type=int)
    p.add_argument("--input_sr", type=int)
    p.add_argument("--target_sr", type=int)
    p.add_argument("--resume", type=str)
    p.add_argument("--accumulation-steps", type=int)
    p.add_argument("--disable_amp", action="store_true")
    p.add_argument("--output", default="logs", type=str)
    p.add_argument("--tag", default=time.strftime("%Y%m%d%H%M%S", time.localtime()))
    p.add_argument("--eval", action="store_true")
    p.add_argument("--inference", action="store_true")
    p.add_argument("--input", type=str)
    p.add_argument("--throughput", action="store_true")
    p.add_argument("--synthetic", type=int, default=64, help="number of synthetic VCTK-shaped clips per epoch")
    p.add_argument("--epochs", type=int, help="override TRAIN.EPOCHS")
    p.add_argument("--no-graphs", action="store_true", help="run the step eagerly instead of replaying HIP graphs")
    args = p.parse_args(argv)
    from vm_asr_amd.config import get_config
    opts = list(args.opts or [])
    if args.target_sr:
        opts += ["DATA.TARGET_SR", args.target_sr]
    if args.epochs:
        opts += ["TRAIN.EPOCHS", args.epochs]
    config = get_config(args.cfg, opts, batch_size=args.batch_size, resume=args.resume, accumulation_steps=args.accumulation_steps,
                        disable_amp=args.disable_amp, output=args.output, tag=args.tag, eval=args.eval, inference=args.inference,
                        throughput=args.throughput, input_sr=args.input_sr)
    return args, config


def main(args, config):
    import vm_asr_amd
    from vm_asr_amd.trainer import (CosineWarmupScheduler, SyntheticVCTK, Trainer, _Logger, build_optimizer, default_metric_ftns,
                                    init_distributed)
    log = _Logger()
    if config.INFERENCE_MODE:
        raise SystemExit("--inference is not built (wav file I/O is out of scope, DESIGN.md §7); use --eval")
    if config.THROUGHPUT_MODE:
        import subprocess
        raise SystemExit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py")]))   # child process, exit with its code
    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("vm_asr_amd needs a GPU (there is no CPU path)")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    torch.manual_seed(config.SEED)
    models = vm_asr_amd.get_model(config)
    metrics = default_metric_ftns(config)
    sr_in = args.input_sr or (16000 if config.DATA.TARGET_SR == 48000 else 8000)
    if config.EVAL_MODE:
        from vm_asr_amd.tester import Tester
        ds = SyntheticVCTK(config, length=args.synthetic, sr_in=sr_in, seed=config.SEED + 10_000)
        loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
        res = Tester({"generator": models["generator"]}, metrics, config, device, loader, log).evaluate()
        print({k: round(v, 4) if isinstance(v, float) else v for k, v in res.items()})
        return
    ds = SyntheticVCTK(config, length=args.synthetic, sr_in=sr_in, seed=config.SEED + 1000 * rank)
    loader = torch.utils.data.DataLoader(ds, batch_size=config.DATA.BATCH_SIZE, shuffle=False, drop_last=True)
    gan = config.TRAIN.ADVERSARIAL.ENABLE
    graphs = not args.no_graphs and config.TRAIN.ACCUMULATION_STEPS == 1
    for m in models.values():
        if m is not None:
            m.to(device)
    opts = {"generator": build_optimizer(config, models["generator"], capturable=graphs)}
    if gan:
        opts["discriminator"] = build_optimizer(config, [models[d] for d in config.TRAIN.ADVERSARIAL.DISCRIMINATORS], capturable=graphs)
    steps = max(1, len(loader) // config.TRAIN.ACCUMULATION_STEPS)
    sched = {k: CosineWarmupScheduler(o, config.TRAIN.EPOCHS * steps, config.TRAIN.WARMUP_EPOCHS * steps, config.TRAIN.BASE_LR,
                                      config.TRAIN.MIN_LR, config.TRAIN.LR_SCHEDULER.WARMUP_PREFIX) for k, o in opts.items()}
    tr = Trainer(models, metrics, opts, config, device, loader, None, sched, amp=config.AMP_ENABLE, gan=gan, logger=log)
    if graphs:
        first = next(iter(loader))
        tr.enable_graphs(tr._to_dev(first))
    tr.train()


if __name__ == "__main__":
    main(*parse_option())
