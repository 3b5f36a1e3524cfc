"""Training harness: the reference's BaseTrainer / Trainer surface on one process per GPU.

Mirrors base/base_trainer.py:12-231 and trainer/trainer.py:10-495 (constructor arguments,
`train`, `_train_epoch`, `_valid_epoch`, `_save_checkpoint`, `_resume_checkpoint`, checkpoint
file names and dict layout), re-designed for MI355X data parallelism, which the reference does
not have (README.md:31):

  * one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI); the clip batch
    is sharded across ranks; every model's gradients are packed into ONE flat fp32 buffer and
    all-reduced by one call per model per step (dp_mode "flat": few large messages suit the
    point-to-point xGMI links, and the step stays HIP-graph capturable); torch DDP buckets
    overlapped with backward are the alternative dp_mode "ddp".  No data-path collective;
  * the 129 `layers_decoder_phase` tensors never receive a gradient in the reference either
    (model/model.py:1187) — they are left out of the flat buffer / the DDP reducer statically
    instead of paying `find_unused_parameters` every step;
  * the step is replayed as two HIP graphs (graph_step.py): forward + losses + both backwards +
    gradient packing — with the discriminator on a side stream BESIDE the generator (fork / join inside the graph:
    _two_streams, _forward_losses, _backward_two) —, then the fused AdamW steps + bf16 shadow-weight refresh;
  * the generator's adversarial/feature losses run through the discriminator with its
    parameters frozen, so the 164 MB MPD gradient is produced and all-reduced once per step
    (the reference fills and discards it during the G update, trainer/trainer.py:428-438);
  * bf16 autocast without GradScaler (the reference: fp16 + GradScaler, :106-107) over the SAME
    scope as the reference by default — the generator forward only (:138-139); losses and the
    discriminator run in fp32 (`amp_scope="step"` widens it); the scan still runs fp32 (forward
    type v5, model/vmamba.py:842-848);
  * no anomaly mode (:320) and no per-loss `.item()` syncs (:158-176): scalars are read every
    PRINT_FREQ steps.
"""
import math
import os
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from . import metric as metric_mod
from .loss import HiFiGANLoss, MultiResolutionSTFTLoss, mae_loss, mse_loss

__all__ = ["BaseTrainer", "Trainer", "SyntheticVCTK", "CosineWarmupScheduler", "build_optimizer",
           "set_weight_decay", "init_distributed", "unwrap"]


def init_distributed():
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # "nccl" is RCCL on ROCm.  VMASR_DIST_BACKEND=gloo lets the multi-process path be exercised
        # on a machine with fewer GPUs than ranks (ranks then share devices; test use only).
        backend = os.environ.get("VMASR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            if backend == "nccl":
                # bind the communicator to THIS rank's device up front (eager RCCL init on the right GPU: no lazy init on the
                # first collective, no "guessing device" warning, and graph capture never races a communicator bootstrap)
                kw["device_id"] = torch.device("cuda", local % torch.cuda.device_count())
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("VMASR_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def distributed_info(device):
    """What the process group actually is, gathered from every rank (bench.py records it so that an N-GPU line proves N ranks on
    N different devices): world size, backend, RCCL version, and per rank the device index / name / UUID / PCI bus id."""
    info = {"world_size": 1, "backend": None, "rccl_version": None}
    mine = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()}
    if device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        mine.update(device_index=device.index, name=p.name, uuid=str(getattr(p, "uuid", "")),
                    pci_bus_id=getattr(p, "pci_bus_id", None), pci_device_id=getattr(p, "pci_device_id", None))
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    if dist.is_available() and dist.is_initialized():
        info["world_size"], info["backend"] = dist.get_world_size(), dist.get_backend()
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, mine)
        info["ranks"] = ranks
        info["distinct_devices"] = len({(r.get("uuid") or r.get("pci_bus_id"), r.get("device_index")) for r in ranks})
    else:
        info["ranks"] = [mine]
        info["distinct_devices"] = 1
    return info


def unwrap(m):
    return m.module if isinstance(m, DDP) else m


class SyntheticVCTK(torch.utils.data.Dataset):
    """Synthetic clips with the reference's batch contract
    `(wave_in (1,T), wave_tgt (1,T), highcut int64, name, pad)` — CustomVCTK_092._load_sample
    (data_loader/data_loaders.py:490-513); T = int(SEGMENT * TARGET_SR) (:138-140);
    highcut = int((n_fft/2+1) * sr_in / sr_tgt) (:482-486).  Seeds follow SURVEY.md §8d."""

    def __init__(self, config, length=64, sr_in=16000, seed=123):
        self.T = int(config.DATA.SEGMENT * config.DATA.TARGET_SR)
        self.n = length
        self.seed = seed
        self.highcut = int((config.DATA.STFT.N_FFT // 2 + 1) * sr_in / config.DATA.TARGET_SR)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        tgt = 0.1 * torch.randn(1, self.T, generator=g)
        g2 = torch.Generator().manual_seed(self.seed + 1 + 7919 * (i + 1))
        inp = 0.1 * torch.randn(1, self.T, generator=g2)
        return inp, tgt, torch.tensor(self.highcut, dtype=torch.int64), f"synthetic_{i:06d}", 0


def set_weight_decay(models):
    """1-D tensors, biases and `_no_weight_decay` params get weight_decay 0 (utils/optimizer.py:53-77)."""
    decay, no_decay = [], []
    for model in models:
        for name, p in unwrap(model).named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if (p.ndim == 1 or name.endswith(".bias") or getattr(p, "_no_weight_decay", False))
             else decay).append(p)
    return [{"params": decay}, {"params": no_decay, "weight_decay": 0.0}]


def build_optimizer(config, models, capturable=False):
    if not isinstance(models, (list, tuple)):
        models = [models]
    groups = set_weight_decay(models)
    name = config.TRAIN.OPTIMIZER.NAME.lower()
    if name == "adamw":
        on_gpu = any(p.is_cuda for g in groups for p in g["params"])
        # capturable: the learning rate is a DEVICE tensor the fused kernel reads at run time, so the schedule keeps
        # working when the step is replayed from a HIP graph (a CPU tensor is read with .item() at capture and
        # frozen into the graph); Trainer.lr_to_device() re-establishes this after .to(device) / load_state_dict
        dev = next((p.device for g in groups for p in g["params"]), torch.device("cpu"))
        lr = torch.tensor(float(config.TRAIN.BASE_LR), device=dev) if capturable else config.TRAIN.BASE_LR
        # fused: one multi-tensor kernel per step instead of ~10 foreach passes over 44 M parameters
        extra = dict(fused=True) if (capturable and on_gpu and os.environ.get("VMASR_FUSED_ADAMW", "1") == "1") \
            else dict(foreach=True if capturable else None)
        return torch.optim.AdamW(groups, lr=lr, eps=config.TRAIN.OPTIMIZER.EPS,
                                 betas=tuple(config.TRAIN.OPTIMIZER.BETAS), weight_decay=config.TRAIN.WEIGHT_DECAY,
                                 capturable=capturable, **extra)
    if name == "sgd":
        return torch.optim.SGD(groups, lr=config.TRAIN.BASE_LR, momentum=config.TRAIN.OPTIMIZER.MOMENTUM,
                               nesterov=True, weight_decay=config.TRAIN.WEIGHT_DECAY)
    raise NotImplementedError(name)


class CosineWarmupScheduler:
    """Linear warm-up from MIN_LR then one cosine cycle to MIN_LR, stepped per update
    (`step_update`), like the timm scheduler the reference configures (utils/lr_scheduler.py:15-41)."""

    def __init__(self, optimizer, total_steps, warmup_steps, base_lr, min_lr, warmup_prefix=True):
        self.opt, self.base_lr, self.min_lr = optimizer, base_lr, min_lr
        self.warm = max(0, int(warmup_steps))
        self.t_initial = max(1, int(total_steps - self.warm if warmup_prefix else total_steps))
        self.prefix = warmup_prefix
        self.step_update(0)

    def lr_at(self, t):
        if t < self.warm:
            return self.min_lr + (self.base_lr - self.min_lr) * t / max(1, self.warm)
        tt = t - self.warm if self.prefix else t
        if tt >= self.t_initial:
            return self.min_lr
        return self.min_lr + 0.5 * (self.base_lr - self.min_lr) * (1 + math.cos(math.pi * tt / self.t_initial))

    def step_update(self, num_updates):
        lr, done = self.lr_at(num_updates), set()
        for g in self.opt.param_groups:
            if torch.is_tensor(g["lr"]):
                if g["lr"].data_ptr() not in done:      # capturable optimisers keep ONE lr tensor on the device
                    g["lr"].fill_(lr)
                    done.add(g["lr"].data_ptr())
            else:
                g["lr"] = lr


def lr_to_device(optimizer, device):
    """Capturable optimisers: every param group shares one lr tensor that lives on `device` (see build_optimizer).
    Needed after the models moved to the GPU and after `optimizer.load_state_dict` (which restores a CPU value)."""
    if optimizer is None or not optimizer.defaults.get("capturable", False):
        return
    shared = {}
    for g in optimizer.param_groups:
        v = float(g["lr"])
        if v not in shared:
            shared[v] = torch.tensor(v, dtype=torch.float32, device=device)
        g["lr"] = shared[v]


_RUNTIME_GROUP_KEYS = ("capturable", "fused", "foreach", "differentiable")


def portable_optimizer_state(optimizer):
    """optimizer.state_dict() in the form ANY torch.optim.AdamW accepts — in particular the reference's plain one
    (main.py:168-201 builds it non-fused, non-capturable, float lr): `lr` as a Python float, the runtime flags of this
    package's optimisers (capturable / fused / foreach) reset to their defaults, `step` counters as CPU tensors.
    `load_optimizer_state` below is the inverse for an optimiser built by build_optimizer()."""
    sd = optimizer.state_dict()
    groups = []
    for g in sd["param_groups"]:
        g = dict(g)
        g["lr"] = float(g["lr"])
        if "capturable" in g:
            g["capturable"] = False
        for k in ("fused", "foreach"):
            if k in g:
                g[k] = None
        groups.append(g)
    state = {pid: {k: (v.detach().cpu() if (k == "step" and torch.is_tensor(v)) else v) for k, v in st.items()}
             for pid, st in sd["state"].items()}
    return {"state": state, "param_groups": groups}


def load_optimizer_state(optimizer, state_dict, device):
    """optimizer.load_state_dict that keeps THIS optimiser's runtime flags (a checkpoint — ours or the reference's —
    carries its writer's), puts the step counters where a capturable optimiser needs them and re-creates the shared
    device learning-rate tensor."""
    keep = [{k: g[k] for k in _RUNTIME_GROUP_KEYS if k in g} for g in optimizer.param_groups]
    optimizer.load_state_dict(state_dict)
    for g, k in zip(optimizer.param_groups, keep):
        g.update(k)
    if optimizer.defaults.get("capturable", False):
        for st in optimizer.state.values():
            if "step" in st:
                st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).to(device)
    lr_to_device(optimizer, device)


class _Logger:
    """stderr only: stdout belongs to callers that print machine-readable results (bench.py)."""

    def info(self, m):
        import sys
        print(m, file=sys.stderr, flush=True)

    warning = info


class BaseTrainer:
    def __init__(self, models, metric_ftns, optimizer, config, logger=None, resume_now=True):
        self.config, self.logger = config, logger or _Logger()
        self.models, self.metric_ftns, self.optimizer = models, metric_ftns, optimizer
        self.epochs, self.monitor = config.TRAIN.EPOCHS, config.MONITOR
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if self.monitor == "off":
            self.mnt_mode, self.mnt_best = "off", 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ("min", "max")
            self.mnt_best = math.inf if self.mnt_mode == "min" else -math.inf
            self.early_stop = config.TRAIN.EARLY_STOPPING if config.TRAIN.EARLY_STOPPING > 0 else math.inf
        self.start_epoch = 1
        self.log_dir = config.OUTPUT
        self.epoch_log = {}
        self.do_validation = False
        # (Trainer resumes at the END of its own constructor, once the models sit on their device: the optimiser
        #  state then loads next to the parameters instead of staying on the host)
        if config.MODEL.RESUME_PATH is not None and resume_now:
            self._resume_checkpoint()

    def _train_epoch(self, epoch):
        raise NotImplementedError

    def _valid_epoch(self, epoch):
        raise NotImplementedError

    def train(self):
        not_improved = 0
        for epoch in range(self.start_epoch, self.epochs + 1):
            self._train_epoch(epoch)
            if self.do_validation:
                self._valid_epoch(epoch)
            self._sync_epoch_log()
            log = {"epoch": epoch, **self.epoch_log}
            self._log_epoch(log)
            best = False
            if self.mnt_mode != "off":
                if self.mnt_metric not in log:
                    self.logger.warning(f"Metric '{self.mnt_metric}' is not found. Monitoring is disabled.")
                    self.mnt_mode = "off"
                else:
                    v = log[self.mnt_metric]
                    improved = v <= self.mnt_best if self.mnt_mode == "min" else v >= self.mnt_best
                    if improved:
                        self.mnt_best, not_improved, best = v, 0, True
                    else:
                        not_improved += 1
                    if not_improved > self.early_stop:
                        self.logger.info(f"Validation performance didn't improve for {self.early_stop} epochs. Stop.")
                        break
            self._save_checkpoint(epoch, save_best=best)

    def _sync_epoch_log(self):
        """Multi-process runs: every rank continues with the MEAN of the ranks' epoch scalars, so that 'improved',
        early stopping and the NaN abort below are the same decision everywhere (a rank that left train() alone
        would strand the others in the next all-reduce)."""
        if self.world <= 1 or not dist.is_initialized():
            return
        keys = sorted(k for k, v in self.epoch_log.items() if isinstance(v, (int, float)))
        dev = getattr(self, "device", torch.device("cpu"))
        if dist.get_backend() == "gloo":
            dev = torch.device("cpu")
        t = torch.tensor([float(self.epoch_log[k]) for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(t)                      # NaN / inf on any rank propagates to all
        for k, v in zip(keys, (t / self.world).tolist()):
            self.epoch_log[k] = v

    def _save_checkpoint(self, epoch, save_best=False):
        """checkpoint-{latest,best,epoch-N}-{G|mpd}.pth with the reference's dict layout
        (base/base_trainer.py:130-179), `config` included as an object the reference's loader can `.defrost()`
        (utils/utils.py:141-145; config.to_yacs); rank 0 writes."""
        if self.rank != 0:
            return
        from .config import to_yacs, yacs_pickle_compat
        os.makedirs(self.log_dir, exist_ok=True)
        cfg_obj = to_yacs(self.config) if hasattr(self.config, "is_frozen") else self.config
        for key, model in self.models.items():
            if model is None:
                continue
            name = "G" if key == "generator" else key
            mtype = "generator" if key == "generator" else "discriminator"
            state = {"name": name, "epoch": epoch, "state_dict": unwrap(model).state_dict(),
                     "optimizer": portable_optimizer_state(self.optimizer[mtype]), "monitor_best": self.mnt_best,
                     "config": cfg_obj}
            with yacs_pickle_compat():
                torch.save(state, os.path.join(self.log_dir, f"checkpoint-latest-{name}.pth"))
                if self.config.SAVE_EPOCH_FREQ != -1 and epoch % self.config.SAVE_EPOCH_FREQ == 0:
                    torch.save(state, os.path.join(self.log_dir, f"checkpoint-epoch-{epoch}-{name}.pth"))
                if save_best:
                    torch.save(state, os.path.join(self.log_dir, f"checkpoint-best-{name}.pth"))

    def _resume_checkpoint(self):
        """Loads `checkpoint-best-*.pth` (falls back to latest) from MODEL.RESUME_PATH
        (utils/utils.py:112-178): state_dict strict, optimizer state, epoch, monitor_best; then, as the reference does AFTER
        everything has been constructed from the CLI config, `self.config` becomes the generator checkpoint's
        (base/base_trainer.py:181-192).  Reads the reference's own files too (their pickled yacs CfgNode resolves through
        config.yacs_pickle_compat)."""
        from .config import from_yacs, yacs_pickle_compat
        path = self.config.MODEL.RESUME_PATH
        for key, model in self.models.items():
            if model is None:
                continue
            name = "G" if key == "generator" else key
            mtype = "generator" if key == "generator" else "discriminator"
            for kind in ("best", "latest"):
                f = os.path.join(path, f"checkpoint-{kind}-{name}.pth")
                if os.path.exists(f):
                    with yacs_pickle_compat():
                        ck = torch.load(f, map_location="cpu", weights_only=False)
                    unwrap(model).load_state_dict(ck["state_dict"], strict=True)
                    if self.optimizer and mtype in self.optimizer and "optimizer" in ck:
                        # state tensors follow their parameter's device; this optimiser's runtime flags are kept
                        load_optimizer_state(self.optimizer[mtype], ck["optimizer"], next(unwrap(model).parameters()).device)
                    if key == "generator":
                        self.start_epoch = ck["epoch"] + 1
                        self.mnt_best = ck.get("monitor_best", self.mnt_best)
                        # base/base_trainer.py:181-192 + utils/utils.py:141-145: models, optimisers, schedulers, `self.epochs` and
                        # the log directory were built from the CLI config BEFORE this point and stay as built (so
                        # `--resume DIR --epochs N`, `--output`, `--batch-size`, moved data paths ... take effect); only
                        # `self.config` — what the step reads from now on (losses, print / save frequencies) — becomes the
                        # generator checkpoint's, with RESUME_PATH re-pointed.  Evaluation keeps the CLI config (:154-176).
                        self.checkpoint_config = from_yacs(ck["config"]) if ck.get("config") is not None else None
                        if self.checkpoint_config is not None and not getattr(self.config, "EVAL_MODE", False) \
                                and not getattr(self.config, "INFERENCE_MODE", False):
                            cfg = self.checkpoint_config.clone()
                            cfg.defrost()
                            cfg.MODEL.RESUME_PATH = path
                            cfg.freeze()
                            self._check_resumed_config(cfg)
                            self.config = cfg
                    self.logger.info(f"Resumed {name} from {f} (epoch {ck['epoch']})")
                    break
        if getattr(self, "_shadow_dst", None):
            self._refresh_shadows()

    # what a step READS from `self.config` while everything it runs on (loss modules, discriminators, gradient accumulation, flat
    # buffers, a captured graph) was BUILT from the CLI config: these must agree between the two, or the resumed run would either
    # raise in the middle of a step or silently train another loss set than the one it was built for
    _STEP_CONFIG_FIELDS = ("TRAIN.LOSSES", "TRAIN.ADVERSARIAL", "TRAIN.ACCUMULATION_STEPS")

    def _check_resumed_config(self, stored):
        """The reference swaps the config without looking (base/base_trainer.py:181-192).  Here a checkpoint whose step-level
        fields differ from the CLI config's is refused with the list of differences (VMASR_RESUME_CONFIG_MISMATCH=warn: logged,
        and the CLI config's values are kept for those fields, so that the step matches what was built)."""
        def get(cfg, dotted):
            for part in dotted.split("."):
                if cfg is None or part not in cfg:
                    return None
                cfg = cfg[part]
            return cfg

        def flat(node, prefix):
            if hasattr(node, "items"):
                out = {}
                for k, v in node.items():
                    out.update(flat(v, f"{prefix}.{k}"))
                return out
            return {prefix: list(node) if isinstance(node, (list, tuple)) else node}
        diffs = []
        for f in self._STEP_CONFIG_FIELDS:
            a, b = flat(get(self.config, f), f), flat(get(stored, f), f)
            diffs += [f"{k}: built {a.get(k)!r}, checkpoint {b.get(k)!r}" for k in sorted(set(a) | set(b)) if a.get(k) != b.get(k)]
        if not diffs:
            return
        msg = ("the checkpoint's config differs from the config this trainer was built from in fields the train step reads: "
               + "; ".join(diffs))
        if os.environ.get("VMASR_RESUME_CONFIG_MISMATCH", "raise") != "warn":
            raise ValueError(msg + " (pass the matching --cfg / --opts, or set VMASR_RESUME_CONFIG_MISMATCH=warn to keep the built values)")
        self.logger.warning(msg + " — keeping the values the trainer was built from")
        stored.defrost()
        for f in self._STEP_CONFIG_FIELDS:
            parent, leaf = f.rsplit(".", 1)
            src = get(self.config, f)
            if src is not None:
                get(stored, parent)[leaf] = src.clone() if hasattr(src, "clone") else src
        stored.freeze()

    def _log_epoch(self, logs):
        if self.rank == 0:
            self.logger.info(" | ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}" for k, v in logs.items()))
        bad = [k for k, v in logs.items() if isinstance(v, float) and (math.isnan(v) or math.isinf(v))]
        if bad:
            self.logger.warning(f"Found invalid values: {bad}. Terminating.")
            raise SystemExit(1)


class _StreamWork:
    """The join handle of a collective issued on a stream of ours: wait() = the current stream waits for its event."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)


class Trainer(BaseTrainer):
    def __init__(self, models, metric_ftns, optimizers, config, device, data_loader_train, data_loader_val=None,
                 lr_schedulers=None, amp=False, gan=False, logger=None, len_epoch=None, dp_mode="flat",
                 amp_scope="generator"):
        """dp_mode: "flat" = gradients live in one flat buffer per model and are all-reduced with ONE
        RCCL call per optimiser per step (also what makes the step HIP-graph capturable);
        "ddp" = torch DistributedDataParallel buckets overlapped with backward.
        amp_scope: what autocast covers when `amp` is on.  "generator" = the reference's scope
        (trainer/trainer.py:138-139: only the generator forward; the losses and the discriminator run
        outside autocast, i.e. in fp32); "step" = generator, losses and discriminator (bf16 MPD)."""
        super().__init__(models, metric_ftns, optimizers, config, logger, resume_now=False)
        self.dp_mode = dp_mode
        if amp_scope not in ("generator", "step"):
            raise ValueError(f"amp_scope='{amp_scope}'")
        self.amp_scope = amp_scope
        self._flat, self._flat_params, self._flat_views = {}, {}, {}
        self._acc = max(1, int(config.TRAIN.ACCUMULATION_STEPS))
        self._gather = self._acc == 1   # fresh grads are packed, not accumulated in place
        self._micro = 0                 # micro-batches seen (gradient accumulation)
        self._pending = []              # in-flight gradient all-reduces (async work handles)
        self._graphed = None
        self.time_reduces, self._reduce_events = False, []
        self._flat_lp = {}
        self.device = device[0] if isinstance(device, (tuple, list)) else device
        self.data_loader, self.data_loader_val = data_loader_train, data_loader_val
        self.len_epoch = len_epoch if len_epoch is not None else (len(data_loader_train) if data_loader_train is not None else 0)
        self.do_validation = data_loader_val is not None and config.DATA.VALID_SPLIT > 0.0
        self.amp, self.gan = amp, gan
        lr_schedulers = lr_schedulers or {}
        self.optimizer_G = optimizers["generator"]
        self.lr_scheduler_G = lr_schedulers.get("generator")
        if self.gan:
            self.optimizer_D = optimizers["discriminator"]
            self.lr_scheduler_D = lr_schedulers.get("discriminator")
        self._init_losses()
        for k, m in list(self.models.items()):
            if m is not None:
                self.models[k] = m.to(self.device)
        if self.dp_mode == "ddp":
            self._wrap_ddp()
        elif self.world > 1:
            self._broadcast_state()
        self.global_step = 0
        for opt in (self.optimizer_G, getattr(self, "optimizer_D", None)):
            lr_to_device(opt, self.device)
        self._make_shadows()
        if config.MODEL.RESUME_PATH is not None:
            self._resume_checkpoint()      # after .to(device): optimiser state lands next to the parameters

    # ---- distributed -----------------------------------------------------------------------
    def _wrap_ddp(self):
        if self.world <= 1:
            return
        gen = self.models["generator"]
        if not isinstance(gen, DDP):
            ignore = []
            if getattr(gen, "concat_skip", False) and getattr(gen, "interact", "dual") != "single":
                ignore = [n for n, _ in gen.named_parameters()
                          if n.startswith("layers_decoder_phase.") and not n.startswith("layers_decoder_phase.0.")]
            DDP._set_params_and_buffers_to_ignore_for_model(gen, ignore)
            ids = [self.device.index] if self.device.type == "cuda" else None
            # xGMI is point-to-point: a ring all-reduce is bound by one link, so prefer few large
            # buckets (whole generator = 12 MB in one bucket; MPD 164 MB in ~4)
            self.models["generator"] = DDP(gen, device_ids=ids, bucket_cap_mb=48, gradient_as_bucket_view=True,
                                           broadcast_buffers=False)
        if self.gan and self.models.get("mpd") is not None and not isinstance(self.models["mpd"], DDP):
            ids = [self.device.index] if self.device.type == "cuda" else None
            self.models["mpd"] = DDP(self.models["mpd"], device_ids=ids, bucket_cap_mb=48,
                                     gradient_as_bucket_view=True, broadcast_buffers=True)

    # ---- losses (trainer/trainer.py:88-96, 318-399) ----------------------------------------
    def _init_losses(self):
        a = self.config.TRAIN.ADVERSARIAL
        self.multi_resolution_stft = MultiResolutionSTFTLoss(factor_sc=a.STFT_LOSS.SC_FACTOR,
                                                             factor_mag=a.STFT_LOSS.MAG_FACTOR,
                                                             emphasize_high_freq=a.STFT_LOSS.EMPHASIZE_HIGH_FREQ)
        self.higi_gan_loss = HiFiGANLoss(gan_loss_type=a.GAN_LOSS_TYPE, gp_weight=a.GP_LAMBDA)

    def _get_stft_loss(self, wave_out, wave_target):
        sc, mag = self.multi_resolution_stft(wave_out.squeeze(1), wave_target.squeeze(1))
        return sc + mag

    def _generator_losses(self, wave_out, wave_target, fmap_real=None, fake_pass=None, parts=("signal", "mpd")):
        """parts: "signal" = the losses on the waveform itself, "mpd" = the ones through the period discriminator (the two-stream
        step evaluates them on different streams; the dict keeps the reference's order either way)."""
        cfg, out = self.config.TRAIN, {}
        wave_out = wave_out.float()
        if "signal" in parts:
            if "l1" in cfg.LOSSES.GEN:
                out["l1"] = mae_loss(wave_out, wave_target)
            if "l2" in cfg.LOSSES.GEN:
                out["l2"] = mse_loss(wave_out, wave_target)
            if "multi_resolution_stft" in cfg.LOSSES.GEN:
                out["multi_resolution_stft"] = self._get_stft_loss(wave_out, wave_target)
        if "mpd" in parts and self.gan and "mpd" in cfg.ADVERSARIAL.DISCRIMINATORS:
            mpd = unwrap(self.models["mpd"])  # weights used as constants: no MPD gradients, no DDP hooks
            if fake_pass is not None:      # shared fake pass (see _forward_backward): scores / features already there
                y_gen, fmap_gen = fake_pass
            else:
                if fmap_real is None:  # the reference recomputes the real-signal features here
                    with torch.no_grad():
                        _, fmap_real = mpd.forward_single(wave_target, detach_weights=True)
                y_gen, fmap_gen = mpd.forward_single(wave_out, detach_weights=True)
            if not cfg.ADVERSARIAL.ONLY_FEATURE_LOSS:
                out["adversarial_mpd"] = self.higi_gan_loss.generator_loss(y_gen)
            if not cfg.ADVERSARIAL.ONLY_ADVERSARIAL_LOSS:
                out["features_mpd"] = cfg.ADVERSARIAL.FEATURE_LOSS_LAMBDA * self.higi_gan_loss.feature_loss(fmap_real, fmap_gen)
        return out

    def _discriminator_losses(self, wave_out, wave_target):
        """-> (losses, detached real-signal feature maps for the generator's feature-matching loss:
        same discriminator weights, same input, so they equal what the reference recomputes)."""
        out, fmap_real = {}, None
        if self.gan and "mpd" in self.config.TRAIN.ADVERSARIAL.DISCRIMINATORS:
            fake = wave_out.detach().float()
            mpd = self.models["mpd"]
            y_real, y_gen, fr, _ = (mpd(wave_target, fake) if isinstance(mpd, DDP) else mpd.forward_pair(wave_target, fake))
            fmap_real = fr.detach() if hasattr(fr, "stacks") else [[f.detach() for f in fs] for fs in fr]
            d = self.higi_gan_loss.discriminator_loss(y_real, y_gen)
            if self.config.TRAIN.ADVERSARIAL.GAN_LOSS_TYPE == "wgan-gp":
                # double backward: runs the discriminator on plain (twice differentiable) torch operators
                d = d + self.higi_gan_loss.gradient_penalty(wave_target, fake, unwrap(self.models["mpd"]))
            out["mpd"] = d
        return out, fmap_real

    # ---- flat gradient buffers / single-call all-reduce -------------------------------------
    def _broadcast_state(self):
        """Rank 0's parameters and buffers everywhere (what DDP does at construction)."""
        for m in self.models.values():
            if m is None:
                continue
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=0)

    def _setup_flat(self, key, optimizer):
        """After a first backward: give every parameter that received a gradient a view into one
        flat fp32 buffer (never-used parameters keep grad None, as in the reference)."""
        model = unwrap(self.models[key])
        used = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
        flat = torch.zeros(sum(p.numel() for p in used), dtype=torch.float32, device=self.device)
        off = 0
        for p in used:
            n = p.numel()
            view = flat[off:off + n].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            off += n
        self._flat[key] = flat
        self._flat_params[key] = used
        self._flat_views[key] = [p.grad for p in used]
        return flat

    def _zero_grads(self, key, optimizer):
        """Before a backward.  With a flat buffer and no gradient accumulation the parameters' grads are
        dropped so that autograd hands over each gradient tensor as produced (no `grad += g` kernel per
        parameter: 640 launches a step); _gather_grads() then packs them into the flat buffer."""
        if key in self._flat:
            if self._gather:
                for p in self._flat_params[key]:
                    p.grad = None
            else:
                self._flat[key].zero_()
        else:
            optimizer.zero_grad(set_to_none=True)

    def _gather_grads(self, key):
        """After a backward: multi-tensor copy of the fresh gradients into the flat buffer's views, which
        become the parameters' .grad again (what the all-reduce and the fused AdamW read)."""
        if key not in self._flat or not self._gather:
            return
        params, views = self._flat_params[key], self._flat_views[key]
        src, dst = [], []
        for p, v in zip(params, views):
            if p.grad is None:
                v.zero_()
            else:
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        torch._foreach_copy_(dst, src)

    def _comm_dtype(self, key):
        """Wire dtype of `key`'s gradient all-reduce.  VMASR_GRAD_COMM: "fp32" (default: what the reference's DDP sends, so N-rank and
        1-rank training agree to fp32 rounding) | "mpd-bf16": the period discriminator's 164 MB buffer travels as bf16 (82 MB; the
        fp32 flat buffer stays the optimiser's input, AdamW's moments and the weights stay fp32 — DDP's bf16 compression hook,
        SURVEY.md 8(e)), the generator's 9 MB as fp32 | "bf16": both.  The 16-bit wire is a NUMERICS CHANGE (3e-4 ... 9e-4 on the
        losses of one step) and is opt-in until a multi-GPU run has shown loss parity with the fp32 wire.  RCCL only: gloo is
        the CPU test backend."""
        mode = os.environ.get("VMASR_GRAD_COMM", "fp32")
        if mode not in ("bf16", "mpd-bf16") or self.device.type != "cuda" or dist.get_backend() != "nccl":
            return torch.float32
        return torch.bfloat16 if (mode == "bf16" or key != "generator") else torch.float32

    def _reduce_grads(self, key, async_op=False):
        """ONE all-reduce (mean) per model per step over RCCL/xGMI (generator 9 MB fp32, MPD 82 MB bf16 / 164 MB fp32).
        async_op: the call returns at once and the collective runs on RCCL's own stream, ordered after the CURRENT stream's work so
        far — the caller overlaps it with further work and joins it with _wait_reduces() before the optimiser reads the gradients.
        Capturable (RCCL): inside a stream capture the collective becomes a branch of the graph."""
        emu = os.environ.get("VMASR_GRAD_COMM_EMULATE") if self.world == 1 else None
        if emu and key in self._flat and (emu == "bf16" or (emu == "mpd-bf16" and key != "generator")):
            # one rank, no wire: the 16-bit wire's ROUNDING applied to this rank's own gradient (tools/wire_dtype_run.py compares the
            # loss curves of 200 steps with and without it — the numerics question of the bf16 wire, answerable without a second GPU)
            flat = self._flat[key]
            flat.copy_(flat.to(torch.bfloat16))
            return
        if self.world > 1 and self.dp_mode == "flat":
            if key not in self._flat:
                self._setup_flat(key, None)
            flat = self._flat[key]
            if os.environ.get("VMASR_OVERLAP_REDUCE", "1") != "1":
                async_op = False                    # escape hatch: collectives strictly between the graphs, no overlap
            avg = dist.get_backend() == "nccl"      # RCCL averages in the collective; gloo has no AVG
            buf = flat
            if self._comm_dtype(key) != flat.dtype:
                lp = self._flat_lp.get(key)
                if lp is None:                      # (allocated before any capture: GraphedTrainStep's warm-up steps reduce too)
                    lp = self._flat_lp[key] = torch.empty_like(flat, dtype=self._comm_dtype(key))
                lp.copy_(flat)
                buf = lp
            if getattr(self, "_direct_rccl", None) is not None and (torch.cuda.is_current_stream_capturing()
                                                                     or os.environ.get("VMASR_RCCL_DIRECT", "0") == "1"):
                # RCCL's C API on a stream of its own, forked from the current one (vm_asr_amd/rccl.py: the process group's watchdog
                # cannot live with captured collectives): a branch of the graph being captured
                cs, cur = self._comm_stream(), torch.cuda.current_stream(self.device)
                cs.wait_stream(cur)
                self._direct_rccl.all_reduce_(buf, avg=True, stream=cs)
                done = torch.cuda.Event()
                done.record(cs)
                self._pending.append((_StreamWork(done, self.device), flat, buf, False))
                return
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=async_op)
            self._pending.append((work if async_op else None, flat, buf, not avg))
            if not async_op:
                self._wait_reduces()

    def _comm_stream(self):
        if getattr(self, "_comm_st", None) is None:
            self._comm_st = torch.cuda.Stream(self.device)
        return self._comm_st

    def enable_direct_rccl(self):
        """Create this trainer's own RCCL communicator (collective over the process group: every rank calls it, outside any capture)."""
        if getattr(self, "_direct_rccl", None) is None:
            from .rccl import RcclComm
            self._direct_rccl = RcclComm(self.device)
            self._comm_stream()
        return self._direct_rccl

    def _wait_reduces(self):
        """Join the pending collectives.  With `time_reduces` (bench.py, N > 1, collectives between the graphs) an event pair
        brackets the join on the compute stream: the time between them is what the collectives cost the step AFTER the overlap
        — the EXPOSED all-reduce time (`reduce_exposed_ms()`).  Inside a capture nothing is timed."""
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        timed = getattr(self, "time_reduces", False) and self.device.type == "cuda" and bool(self._pending) and not capturing
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, flat, buf, divide in self._pending:
            if work is not None:
                work.wait()                # the current stream waits for the collective (no host block on RCCL)
            if buf is not flat:
                flat.copy_(buf)            # bf16 wire buffer -> the optimiser's fp32 gradients
            if divide:
                flat.div_(self.world)
        if timed:
            e1.record()
            self._reduce_events.append((e0, e1))
        self._pending = []

    def reduce_exposed_ms(self):
        """Sum of the bracketed join times since the last call (synchronises)."""
        torch.cuda.synchronize(self.device)
        ms = sum(a.elapsed_time(b) for a, b in self._reduce_events)
        self._reduce_events = []
        return ms

    # ---- one optimisation step (the unit bench.py times) ------------------------------------
    def _set_deferred_reductions(self):
        """LayerNorm's dgamma / dbeta of a whole backward pass in one launch (layernorm.DEFER_REDUCE): only when no
        parameter gradient is read or accumulated in place before the pass ends — flat gradient buffers, no accumulation."""
        from . import layernorm
        layernorm.DEFER_REDUCE = (self.device.type == "cuda" and self.dp_mode == "flat" and self._gather
                                  and os.environ.get("VMASR_LN_DEFER", "1") == "1")
        layernorm.reset_uses()

    def _two_streams(self):
        """The step on two HIP streams: the period discriminator's chip-filling MFMA kernels on a side stream,
        the generator's ~1400 small launches on the main one.  Needs the shared fake pass (GPU, flat gradient buffers);
        off in deterministic mode (the ordered-accumulation tickets are per kernel, not per stream)."""
        from . import _lib, hip_env
        mode = os.environ.get("VMASR_TWO_STREAM", "1")
        # ("force": test hook — the EAGER two-stream step in deterministic mode: generator and discriminator share no ticketed kernel id,
        #  so their ordered tails cannot meet; tests/test_determinism.py uses it to pin the stream layout's arithmetic bit for bit)
        if mode not in ("1", "force") or (_lib.det_mode() and mode != "force"):     # (the mode switched on by env var OR through the library)
            return False
        if not self._share_fake_pass():
            return False
        if not hip_env.streamk_dp_in_force():
            # hipBLASLt's stream-K GEMMs of two concurrent streams can stop the device for good (vm_asr_amd/hip_env.py)
            if not getattr(self, "_warned_streamk", False):
                self._warned_streamk = True
                import warnings
                warnings.warn("TENSILE_STREAMK_DATA_PARALLEL=1 is not known to be in force: the train step stays on one HIP stream "
                              "(vm_asr_amd/hip_env.py sets it when the package is imported before the GPU is initialised)")
            return False
        return True

    def side_cu_limits(self, lanes=False):
        """(forward, backward) CU limits of the discriminator's convolution kernels while the two streams overlap, DERIVED from the
        device and the model instead of fixed numbers: the convolutions are compute-bound (time ~ 1 / CUs), the generator's small
        kernels are not, so the best split gives the side stream the share of the chip that balances the two phases' ends —
        3/8 of the CUs beside the generator's forward (D(real) has slack there: everything waits for the generator's output),
        5/8 beside its backward, 1/2 when the generator is the wide one (DIMS >= 32: its backward is 2x the work).  Measured flat in
        the batch (2, 4, 8) and within 1 % of the best point of every sweep: profiles/r05_side_cus_sweep_*.log (fwd 96 / bwd 160 of
        256 CUs: 172.1 clips/s at batch 4, 198.5 at batch 8; DIMS 32: fwd 96 / bwd 128: 153.7).  Rounded to 8 CUs (one per XCD).
        lanes: the generator's phase branch runs on a stream of its own (captured steps, model._lanes): its backward then ends sooner,
        and the discriminator's backward can be the longer side — 5/8 or 3/4 of the CUs, whichever capture enable_graphs() times faster
        (profiles/r05_side_cus_sweep_lanes.log: batch 4: 185.2 at 160, 193.0 at 192, 193.9 at 208, 184.7 at 240; batch 2: 139.1 / 143.2
        at 160 / 192; n_fft 2048 at batch 8: 176.9 / 171.1 — stays at 5/8; DIMS 32: 170.5 at 128 and at 160 — stays at 1/2)."""
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        dims = self.config.MODEL.VSSM.DIMS
        dims = dims[0] if isinstance(dims, (list, tuple)) else dims
        r8 = lambda v: max(8, int(round(v / 8.0)) * 8)      # noqa: E731
        batch = getattr(self, "_step_batch", None) or self.config.DATA.BATCH_SIZE    # (the batch actually built: enable_graphs() notes it)
        bwd = 1 / 2 if dims >= 32 else (3 / 4 if lanes and batch <= 4 else 5 / 8)
        share = getattr(self, "_lane_bwd_share", None)       # enable_graphs(): the share the timed captures preferred
        if lanes and share is not None:
            bwd = share
        return r8(cus * 3 / 8), r8(cus * bwd)

    def _side_cus(self, forward=False):
        """-> context: CUs the discriminator's convolution kernels may take while the generator's kernels run beside them
        (side_cu_limits(); soft by 24: csrc/convgemm.hip cg_grid).  Overrides: VMASR_SIDE_CUS (backward, and forward unless
        VMASR_SIDE_CUS_FWD is given too), 0 = no limit; VMASR_SIDE_CUS_MINC: backward, only layers at least that wide."""
        from . import convgemm
        lanes = (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
                 and bool(getattr(unwrap(self.models["generator"]), "phase_lane", False)))
        fwd_auto, bwd_auto = self.side_cu_limits(lanes)
        cus = os.environ.get("VMASR_SIDE_CUS", str(bwd_auto))
        if forward:
            cus = os.environ.get("VMASR_SIDE_CUS_FWD", str(fwd_auto) if "VMASR_SIDE_CUS" not in os.environ else cus)
        return convgemm.cu_limit(int(cus), 0 if forward else int(os.environ.get("VMASR_SIDE_CUS_MINC", "0")))

    def _mark(self, name, stream=None):
        """VMASR_PHASE_EVENTS=1 (dev aid, tools/phase_probe.py): the device clock written to a buffer when `stream` reaches this point of the step
        (vmasr_mark_time: a one-thread kernel, so it is captured into the step's graph and a replay yields the step's real timeline)."""
        if os.environ.get("VMASR_PHASE_EVENTS") != "1":
            return
        from . import _lib
        if not hasattr(self, "phase_marks"):
            self.phase_marks, self._phase_buf = {}, torch.zeros(64, dtype=torch.int64, device=self.device)
        slot = self.phase_marks.setdefault(name, len(self.phase_marks))
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib().vmasr_mark_time(self._phase_buf.data_ptr() + 8 * slot, st.cuda_stream), "mark_time")

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def _forward_losses(self, wave_input, wave_target, highcut):
        """generator forward -> discriminator + generator losses (one autograd graph; the D graph is built on the
        same D weights the G pass sees, so both backwards can run before either optimiser step).
        Returns the state the two backward phases consume."""
        acc = self._acc
        self._set_deferred_reductions()
        shared = self._share_fake_pass()
        two = shared and self._two_streams()
        dev_t = self.device.type
        step_amp = lambda: torch.autocast(device_type=dev_t, dtype=torch.bfloat16, enabled=self.amp and self.amp_scope == "step")  # noqa: E731
        d_losses = {}
        if two:
            # main stream: generator forward, losses on the waveform.   side stream: D(real) meanwhile, then D(fake) and
            # the losses on its outputs.  autograd runs every backward node on its forward's stream; the ORDER of the three
            # backward pieces is set by _backward_two() (the graph is cut at the waveform for that).
            import contextlib
            main, side = torch.cuda.current_stream(self.device), self._side_stream()
            mpd = unwrap(self.models["mpd"])
            with contextlib.ExitStack() as weights:
                self._mark("start", main)
                side.wait_stream(main)                       # fork: last step's optimiser, this step's inputs
                with torch.cuda.stream(side), step_amp(), self._side_cus(forward=True):       # (beside the generator's forward)
                    weights.enter_context(self._mpd_weights_once())
                    y_real, f_real = mpd.forward_single(wave_target)
                    self._mark("d_real_fwd_end", side)
                with torch.autocast(device_type=dev_t, dtype=torch.bfloat16, enabled=self.amp):
                    wave_out = self.models["generator"](wave_input, highcut)
                wave_f = wave_out.float()
                self._mark("g_fwd_end", main)
                # the autograd graph is CUT here: the discriminator sees a leaf, whose gradient _backward_two() hands to the
                # generator's backward — so the order in which the three backward pieces are issued is ours, not the engine's
                wave_d = wave_f.detach().requires_grad_(True)
                side.wait_stream(main)
                wave_d.record_stream(side)
                with torch.cuda.stream(side), step_amp():
                    y_fake, f_fake = mpd.forward_single(wave_d)
                    d_losses = {"mpd": self.higi_gan_loss.discriminator_loss(y_real, y_fake)}
                    f_real = f_real.detach() if hasattr(f_real, "stacks") else [[f.detach() for f in fs] for fs in f_real]
                    g_side = self._generator_losses(wave_d, wave_target, f_real, fake_pass=(y_fake, f_fake), parts=("mpd",))
                    total_d = sum(d_losses.values()) / acc
                    g_mpd = sum(g_side.values()) / acc if g_side else None
                    self._mark("d_fake_fwd_losses_end", side)
                with step_amp():
                    g_losses = self._generator_losses(wave_f, wave_target, parts=("signal",))
                    g_sig = sum(g_losses.values()) / acc if g_losses else None
            main.wait_stream(side)                           # (the side stream's scalar losses are summed for the log on the main one)
            for v in list(g_side.values()) + [total_d]:
                v.record_stream(main)
            g_losses.update(g_side)
            two = {"wave_f": wave_f, "wave_d": wave_d, "g_sig": g_sig, "g_mpd": g_mpd}
        else:
            with torch.autocast(device_type=dev_t, dtype=torch.bfloat16, enabled=self.amp):
                wave_out = self.models["generator"](wave_input, highcut)
            with step_amp():
                # D loss first, with the same D weights the G pass sees (reference order, trainer/trainer.py:369-399)
                with self._mpd_weights_once():
                    if shared:
                        # The reference runs D(fake) twice per step on identical values (detached for the D loss, attached
                        # for the G loss: trainer/trainer.py:376-384).  Here ONE pass over the fake signal carries both:
                        # the G loss is back-propagated through it for input gradients only, the D loss for weight
                        # gradients only (`backward(inputs=...)` + skip_weight_grads) — 1/8 fewer discriminator FLOPs.
                        mpd = unwrap(self.models["mpd"])
                        y_real, f_real = mpd.forward_single(wave_target)
                        y_fake, f_fake = mpd.forward_single(wave_out.float())
                        d_losses = {"mpd": self.higi_gan_loss.discriminator_loss(y_real, y_fake)}
                        f_real = f_real.detach() if hasattr(f_real, "stacks") else [[f.detach() for f in fs] for fs in f_real]
                        g_losses = self._generator_losses(wave_out, wave_target, f_real, fake_pass=(y_fake, f_fake))
                    else:
                        d_losses, fmap_real = self._discriminator_losses(wave_out, wave_target)
                        g_losses = self._generator_losses(wave_out, wave_target, fmap_real)
            total_d = sum(d_losses.values()) / acc if self.gan else None
        with torch.set_grad_enabled(not two):      # (two streams: the pieces above carry the backward, this sum is the log's)
            total_g = sum(g_losses.values()) / acc
        logs = {"total_loss": total_g.detach()}
        logs.update({f"generator/{k}": v.detach() for k, v in g_losses.items()})
        if self.gan:
            logs["total_disc_loss"] = total_d.detach()
        return {"wave_out": wave_out, "total_g": total_g, "total_d": total_d, "shared": shared, "two": two, "logs": logs}

    def _backward_d(self, st, zero=True):
        """Backward of the discriminator loss (weight gradients of the MPD) + packing into its flat buffer.
        Runs BEFORE the generator's backward so that the 164 MB MPD all-reduce can overlap the latter."""
        if not self.gan:
            return
        if zero:
            self._zero_grads("mpd", self.optimizer_D)
        if st["shared"]:
            st["total_d"].backward(inputs=self._grad_targets("mpd"), retain_graph=True)
        else:
            st["total_d"].backward()
        self._gather_grads("mpd")

    def _backward_g(self, st, zero=True):
        """Backward of the generator loss — through the (shared) discriminator pass for input gradients only."""
        from . import layernorm
        try:
            if zero:
                self._zero_grads("generator", self.optimizer_G)
            if st["shared"]:
                from .discriminator import skip_weight_grads
                with skip_weight_grads():
                    st["total_g"].backward(inputs=self._grad_targets("generator"))
            else:   # the generator's pass saw the discriminator weights as constants (detach_weights)
                st["total_g"].backward()
        finally:
            # the step's last backward is over (or failed): deferral is a property of THIS step's passes, not of the process —
            # any other backward (another trainer, a tester, user code) must not inherit it; an aborted pass' queue is dropped
            layernorm.DEFER_REDUCE = False
            layernorm.reset_uses()
        self._gather_grads("generator")

    def _backward_two(self, st, zero=True, reduce=False):
        """The backward passes of the two-stream step, issued in the order that lets them overlap (capture order is replay
        enqueue order):  side: G losses -> through the discriminator -> d/d(wave)  [input gradients only]
                         side: D loss -> discriminator weight gradients, packed            } beside each other
                         main: d/d(wave) + the waveform losses -> generator, packed        }
        then the main stream joins the side stream."""
        from . import layernorm
        from .discriminator import scores_only, skip_weight_grads
        tw = st["two"]
        main, side = torch.cuda.current_stream(self.device), self._side_stream()
        try:
            with torch.cuda.stream(side):
                if tw["g_mpd"] is not None:
                    with skip_weight_grads():
                        tw["g_mpd"].backward(inputs=[tw["wave_d"]], retain_graph=True)
                handed = torch.cuda.Event()
                handed.record(side)
                self._mark("g_dgrad_end", side)
                # A dependency on the main stream that costs nothing (everything main has issued so far — the generator's forward, the
                # waveform losses — ended before the discriminator's forward did) but places the D loss' backward BEHIND the generator's
                # forward in the captured graph's topology.  The runtime maps the branches of an instantiated graph to its own streams by
                # that topology: without the edge, the generator's phase branch (model._lanes) shares a hardware queue with this
                # stream's long kernels in the backward and waits behind them (profiles/r05_lane_trace_gen2.log: 147 clips/s);
                # with it the discriminator keeps a queue to itself (r05_lane_trace_gen2_edge.log: 186).
                side.wait_stream(main)
                if zero:
                    self._zero_grads("mpd", self.optimizer_D)
                with self._side_cus(), scores_only():                             # (beside the generator's backward)
                    st["total_d"].backward(inputs=self._grad_targets("mpd"))
                self._gather_grads("mpd")
                self._mark("d_bwd_end", side)
                if reduce and self.gan:
                    # the MPD all-reduce starts HERE, ordered after the side stream: it runs on RCCL's stream beside whatever is
                    # left of the generator's backward on the main stream (and, captured, as a third branch of the step's graph)
                    self._reduce_grads("mpd", async_op=True)
            main.wait_event(handed)
            if zero:
                self._zero_grads("generator", self.optimizer_G)
            roots, seeds = [], []
            if tw["g_sig"] is not None:
                roots.append(tw["g_sig"])
                seeds.append(None)
            gw = tw["wave_d"].grad
            if gw is not None:
                gw.record_stream(main)
                roots.append(tw["wave_f"])
                seeds.append(gw)
            torch.autograd.backward(roots, seeds, inputs=self._grad_targets("generator"))
        finally:
            layernorm.DEFER_REDUCE = False
            layernorm.reset_uses()
        self._gather_grads("generator")
        self._mark("g_bwd_end", main)
        main.wait_stream(side)
        self._mark("join", main)

    def _backward_both(self, st, zero=True, reduce=False):
        """Both backward passes in the order the step's stream layout wants; `reduce`: start each model's gradient all-reduce as
        soon as its flat buffer is packed.
        One stream : D loss, [the MPD all-reduce starts: it overlaps the whole generator backward], G loss.
        Two streams: _backward_two() — the MPD all-reduce is issued on the SIDE stream right behind the D loss' backward."""
        if st.get("two"):
            self._backward_two(st, zero, reduce)
            return
        self._backward_d(st, zero)
        if reduce and self.gan:
            self._reduce_grads("mpd", async_op=True)
        self._backward_g(st, zero)

    def _forward_backward(self, wave_input, wave_target, highcut, zero=True):
        """forward -> losses -> both backward passes (no collectives, no optimiser)."""
        st = self._forward_losses(wave_input, wave_target, highcut)
        self._backward_both(st, zero)
        return st["wave_out"].detach(), st["logs"]

    def _share_fake_pass(self):
        """One discriminator pass over the generated signal for both losses: GPU, flat gradient buffers (no DDP
        hooks), LSGAN / WGAN without gradient penalty, and the layer-synchronous discriminator path."""
        if not (self.gan and self.device.type == "cuda" and self.dp_mode == "flat"):
            return False
        if self.config.TRAIN.ADVERSARIAL.GAN_LOSS_TYPE == "wgan-gp" or "mpd" not in self.config.TRAIN.ADVERSARIAL.DISCRIMINATORS:
            return False
        return os.environ.get("VMASR_SHARE_FAKE_PASS", "1") == "1" and hasattr(unwrap(self.models["mpd"]), "_forward_batched")

    def _grad_targets(self, key):
        if key in self._flat_params:
            return self._flat_params[key]
        return [p for p in unwrap(self.models[key]).parameters() if p.requires_grad]

    def _mpd_weights_once(self):
        """The reference evaluates every spectrally-normalised MPD weight four times per step (real and
        fake in the D pass, real and fake again in the G pass: trainer/trainer.py:376-384, each
        model/discriminator.py:129-147 call running every discriminator on both signals), one power
        iteration each, although the weights do not change in between.  Here the four power iterations run
        back to back at the first use (same u/v state after the step) and the normalised weight is computed
        once and reused (torch.nn.utils.parametrize.cached): all passes see the sigma of the fourth
        iteration instead of the first ... fourth — a difference that vanishes as u, v converge."""
        import contextlib
        from torch.nn.utils import parametrize
        mpd = unwrap(self.models.get("mpd")) if self.gan else None
        if mpd is None or not mpd.training or not hasattr(mpd, "spectral_norms"):
            return contextlib.nullcontext()
        sns = mpd.spectral_norms()

        @contextlib.contextmanager
        def ctx():
            saved = [m.n_power_iterations for m in sns]
            n_it = 4 * saved[0] if saved else 4     # the reference's four D forwards per step
            batched = hasattr(mpd, "power_iterate_all") and mpd.power_iterate_all(n_it, with_sigma=True)   # 18 launches for all 30 weights
            for m in sns:
                m.n_power_iterations = 0 if batched else n_it
            try:
                with parametrize.cached(), (mpd.frozen_weights() if hasattr(mpd, "frozen_weights") else contextlib.nullcontext()):
                    yield
            finally:
                for m, n in zip(sns, saved):
                    m.n_power_iterations = n
                if batched:
                    mpd.clear_sigmas()
        return ctx()

    def _optimizer_steps(self):
        """AdamW for G (and D), then refresh the low-precision shadow weights (what graph B captures).
        With flat gradient buffers and capturable AdamW the update of each model is ONE launch of the HIP library
        (fused_adamw.HipAdamWStep: same state tensors, same arithmetic, bf16 shadows written in the same pass)."""
        fused = self._hip_adamw_steps()
        if fused is not None:
            for f in fused:       # (the kernel also writes the bf16 shadow of every parameter it updates;
                f.step()          #  parameters without a gradient do not change, their shadows stay valid)
            return
        self.optimizer_G.step()
        if self.gan:
            self.optimizer_D.step()
        self._refresh_shadows()

    def _hip_adamw_steps(self):
        """The HipAdamWStep objects of this trainer's optimisers, built once the optimiser states exist (after the first
        torch step) and the gradients live in the flat buffers; None -> use optimizer.step()."""
        if self.device.type != "cuda" or self.dp_mode != "flat" or os.environ.get("VMASR_HIP_ADAMW", "1") != "1":
            return None
        cur = getattr(self, "_hip_adamw", None)
        if cur is not None and all(f.still_valid() for f in cur):
            return cur
        if getattr(self, "_hip_adamw_failed", False):
            return None
        from .fused_adamw import HipAdamWStep
        opts = [("generator", self.optimizer_G)] + ([("mpd", self.optimizer_D)] if self.gan else [])
        if any(k not in self._flat for k, _ in opts):
            return None
        shadows = {id(src): dst for src, dst in zip(self._shadow_params, self._shadow_dst)}
        try:
            built = [HipAdamWStep(o, shadows, getattr(self, "_shadow_t", None)) for _, o in opts]
        except ValueError as e:
            if "not initialised" in str(e):
                return None                       # first step: torch creates the state, the next call builds the table
            self._hip_adamw_failed = True
            return None
        self._hip_adamw = built
        return built

    def _reduce_and_step(self):
        if self.gan:
            self._reduce_grads("mpd", async_op=True)
        self._reduce_grads("generator", async_op=True)
        self._wait_reduces()
        self._optimizer_steps()

    # ---- low-precision shadow weights ----------------------------------------------------------
    def _make_shadows(self):
        """Under AMP on the GPU every parameter gets a bf16 shadow copy (attribute linear.LP_ATTR) that
        vm_asr_amd.linear uses instead of casting the fp32 weight in every forward (300 cast kernels
        per step); the copies are refreshed by one multi-tensor copy after the optimiser steps."""
        from .linear import LP_ATTR, LPT_ATTR
        self._shadow_src, self._shadow_dst, self._shadow_params = [], [], []
        self._shadow_t, self._shadow_t_view, self._shadow_t_src = {}, [], []      # transposed shadows of the 2-D weights
        if not (self.amp and self.device.type == "cuda") or os.environ.get("VMASR_LP_SHADOWS", "1") != "1":
            return
        for key, m in self.models.items():
            if m is None:
                continue
            if key != "generator" and self.amp_scope != "step":
                continue          # the discriminator runs outside autocast in the reference's AMP scope: no bf16 reader
            for p in unwrap(m).parameters():
                if p.requires_grad and p.dtype == torch.float32:
                    lp = p.detach().to(torch.bfloat16)
                    setattr(p, LP_ATTR, lp)
                    self._shadow_src.append(p.detach())
                    self._shadow_dst.append(lp)
                    self._shadow_params.append(p)
                    if p.dim() == 2 and os.environ.get("VMASR_LP_SHADOWS_T", "1") == "1":
                        lpt = lp.t().contiguous()          # refreshed with lp: by the AdamW kernel, or _refresh_shadows
                        setattr(p, LPT_ATTR, lpt)
                        self._shadow_t[id(p)] = lpt
                        self._shadow_t_view.append(lpt.t())
                        self._shadow_t_src.append(p.detach())

    def _refresh_shadows(self):
        """bf16 shadows <- fp32 parameters (one multi-tensor copy): after optimizer.step() / load_state_dict."""
        if self._shadow_dst:
            torch._foreach_copy_(self._shadow_dst, self._shadow_src)
        if getattr(self, "_shadow_t_view", None):
            torch._foreach_copy_(self._shadow_t_view, self._shadow_t_src)

    def train_step(self, wave_input, wave_target, highcut):
        """One optimisation step of G (and D); returns (wave_out, dict of loss tensors)."""
        if self._graphed is not None:
            out = self._graphed(wave_input, wave_target, highcut)
            self.global_step += 1
            return out
        # gradient accumulation (TRAIN.ACCUMULATION_STEPS): gradients of `acc` consecutive micro-batches (each
        # loss / acc) are summed in place; collectives and optimisers run on the last one.  The reference divides
        # the loss the same way but calls backward only on every acc-th batch (trainer/trainer.py:146-156), i.e. it
        # drops the other batches; accumulating them is what the option's name promises.
        first, last = self._micro % self._acc == 0, (self._micro + 1) % self._acc == 0
        self._micro += 1
        st = self._forward_losses(wave_input, wave_target, highcut)
        self._backward_both(st, zero=first, reduce=last)
        if last:
            self._reduce_grads("generator", async_op=True)
            self._wait_reduces()
            self._optimizer_steps()
            self.global_step += 1
        return st["wave_out"].detach(), st["logs"]

    def _snapshot_training_state(self):
        """Copies of everything a training step changes: parameters and buffers (spectral-norm u / v) of every model and
        the optimisers' per-parameter state.  (None for a parameter whose optimiser state does not exist yet.)"""
        tensors = []
        for m in self.models.values():
            if m is not None:
                tensors += [(t, t.detach().clone()) for t in list(unwrap(m).parameters()) + list(unwrap(m).buffers())]
        opt = []
        for o in [self.optimizer_G] + ([self.optimizer_D] if self.gan else []):
            for g in o.param_groups:
                for p in g["params"]:
                    st = o.state.get(p)
                    opt.append((o, p, None if not st else {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}))
        return tensors, opt, self.global_step, self._micro

    @torch.no_grad()
    def _restore_training_state(self, snap):
        """In place (copy_ / zero_): the pointers captured into the HIP graphs and the AdamW table stay valid."""
        tensors, opt, self.global_step, self._micro = snap
        for t, c in tensors:
            t.copy_(c)
        for o, p, saved in opt:
            st = o.state.get(p)
            if not st:
                continue
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.zero_() if saved is None else v.copy_(saved[k])     # no state before == a fresh optimiser: m = v = step = 0
        self._refresh_shadows()

    def enable_graphs(self, example_batch, warmup=3, preserve_state=True):
        """Capture forward+backward (one HIP graph) and the optimiser steps (a second one); the
        gradient all-reduce stays an eager RCCL call between them.  Returns True if capture worked;
        on any failure the trainer stays in eager mode.

        The warm-up iterations and the capture run REAL optimiser steps on `example_batch` (the caching allocator, the
        flat gradient buffers and the optimiser states have to exist before anything is captured).  With
        `preserve_state` (default) weights, buffers, Adam moments and step counters are copied back in place afterwards,
        so that training — fresh or resumed — starts from exactly the state it was given and `global_step` / the LR
        schedule count only real training steps.
        With preserve_state=False all of it STAYS: up to three captures x (warm-up + 8 timed replays) = ~30 extra optimisation steps on
        `example_batch`, and the layout is chosen by timings that differ by 1-5 % (it can differ between runs, which changes the order of
        the atomic additions): pin it with VMASR_STEP_VARIANT (one | lane | lane:<share>) for reproducible runs."""
        # No cyclic garbage collection from here until the kept capture has been replayed: a collection that runs between a capture and
        # its first replays — it destroys whatever cycles the capture left behind: autograd contexts with their closures, an earlier
        # trainer's graphs — crashed the process inside hipGraphLaunch in 5 of 28 runs of tests/test_trainer.py's layout test (host-side
        # segmentation fault, with and without round 6's changes); with the collector off across captures AND timed replays: 0 of 38
        # (profiles/r06_replay_segfault.md; a device synchronisation after the collection alone did not help: 2 of 12).  One
        # collection at the end, behind a device synchronisation.  VMASR_GRAPH_GC_GUARD=0: off.
        import gc as _gc
        guard = _gc.isenabled() and os.environ.get("VMASR_GRAPH_GC_GUARD", "1") == "1"
        if guard:
            _gc.collect()
            _gc.disable()
        try:
            return self._enable_graphs(example_batch, warmup, preserve_state, first_replay=True)      # (every rank alike: the replay holds collectives)
        finally:
            if guard:
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                _gc.collect()
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                _gc.enable()

    def _enable_graphs(self, example_batch, warmup, preserve_state, first_replay):
        from .graph_step import GraphedTrainStep
        snap = self._snapshot_training_state() if preserve_state else None
        multi = self.world > 1 and dist.is_initialized()
        if multi:
            dist.barrier()          # every rank has built its models / communicator before anybody starts capturing
        # The generator's phase branch on a second stream inside the captured step (model._lanes; VMASR_GEN_STREAMS=1 / 2 restricts the
        # choice).  Generator-only steps: always (+14 ... +24 % at batch 35 ... 4).  GAN step on two streams: whether it pays, and how many
        # CUs the discriminator's backward should then keep, depends on the configuration (batch 4: 174 / 185 / 193 clips/s for one
        # generator stream / lane with 5/8 / lane with 3/4 of the CUs; batch 8: 201 / 200 / 213; n_fft 2048 at batch 8: 161 / 177 / 171:
        # profiles/r05_gen_streams_ab.log, r05_side_cus_sweep_lanes.log) — so the variants are captured, replayed a few times, and the
        # fastest stays.
        from . import _lib, hip_env
        gen = unwrap(self.models["generator"])
        mode = os.environ.get("VMASR_GEN_STREAMS", "auto")
        self._step_batch = int(example_batch[0].shape[0])      # (not config.DATA.BATCH_SIZE: after a resume that is the checkpoint's)
        # never in deterministic mode: both branches launch the same ticketed kernels (sscan, xproj, dwconv, mlp, ss2d_glue) and the
        # ordered-accumulation tickets are per kernel id, ONE STREAM ONLY (csrc/common.h)
        lanes_possible = (getattr(gen, "interact", "single") != "single" and hip_env.streamk_dp_in_force() and not _lib.det_mode()
                          and mode in ("auto", "2"))
        # a variant = (phase lane?, share of the CUs for the discriminator's backward beside the generator's; None: side_cu_limits()'s own)
        dims = self.config.MODEL.VSSM.DIMS
        dims = dims[0] if isinstance(dims, (list, tuple)) else dims
        lane_variants = [(True, None)] if dims >= 32 else [(True, 5 / 8), (True, 3 / 4)]
        if not lanes_possible:
            candidates = [(False, None)]
        elif not self.gan:
            candidates = [(True, None)]
        elif multi and dist.get_backend() != "nccl":
            candidates = [(False, None)]     # (gloo: the CPU-transport test mode, where ranks may share one GPU and its hardware queues)
        elif not self._two_streams():
            candidates = [(False, None)]
        elif mode == "2":
            candidates = lane_variants
        else:
            candidates = [(False, None)] + lane_variants
        # VMASR_STEP_VARIANT pins the layout (reproducible runs: the timed choice below rests on differences of 1-5 % and changes the
        # order of the atomic additions): "one" | "lane" | "lane:<share of the CUs for the discriminator's backward>", e.g. lane:0.75
        pin = os.environ.get("VMASR_STEP_VARIANT")
        if pin and len(candidates) > 1:
            want = (False, None) if pin == "one" else (True, float(pin.split(":", 1)[1]) if ":" in pin else None)
            if want[0] and want[1] is None:
                want = next((c for c in candidates if c[0]), want)
            if want in candidates or (want[0] and lanes_possible):
                candidates = [want]

        def use(variant):
            gen.phase_lane, self._lane_bwd_share = variant

        def label(variant):
            return "one_generator_stream_ms" if not variant[0] else ("phase_lane_ms" if variant[1] is None else f"phase_lane_d_bwd_{variant[1]:.3f}_ms")

        def attempt():
            try:
                self._graphed = GraphedTrainStep(self, example_batch, warmup)
                ok = True
            except Exception as e:  # pragma: no cover - depends on the runtime
                self._graphed = None
                self.graph_error = f"{type(e).__name__}: {e}"
                self.logger.warning(f"HIP graph capture unavailable ({self.graph_error})")
                ok = False
            if multi:
                # all ranks replay graphs or none does: a rank that fell back to the eager step would issue its collectives in a
                # different order relative to its kernels and run ~2x slower — the others would sit in the all-reduce
                dev = torch.device("cpu") if dist.get_backend() == "gloo" else self.device
                flag = torch.tensor([1.0 if ok else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if ok and flag.item() == 0.0:
                    self._graphed, ok = None, False
                    self.graph_error = "graph capture failed on another rank"
            return ok
        def timed_replays(n=6):
            """ms per replayed step of the current capture (real optimiser steps on the example batch: undone below with the warm-up's)"""
            for _ in range(2):
                self._graphed(*example_batch)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(n):
                self._graphed(*example_batch)
            torch.cuda.synchronize(self.device)
            t = torch.tensor([1e3 * (time.perf_counter() - t0) / n], dtype=torch.float64,
                             device=torch.device("cpu") if not multi or dist.get_backend() == "gloo" else self.device)
            if multi:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank takes the same decision: the slowest rank's time
            return float(t.item())

        use(candidates[0])
        ok = attempt()
        if (not ok and multi and dist.get_backend() == "nccl" and self.graph_collectives()):
            # the gradient all-reduces were to be captured into the step's graph (VMASR_GRAPH_COLLECTIVES=1, graph_step.py): if that
            # capture fails on any rank, every rank tries once more with the collectives BETWEEN the graphs before giving graphs up
            self.logger.warning("retrying the capture with the collectives between the graphs")
            self._graph_collectives = False          # (this trainer's choice: the process environment is left alone)
            torch.cuda.synchronize(self.device)
            ok = attempt()
        self.graph_variants = None
        if ok and len(candidates) > 1:
            # Every variant is captured and replayed a few times; the fastest stays.  Only the LAST capture is replayed afterwards (a
            # capture's warm-up steps re-create the flat gradient buffers the previous capture's graphs point into), so unless the
            # winner was captured last it is captured again.
            times, last = {0: timed_replays()}, 0
            for i in range(1, len(candidates)):
                self._graphed = None
                use(candidates[i])
                if attempt():
                    times[i], last = timed_replays(), i
                else:
                    self.graph_error = None
            self.graph_variants = {label(candidates[i]): round(t, 3) for i, t in times.items()}
            best = self._pick_variant([candidates[i] for i in times], [times[i] for i in times])
            self.logger.info(f"captured step: {self.graph_variants} -> {label(best)[:-3]}")
            use(best)
            if best != candidates[last] or self._graphed is None:
                self._graphed = None
                ok = attempt()
            torch.cuda.empty_cache()
        if not ok:
            self.logger.warning("running eagerly")
        elif first_replay:
            self._graphed(*example_batch)          # the kept capture's first replay happens HERE, before the collector runs again (enable_graphs)
            torch.cuda.synchronize(self.device)
        if snap is not None:
            torch.cuda.synchronize(self.device)
            self._restore_training_state(snap)
        return ok

    def graph_collectives(self):
        """Capture the gradient all-reduces INTO the step's graph (RCCL's C API on a communicator of this trainer, vm_asr_amd/rccl.py)?
        Opt-in (VMASR_GRAPH_COLLECTIVES=1): that path has no process-group watchdog behind it and has not run with more than one
        real rank yet; the default keeps the collectives between the graphs on torch.distributed's communicator."""
        own = getattr(self, "_graph_collectives", None)
        return os.environ.get("VMASR_GRAPH_COLLECTIVES", "0") == "1" if own is None else bool(own)

    @staticmethod
    def _pick_variant(variants, ms):
        """the fastest of the captured step layouts (tests override it to force one)"""
        return variants[min(range(len(ms)), key=lambda i: ms[i])]

    def _to_dev(self, batch):
        wave_input, wave_target, highcut = batch[0], batch[1], batch[2]
        return (wave_input.to(self.device, non_blocking=True), wave_target.to(self.device, non_blocking=True),
                highcut.to(self.device, non_blocking=True))

    def _metrics(self, wave_out, wave_target, highcut):
        out = {}
        for met in self.metric_ftns or []:
            out[met.__name__] = float(met(wave_out.float().squeeze(1), wave_target.squeeze(1), hf=highcut))
        return out

    def _train_epoch(self, epoch):
        for m in self.models.values():
            if m is not None:
                m.train()
        sums, count, t0 = {}, 0, time.time()
        self._micro = 0            # accumulation is keyed on the per-epoch batch index, as in the reference (trainer/trainer.py:146-156)
        last_idx = 0
        for batch_idx, batch in enumerate(self.data_loader):
            if batch_idx >= self.len_epoch:
                break
            last_idx = batch_idx   # index of the last PROCESSED batch (the loop variable is one past it after a `break`)
            wave_input, wave_target, highcut = self._to_dev(batch)
            wave_out, logs = self.train_step(wave_input, wave_target, highcut)
            if batch_idx % self.config.PRINT_FREQ == 0 or batch_idx == self.len_epoch - 1:
                vals = {k: float(v) for k, v in logs.items()}
                vals.update(self._metrics(wave_out, wave_target, highcut))
                for k, v in vals.items():
                    sums[k] = sums.get(k, 0.0) + v
                count += 1
                if self.rank == 0:
                    self.logger.info(f"Epoch {epoch} [{batch_idx + 1}/{self.len_epoch}] " +
                                     " ".join(f"{k}={v:.4f}" for k, v in vals.items()))
        # the learning-rate schedule advances once per EPOCH, with the reference's update index
        # (trainer/trainer.py:196-218: `(epoch * num_steps + batch_idx) // ACCUMULATION_STEPS` after the batch loop)
        if count:
            num_steps = self.len_epoch // self._acc
            upd = (epoch * num_steps + last_idx) // self._acc
            if self.lr_scheduler_G is not None:
                self.lr_scheduler_G.step_update(upd)
            if self.gan and getattr(self, "lr_scheduler_D", None) is not None:
                self.lr_scheduler_D.step_update(upd)
        self.epoch_log = {k: v / max(1, count) for k, v in sums.items()}
        self.epoch_log["epoch_seconds"] = time.time() - t0

    @torch.no_grad()
    def _valid_epoch(self, epoch):
        for m in self.models.values():
            if m is not None:
                m.eval()
        sums, count = {}, 0
        for batch in self.data_loader_val:
            wave_input, wave_target, highcut = self._to_dev(batch)
            with torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.amp):
                wave_out = unwrap(self.models["generator"])(wave_input, highcut)
            vals = {"total_loss": float(sum(self._generator_losses(wave_out, wave_target).values()))} if not self.gan else {}
            vals.update(self._metrics(wave_out, wave_target, highcut))
            for k, v in vals.items():
                sums[k] = sums.get(k, 0.0) + v
            count += 1
        if self.world > 1:
            keys = sorted(sums)
            t = torch.tensor([sums[k] for k in keys] + [float(count)], device=self.device, dtype=torch.float64)
            dist.all_reduce(t)
            sums, count = {k: t[i].item() for i, k in enumerate(keys)}, int(t[-1].item())
        # validation values go in under "val_*" only: MONITOR ("min lsd", config.py:223) keeps watching the TRAINING
        # metric, exactly as the reference does (trainer/trainer.py:312-313)
        self.epoch_log.update({f"val_{k}": v / max(1, count) for k, v in sums.items()})


def default_metric_ftns(config):
    return [getattr(metric_mod, m) for m in config.TRAIN.METRICS]
