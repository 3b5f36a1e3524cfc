"""Minimal attribute-dict config that loads the reference's configs/vm_asr_*.yaml unchanged.

The reference uses yacs (config.py:1-344), which is not a dependency here.  Same rules:
defaults (config.py:5-249) <- yaml file with optional BASE includes (:252-264) <- `--opts`
KEY VALUE pairs (:271-272) <- derived fields (hop length / resample ranges by TARGET_SR
:313-320, output dir :307-310, single LPF unless MULTIFILTER :330-332); then frozen.
Only keys that exist in the defaults may be set (yacs semantics).
"""
import ast
import copy
import os

import yaml

__all__ = ["CfgNode", "get_default_config", "get_config", "update_config", "to_yacs", "from_yacs", "yacs_pickle_compat"]


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def clone(self):
        c = CfgNode(copy.deepcopy(dict(self)))
        return c

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def merge_from_dict(self, other, path=""):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{full}: expected a mapping")
                self[k].merge_from_dict(v, full)
            else:
                self[k] = _coerce(v, self[k], full)

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("opts must be KEY VALUE pairs")
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(val, node[parts[-1]], key)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self))


# ---- checkpoint interchange with the reference (base/base_trainer.py:146-153, utils/utils.py:141-145) ------------
# The reference stores its yacs CfgNode OBJECT under checkpoint["config"] and calls `.defrost()` on what it loads.
# yacs is not a dependency here, so checkpoints are written with a stand-in that pickles BY REFERENCE as
# `yacs.config.CfgNode` with yacs's own instance state (a dict subclass whose __dict__ holds __immutable__,
# __deprecated_keys__, __renamed_keys__, __new_allowed__): the reference unpickles it into a real yacs node, and
# reference checkpoints unpickle here into the stand-in, which `from_yacs` turns into this module's CfgNode.
class _YacsNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def is_frozen(self):
        return self.__dict__.get("__immutable__", False)

    def _imm(self, flag):
        self.__dict__["__immutable__"] = flag
        for v in self.values():
            if isinstance(v, _YacsNode):
                v._imm(flag)

    def defrost(self):
        self._imm(False)

    def freeze(self):
        self._imm(True)


_YacsNode.__module__, _YacsNode.__qualname__, _YacsNode.__name__ = "yacs.config", "CfgNode", "CfgNode"


def to_yacs(cfg):
    """This module's CfgNode -> the node the reference's `torch.load` turns into a yacs CfgNode."""
    try:
        from yacs.config import CfgNode as Y      # a real yacs, if the environment has one
    except ImportError:
        Y = _YacsNode
    n = Y.__new__(Y)
    dict.__init__(n)
    n.__dict__.update({"__immutable__": bool(cfg.is_frozen()), "__deprecated_keys__": set(), "__renamed_keys__": {},
                       "__new_allowed__": False})
    for k, v in cfg.items():
        dict.__setitem__(n, k, to_yacs(v) if isinstance(v, CfgNode) else copy.deepcopy(v))
    return n


def from_yacs(node):
    """A config loaded from a checkpoint (yacs CfgNode, the stand-in, a plain dict or a yaml dump) -> CfgNode."""
    if isinstance(node, CfgNode):
        return node
    if isinstance(node, str):
        node = yaml.safe_load(node)
    frozen = bool(getattr(node, "__dict__", {}).get("__immutable__", False))
    c = CfgNode(node)
    if frozen:
        c.freeze()
    return c


class yacs_pickle_compat:
    """While active, `yacs.config.CfgNode` resolves (for pickle, both directions) to the real yacs class when yacs
    is installed and to the stand-in otherwise."""

    def __enter__(self):
        import sys
        import types
        self._added = []
        try:
            import yacs.config  # noqa: F401
        except ImportError:
            for name in ("yacs", "yacs.config"):
                if name not in sys.modules:
                    sys.modules[name] = types.ModuleType(name)
                    self._added.append(name)
            sys.modules["yacs"].config = sys.modules["yacs.config"]
            sys.modules["yacs.config"].CfgNode = _YacsNode
        return self

    def __exit__(self, *exc):
        import sys
        for name in self._added:
            sys.modules.pop(name, None)
        return False


def _coerce(new, old, key):
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and isinstance(new, (int, float)) and key.endswith("SSM_DT_RANK"):
        return new
    raise ValueError(f"Type mismatch for {key}: {type(old).__name__} vs {type(new).__name__}")


_LPFS = [["cheby1", 6], ["cheby1", 8], ["cheby1", 10], ["cheby1", 12], ["bessel", 6], ["bessel", 12],
         ["ellip", 6], ["ellip", 12]]


def get_default_config():
    d = {
        "BASE": [""],
        "DATA": {
            "BATCH_SIZE": 24, "DATA_PATH": "data/", "DATASET": "VCTK_092", "MIC_ID": "mic1", "RESAMPLER": "scipy",
            "SHUFFLE": True, "NUM_WORKERS": 1, "USE_QUANTITY": 0.1, "TRAIN_SPLIT": [100, 8], "VALID_SPLIT": 0.1,
            "TARGET_SR": 48000, "RANDOM_RESAMPLE": [8000, 48000],
            "WEIGHTED_SR": {"ENABLE": False, "RANGES": [[8000, 16000], [16000, 24000], [24000, 48000]],
                            "WEIGHTS": [0.5, 0.3, 0.2]},
            "SEGMENT": 2.555, "PAD_WHITENOISE": 1e-32,
            "STFT": {"N_FFT": 1024, "HOP_LENGTH": 240, "WIN_LENGTH": 1024, "SCALE": "log2"},
            "LPF": {"MULTIFILTER": False, "LPF_TRAIN": copy.deepcopy(_LPFS), "LPF_TEST": [["cheby1", 6]]},
            "FLAC2WAV": {"SRC_SR": 48000, "SRC_PATH": "data/",
                         "DST_PATH": "VCTK-Corpus-0.92/wav48_silence_trimmed_wav",
                         "TIMESTAMPS": "./vctk-silence-labels/vctk-silences.0.92.txt"},
        },
        "MODEL": {
            "TYPE": "VM_ASR", "NAME": "VM_ASR_BASIC", "RESUME_PATH": None, "DROP_RATE": 0.0,
            "VSSM": {"IN_CHANS": 1, "PATCH_SIZE": 4, "DEPTHS": [2, 2, 2, 2], "DIMS": 16, "SSM_D_STATE": 1,
                     "SSM_RATIO": 2.0, "SSM_DT_RANK": "auto", "SSM_ACT_LAYER": "silu", "SSM_CONV": 3,
                     "SSM_CONV_BIAS": True, "SSM_DROP_RATE": 0.0, "SSM_INIT": "v0", "SSM_FORWARDTYPE": "v5",
                     "MLP_RATIO": 4.0, "MLP_ACT_LAYER": "gelu", "MLP_DROP_RATE": 0.0, "GMLP": False,
                     "DROP_PATH_RATE": 0.1, "PATCH_NORM": True, "NORM_LAYER": "LN", "PATCHEMBED": "v2",
                     "DOWNSAMPLE": "v1", "UPSAMPLE": "v1", "OUTPUT": "v3", "CONCAT_SKIP": True, "INTERACT": "dual"},
        },
        "TRAIN": {
            "START_EPOCH": 0, "EPOCHS": 50, "WARMUP_EPOCHS": 10, "EARLY_STOPPING": 10, "WEIGHT_DECAY": 0.0,
            "BASE_LR": 1e-3, "MAX_LR": 1e-3, "MIN_LR": 1e-5, "CYCLE_MULT": 1.0, "ENABLE_GAN": False,
            "LOSSES": {"GEN": ["multi_resolution_stft"]}, "METRICS": ["snr", "lsd", "lsd_hf", "lsd_lf"],
            "LOW_FREQ_REPLACEMENT": False, "AUTO_RESUME": True, "ACCUMULATION_STEPS": 1,
            "OPTIMIZER": {"NAME": "adamw", "EPS": 1e-8, "BETAS": (0.9, 0.999), "MOMENTUM": 0.9},
            "LR_SCHEDULER": {"NAME": "cosine", "DECAY_EPOCHS": 30, "DECAY_RATE": 0.1, "WARMUP_PREFIX": True,
                             "GAMMA": 0.1, "MULTISTEPS": []},
            "ADVERSARIAL": {"ENABLE": False, "DISCRIMINATORS": [""],
                            "STFT_LOSS": {"SC_FACTOR": 0.5, "MAG_FACTOR": 0.5, "EMPHASIZE_HIGH_FREQ": False},
                            "MPD_HIDDEN": 32, "FEATURE_LOSS_LAMBDA": 100, "ONLY_FEATURE_LOSS": False,
                            "ONLY_ADVERSARIAL_LOSS": False, "GAN_LOSS_TYPE": "lsgan", "GP_LAMBDA": 10},
        },
        "TEST": {"RESULTS_DIR": "results", "OVERLAP": 2000, "SAVE_RESULT": True},
        "INFERENCE": {"RESULTS_DIR": "results_inference", "OVERLAP": 2000},
        "DEBUG": False, "DEBUG_OUTPUT": "debug", "N_GPU": 1, "AMP_ENABLE": True, "OUTPUT": "logs",
        "TAG": "default", "MONITOR": "min lsd", "SAVE_EPOCH_FREQ": -1, "PRINT_FREQ": 10, "SEED": 123,
        "EVAL_MODE": False, "THROUGHPUT_MODE": False,
        "WANDB": {"ENABLE": False, "PROJECT": "VM_ASR", "ENTITY": None, "MODE": "online", "LOG": "all",
                  "RESUME": False, "TAGS": []},
        "TENSORBOARD": {"ENABLE": True, "LOG_ITEMS": ["audio", "waveform", "spectogram"]},
        "INFERENCE_MODE": False,
    }
    return CfgNode(d)


def _merge_file(config, cfg_file):
    with open(cfg_file) as f:
        y = yaml.safe_load(f) or {}
    for base in y.get("BASE", [""]) or [""]:
        if base:
            _merge_file(config, os.path.join(os.path.dirname(cfg_file), base))
    config.merge_from_file(cfg_file)


def update_config(config, cfg=None, opts=None, **args):
    """`args` mirrors main.py's named flags: batch_size, resume, accumulation_steps, disable_amp,
    output, tag, eval, inference, throughput, optim, input_sr."""
    config.defrost()
    if cfg:
        _merge_file(config, cfg)
    if opts:
        config.merge_from_list(list(opts))
    g = lambda n: args.get(n)  # noqa: E731
    if g("batch_size"):
        config.DATA.BATCH_SIZE = g("batch_size")
    if g("resume"):
        config.MODEL.RESUME_PATH = g("resume")
        if not config.EVAL_MODE:
            config.WANDB.RESUME = True
    if g("accumulation_steps"):
        config.TRAIN.ACCUMULATION_STEPS = g("accumulation_steps")
    if g("disable_amp"):
        config.AMP_ENABLE = False
    if g("output"):
        config.OUTPUT = g("output")
    if g("tag"):
        config.TAG = g("tag")
    if g("eval"):
        config.EVAL_MODE = True
    if g("inference"):
        config.INFERENCE_MODE = True
    if g("throughput"):
        config.THROUGHPUT_MODE = True
    if g("optim"):
        config.TRAIN.OPTIMIZER.NAME = g("optim")
    if config.MODEL.RESUME_PATH is None:
        config.OUTPUT = os.path.join(config.OUTPUT, config.MODEL.NAME, config.TAG)
    else:
        config.OUTPUT = config.MODEL.RESUME_PATH
    if config.DATA.TARGET_SR == 48000:
        config.DATA.RANDOM_RESAMPLE = [8000, 48000]
        config.DATA.STFT.HOP_LENGTH = 240
        config.DATA.WEIGHTED_SR.RANGES = [[8000, 16000], [16000, 24000], [24000, 48000]]
    else:
        config.DATA.RANDOM_RESAMPLE = [2000, 16000]
        config.DATA.STFT.HOP_LENGTH = 80
        config.DATA.WEIGHTED_SR.RANGES = [[2000, 8000], [8000, 12000], [12000, 16000]]
    if g("input_sr"):
        if config.DATA.TARGET_SR == 48000 and g("input_sr") >= config.DATA.TARGET_SR:
            raise ValueError(f"Input sample rate should be less than {config.DATA.TARGET_SR}")
        config.DATA.RANDOM_RESAMPLE = [g("input_sr")]
    if not config.EVAL_MODE and not config.DATA.LPF.MULTIFILTER:
        config.DATA.LPF.LPF_TRAIN = [config.DATA.LPF.LPF_TRAIN[0]]
    config.freeze()
    return config


def get_config(cfg=None, opts=None, **args):
    return update_config(get_default_config(), cfg, opts, **args)
