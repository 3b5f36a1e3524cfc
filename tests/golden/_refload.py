"""Import the read-only reference (/root/reference) on CPU with stubbed third-party modules.

THIS FILE ONLY RUNS IN THE BUILD CONTAINER.  It is used by ``make_golden.py`` to
generate the golden fixtures under ``tests/golden/*.npz``; nothing at test/bench/
smoke time imports it, and ``/root/reference`` does not exist on the GPU box.
No reference source is copied: modules are loaded from where they lie.

Recipe: timm / fvcore / torchinfo / torchaudio / termcolor are
absent here, so tiny stand-in *modules* (not reference code) are registered before
loading ``utils/stft.py``, ``model/vmamba.py``, ``model/model.py``,
``model/metric.py`` and ``model/loss.py`` by file path.  ``selective_scan_ref`` is
extracted from ``kernels/selective_scan/test_selective_scan.py`` by AST because
that file runs GPU tests at import time.
"""
import ast
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("VMASR_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    """Stochastic depth stand-in with timm's semantics (identity in eval / p=0)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _install_stubs():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_vmasr_stub", False):
        return
    layers = _mod("timm.models.layers", DropPath=_DropPath,
                  trunc_normal_=torch.nn.init.trunc_normal_)
    models = _mod("timm.models", layers=layers)
    _mod("timm", models=models, _vmasr_stub=True)
    _mod("fvcore.nn", FlopCountAnalysis=None, flop_count_str=None, flop_count=None,
         parameter_count=None)
    _mod("fvcore", nn=sys.modules["fvcore.nn"])
    _mod("torchinfo", summary=None)

    class _AmplitudeToDB(nn.Module):  # only the unused dB branch touches it
        def __init__(self, *a, **k):
            super().__init__()

    tr = _mod("torchaudio.transforms", AmplitudeToDB=_AmplitudeToDB)
    fn = _mod("torchaudio.functional")
    _mod("torchaudio", transforms=tr, functional=fn)
    _mod("termcolor", colored=lambda s, *a, **k: s)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference modules loaded on CPU."""
    if _CACHE:
        return _CACHE["ns"]
    _install_stubs()
    # `base` package: only BaseModel is needed (base/__init__ pulls logger/termcolor).
    base_model = _load("base.base_model", os.path.join(REF, "base/base_model.py"))
    _mod("base", BaseModel=base_model.BaseModel, base_model=base_model)
    stft = _load("utils.stft", os.path.join(REF, "utils/stft.py"))
    _mod("utils", stft=stft)
    sys.path.insert(0, os.path.join(REF, "model"))  # for vmamba's fallback `from csm_triton import`
    try:
        vmamba = _load("vmamba", os.path.join(REF, "model/vmamba.py"))
        model = _load("refmodel", os.path.join(REF, "model/model.py"))
    finally:
        sys.path.pop(0)
    metric = _load("refmetric", os.path.join(REF, "model/metric.py"))
    try:
        loss = _load("refloss", os.path.join(REF, "model/loss.py"))
    except Exception as e:  # pragma: no cover
        loss = None
        print("WARNING: reference loss not importable:", e)

    # selective_scan_ref by AST (the file itself executes CUDA tests on import).
    src = open(os.path.join(REF, "kernels/selective_scan/test_selective_scan.py")).read()
    tree = ast.parse(src)
    fn_nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef)
                and n.name == "selective_scan_ref"]
    assert len(fn_nodes) == 1
    code = compile(ast.Module(body=fn_nodes, type_ignores=[]), "selective_scan_ref", "exec")
    import torch.nn.functional as F
    from einops import rearrange, repeat
    g = dict(torch=torch, F=F, rearrange=rearrange, repeat=repeat)
    exec(code, g)

    # float64 evaluation of the SAME function: every `.float()` in its body becomes `.double()` (AST
    # rewrite, nothing else touched) -> the adjudicator for "which fp32 result is closer to exact"
    class _ToDouble(ast.NodeTransformer):
        def visit_Attribute(self, node):
            self.generic_visit(node)
            if node.attr == "float":
                node.attr = "double"
            return node
    tree64 = ast.fix_missing_locations(_ToDouble().visit(ast.Module(body=fn_nodes, type_ignores=[])))
    g64 = dict(torch=torch, F=F, rearrange=rearrange, repeat=repeat)
    exec(compile(tree64, "selective_scan_ref64", "exec"), g64)

    try:
        disc = _load("refdisc", os.path.join(REF, "model/discriminator.py"))
    except Exception as e:  # pragma: no cover
        disc = None
        print("WARNING: reference discriminator not importable:", e)
    ns = types.SimpleNamespace(vmamba=vmamba, model=model, stft=stft, metric=metric, discriminator=disc,
                               loss=loss, selective_scan_ref=g["selective_scan_ref"],
                               selective_scan_ref64=g64["selective_scan_ref"])
    _CACHE["ns"] = ns
    return ns


class RefScanAdapter:
    """Gives selective_scan_ref the `.apply` signature SS2D.forward_corev2 expects
    (model/vmamba.py:1440-1455)."""

    ref = None

    @classmethod
    def apply(cls, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False,
              nrows=1, backnrows=1, oflex=True):
        return cls.ref(u, delta, A, B, C, D, None, delta_bias, delta_softplus)


def patch_ss2d_to_cpu(ns, module):
    """Rewire every SS2D in `module` to the CPU oracle through the reference's own
    keyword hooks (model/vmamba.py:1398-1400)."""
    from functools import partial
    RefScanAdapter.ref = staticmethod(ns.selective_scan_ref)
    for m in module.modules():
        if isinstance(m, ns.vmamba.SS2D):
            m.forward_core = partial(m.forward_corev2, force_fp32=True,
                                     SelectiveScan=RefScanAdapter,
                                     CrossScan=ns.vmamba.CrossScan,
                                     CrossMerge=ns.vmamba.CrossMerge)
    return module


class _RefScan64:
    """`.apply` adapter for one SS2D evaluated in float64: forward_corev2 hands over As / Ds / delta_bias
    after a `.to(torch.float)` (model/vmamba.py:1481-1485); the adapter rebuilds them in double from the
    module's own parameters so that no fp32 arithmetic is left on the path."""

    def __init__(self, ref64, module):
        self.ref64, self.m = ref64, module

    def apply(self, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1, backnrows=1,
              oflex=True):
        m = self.m
        return self.ref64(u.double(), delta.double(), -torch.exp(m.A_logs.double()), B.double(), C.double(),
                          m.Ds.double(), None, m.dt_projs_bias.view(-1).double(), delta_softplus)


def patch_ss2d_to_cpu64(ns, module):
    """float64 evaluation: module.double() + every SS2D on selective_scan_ref64 with force_fp32 off."""
    from functools import partial
    module.double()
    for m in module.modules():
        if isinstance(m, ns.vmamba.SS2D):
            m.forward_core = partial(m.forward_corev2, force_fp32=False, SelectiveScan=_RefScan64(ns.selective_scan_ref64, m),
                                     CrossScan=ns.vmamba.CrossScan, CrossMerge=ns.vmamba.CrossMerge)
    return module


class hann_in_double:
    """utils/stft.py builds `torch.hann_window(win_length)` in fp32 (:36,:83); inside this context the window
    is float64 so the float64 evaluation carries no fp32-rounded constant."""

    def __enter__(self):
        self._orig = torch.hann_window
        torch.hann_window = lambda n, *a, **k: self._orig(n, *a, **{**k, "dtype": torch.float64})
        return self

    def __exit__(self, *exc):
        torch.hann_window = self._orig
        return False


if __name__ == "__main__":
    ns = load_reference()
    print("reference loaded:", [k for k in vars(ns)])
