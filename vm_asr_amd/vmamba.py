"""SS2D / VSSBlock / VSSM modules on the HIP operators.

API surface and parameter names follow the reference's model/vmamba.py so that its
checkpoints (`...op.{in_proj,conv2d,x_proj_weight,dt_projs_weight,dt_projs_bias,A_logs,Ds,
out_norm,out_proj}`) load with strict=True and its configs drop in unchanged:

    SS2D(d_model, d_state, ssm_ratio, dt_rank, act_layer, d_conv, conv_bias, dropout, bias,
         dt_min, dt_max, dt_init, dt_scale, dt_init_floor, initialize, forward_type,
         channel_first)                                            model/vmamba.py:544-571
    VSSBlock(hidden_dim, drop_path, norm_layer, channel_first, ssm_*, mlp_*, use_checkpoint,
             post_norm)                                            model/vmamba.py:1753-1843
    VSSM(...)  thin backbone kept for API compatibility            model/vmamba.py:1846-2299

Only the `__initv2__` family is built (forward types v2 / v5 and their `no32` / `noz` /
`nozact` postfixes share `forward_corev2`); every shipped yaml uses "v5" (config.py:108).
The scan, cross-scan/merge and depthwise-conv+SiLU run in libvmasr_hip.so; Linear, einsum
and LayerNorm stay in PyTorch (hipBLASLt / MIOpen).

`forward_corev2` keeps the reference's keyword hooks (`SelectiveScan=`, `CrossScan=`,
`CrossMerge=`, model/vmamba.py:1398-1400) — tests use them to plug in the CPU oracle;
the product default is always the HIP operator set.
"""
import math
import os
from functools import partial
from typing import Any

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from .csm import CrossMergeHIP, CrossScanF32, CrossScanHIP
from . import ss2d_core as _ss2d
from . import ss2d_deep as _deep
from . import ss2d_glue as _glue
from . import xproj as _xproj
from . import mlp as _mlp
from . import inproj as _inproj
from . import outproj as _outproj
from .dwconv import dwconv3x3_silu
from .layernorm import LayerNorm
from .linear import Linear as _Linear
from .selective_scan import SelectiveScanCore

__all__ = ["SS2D", "VSSBlock", "VSSM", "Mlp", "DropPath", "LayerNorm2d", "Linear2d", "Permute"]


class DropPathPool:
    """One Bernoulli draw per forward for ALL stochastic-depth layers of a model (instead of a
    bernoulli_ + div_ pair per layer: 140 tiny launches per step).  `masks[i, b]` is layer i's keep mask
    for sample b, already scaled by 1/keep (timm's scale_by_keep).  Plain object: no parameters/buffers."""

    def __init__(self, drop_probs):
        self.keep = 1.0 - torch.tensor(drop_probs, dtype=torch.float32).view(-1, 1)
        self.masks = None

    def refresh(self, batch, device):
        if self.keep.device != device:
            self.keep = self.keep.to(device)
        self.masks = torch.bernoulli(self.keep.expand(-1, batch)) / self.keep
        self._cast = {}

    def get(self, index, batch, ndim, dtype=torch.float32):
        """layer `index`'s mask as a view of ONE cast of the whole table per dtype and forward (a `.to(bf16)` per layer was 28 launches a step)"""
        m = self.masks
        if dtype != torch.float32:
            m = self._cast.get(dtype)
            if m is None:
                m = self._cast[dtype] = self.masks.to(dtype)
        return m[index, :batch].view((batch,) + (1,) * (ndim - 1))


class DropPath(nn.Module):
    """Stochastic depth per sample (timm semantics: identity in eval or at p=0)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep
        self._pool, self._index = None, -1          # set by attach_drop_path_pool()

    def active(self):
        return self.drop_prob != 0.0 and self.training

    def _mask(self, x, dtype=torch.float32):
        pool = self._pool
        if pool is not None and pool.masks is not None and pool.masks.shape[1] >= x.shape[0] and pool.masks.device == x.device:
            return pool.get(self._index, x.shape[0], x.ndim, dtype)
        keep = 1.0 - self.drop_prob
        mask = torch.empty((x.shape[0],) + (1,) * (x.ndim - 1), dtype=torch.float32, device=x.device).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        if not self.active():
            return x
        return x * self._mask(x, x.dtype).to(x.dtype)

    def residual(self, x, y):
        """x + drop_path(y) as ONE kernel (addcmul with the per-sample mask)."""
        if not self.active():
            return x + y
        dt = torch.promote_types(x.dtype, y.dtype)
        return torch.addcmul(x, y, self._mask(y, dt).to(dt))

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:0.3f}"


def attach_drop_path_pool(model):
    """Give every active DropPath of `model` a slot in one shared DropPathPool; returns the pool (or None)."""
    layers = [m for m in model.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0 and m.scale_by_keep and m.drop_prob < 1.0]
    if not layers:
        return None
    pool = DropPathPool([m.drop_prob for m in layers])
    for i, m in enumerate(layers):
        m._pool, m._index = pool, i
    return pool


class Linear2d(nn.Linear):
    """1x1 conv expressed as a Linear over the channel axis of (B,C,H,W)."""

    def forward(self, x: torch.Tensor):
        return F.conv2d(x, self.weight[:, :, None, None], self.bias)


class LayerNorm2d(nn.LayerNorm):
    def forward(self, x: torch.Tensor):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class Permute(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.args = args

    def forward(self, x: torch.Tensor):
        return x.permute(*self.args)


class Mlp(nn.Module):
    """model/vmamba.py:483-509."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 channels_first=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        Linear = Linear2d if channels_first else _Linear
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def _strip(tag: str, value: str):
    if value.endswith(tag):
        return True, value[: -len(tag)]
    return False, value


# d_inner up to which x_proj / dt_proj run as the HIP map (csrc/xproj.hip) rather than as einsums.  As batched GEMMs the
# deep stages' weight gradients are (64 x 4) products over a 16384-long contraction that hipBLASLt runs on one CU per
# direction (0.1 ms per call); the map's row-parallel kernels (d_inner >= 64) take 15-30 us per pass.  Train step of the
# headline workload: einsums for d_inner >= 64: 45.2 ms; map up to 128: 41.8 ms; up to 256 (all it supports): 41.4 ms.
_XPROJ_MAX_D = int(os.environ.get("VMASR_XPROJ_MAX_D", "512"))


class SS2D(nn.Module):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU,
                 d_conv=3, conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, initialize="v0", forward_type="v2",
                 channel_first=False, **kwargs):
        super().__init__()
        d_inner = int(ssm_ratio * d_model)
        dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.d_model, self.d_state, self.d_inner, self.dt_rank = d_model, d_state, d_inner, dt_rank
        self.d_conv = d_conv
        self.channel_first = channel_first
        self.k_group = 4
        Linear = Linear2d if channel_first else _Linear  # nn.Linear subclass (HIP row map for tiny widths)

        self.disable_force32, forward_type = _strip("no32", forward_type)
        self.disable_z, forward_type = _strip("noz", forward_type)
        self.disable_z_act, forward_type = _strip("nozact", forward_type)
        if forward_type not in ("v2", "v5"):
            raise NotImplementedError(
                f"forward_type '{forward_type}': only the v2/v5 family (SelectiveScanCore) is built; "
                "no shipped config selects anything else (config.py:108)")
        if channel_first:
            self.out_norm_shape = "v1"
            self.out_norm = LayerNorm2d(d_inner)
        else:
            self.out_norm_shape = "v0"
            self.out_norm = LayerNorm(d_inner)

        # the reference's v2 uses the PyTorch cross-scan, v5 the Triton one; both map to HIP here
        self.forward_core = partial(self.forward_corev2, force_fp32=(not self.disable_force32),
                                    SelectiveScan=SelectiveScanCore, CrossScan=CrossScanHIP,
                                    CrossMerge=CrossMergeHIP)

        d_proj = d_inner if self.disable_z else d_inner * 2
        self.in_proj = Linear(d_model, d_proj, bias=bias)
        self.act = act_layer()
        self._act_is_silu = isinstance(self.act, nn.SiLU)
        if d_conv > 1:
            self.conv2d = nn.Conv2d(d_inner, d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                    groups=d_inner, bias=conv_bias)

        K, N, R = self.k_group, d_state, dt_rank
        # x_proj: K independent Linear(d_inner -> R + 2N) weights stacked (default Linear init)
        xw = torch.empty(K, R + 2 * N, d_inner)
        for k in range(K):
            nn.init.kaiming_uniform_(xw[k], a=math.sqrt(5))
        self.x_proj_weight = nn.Parameter(xw)

        self.out_proj = Linear(d_inner, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()

        if initialize == "v0":
            ws, bs = zip(*[self.dt_init(R, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor)
                           for _ in range(K)])
            self.dt_projs_weight = nn.Parameter(torch.stack(ws, 0))  # (K, d_inner, R)
            self.dt_projs_bias = nn.Parameter(torch.stack(bs, 0))    # (K, d_inner)
            self.A_logs = self.A_log_init(N, d_inner, copies=K, merge=True)  # (K*d_inner, N)
            self.Ds = self.D_init(d_inner, copies=K, merge=True)             # (K*d_inner,)
        elif initialize == "v1":
            self.Ds = nn.Parameter(torch.ones(K * d_inner))
            self.A_logs = nn.Parameter(torch.randn(K * d_inner, N))
            self.dt_projs_weight = nn.Parameter(torch.randn(K, d_inner, R))
            self.dt_projs_bias = nn.Parameter(torch.randn(K, d_inner))
        elif initialize == "v2":
            self.Ds = nn.Parameter(torch.ones(K * d_inner))
            self.A_logs = nn.Parameter(torch.zeros(K * d_inner, N))
            self.dt_projs_weight = nn.Parameter(0.1 * torch.rand(K, d_inner, R))
            self.dt_projs_bias = nn.Parameter(0.1 * torch.rand(K, d_inner))
        else:
            raise NotImplementedError(f"initialize='{initialize}'")

    # ---- initialisers (model/vmamba.py:1204-1267) ------------------------------------------
    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4):
        """-> (weight (d_inner, dt_rank), bias (d_inner,)); softplus(bias) is log-uniform in
        [dt_min, dt_max]."""
        std = dt_rank ** -0.5 * dt_scale
        w = torch.empty(d_inner, dt_rank)
        if dt_init == "constant":
            nn.init.constant_(w, std)
        elif dt_init == "random":
            nn.init.uniform_(w, -std, std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = dt.clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))  # softplus^-1
        return w, inv_dt

    @staticmethod
    def A_log_init(d_state, d_inner, copies=-1, device=None, merge=True):
        A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 0:
            A_log = A_log[None].repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        A_log = nn.Parameter(A_log.contiguous())
        A_log._no_weight_decay = True
        return A_log

    @staticmethod
    def D_init(d_inner, copies=-1, device=None, merge=True):
        D = torch.ones(d_inner, device=device)
        if copies > 0:
            D = D[None].repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        D = nn.Parameter(D.contiguous())
        D._no_weight_decay = True
        return D

    # ---- core (model/vmamba.py:1377-1531) --------------------------------------------------
    def forward_corev2(self, x: torch.Tensor = None, delta_softplus=True, to_dtype=True, force_fp32=False,
                       nrows=-1, backnrows=-1, ssoflex=True, SelectiveScan=None, CrossScan=CrossScanHIP,
                       CrossMerge=CrossMergeHIP, no_einsum=False, merged_only=False, **kwargs):
        """merged_only: return the cross-merged (B, D, L) fp32 tensor and leave out_norm to the caller (the fused
        LayerNorm-gate operator of forward())."""
        x_proj_weight, dt_projs_weight, dt_projs_bias = self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias
        A_logs, Ds = self.A_logs, self.Ds
        out_norm = getattr(self, "out_norm", None)
        B, D, H, W = x.shape
        D, N = A_logs.shape
        K, D, R = dt_projs_weight.shape
        L = H * W

        hip_default = (x.is_cuda and force_fp32 and not no_einsum and CrossScan is CrossScanHIP
                       and SelectiveScan is SelectiveScanCore and CrossMerge is CrossMergeHIP)
        if hip_default and delta_softplus and K == 4 and _ss2d.supported(N, R, D, H, W):
            # the whole core (cross-scan, x_proj, dt_proj, 4 scans, cross-merge) as one fused operator: the
            # high-resolution stages (d_state 1, dt_rank 1, d_inner <= 32) — csrc/ss2d.hip
            y = _ss2d.ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
            return y if merged_only else self._merge_norm(y, x, B, H, W, to_dtype)
        if hip_default and delta_softplus and K == 4 and _deep.supported(N, R, D, H, W, x.dtype):
            # the deep stages (dt_rank 2..8, d_inner 64..512, H*W <= 4096): whole rows per workgroup, cross-scan / cross-merge
            # through an LDS image, x_proj as a small kernel in front — csrc/ss2d_deep.hip
            y = _deep.ss2d_deep(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
            return y if merged_only else self._merge_norm(y, x, B, H, W, to_dtype)
        if (hip_default and _xproj.supported(N, R, D) and D <= _XPROJ_MAX_D):
            # (position-parallel kernels for d_inner <= 32, row-parallel ones above; see _XPROJ_MAX_D)
            # HIP fast path: the scan streams are produced in fp32 directly and x_proj/dt_proj are one
            # memory-bound kernel writing scan-ready fp32 tensors (no einsum / contiguous / cast passes)
            xs = CrossScanF32.apply(x)                                         # (B, K, D, L) fp32
            dts, Bs, Cs = _xproj.x_proj_dt(xs, x_proj_weight, dt_projs_weight, N)
            ys = SelectiveScan.apply(xs.view(B, -1, L), dts, -torch.exp(A_logs.to(torch.float)), Bs, Cs,
                                     Ds.to(torch.float), dt_projs_bias.view(-1).to(torch.float), delta_softplus,
                                     nrows, backnrows, ssoflex).view(B, K, -1, H, W)
            y = CrossMerge.apply(ys)
            return y if merged_only else self._merge_norm(y, x, B, H, W, to_dtype)

        xs = CrossScan.apply(x)  # (B, K, D, L)
        if no_einsum:
            x_dbl = F.conv1d(xs.view(B, -1, L), x_proj_weight.view(-1, D, 1), groups=K)
            dts, Bs, Cs = torch.split(x_dbl.view(B, K, -1, L), [R, N, N], dim=2)
            dts = F.conv1d(dts.contiguous().view(B, -1, L), dt_projs_weight.view(K * D, -1, 1), groups=K)
        else:
            x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, x_proj_weight)
            dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
            dts = torch.einsum("b k r l, k d r -> b k d l", dts, dt_projs_weight)

        xs = xs.view(B, -1, L)
        dts = dts.contiguous().view(B, -1, L)
        # (a float64 module — the tests' adjudicator runs — keeps float64; everything else is fp32 as in the reference)
        f32 = (lambda t: t) if A_logs.dtype == torch.float64 else (lambda t: t.to(torch.float))
        As = -torch.exp(f32(A_logs))  # (K*D, N)
        Bs = Bs.contiguous().view(B, K, N, L)
        Cs = Cs.contiguous().view(B, K, N, L)
        Ds = f32(Ds)
        delta_bias = f32(dt_projs_bias.view(-1))
        if force_fp32:
            xs, dts, Bs, Cs = xs.to(torch.float), dts.to(torch.float), Bs.to(torch.float), Cs.to(torch.float)

        ys = SelectiveScan.apply(xs, dts, As, Bs, Cs, Ds, delta_bias, delta_softplus, nrows, backnrows,
                                 ssoflex).view(B, K, -1, H, W)
        y = CrossMerge.apply(ys)
        return y if merged_only else self._merge_norm(y, x, B, H, W, to_dtype)

    def _merge_norm(self, y, x, B, H, W, to_dtype):
        """out_norm on the merged (B, D, L) tensor -> (B, H, W, D)  (model/vmamba.py:1517-1531)."""
        out_norm = getattr(self, "out_norm", None)
        if self.channel_first:
            y = y.view(B, -1, H, W)
            if self.out_norm_shape == "v1":
                y = out_norm(y)
            else:
                y = out_norm(y.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            return y.to(x.dtype) if to_dtype else y
        if self.out_norm_shape == "v1":
            y = out_norm(y.view(B, -1, H, W)).permute(0, 2, 3, 1)
        else:
            y = out_norm(y.transpose(1, 2).contiguous()).view(B, H, W, -1)
        return y.to(x.dtype) if to_dtype else y

    # ---- module forward (model/vmamba.py:1533-1552) ----------------------------------------
    def _conv_act(self, x):
        """conv2d + act; the SiLU case is one fused HIP kernel.  `conv_act_fn` is a test hook in
        the spirit of forward_corev2's operator keywords (the CPU oracle plugs in there)."""
        hook = getattr(self, "conv_act_fn", None)
        if hook is not None:
            return hook(x, self.conv2d.weight, self.conv2d.bias)
        if self.d_conv == 3 and self._act_is_silu and x.is_cuda:
            return dwconv3x3_silu(x, self.conv2d.weight, self.conv2d.bias)
        if self.d_conv == 3 and self._act_is_silu:
            raise RuntimeError("SS2D: vm_asr_amd has no CPU path (expected a CUDA/HIP tensor)")
        return self.act(self.conv2d(x))  # non-default kernel size / activation: MIOpen

    def _fused_glue_ok(self, x):
        """The HIP glue operators apply: GPU, channel-last, default operator set (no test hooks), z gate with SiLU,
        3x3 conv, channel-last LayerNorm as out_norm."""
        fc = self.forward_core
        return (x.is_cuda and not self.channel_first and not self.disable_z and not self.disable_z_act and self._act_is_silu
                and self.d_conv == 3 and self.out_norm_shape == "v0" and isinstance(self.out_norm, LayerNorm)
                and getattr(self, "conv_act_fn", None) is None and isinstance(fc, partial)
                and fc.keywords.get("SelectiveScan") is SelectiveScanCore and fc.keywords.get("CrossScan") is CrossScanHIP
                and fc.keywords.get("CrossMerge") is CrossMergeHIP and fc.keywords.get("force_fp32", False)
                # the fused cores are softplus / x_proj-as-einsum semantics only (forward_corev2's `hip_default` predicate): a
                # forward_type that sets delta_softplus=False or no_einsum=True must take the operator chain
                and fc.keywords.get("delta_softplus", True) and not fc.keywords.get("no_einsum", False))

    def _finish(self, y, residual):
        """out_proj (+ dropout) and — when the calling block handed over (stream, DropPath) — its stochastic-depth residual
        add: one MFMA kernel where vm_asr_amd/outproj.py applies (model/vmamba.py:1551 + :1826-1827)."""
        if residual is None:
            return self.dropout(self.out_proj(y))
        res, dp = residual
        if y.dim() == 4 and _outproj.supported(y, self.out_proj, res, self.dropout):
            return _outproj.fused_out_proj_residual(y, self.out_proj, res, dp._mask(res) if dp.active() else None)
        return dp.residual(res, self.dropout(self.out_proj(y)))

    def forward(self, x: torch.Tensor, pre_norm=None, residual=None, **kwargs):
        """pre_norm: the block's LayerNorm (or nn.Identity) when the caller hands over the UN-normalised stream, so that
        LayerNorm + in_proj + chunk + SiLU(z) + the channel-first copy can run as one MFMA kernel (vm_asr_amd/inproj.py).
        residual: (stream, DropPath) of the calling VSSBlock — the result is then stream + drop_path(branch)."""
        fused_in = (pre_norm is not None and x.dim() == 4 and self._fused_glue_ok(x) and _inproj.supported(x, pre_norm, self.in_proj)
                    and _glue.supported(self.d_inner, x.shape[1] * x.shape[2], torch.bfloat16))
        if not fused_in:
            x = self.in_proj(x if pre_norm is None else pre_norm(x))
        if fused_in or (x.dim() == 4 and self._fused_glue_ok(x) and _glue.supported(self.d_inner, x.shape[1] * x.shape[2], x.dtype)):
            # chunk + SiLU(z) + layout copy as one kernel; LayerNorm + cast + gate (and the layout copy in front of
            # them) as another (csrc/ss2d_glue.hip)
            xT, sz = _inproj.fused_in_proj(x, pre_norm, self.in_proj) if fused_in else _glue.ss2d_pre(x)
            u = self._conv_act(xT)
            Bn, H, W = x.shape[0], x.shape[1], x.shape[2]
            if (self.k_group == 4 and _ss2d.supported(self.d_state, self.dt_rank, self.d_inner, H, W)
                    and _glue.pairs_supported(self.d_inner, H, W, sz.dtype)):
                # fused core + LayerNorm-gate without the merged tensor: the core leaves its two pair outputs, ln_gate adds them
                y02, y13 = _ss2d.ss2d_core_pairs(u, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds)
                y = _glue.ln_gate_pairs(y02, y13, sz, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps)
            else:
                y = self.forward_core(u, merged_only=True)                        # (B, D, L) fp32
                y = _glue.ln_gate(y, sz, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps)
            return self._finish(y, residual)
        z = None
        if not self.disable_z:
            x, z = x.chunk(2, dim=(1 if self.channel_first else -1))
            if not self.disable_z_act:
                z = self.act(z)
        if not self.channel_first:
            x = x.permute(0, 3, 1, 2).contiguous()
        if self.d_conv > 1:
            x = self._conv_act(x)
        else:
            x = self.act(x)
        y = self.forward_core(x)
        if z is not None:
            y = y * z
        return self._finish(y, residual)

    forwardv2 = forward


class VSSBlock(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer=LayerNorm, channel_first=False,
                 ssm_d_state: int = 16, ssm_ratio=2.0, ssm_dt_rank: Any = "auto", ssm_act_layer=nn.SiLU,
                 ssm_conv: int = 3, ssm_conv_bias=True, ssm_drop_rate: float = 0, ssm_init="v0",
                 forward_type="v2", mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate: float = 0.0,
                 gmlp=False, use_checkpoint: bool = False, post_norm: bool = False, **kwargs):
        super().__init__()
        self.ssm_branch = ssm_ratio > 0
        self.mlp_branch = mlp_ratio > 0
        self.use_checkpoint = use_checkpoint
        self.post_norm = post_norm
        if self.ssm_branch:
            self.norm = norm_layer(hidden_dim)
            self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank,
                           act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias,
                           dropout=ssm_drop_rate, initialize=ssm_init, forward_type=forward_type,
                           channel_first=channel_first)
        self.drop_path = DropPath(drop_path)
        if self.ssm_branch and isinstance(self.norm, LayerNorm) and not post_norm and not channel_first:
            self.norm.feeds_gemm = True   # -> SS2D.in_proj
        if self.mlp_branch:
            if gmlp:
                raise NotImplementedError("gMlp is not reachable from any shipped config (config.py:112)")
            self.norm2 = norm_layer(hidden_dim)
            self.mlp = Mlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio),
                           act_layer=mlp_act_layer, drop=mlp_drop_rate, channels_first=channel_first)
            if isinstance(self.norm2, LayerNorm) and not post_norm and not channel_first:
                self.norm2.feeds_gemm = True  # -> Mlp.fc1

    def _forward(self, input: torch.Tensor):
        x = input
        if self.ssm_branch:
            if self.post_norm:
                x = input + self.drop_path(self.norm(self.op(input)))
            else:
                x = self.op(input, pre_norm=self.norm, residual=(input, self.drop_path))
        if self.mlp_branch:
            if self.post_norm:
                x = x + self.drop_path(self.norm2(self.mlp(x)))
            elif _mlp.supported(x, self.norm2, self.mlp):
                # LayerNorm + fc1 + GELU + fc2 + residual (+ the stochastic-depth scale) as ONE MFMA kernel (csrc/mlp.hip)
                x = _mlp.fused_mlp_residual(x, self.norm2, self.mlp, self.drop_path._mask(x) if self.drop_path.active() else None)
            else:
                x = self.drop_path.residual(x, self.mlp(self.norm2(x)))
        return x

    def forward(self, input: torch.Tensor):
        if self.use_checkpoint:
            return checkpoint.checkpoint(self._forward, input, use_reentrant=False)
        return self._forward(input)


class PatchMerging2D(nn.Module):
    """(B,H,W,C) -> (B,H/2,W/2,2C)  — model/model.py:57-89, model/vmamba.py PatchMerging2D."""

    def __init__(self, dim, out_dim=-1, norm_layer=LayerNorm, **kwargs):
        super().__init__()
        self.dim = dim
        self.reduction = _Linear(4 * dim, (2 * dim) if out_dim < 0 else out_dim, bias=False)
        self.norm = norm_layer(4 * dim)
        if isinstance(self.norm, LayerNorm):
            self.norm.feeds_gemm = True  # -> reduction

    @staticmethod
    def _patch_merging_pad(x: torch.Tensor):
        H, W, _ = x.shape[-3:]
        if (W % 2 != 0) or (H % 2 != 0):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        # cat([x[0::2, 0::2], x[1::2, 0::2], x[0::2, 1::2], x[1::2, 1::2]], -1) of the reference (model/vmamba.py:80-88) as ONE
        # permuting copy: channel block j = 2 (w % 2) + (h % 2).  Same elements in the same order; forward 1 launch instead of 5,
        # backward 1 instead of 4 zero fills + 4 strided copies + 3 adds (66 launches per generator step)
        *lead, H, W, C = x.shape
        y = x.reshape(*lead, H // 2, 2, W // 2, 2, C)
        n = len(lead)
        y = y.permute(*range(n), n, n + 2, n + 3, n + 1, n + 4)       # (..., H/2, W/2, w%2, h%2, C)
        return y.reshape(*lead, H // 2, W // 2, 4 * C)

    def forward(self, x):
        return self.reduction(self.norm(self._patch_merging_pad(x)))


class VSSM(nn.Module):
    """Thin VMamba backbone kept for API-surface compatibility (model/vmamba.py:1846-2299).
    VM-ASR itself never instantiates it (model/__init__.py:11-55 builds the U-Net); the
    classifier is functional but checkpoint-renaming / openmmlab glue are not reproduced."""

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=[2, 2, 9, 2],
                 dims=[96, 192, 384, 768], ssm_d_state=16, ssm_ratio=2.0, ssm_dt_rank="auto",
                 ssm_act_layer="silu", ssm_conv=3, ssm_conv_bias=True, ssm_drop_rate=0.0, ssm_init="v0",
                 forward_type="v2", mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False,
                 drop_path_rate=0.1, patch_norm=True, norm_layer="LN", downsample_version="v1",
                 patchembed_version="v1", use_checkpoint=False, **kwargs):
        super().__init__()
        if norm_layer.lower() != "ln":
            raise NotImplementedError("VSSM: only channel-last LayerNorm is built")
        acts = dict(silu=nn.SiLU, gelu=nn.GELU, relu=nn.ReLU, sigmoid=nn.Sigmoid)
        ssm_act, mlp_act = acts[ssm_act_layer.lower()], acts[mlp_act_layer.lower()]
        self.num_classes, self.num_layers = num_classes, len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.dims, self.num_features = dims, dims[-1]
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.patch_embed = nn.Sequential(
            nn.Conv2d(in_chans, dims[0], kernel_size=patch_size, stride=patch_size, bias=True),
            Permute(0, 2, 3, 1), LayerNorm(dims[0]) if patch_norm else nn.Identity())
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            blocks = [VSSBlock(hidden_dim=dims[i], drop_path=dpr[sum(depths[:i]) + j], norm_layer=LayerNorm,
                               ssm_d_state=ssm_d_state, ssm_ratio=ssm_ratio, ssm_dt_rank=ssm_dt_rank,
                               ssm_act_layer=ssm_act, ssm_conv=ssm_conv, ssm_conv_bias=ssm_conv_bias,
                               ssm_drop_rate=ssm_drop_rate, ssm_init=ssm_init, forward_type=forward_type,
                               mlp_ratio=mlp_ratio, mlp_act_layer=mlp_act, mlp_drop_rate=mlp_drop_rate,
                               gmlp=gmlp, use_checkpoint=use_checkpoint) for j in range(depths[i])]
            down = PatchMerging2D(dims[i], dims[i + 1]) if i < self.num_layers - 1 else nn.Identity()
            self.layers.append(nn.Sequential(nn.Sequential(*blocks), down))
        self.classifier = nn.Sequential(LayerNorm(self.num_features), Permute(0, 3, 1, 2),
                                        nn.AdaptiveAvgPool2d(1), nn.Flatten(1),
                                        nn.Linear(self.num_features, num_classes))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m: nn.Module):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x: torch.Tensor):
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x)
        return self.classifier(x)
