"""Small nn helpers whose *module naming* is part of the checkpoint contract:
`Sequential` (children named "0","1",...; det3d/models/utils/misc.py:22-98), `build_norm_layer`
(det3d/models/utils/norm.py:59-109), and the two weight-init rules the reference applies
(det3d/torchie/cnn/weight_init.py: kaiming_init / xavier_init)."""
from collections import OrderedDict

import torch.nn as nn


class Sequential(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        n = len(self._modules)
        if not (-n <= idx < n):
            raise IndexError(f"index {idx} is out of range")
        return list(self._modules.values())[idx % n]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


_NORMS = {"BN": ("bn", nn.BatchNorm2d), "BN1d": ("bn1d", nn.BatchNorm1d),
          "GN": ("gn", nn.GroupNorm), "LN": ("ln", nn.LayerNorm)}


def build_norm_layer(cfg, num_features, postfix=""):
    assert isinstance(cfg, dict) and "type" in cfg
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if kind not in _NORMS:
        raise KeyError(f"Unrecognized norm type {kind}")
    abbr, cls = _NORMS[kind]
    assert isinstance(postfix, (int, str))
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if kind == "GN":
        assert "num_groups" in cfg
        layer = cls(num_channels=num_features, **cfg)
    else:
        layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


def kaiming_init(module, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    assert distribution in ("uniform", "normal")
    init = nn.init.kaiming_uniform_ if distribution == "uniform" else nn.init.kaiming_normal_
    init(module.weight, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    assert distribution in ("uniform", "normal")
    init = nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_
    init(module.weight, gain=gain)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


# ---- synchronised batch norm (data-parallel training) --------------------------------------------
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


SYNC_BN_COLLECTIVES = [0]      # all-reduces issued by SyncBatchNorm in this process (forward + backward), for reports


class _SyncBNFunction(torch.autograd.Function):
    """Batch norm over the rows of ALL ranks.  Forward: one all-reduce of [sum, sum of squares,
    count] per layer; backward: one all-reduce of [sum dy, sum dy*(x-mean)].  Row counts may differ
    between ranks (sparse levels hold a different number of active sites per sample)."""

    @staticmethod
    def forward(ctx, x2, weight, bias, eps, group):
        C = x2.shape[1]
        xf = x2.float()
        stats = torch.cat([xf.sum(0), (xf * xf).sum(0), xf.new_full((1,), float(x2.shape[0]))])
        dist.all_reduce(stats, group=group)
        SYNC_BN_COLLECTIVES[0] += 1
        n = stats[2 * C]
        mean = stats[:C] / n
        var = (stats[C:2 * C] / n - mean * mean).clamp_(min=0)
        invstd = torch.rsqrt(var + eps)
        ctx.save_for_backward(xf, weight, mean, invstd, n)
        ctx.group = group
        y = (xf - mean) * (invstd * weight.float()) + bias.float()
        return y.to(x2.dtype), mean, var, n

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dn):
        xf, weight, mean, invstd, n = ctx.saved_tensors
        C = xf.shape[1]
        dyf = dy.float()
        xmu = xf - mean
        sums = torch.cat([dyf.sum(0), (dyf * xmu).sum(0)])
        local = sums.clone()
        dist.all_reduce(sums, group=ctx.group)
        SYNC_BN_COLLECTIVES[0] += 1
        sum_dy, sum_dy_xmu = sums[:C], sums[C:]
        dx = (dyf - sum_dy / n - xmu * (invstd * invstd) * (sum_dy_xmu / n)) * (invstd * weight.float())
        # parameter gradients stay local sums: the gradient exchange averages them like any other
        dw = local[C:] * invstd
        db = local[:C]
        return dx.to(dy.dtype), dw.to(weight.dtype), db.to(weight.dtype), None, None


class SyncBatchNorm(nn.modules.batchnorm._BatchNorm):
    """apex `convert_syncbn_model` semantics (det3d/torchie/apis/train.py:267-268) for [N, C] rows and
    [B, C, H, W] maps on any backend (RCCL on the GPU, gloo in CPU tests).  Same parameters and buffers
    as BatchNorm1d/2d, so state-dict keys do not change.  Without an initialised process group, or in
    eval mode, it is plain batch norm."""

    def _check_input_dim(self, x):
        if x.dim() not in (2, 4):
            raise ValueError(f"expected 2D or 4D input (got {x.dim()}D input)")

    def forward(self, x):
        self._check_input_dim(x)
        synced = self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not synced:
            return super().forward(x)
        x2 = x if x.dim() == 2 else x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        y, mean, var, n = _SyncBNFunction.apply(x2, self.weight, self.bias, self.eps, None)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1).clamp(min=1))
                self.running_mean.mul_(1 - m).add_(mean.to(self.running_mean.dtype), alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased.to(self.running_var.dtype), alpha=m)
        if x.dim() == 2:
            return y
        B, C, H, W = x.shape
        return y.view(B, H, W, C).permute(0, 3, 1, 2)


def convert_syncbn_model(module):
    """Swap every BatchNorm1d / BatchNorm2d for `SyncBatchNorm`, in place, keeping parameters, buffers
    and names (apex.parallel.convert_syncbn_model as called at det3d/torchie/apis/train.py:268)."""
    for name, child in list(module.named_children()):
        if isinstance(child, (nn.BatchNorm1d, nn.BatchNorm2d)) and not isinstance(child, SyncBatchNorm):
            new = SyncBatchNorm(child.num_features, child.eps, child.momentum, child.affine,
                                child.track_running_stats)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var = child.running_mean, child.running_var
            new.num_batches_tracked = child.num_batches_tracked
            new.train(child.training)
            module._modules[name] = new
        else:
            convert_syncbn_model(child)
    return module
