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
