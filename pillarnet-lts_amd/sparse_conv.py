"""Sparse-conv modules with the `spconv.pytorch` constructor signatures and state-dict layout the
reference's backbone uses (det3d/models/backbones/base.py:3-4,38-63, PillarResNet.py:4,81-108):
`SubMConv2d`, `SparseConv2d`, `SparseSequential`, `SparseReLU`, `SparseModule`.

Weights keep spconv 2.x's KRSC layout `[out, kh, kw, in]` (+ `bias [out]`), so reference
checkpoints load key-for-key (det3d/torchie/trainer/checkpoint.py:49-64 describes the layouts).
Compute goes through the HIP C ABI (`pn_spconv_*`); eval-mode `conv -> BatchNorm1d -> ReLU`
chains inside a `SparseSequential` run as ONE kernel with the norm folded into the epilogue.
"""
import math

import os

import torch
import torch.nn as nn

from . import _lib
from .spconv import SparseConvTensor


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.PN_DTYPE_F32
    if t.dtype == torch.bfloat16:
        return _lib.PN_DTYPE_BF16
    raise _lib.PillarnetHipError(f"sparse features must be fp32 or bf16, got {t.dtype}")


class SparseModule(nn.Module):
    """marker base class, as in spconv"""


class SparseReLU(SparseModule):
    def forward(self, x):
        if isinstance(x, SparseConvTensor):
            return x.replace_feature(torch.relu(x.features_cap))
        return torch.relu(x)


class SparseConvolution(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, indice_key=None):
        super().__init__()
        if kernel_size not in (3, (3, 3), [3, 3]) or dilation not in (1, (1, 1), [1, 1]) or groups != 1:
            raise NotImplementedError("the HIP sparse conv implements 3x3, dilation 1, groups 1")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (3, 3)
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.padding = padding if isinstance(padding, int) else padding[0]
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = {}

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(9 * self.in_channels)
            nn.init.uniform_(self.bias, -bound, bound)

    def packed(self, dtype, bn=None):
        """([9, out, in] weight in `dtype`, fp32 scale, fp32 shift) with the conv bias and an
        eval-mode norm folded; cached until a parameter or buffer changes."""
        key = (dtype, self.weight._version, None if self.bias is None else self.bias._version,
               self.weight.device, id(bn))
        if bn is not None:
            key += (bn.weight._version, bn.bias._version, bn.running_mean._version,
                    bn.running_var._version)
        hit = self._packed.get(dtype)
        if hit is not None and hit[0] == key:
            return hit[1:]
        with torch.no_grad():
            w = self.weight.detach().permute(1, 2, 0, 3).reshape(9, self.out_channels, self.in_channels)
            w = w.to(dtype).contiguous()
            if bn is not None:
                from .pillar_ops import fold_batchnorm
                scale, shift = fold_batchnorm(bn, self.bias)
            else:
                scale = torch.ones(self.out_channels, device=w.device)
                shift = (self.bias.detach().float().clone() if self.bias is not None
                         else torch.zeros(self.out_channels, device=w.device))
        self._packed[dtype] = (key, w, scale.contiguous(), shift.contiguous())
        return w, scale, shift

    def forward(self, x, bn=None, relu=False, residual=None):
        return conv_forward(x, self, bn, relu, residual)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, True, indice_key)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, False, indice_key)


# bench.py's roofline leg: when set to (cin, cout), every launch of that conv shape is bracketed by
# HIP events on the launch stream and the rulebook is kept so the active pairs can be counted.
EVENT_PROBE = {"shape": None, "events": [], "rules": None, "rows": None}


def _launch_conv(dt, feats, cin, nbr, stride, num, cap, w, scale, shift, res, relu, out, cout, dev):
    probe = EVENT_PROBE["shape"] == (cin, cout)
    if probe:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    rc = _lib.lib().pn_spconv_forward(dt, _lib.ptr(feats), cin, feats.shape[0], _lib.ptr(nbr), stride, _lib.ptr(num), cap,
                                      _lib.ptr(w), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res),
                                      int(relu), _lib.ptr(out), cout, _lib.current_stream(dev))
    if probe:
        e1.record(torch.cuda.current_stream(dev))
        EVENT_PROBE["events"].append((e0, e1))
        EVENT_PROBE["rules"], EVENT_PROBE["rows"] = nbr, num
    if _lib.FLOP_LOG is not None:
        _lib.log_flops(f"sparse {cin}->{cout}", cin, cout, (nbr[:, :int(num.item())] >= 0).sum())
    _lib.check(rc, "pn_spconv_forward")


def _subm_rules(x: SparseConvTensor, key, stream=None):
    """stream: raw handle to launch on instead of the current stream (the tensors are still allocated under the
    current one -- see rules_ahead)."""
    cached = x.indice_dict.get(key) if key is not None else None
    if cached is not None:
        return cached
    dev = x.indices_cap.device
    cap = x.capacity
    H, W = x.spatial_shape
    nbr = torch.empty((9, cap), dtype=torch.int32, device=dev)
    rc = _lib.lib().pn_spconv_subm_rules(_lib.ptr(x.indices_cap), _lib.ptr(x.num_active), cap,
                                         x.batch_size, H, W, _lib.ptr(x.occ_bits),
                                         _lib.ptr(x.occ_rank), _lib.ptr(nbr),
                                         stream if stream is not None else _lib.current_stream(dev))
    _lib.check(rc, "pn_spconv_subm_rules")
    if key is not None:
        x.indice_dict[key] = nbr
    return nbr


# Inference, one frame at a time: the index chain of the strided levels (output sites, occupancy index, rulebooks of
# levels 2..4 -- a dozen launches of ~6 us that depend on level 1's INDICES only) runs on a side stream beside
# level 1's convolutions instead of between the levels.  Off by default: it takes 20-30 us off a 1.42 ms frame, a
# graph with a fork costs the host six times as long to launch, and with several frames in flight the other frame's
# kernels already fill those gaps.
RULES_AHEAD = False
_SIDE_STREAMS = {}


class _Ahead:
    __slots__ = ("event", "out_idx", "out_num", "out_bits", "out_rank", "nbr", "shape", "out_cap", "indice_dict", "keep")


def rules_ahead(x: SparseConvTensor, stages):
    """stages: [(strided conv module, indice_key of the SubM convs behind it)] in network order, starting at `x`.
    Returns one record per stage; `conv_forward` of that stage's strided conv takes it from `tensor._ahead`, waits for
    its event and uses its tensors.  Every tensor is allocated under the CURRENT stream and only the launches go to the
    side stream, so the allocator's stream bookkeeping stays that of an ordinary frame; scratch buffers are kept in the
    records until the caller drops them (behind the last join)."""
    dev = x.indices_cap.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)      # default priority
    side.wait_stream(main)                                              # the indices come from the reader
    import ctypes
    h = ctypes.c_void_p(side.cuda_stream)
    recs, t = [], x
    for conv, key in stages:
        r = _Ahead()
        r.keep = []
        r.out_idx, r.out_num, r.out_bits, r.out_rank, r.nbr, r.shape, r.out_cap = _down_rules(
            t, conv.stride, conv.padding, stream=h, keep=r.keep)
        t = SparseConvTensor(None, r.out_idx, r.shape, x.batch_size, r.out_num, r.out_bits, r.out_rank)
        if key is not None:
            _subm_rules(t, key, stream=h)
        r.indice_dict = t.indice_dict
        r.event = torch.cuda.Event()
        r.event.record(side)
        recs.append(r)
    return recs


def _down_rules(x: SparseConvTensor, stride, pad, stream=None, keep=None):
    dev = x.indices_cap.device
    H, W = x.spatial_shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    B = x.batch_size
    # an input site feeds ceil(3/stride)^2 output cells at most (4 for k3 s2)
    fan = ((3 + stride - 1) // stride) ** 2
    out_cap = max(1, min(x.capacity * fan, B * Ho * Wo))
    words = (B * Ho * Wo + 63) // 64
    L = _lib.lib()
    out_idx = torch.empty((out_cap, 3), dtype=torch.int32, device=dev)
    out_num = torch.empty(1, dtype=torch.int32, device=dev)   # always written by the emit kernel
    out_bits = torch.empty(words, dtype=torch.int64, device=dev)
    out_rank = torch.empty(words, dtype=torch.int32, device=dev)
    nbr = torch.empty((9, out_cap), dtype=torch.int32, device=dev)
    sb = L.pn_spconv_down_scratch_bytes(B, Ho, Wo)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    rc = L.pn_spconv_down_rules(_lib.ptr(x.indices_cap), _lib.ptr(x.num_active), x.capacity, B, H, W,
                                _lib.ptr(x.occ_bits), _lib.ptr(x.occ_rank), stride, pad, Ho, Wo,
                                _lib.ptr(out_idx), _lib.ptr(out_num), out_cap, _lib.ptr(out_bits),
                                _lib.ptr(out_rank), _lib.ptr(nbr), _lib.ptr(scratch), sb,
                                stream if stream is not None else _lib.current_stream(dev))
    _lib.check(rc, "pn_spconv_down_rules")
    if keep is not None:
        keep.append(scratch)      # launched on another stream: must outlive this call
    return out_idx, out_num, out_bits, out_rank, nbr, (Ho, Wo), out_cap


def rulebook_wgrad(feats, nbr, g, n_out):
    """dW[co][tap][ci] = sum_m g[m][co] * feats[nbr[tap][m]][ci] -> [cout, ntaps, cin] fp32 (pn_spconv_wgrad).
    feats / g: fp32 rows (row stride = shape[1] of a contiguous 2-D tensor, or any 2-D view with unit channel
    stride); nbr [ntaps, stride] int32 or None (identity, one tap)."""
    cin, cout = feats.shape[1], g.shape[1]
    ntaps = nbr.shape[0] if nbr is not None else 1
    dev = g.device
    assert feats.dtype == torch.float32 and g.dtype == torch.float32 and feats.stride(1) == 1 and g.stride(1) == 1
    if n_out == 0:
        return torch.zeros((cout, ntaps, cin), dtype=torch.float32, device=dev)
    if g.device.type != "cuda":
        raise _lib.PillarnetHipError("rulebook_wgrad: tensors must live on the GPU (no CPU fallback)")
    if cin % 32 or cout % 32:
        # widths outside the MFMA kernel's 32-channel granularity (no live preset has one): nine gathered library
        # GEMMs on the GPU, missing neighbours masked to zero rows instead of compacted (no host wait)
        gw = torch.empty((cout, ntaps, cin), dtype=torch.float32, device=dev)
        gk = g[:n_out]
        for k in range(ntaps):
            if nbr is None:
                gw[:, k] = gk.t() @ feats[:n_out]
                continue
            idx = nbr[k, :n_out].long()
            live = (idx >= 0).to(gk.dtype).unsqueeze(1)
            gw[:, k] = (gk * live).t() @ feats.index_select(0, idx.clamp(min=0))
        return gw
    L = _lib.lib()
    gw = torch.empty((cout, ntaps, cin), dtype=torch.float32, device=dev)
    ws = torch.empty(max(int(L.pn_spconv_wgrad_workspace_bytes(n_out, cin, cout, ntaps)), 1), dtype=torch.uint8, device=dev)
    rc = L.pn_spconv_wgrad(_lib.ptr(feats), cin, feats.stride(0), _lib.ptr(nbr) if nbr is not None else None,
                           nbr.shape[1] if nbr is not None else 0, ntaps, None, n_out, _lib.ptr(g), cout, g.stride(0),
                           _lib.ptr(gw), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "pn_spconv_wgrad")
    return gw


_CONST = {}


def _const(value, n, dev):
    """cached fp32 [n] constant (identity scale / zero shift of the un-folded training convs)"""
    key = (value, n, dev)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), float(value), dtype=torch.float32, device=dev)
    return t


class SparseConvFunction(torch.autograd.Function):
    """Differentiable sparse conv on exact fp32 rows.  Forward and data gradient run the HIP
    implicit GEMM (the gradient is the same gather-GEMM over the INVERSE rulebook with transposed
    tap matrices); the weight gradient is one fp32 MFMA kernel over the rulebook (pn_spconv_wgrad)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, nbr, n_out, inv, count_in, count_out):
        cout, cin = weight.shape[0], weight.shape[3]
        dev = feats.device
        w = weight.detach().permute(1, 2, 0, 3).reshape(9, cout, cin).contiguous().float()
        ones = _const(1, cout, dev)
        shift = bias.detach().float().contiguous() if bias is not None else _const(0, cout, dev)
        out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
        if n_out > 0:
            rc = _lib.lib().pn_spconv_forward(_lib.PN_DTYPE_F32, _lib.ptr(feats), cin, feats.shape[0], _lib.ptr(nbr),
                                              nbr.shape[1], _lib.ptr(count_out), n_out, _lib.ptr(w), _lib.ptr(ones),
                                              _lib.ptr(shift), None, 0, _lib.ptr(out), cout,
                                              _lib.current_stream(dev))
            _lib.check(rc, "pn_spconv_forward")
        ctx.save_for_backward(feats, weight, nbr, inv, count_in)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        feats, weight, nbr, inv, count_in = ctx.saved_tensors
        g = g.contiguous().float()
        cout, cin = weight.shape[0], weight.shape[3]
        dev = g.device
        n_in, n_out = feats.shape[0], g.shape[0]
        grad_feats = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_feats = torch.zeros((n_in, cin), dtype=torch.float32, device=dev)
            if n_in > 0 and n_out > 0:
                wd = weight.detach().permute(1, 2, 3, 0).reshape(9, cin, cout).contiguous().float()   # W_k^T
                ones, zeros = _const(1, cin, dev), _const(0, cin, dev)
                rc = _lib.lib().pn_spconv_forward(_lib.PN_DTYPE_F32, _lib.ptr(g), cout, n_out, _lib.ptr(inv),
                                                  inv.shape[1], _lib.ptr(count_in), n_in, _lib.ptr(wd),
                                                  _lib.ptr(ones), _lib.ptr(zeros), None, 0, _lib.ptr(grad_feats),
                                                  cin, _lib.current_stream(dev))
                _lib.check(rc, "pn_spconv_forward (data gradient)")
        if ctx.needs_input_grad[1]:
            grad_w = rulebook_wgrad(feats, nbr, g, n_out).view(cout, 3, 3, cin).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = g.sum(0)
        return grad_feats, grad_w, grad_b, None, None, None, None, None


def _conv_forward_train(x: SparseConvTensor, conv: SparseConvolution):
    """Training-mode conv (no folding): exact fp32 rows in, exact rows out, autograd-connected."""
    x = x.exact()
    feats = x.features_cap
    if feats.dtype != torch.float32:
        raise NotImplementedError("training runs in fp32")
    dev = feats.device
    n_in = feats.shape[0]
    if conv.subm:
        nbr = _subm_rules(x, conv.indice_key)
        inv = nbr.flip(0).contiguous()        # inv_k = nbr_{8-k} for a submanifold conv
        out = SparseConvFunction.apply(feats, conv.weight, conv.bias, nbr, n_in, inv, x.num_active, x.num_active)
        return x.replace_feature(out)
    out_idx, out_num, out_bits, out_rank, nbr, shape, out_cap = _down_rules(x, conv.stride, conv.padding)
    n_out = int(out_num.item())
    nbr = nbr[:, :n_out].contiguous() if n_out else nbr[:, :1].contiguous()
    inv = torch.empty((9, max(n_in, 1)), dtype=torch.int32, device=dev)
    rc = _lib.lib().pn_spconv_invert_rules(_lib.ptr(nbr), nbr.shape[1], _lib.ptr(out_num), max(n_out, 1),
                                           _lib.ptr(inv), inv.shape[1], _lib.current_stream(dev))
    _lib.check(rc, "pn_spconv_invert_rules")
    out = SparseConvFunction.apply(feats, conv.weight, conv.bias, nbr, n_out, inv, x.num_active, out_num)
    y = SparseConvTensor(out, out_idx[:n_out].contiguous(), shape, x.batch_size, out_num, out_bits, out_rank)
    y._n_host = n_out
    return y


class ToDenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, x):
        ctx.idx = x.indices_cap[: feats.shape[0]].long()
        out = to_dense(x.replace_feature(feats), channels_last=True)
        return out.contiguous() if os.environ.get("PN_TRAIN_NCHW") else out

    @staticmethod
    def backward(ctx, g):
        i = ctx.idx
        return g[i[:, 0], :, i[:, 1], i[:, 2]].contiguous(), None


def conv_forward(x: SparseConvTensor, conv: SparseConvolution, bn=None, relu=False, residual=None):
    """conv (+ folded eval-mode norm) (+ residual rows) (+ ReLU) as one kernel; in training mode the
    plain differentiable conv (the norm / activation then run as torch ops on the rows)."""
    feats = x.features_cap
    if conv.training or (bn is not None and bn.training):
        if feats.device.type != "cuda":
            raise _lib.PillarnetHipError("sparse conv: features must live on the GPU (no CPU fallback)")
        y = _conv_forward_train(x, conv)
        f = y.features_cap
        if bn is not None:
            f = bn(f)
        if residual is not None:
            r = residual.features_cap if isinstance(residual, SparseConvTensor) else residual
            f = f + r[: f.shape[0]]
        if relu:
            f = torch.relu(f)
        return y.replace_feature(f) if f is not y.features_cap else y
    if feats.device.type != "cuda":
        raise _lib.PillarnetHipError("sparse conv: features must live on the GPU (no CPU fallback)")
    dt = _dtype_code(feats)
    w, scale, shift = conv.packed(feats.dtype, bn)
    dev = feats.device
    if conv.subm:
        assert conv.stride == 1 and conv.padding == 1, "SubMConv2d 3x3 needs stride 1, padding 1"
        nbr = _subm_rules(x, conv.indice_key)
        cap, num = x.capacity, x.num_active
        out = torch.empty((cap, conv.out_channels), dtype=feats.dtype, device=dev)
        res = residual.features_cap if isinstance(residual, SparseConvTensor) else residual
        _launch_conv(dt, feats, conv.in_channels, nbr, cap, num, cap, w, scale, shift, res, relu, out,
                     conv.out_channels, dev)
        return x.replace_feature(out)
    assert residual is None
    ahead = x.__dict__.pop("_ahead", None)
    if ahead is not None:          # built beside the previous level's convolutions (rules_ahead)
        torch.cuda.current_stream(dev).wait_event(ahead.event)
        out_idx, out_num, out_bits, out_rank, nbr, shape, out_cap = (
            ahead.out_idx, ahead.out_num, ahead.out_bits, ahead.out_rank, ahead.nbr, ahead.shape, ahead.out_cap)
    else:
        out_idx, out_num, out_bits, out_rank, nbr, shape, out_cap = _down_rules(x, conv.stride, conv.padding)
    out = torch.empty((out_cap, conv.out_channels), dtype=feats.dtype, device=dev)
    _launch_conv(dt, feats, conv.in_channels, nbr, out_cap, out_num, out_cap, w, scale, shift, None, relu,
                 out, conv.out_channels, dev)
    y = SparseConvTensor(out, out_idx, shape, x.batch_size, out_num, out_bits, out_rank)
    if ahead is not None:
        y.indice_dict = ahead.indice_dict
    return y


class SparseSequential(SparseModule):
    """spconv's SparseSequential: sparse modules get the tensor, plain modules get `.features`.
    In eval mode a `conv, BatchNorm1d[, ReLU/SparseReLU]` run is dispatched as one fused kernel."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        from collections import OrderedDict
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
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(str(len(self._modules)) if name is None else name, module)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution):
                bn, relu, step = None, False, 1
                if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.modules.batchnorm._BatchNorm)
                        and not mods[i + 1].training and not m.training):
                    bn, step = mods[i + 1], 2
                    if i + 2 < len(mods) and isinstance(mods[i + 2], (nn.ReLU, SparseReLU)):
                        relu, step = True, 3
                x = conv_forward(x, m, bn, relu)
                i += step
            elif isinstance(m, SparseModule):
                x = m(x)
                i += 1
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(m(x.features_cap))
                i += 1
            else:
                x = m(x)
                i += 1
        return x


def to_dense(x: SparseConvTensor, channels_last=True, out_channels=None):
    """out_channels > C (channels-last only): the rows land in the first C channels of a
    [B,H,W,out_channels] buffer (the rest is left for a later producer: concat without a copy);
    the returned tensor is the [B,C,H,W] view of those channels."""
    feats = x.features_cap
    H, W = x.spatial_shape
    B, C = x.batch_size, feats.shape[1]
    ld = C if out_channels is None else int(out_channels)
    if channels_last:
        out = torch.empty((B, H, W, ld), dtype=feats.dtype, device=feats.device)
    else:
        assert ld == C
        out = torch.empty((B, C, H, W), dtype=feats.dtype, device=feats.device)
    rc = _lib.lib().pn_sparse_to_dense(_dtype_code(feats), _lib.ptr(feats), _lib.ptr(x.indices_cap),
                                       _lib.ptr(x.num_active), x.capacity, B, H, W, C,
                                       int(channels_last), ld,
                                       _lib.ptr(x.occ_bits) if channels_last else None,
                                       _lib.ptr(x.occ_rank) if channels_last else None, _lib.ptr(out),
                                       _lib.current_stream(feats.device))
    _lib.check(rc, "pn_sparse_to_dense")
    return out[..., :C].permute(0, 3, 1, 2) if channels_last else out
