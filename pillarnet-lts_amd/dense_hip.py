"""Dense conv stacks (conv5, RPN necks, CenterHead) on the HIP gather-GEMM instead of MIOpen.

The dense layers of the reference are plain 3x3 convs / 2x2 stride-2 transposed convs with
BatchNorm + ReLU (det3d/models/backbones/PillarResNet.py:111-117, det3d/models/necks/rpn.py:147-207,
det3d/models/bbox_heads/center_head.py:14-51,101-127).  On channels-last rows `[B*H*W, C]` such a
conv is the same gather -> MFMA -> fused-epilogue kernel the sparse backbone uses, with a rulebook
that is simply the regular pixel grid: `pn_gather_gemm`.  Measured on MI355X at 180x180x256->256
bf16: 78 us against 244 us for MIOpen's igemm + separate BN and ReLU passes, and bit-reproducible
(MIOpen's strided bf16 conv is not).

This module compiles eval-mode `Sequential`s into lists of fused launches ("plans"); the modules
and their state dicts are untouched (plans are rebuilt when a parameter changes).  bf16 only -- the
fp32 parity path keeps running the torch modules.
"""
import weakref

import torch
import torch.nn as nn

from . import _lib
from .pillar_ops import fold_batchnorm
from .sparse_conv import SparseReLU

_RULES = {}


def grid_rules(B, H, W, stride, device):
    """nbr[k, o] = input pixel row of tap k (ky*3+kx, pad 1) of output pixel o, or -1; cached."""
    key = (B, H, W, stride, str(device))
    hit = _RULES.get(key)
    if hit is not None:
        return hit
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    b = torch.arange(B, device=device).view(B, 1, 1)
    oy = torch.arange(Ho, device=device).view(1, Ho, 1)
    ox = torch.arange(Wo, device=device).view(1, 1, Wo)
    nbr = torch.empty((9, B * Ho * Wo), dtype=torch.int32, device=device)
    for k in range(9):
        iy, ix = oy * stride - 1 + k // 3, ox * stride - 1 + k % 3
        ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
        idx = (b * H + iy) * W + ix
        nbr[k] = torch.where(ok, idx, torch.full_like(idx, -1)).reshape(-1).to(torch.int32)
    _RULES[key] = (nbr, Ho, Wo)
    return _RULES[key]


def up2_rows(B, H, W, device):
    """output row of input pixel (b,y,x) for each parity (dy,dx) of a k2 s2 transposed conv."""
    key = ("up2", B, H, W, str(device))
    hit = _RULES.get(key)
    if hit is not None:
        return hit
    b = torch.arange(B, device=device).view(B, 1, 1)
    y = torch.arange(H, device=device).view(1, H, 1)
    x = torch.arange(W, device=device).view(1, 1, W)
    maps = []
    for dy in range(2):
        for dx in range(2):
            maps.append(((b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx).reshape(-1).to(torch.int32))
    _RULES[key] = torch.stack(maps).contiguous()           # [4, rows_in]
    return _RULES[key]


class Rows:
    """A dense feature map as channels-last rows: t is a [B*H*W, C] view with strides (ld, 1);
    ld > C when the map is a channel slice of a wider buffer (concat without copies)."""

    def __init__(self, t, B, H, W):
        assert t.dim() == 2 and t.stride(1) == 1
        self.t, self.B, self.H, self.W = t, B, H, W

    @property
    def ld(self):
        return self.t.stride(0)

    @staticmethod
    def from_nchw(x):
        B, C, H, W = x.shape
        r = x.permute(0, 2, 3, 1)
        try:
            t = r.view(B * H * W, C)
            if t.stride(1) != 1:
                raise RuntimeError
        except RuntimeError:
            t = r.contiguous().view(B * H * W, C)
        return Rows(t, B, H, W)

    def nchw(self):
        C = self.t.shape[1]
        return torch.as_strided(self.t, (self.B, C, self.H, self.W),
                                (self.H * self.W * self.ld, 1, self.W * self.ld, self.ld))

    def widen(self, channels):
        """The [rows, channels] view starting at this map's first channel (same buffer)."""
        assert channels <= self.ld
        return Rows(torch.as_strided(self.t, (self.t.shape[0], channels), (self.ld, 1)), self.B, self.H, self.W)

    def cols(self, c0, c1):
        return Rows(self.t[:, c0:c1], self.B, self.H, self.W)


def _launch(x, cin, nbr, ntaps, rows, w, scale, shift, relu, out, out_ld, fp32_out=False, out_rows=None,
            residual=None, res_ld=0, in_ld=0, useful=1.0):
    """x / out: 2-D views with unit channel stride (their data_ptr carries any channel offset).  `useful`: the share
    of the weight matrix that is not structural zeros (flop accounting only)."""
    if _lib.FLOP_LOG is not None:
        _lib.log_flops(f"dense {cin}->{w.shape[1]} x{ntaps}", cin, w.shape[1],
                       rows if nbr is None else (nbr[:, :rows] >= 0).sum(), useful)
    rc = _lib.lib().pn_gather_gemm(_lib.ptr(x), cin, in_ld, x.shape[0], _lib.ptr(nbr), 0 if nbr is None else nbr.shape[1], ntaps,
                                   None, rows, _lib.ptr(w), w.shape[1], _lib.ptr(scale), _lib.ptr(shift),
                                   _lib.ptr(residual), res_ld, int(relu), _lib.ptr(out), out_ld, int(fp32_out),
                                   _lib.ptr(out_rows), _lib.current_stream(x.device))
    _lib.check(rc, "pn_gather_gemm")


class ConvStep:
    """3x3 conv (pad 1, stride s) [+ folded eval BN] [+ ReLU]."""

    def __init__(self, conv, bn, relu, extra_pad=0):
        assert conv.kernel_size == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1
        assert conv.padding[0] + extra_pad == 1 and conv.stride[0] in (1, 2)
        self.stride, self.relu = conv.stride[0], relu
        self.cin, self.cout = conv.in_channels, conv.out_channels
        with torch.no_grad():
            self.w = conv.weight.detach().permute(2, 3, 0, 1).reshape(9, self.cout, self.cin).to(torch.bfloat16).contiguous()
            if bn is not None:
                self.scale, self.shift = fold_batchnorm(bn, conv.bias)
            else:
                self.scale = torch.ones(self.cout, device=conv.weight.device)
                self.shift = (conv.bias.detach().float().clone() if conv.bias is not None
                              else torch.zeros(self.cout, device=conv.weight.device))

    def run(self, x: Rows, out=None, fp32_out=False):
        """out: optional [rows, cout] view (unit channel stride) to write into."""
        assert x.t.shape[1] == self.cin and x.t.dtype == torch.bfloat16
        nbr, Ho, Wo = grid_rules(x.B, x.H, x.W, self.stride, x.t.device)
        rows = x.B * Ho * Wo
        if out is None:
            out = torch.empty((rows, self.cout), dtype=torch.float32 if fp32_out else torch.bfloat16,
                              device=x.t.device)
        assert out.shape == (rows, self.cout) and out.stride(1) == 1
        _launch(x.t, self.cin, nbr, 9, rows, self.w, self.scale, self.shift, self.relu, out, out.stride(0),
                fp32_out, in_ld=x.ld)
        return Rows(out, x.B, Ho, Wo)


class UpStep:
    """ConvTranspose2d(k2, s2, no bias) [+ BN] [+ ReLU]: four 1-tap GEMMs, one per output parity, issued as
    ONE grouped launch (group = parity: its own weight slice and output-row map, the same input rows)."""

    def __init__(self, convt, bn, relu):
        assert convt.kernel_size == (2, 2) and convt.stride == (2, 2) and convt.padding == (0, 0)
        self.cin, self.cout, self.relu = convt.in_channels, convt.out_channels, relu
        with torch.no_grad():
            w = convt.weight.detach()                      # [cin, cout, 2, 2]
            self.w = torch.stack([w[:, :, dy, dx].t().reshape(1, self.cout, self.cin)
                                  for dy in range(2) for dx in range(2)]).to(torch.bfloat16).contiguous()
            if bn is not None:
                scale, shift = fold_batchnorm(bn, convt.bias)
            else:
                scale = torch.ones(self.cout, device=w.device)
                shift = torch.zeros(self.cout, device=w.device)
            self.scale, self.shift = scale.repeat(4).contiguous(), shift.repeat(4).contiguous()

    def run(self, x: Rows, out=None):
        assert x.t.shape[1] == self.cin
        maps = up2_rows(x.B, x.H, x.W, x.t.device)         # [4, rows_in] int32
        rows_in = x.B * x.H * x.W
        if out is None:
            out = torch.empty((rows_in * 4, self.cout), dtype=torch.bfloat16, device=x.t.device)
        assert out.shape == (rows_in * 4, self.cout) and out.stride(1) == 1
        _lib.log_flops(f"up2 {self.cin}->{self.cout}", self.cin, self.cout, 4 * rows_in)
        rc = _lib.lib().pn_gather_gemm_grouped(_lib.ptr(x.t), self.cin, x.ld, x.t.shape[0], 0, None, 0, 1, rows_in,
                                               _lib.ptr(self.w), self.cout, 4, _lib.ptr(self.scale),
                                               _lib.ptr(self.shift), int(self.relu), _lib.ptr(out), out.stride(0), 0, 0,
                                               _lib.ptr(maps), rows_in, _lib.current_stream(x.t.device))
        _lib.check(rc, "pn_gather_gemm_grouped")
        return Rows(out, x.B, 2 * x.H, 2 * x.W)


class PointwiseStep:
    """1x1 Conv2d, or ConvTranspose2d(k1, s1) (the same map with the weight transposed), [+ BN] [+ ReLU]:
    a 1-tap GEMM over the pixel rows."""

    def __init__(self, conv, bn, relu):
        assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
        self.cin, self.cout, self.relu = conv.in_channels, conv.out_channels, relu
        with torch.no_grad():
            w = conv.weight.detach()[:, :, 0, 0]
            if isinstance(conv, nn.ConvTranspose2d):
                w = w.t()                                   # [cin, cout] -> [cout, cin]
            self.w = w.reshape(1, self.cout, self.cin).to(torch.bfloat16).contiguous()
            if bn is not None:
                self.scale, self.shift = fold_batchnorm(bn, conv.bias)
            else:
                self.scale = torch.ones(self.cout, device=w.device)
                self.shift = (conv.bias.detach().float().clone() if conv.bias is not None
                              else torch.zeros(self.cout, device=w.device))

    def run(self, x: Rows, out=None):
        assert x.t.shape[1] == self.cin and x.t.dtype == torch.bfloat16
        rows = x.t.shape[0]
        if out is None:
            out = torch.empty((rows, self.cout), dtype=torch.bfloat16, device=x.t.device)
        assert out.shape == (rows, self.cout) and out.stride(1) == 1
        _launch(x.t, self.cin, None, 1, rows, self.w, self.scale, self.shift, self.relu, out, out.stride(0),
                in_ld=x.ld)
        return Rows(out, x.B, x.H, x.W)


def compile_stack(seq):
    """Sequential of [ZeroPad2d(1)] Conv2d / ConvTranspose2d [BatchNorm2d] [ReLU] -> list of steps."""
    mods = list(seq._modules.values()) if hasattr(seq, "_modules") else list(seq)
    steps, i, pad = [], 0, 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d):
            assert tuple(m.padding) == (1, 1, 1, 1)
            pad, i = 1, i + 1
            continue
        if isinstance(m, (nn.Sequential,)) or (hasattr(m, "_modules") and len(m._modules) and
                                               not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))):
            steps += compile_stack(m)
            i += 1
            continue
        assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)), f"cannot compile {type(m).__name__}"
        bn, relu, j = None, False, i + 1
        if j < len(mods) and isinstance(mods[j], nn.modules.batchnorm._BatchNorm):
            assert not mods[j].training
            bn, j = mods[j], j + 1
        if j < len(mods) and isinstance(mods[j], (nn.ReLU, SparseReLU)):
            relu, j = True, j + 1
        if m.kernel_size == (1, 1):
            steps.append(PointwiseStep(m, bn, relu))
        else:
            steps.append(UpStep(m, bn, relu) if isinstance(m, nn.ConvTranspose2d) else ConvStep(m, bn, relu, pad))
        pad, i = 0, j
    return steps


def run_stack(steps, x: Rows, last_out=None):
    for k, s in enumerate(steps):
        x = s.run(x, last_out) if (k == len(steps) - 1 and last_out is not None) else s.run(x)
    return x


def params_version(module):
    return tuple(p._version for p in module.parameters()) + tuple(b._version for b in module.buffers())


class PlanCache:
    """plans keyed by module, invalidated when any parameter / buffer is modified in place."""

    def __init__(self):
        self._d = {}

    def get(self, module, builder):
        v = params_version(module)
        hit = self._d.get(id(module))
        # id() values are recycled once a module is collected: the entry must belong to THIS module
        if hit is None or hit[0] != v or hit[2]() is not module:
            if len(self._d) > 256:
                self._d = {k: e for k, e in self._d.items() if e[2]() is not None}
            hit = (v, builder(module), weakref.ref(module))
            self._d[id(module)] = hit
        return hit[1]


PLANS = PlanCache()
LAZY_PLANS = PlanCache()      # CenterHead -> LazyHeadPlan (PLANS holds the dense HeadPlan of the same module)


import os
_HEAD_DIRECT = bool(os.environ.get("PN_HEAD_DIRECT"))   # tuning switch, see HeadPlan.run
# per-tap products of the output convs: fp32 (default) or bf16 (PN_HEAD_Z_BF16=1: halves the 84 MB intermediate;
# measured 0.3 % on the frame, so the head keeps its fp32 logits)
_HEAD_Z_FP32 = not os.environ.get("PN_HEAD_Z_BF16")


# ---- CenterHead: shared conv, all first-layer towers as ONE conv, one small conv per task ----------
_ORDER = ("reg", "height", "dim", "rot", "vel", "iou", "hm")


class HeadPlan:
    def __init__(self, head):
        self.share = [compile_stack(sc) for sc in head.share_convs]
        self.groups = []          # per feature level: batched first convs of the tasks on that level
        dev = next(head.parameters()).device
        self.task_slices = []
        per_level = {}
        for t, th in enumerate(head.task_heads):
            names = [n for n in _ORDER if n in th.heads]
            per_level.setdefault(head.task_idx[t], []).append((t, th, names))
        self.levels = {}
        for lvl, items in per_level.items():
            firsts, finals, col = [], [], 0
            for (t, th, names) in items:
                tower_cols, outs = [], []
                for n in names:
                    fc = getattr(th, n)
                    assert len(fc) == 4, "plan expects (conv, bn, relu, conv) towers"
                    step = ConvStep(fc[0], fc[1], True)
                    firsts.append(step)
                    tower_cols.append((col, step.cout))
                    col += step.cout
                    outs.append(fc[3])
                finals.append((t, tower_cols, outs))
            w = torch.cat([s.w for s in firsts], dim=1).contiguous()
            scale = torch.cat([s.scale for s in firsts]).contiguous()
            shift = torch.cat([s.shift for s in firsts]).contiguous()
            task_convs = []
            for (t, tower_cols, outs) in finals:
                k0 = tower_cols[0][0]
                kin = sum(c for _, c in tower_cols)
                nout = sum(o.out_channels for o in outs)
                npad = (nout + 3) // 4 * 4
                wt = torch.zeros((9, npad, kin), dtype=torch.float32, device=dev)
                bias = torch.zeros(npad, dtype=torch.float32, device=dev)
                r = 0
                for (c0, cw), o in zip(tower_cols, outs):
                    with torch.no_grad():
                        wt[:, r:r + o.out_channels, c0 - k0:c0 - k0 + cw] = \
                            o.weight.detach().float().permute(2, 3, 0, 1).reshape(9, o.out_channels, cw)
                        bias[r:r + o.out_channels] = o.bias.detach().float()
                    r += o.out_channels
                task_convs.append((t, k0, kin, nout, npad, wt.to(torch.bfloat16).contiguous(),
                                   torch.ones(npad, device=dev), bias))
            # uniform tasks (same tower width / padded output width): ONE grouped 1-tap GEMM computes the
            # per-tap channel products z = towers . [W_0 | ... | W_8] for every task, and a small
            # tap-sum kernel adds the nine shifted slices + bias.  A direct 3x3 gather conv would
            # re-read nine taps of the 2304-channel tower map per output tile (measured 97 us / task).
            uniform = len({(tc[2], tc[4]) for tc in task_convs}) == 1 and \
                all(tc[1] == task_convs[0][1] + i * tc[2] for i, tc in enumerate(task_convs))
            fused = None
            if uniform:
                kin, npad = task_convs[0][2], task_convs[0][4]
                wz = torch.stack([tc[5].reshape(1, 9 * npad, kin) for tc in task_convs]).contiguous()
                bias = torch.stack([tc[7] for tc in task_convs]).contiguous()
                ones = torch.ones(len(task_convs) * 9 * npad, device=dev)
                zeros = torch.zeros(len(task_convs) * 9 * npad, device=dev)
                fused = (kin, npad, wz, ones, zeros, bias, task_convs[0][1])
            self.levels[lvl] = (firsts[0].cin, col, w, scale, shift, task_convs, fused)

    def run(self, feats):
        """feats: list of Rows per level -> per task (fp32 map [B*H*W, npad], nout, npad, B, H, W)."""
        out = {}
        L = _lib.lib()
        for lvl, (cin, cols, w, scale, shift, task_convs, fused) in self.levels.items():
            x = run_stack(self.share[lvl], feats[lvl])
            nbr, Ho, Wo = grid_rules(x.B, x.H, x.W, 1, x.t.device)
            rows = x.B * Ho * Wo
            dev = x.t.device
            tow = torch.empty((rows, cols), dtype=torch.bfloat16, device=dev)
            _launch(x.t, cin, nbr, 9, rows, w, scale, shift, True, tow, cols, in_ld=x.ld)
            if fused is not None and _HEAD_DIRECT:
                # all tasks' 3x3 output convs as ONE grouped launch of the window kernel (unit stages, fp32 out)
                kin, npad, wz, ones, zeros, bias, k0 = fused
                G = len(task_convs)
                w9 = torch.stack([tc[5] for tc in task_convs]).contiguous()        # [G, 9, npad, kin]
                one = ones[:G * npad].contiguous()
                maps = torch.empty((G, rows, npad), dtype=torch.float32, device=dev)
                _lib.log_flops("head output convs (direct)", kin, npad, G * (nbr[:, :rows] >= 0).sum())
                rc = L.pn_gather_gemm_grouped(_lib.ptr(tow[:, k0:]), kin, cols, rows, kin, _lib.ptr(nbr), nbr.shape[1], 9,
                                              rows, _lib.ptr(w9), npad, G, _lib.ptr(one), _lib.ptr(bias), 0,
                                              _lib.ptr(maps), npad, rows * npad, 1, None, 0, _lib.current_stream(dev))
                _lib.check(rc, "pn_gather_gemm_grouped")
                for g, (t, _k0, _kin, nout, _npad, _wt, _ones, _bias) in enumerate(task_convs):
                    out[t] = (maps[g], nout, npad, x.B, Ho, Wo)
                continue
            if fused is not None:
                kin, npad, wz, ones, zeros, bias, k0 = fused
                G = len(task_convs)
                zdt = torch.float32 if _HEAD_Z_FP32 else torch.bfloat16
                z = torch.empty((rows, G * 9 * npad), dtype=zdt, device=dev)
                _lib.log_flops("head per-tap products", kin, 9 * npad, G * rows)
                rc = L.pn_gather_gemm_grouped(_lib.ptr(tow[:, k0:]), kin, cols, rows, kin, None, 0, 1, rows,
                                              _lib.ptr(wz), 9 * npad, G, _lib.ptr(ones), _lib.ptr(zeros), 0,
                                              _lib.ptr(z), G * 9 * npad, 9 * npad, int(_HEAD_Z_FP32), None, 0,
                                              _lib.current_stream(dev))
                _lib.check(rc, "pn_gather_gemm_grouped")
                maps = torch.empty((G, rows, npad), dtype=torch.float32, device=dev)
                rc = L.pn_tap_sum(_lib.ptr(z), _lib.PN_DTYPE_F32 if _HEAD_Z_FP32 else _lib.PN_DTYPE_BF16, G * 9 * npad,
                                  9 * npad, _lib.ptr(nbr), nbr.shape[1], rows, npad, G, _lib.ptr(bias), _lib.ptr(maps),
                                  npad, rows * npad, _lib.current_stream(dev))
                _lib.check(rc, "pn_tap_sum")
                for g, (t, _k0, _kin, nout, _npad, _wt, _ones, _bias) in enumerate(task_convs):
                    out[t] = (maps[g], nout, npad, x.B, Ho, Wo)
                continue
            for (t, k0, kin, nout, npad, wt, ones, bias) in task_convs:
                m = torch.empty((rows, npad), dtype=torch.float32, device=dev)
                _launch(tow[:, k0:k0 + kin], kin, nbr, 9, rows, wt, ones, bias, False, m, npad, fp32_out=True,
                        in_ld=cols)
                out[t] = (m, nout, npad, x.B, Ho, Wo)
        return [out[t] for t in sorted(out)]


# ---- CenterHead with the regression heads evaluated only at the cells predict() reads ------------------
class LazyHeadPlan:
    """Dense part: shared conv, the heat-map towers of every task as ONE conv, the heat-map output convs
    (per-tap products + tap sum, exactly as HeadPlan does them).  Sparse part, after the cells have been
    ranked on the heat map alone: the regression towers at the 3x3 neighbourhoods of the selected cells
    (<= nms_pre_max_size per task) and the regression output convs at the selected cells, as two grouped
    gather-GEMMs + one tap sum over compact row blocks.  Every value is produced by the same kernels, in
    the same summation order, as the dense plan produces it for that cell.

    Reference semantics: det3d/models/bbox_heads/center_head.py:14-51 (SepHead), :116-127 (forward),
    :353-413 (post_processing reads the regression maps only at the ranked cells)."""

    def __init__(self, head):
        dev = next(head.parameters()).device
        self.ok, self.why = True, ""
        self.share = [compile_stack(sc) for sc in head.share_convs]
        self.levels = {}
        order = [n for n in _ORDER if n != "hm"]
        per_level = {}
        for t, th in enumerate(head.task_heads):
            per_level.setdefault(head.task_idx[t], []).append((t, th))
        for lvl, items in per_level.items():
            tasks = [t for t, _ in items]
            if tasks != list(range(tasks[0], tasks[0] + len(tasks))):
                self.ok, self.why = False, "tasks of one feature level are not consecutive"
                return
            hm_steps, reg = [], []
            for t, th in items:
                names = [n for n in order if n in th.heads]
                for n in names + ["hm"]:
                    if len(getattr(th, n)) != 4:
                        self.ok, self.why = False, "towers are not (conv, bn, relu, conv)"
                        return
                hm_fc = th.hm
                hm_steps.append((ConvStep(hm_fc[0], hm_fc[1], True), hm_fc[3]))
                steps = [(ConvStep(getattr(th, n)[0], getattr(th, n)[1], True), getattr(th, n)[3]) for n in names]
                reg.append(steps)
            tw = hm_steps[0][0].cout
            kin_r = sum(s.cout for s, _ in reg[0])
            nout_r = sum(o.out_channels for _, o in reg[0])
            npad_r = (nout_r + 3) // 4 * 4
            npad_h = (max(o.out_channels for _, o in hm_steps) + 3) // 4 * 4
            if any(s.cout != tw for s, _ in hm_steps) or \
                    any(sum(s.cout for s, _ in r) != kin_r or sum(o.out_channels for _, o in r) != nout_r for r in reg):
                self.ok, self.why = False, "tasks differ in tower / output widths"
                return
            G = len(items)
            with torch.no_grad():
                w_hm = torch.cat([s.w for s, _ in hm_steps], dim=1).contiguous()              # [9, G*tw, cin]
                sc_hm = torch.cat([s.scale for s, _ in hm_steps]).contiguous()
                sh_hm = torch.cat([s.shift for s, _ in hm_steps]).contiguous()
                wz_hm = torch.zeros((G, 9, npad_h, tw), dtype=torch.float32, device=dev)
                b_hm = torch.zeros((G, npad_h), dtype=torch.float32, device=dev)
                for g, (_, o) in enumerate(hm_steps):
                    wz_hm[g, :, :o.out_channels] = o.weight.detach().float().permute(2, 3, 0, 1).reshape(9, o.out_channels, tw)
                    b_hm[g, :o.out_channels] = o.bias.detach().float()
                # all tasks' heat-map output convs as ONE 1-tap GEMM over the [rows, G*tw] tower map: column
                # tap*(G*npad_h) + g*npad_h + c, block g of the K axis; the zero blocks add exact zeros, and the
                # tap sum then reads G*npad_h contiguous floats per (row, tap) instead of npad_h
                wz_all = torch.zeros((9, G, npad_h, G, tw), dtype=torch.float32, device=dev)
                for g in range(G):
                    wz_all[:, g, :, g, :] = wz_hm[g]
                wz_hm = wz_all.reshape(1, 9 * G * npad_h, G * tw).to(torch.bfloat16).contiguous()
                w_reg = torch.stack([torch.cat([s.w for s, _ in r], dim=1) for r in reg]).contiguous()   # [G, 9, kin_r, cin]
                sc_reg = torch.stack([torch.cat([s.scale for s, _ in r]) for r in reg]).contiguous()
                sh_reg = torch.stack([torch.cat([s.shift for s, _ in r]) for r in reg]).contiguous()
                wz_reg = torch.zeros((G, 9, npad_r, kin_r), dtype=torch.float32, device=dev)
                b_reg = torch.zeros((G, npad_r), dtype=torch.float32, device=dev)
                for g, r in enumerate(reg):
                    row = col = 0
                    for s_, o in r:
                        wz_reg[g, :, row:row + o.out_channels, col:col + s_.cout] = \
                            o.weight.detach().float().permute(2, 3, 0, 1).reshape(9, o.out_channels, s_.cout)
                        b_reg[g, row:row + o.out_channels] = o.bias.detach().float()
                        row += o.out_channels
                        col += s_.cout
                wz_reg = wz_reg.reshape(G, 1, 9 * npad_r, kin_r).to(torch.bfloat16).contiguous()
            self.levels[lvl] = dict(tasks=tasks, G=G, cin=hm_steps[0][0].cin, tw=tw, npad_h=npad_h, kin_r=kin_r,
                                    nout_r=nout_r, npad_r=npad_r, w_hm=w_hm, sc_hm=sc_hm, sh_hm=sh_hm, wz_hm=wz_hm,
                                    b_hm=b_hm.contiguous(), w_reg=w_reg, sc_reg=sc_reg, sh_reg=sh_reg, wz_reg=wz_reg,
                                    b_reg=b_reg.contiguous(),
                                    ones_h=torch.ones(G * 9 * npad_h, device=dev), zeros_h=torch.zeros(G * 9 * npad_h, device=dev),
                                    ones_r=torch.ones(9 * npad_r, device=dev), zeros_r=torch.zeros(9 * npad_r, device=dev))

    def run_hm(self, feats):
        """-> per level: (shared Rows, hm maps fp32 [rows, G, npad_h])."""
        L = _lib.lib()
        out = {}
        for lvl, p in self.levels.items():
            x = run_stack(self.share[lvl], feats[lvl])
            nbr, Ho, Wo = grid_rules(x.B, x.H, x.W, 1, x.t.device)
            rows, dev, G = x.B * Ho * Wo, x.t.device, p["G"]
            tow = torch.empty((rows, G * p["tw"]), dtype=torch.bfloat16, device=dev)
            _launch(x.t, p["cin"], nbr, 9, rows, p["w_hm"], p["sc_hm"], p["sh_hm"], True, tow, G * p["tw"], in_ld=x.ld)
            C = G * p["npad_h"]
            z = torch.empty((rows, 9 * C), dtype=torch.float32, device=dev)
            _launch(tow, G * p["tw"], None, 1, rows, p["wz_hm"], p["ones_h"], p["zeros_h"], False, z, 9 * C,
                    fp32_out=True, in_ld=G * p["tw"], useful=1.0 / G)
            hm = torch.empty((rows, G, p["npad_h"]), dtype=torch.float32, device=dev)     # task g: hm[:, g, :]
            rc = L.pn_tap_sum(_lib.ptr(z), _lib.PN_DTYPE_F32, 9 * C, 0, _lib.ptr(nbr), nbr.shape[1], rows, C, 1,
                              _lib.ptr(p["b_hm"]), _lib.ptr(hm), C, 0, _lib.current_stream(dev))
            _lib.check(rc, "pn_tap_sum")
            out[lvl] = (x, hm)
        return out

    def _per_set(self, lvl, set_task, dev):
        """Weights / scale / shift / bias replicated per set (a GEMM group is one set = task [x class] x sample);
        cached per replication pattern."""
        p = self.levels[lvl]
        key = ("sets", tuple(set_task))
        hit = p.get(key)
        if hit is None:
            n = len(set_task)
            if list(set_task) == list(range(p["G"])):
                hit = (p["w_reg"], p["sc_reg"], p["sh_reg"], p["wz_reg"], p["b_reg"])
            else:
                idx = torch.tensor(list(set_task), device=dev)
                hit = tuple(p[k][idx].contiguous() for k in ("w_reg", "sc_reg", "sh_reg", "wz_reg", "b_reg"))
            hit = hit + (p["ones_r"].repeat(n).contiguous(), p["zeros_r"].repeat(n).contiguous())
            p[key] = hit
        return hit

    def run_reg(self, lvl, x, set_task, nbr_tower, nbr_out, row_counts, cap):
        """Regression rows of one level's sets.  set_task[j]: index (within the level) of the task set j belongs
        to; nbr_tower [Ns, 9, 9*cap], nbr_out [Ns, 9, cap], row_counts [Ns] are that level's slices of what
        pn_centerhead_select wrote.  -> fp32 [Ns, cap, npad_r]."""
        L = _lib.lib()
        p = self.levels[lvl]
        dev = x.t.device
        Ns, R = len(set_task), 9 * cap
        w_reg, sc, sh, wz, b, ones, zeros = self._per_set(lvl, set_task, dev)
        kin, npad = p["kin_r"], p["npad_r"]
        st = _lib.current_stream(dev)
        tow = torch.empty((Ns, R, kin), dtype=torch.bfloat16, device=dev)
        if _lib.FLOP_LOG is not None:
            live = torch.arange(R, device=dev).view(1, 1, R) < row_counts.view(Ns, 1, 1)
            _lib.log_flops("lazy regression towers", p["cin"], kin, ((nbr_tower.view(Ns, 9, R) >= 0) & live).sum())
            _lib.log_flops("lazy per-tap products", kin, 9 * npad, row_counts.sum())
        rc = L.pn_gather_gemm_blocks(_lib.ptr(x.t), p["cin"], x.ld, x.t.shape[0], 0, _lib.ptr(nbr_tower), R, 9 * R, 9,
                                     _lib.ptr(row_counts), R, _lib.ptr(w_reg), kin, Ns, _lib.ptr(sc), _lib.ptr(sh), 1,
                                     _lib.ptr(tow), kin, R * kin, 0, st)
        _lib.check(rc, "pn_gather_gemm_blocks (towers)")
        nz = 9 * npad
        z = torch.empty((Ns, R, nz), dtype=torch.float32, device=dev)
        rc = L.pn_gather_gemm_blocks(_lib.ptr(tow), kin, kin, R, R * kin, None, 0, 0, 1, _lib.ptr(row_counts), R,
                                     _lib.ptr(wz), nz, Ns, _lib.ptr(ones), _lib.ptr(zeros), 0, _lib.ptr(z), nz, R * nz, 1,
                                     st)
        _lib.check(rc, "pn_gather_gemm_blocks (per-tap products)")
        reg = torch.empty((Ns, cap, npad), dtype=torch.float32, device=dev)
        rc = L.pn_tap_sum_blocks(_lib.ptr(z), _lib.PN_DTYPE_F32, nz, R * nz, _lib.ptr(nbr_out), cap, 9 * cap, cap, npad,
                                 Ns, _lib.ptr(b), _lib.ptr(reg), npad, cap * npad, st)
        _lib.check(rc, "pn_tap_sum_blocks")
        return reg
