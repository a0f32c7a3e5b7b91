"""Checkpoint load / save with the reference's file contract
(det3d/torchie/trainer/checkpoint.py:49-64 spconv key discovery, :67-137 load_state_dict,
:166-217 load_checkpoint, :220-260 weights_to_cpu / save_checkpoint; resume:
det3d/torchie/trainer/trainer.py:342-356,496-509).

A checkpoint file is `{"meta": dict, "state_dict": OrderedDict of CPU tensors, "optimizer": dict}`
(or a bare state dict).  Files are read with `torch.load(..., weights_only=True)` only: nothing a
file carries is executed.  Sparse-conv weights saved by spconv 1.x (`[kh, kw, Cin, Cout]`) are
re-laid-out to this package's spconv-2.x layout (`[Cout, kh, kw, Cin]`) by shape matching, exactly
where the reference does it; URL schemes (modelzoo://, http://) are not supported offline.
"""
import os
import os.path as osp
import time
from collections import OrderedDict

import torch
import torch.nn as nn


def find_all_spconv_keys(model: nn.Module, prefix=""):
    """Names of every sparse-conv weight in `model` (checkpoint.py:49-64)."""
    from .sparse_conv import SparseConvolution
    found = set()
    for name, child in model.named_children():
        path = f"{prefix}.{name}" if prefix else name
        if isinstance(child, SparseConvolution):
            found.add(f"{path}.weight")
        found |= find_all_spconv_keys(child, prefix=path)
    return found


def _relayout_spconv_weight(param, want_shape):
    """spconv 1.x weight -> the layout of `want_shape`, or `param` unchanged when nothing fits.
    (checkpoint.py:79-91; that code asserts 5-D, i.e. 3-D convs -- the 2-D convs of the pillar
    backbones are 4-D, so the same two candidates are tried for any rank.)"""
    native = param.transpose(-1, -2)                      # [..., Cin, Cout] -> [..., Cout, Cin]
    if native.shape == want_shape:
        return native.contiguous()
    nd = param.dim()
    implicit = param.permute(nd - 1, *range(nd - 1))      # [k.., Cin, Cout] -> [Cout, k.., Cin]
    if implicit.shape == want_shape:
        return implicit.contiguous()
    return param


def _table(rows):
    widths = [max(len(str(r[i])) for r in rows) for i in range(len(rows[0]))]
    line = "+" + "+".join("-" * (w + 2) for w in widths) + "+"
    out = [line]
    for n, r in enumerate(rows):
        out.append("| " + " | ".join(str(c).ljust(w) for c, w in zip(r, widths)) + " |")
        if n == 0:
            out.append(line)
    out.append(line)
    return "\n".join(out)


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def load_state_dict(module, state_dict, strict=False, logger=None):
    """Copy `state_dict` into `module` key by key (checkpoint.py:67-137): unknown keys and shape
    mismatches are collected, not fatal, unless `strict`; `num_batches_tracked` may be missing.
    Returns (missing_keys, unexpected_keys, shape_mismatch_pairs)."""
    unexpected, mismatched = [], []
    own = module.state_dict()
    spconv_keys = find_all_spconv_keys(module)
    with torch.no_grad():
        for name, param in state_dict.items():
            if isinstance(param, nn.Parameter):
                param = param.data
            if name in spconv_keys and name in own and own[name].shape != param.shape:
                param = _relayout_spconv_weight(param, own[name].shape)
            if name not in own:
                unexpected.append(name)
                continue
            if param.size() != own[name].size():
                mismatched.append([name, tuple(own[name].size()), tuple(param.size())])
                continue
            own[name].copy_(param)
    missing = sorted(k for k in set(own) - set(state_dict) if "num_batches_tracked" not in k)

    msg = []
    if unexpected:
        msg.append("unexpected key in source state_dict: {}\n".format(", ".join(unexpected)))
    if missing:
        msg.append("missing keys in source state_dict: {}\n".format(", ".join(missing)))
    if mismatched:
        msg.append("these keys have mismatched shape:\n" +
                   _table([["key", "expected shape", "loaded shape"]] + mismatched))
    if msg and _rank() == 0:
        text = "\n".join(["The model and loaded state dict do not match exactly\n"] + msg)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)
    return missing, unexpected, mismatched


def _read(filename, map_location):
    if filename.startswith(("modelzoo://", "torchvision://", "open-mmlab://", "http://", "https://")):
        raise IOError(f"{filename}: URL checkpoints are not supported (no network); pass a file path")
    if not osp.isfile(filename):
        raise IOError("{} is not a checkpoint file".format(filename))
    return torch.load(filename, map_location=map_location or "cpu", weights_only=True)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    """checkpoint.py:166-217: read, find the state dict, strip a `module.` prefix (DDP saves),
    load non-strictly.  Returns the checkpoint dict (meta / optimizer for `resume`)."""
    checkpoint = _read(filename, map_location)
    if isinstance(checkpoint, OrderedDict) and "state_dict" not in checkpoint:
        state_dict = checkpoint        # any OrderedDict is a bare state dict to the reference (checkpoint.py:203-204)
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    elif isinstance(checkpoint, dict) and checkpoint and "meta" not in checkpoint and "optimizer" not in checkpoint \
            and any(torch.is_tensor(v) for v in checkpoint.values()):
        # a bare state dict (the reference takes any OrderedDict, checkpoint.py:194-197): entries that are not tensors
        # (a version counter some exporters add) are left to load_state_dict's unexpected-key report
        state_dict = OrderedDict((k, v) for k, v in checkpoint.items() if torch.is_tensor(v))
    else:
        raise RuntimeError("No state_dict found in checkpoint file {}".format(filename))
    keys = list(state_dict.keys())
    if keys and keys[0].startswith("module."):
        state_dict = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
    target = model.module if hasattr(model, "module") else model
    load_state_dict(target, state_dict, strict, logger)
    return checkpoint


def weights_to_cpu(state_dict):
    return OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())


def _plain(obj, strict=False, where="meta"):
    """Config dicts / tuples / numpy values -> plain containers and Python numbers, so that the file loads under
    weights_only.  numpy scalars become Python numbers and numpy arrays tensors (an optimiser's hyper-parameters
    must come back as numbers).  What is left over: `strict` raises (optimiser state -- a value silently turned into
    text would break the optimiser a step after resume), otherwise it is stored as its text (free-form `meta`)."""
    import numpy as np
    if isinstance(obj, dict):
        return {(k if isinstance(k, (str, int)) else str(k)): _plain(v, strict, where) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v, strict, where) for v in obj]
    if isinstance(obj, np.generic):           # (before the Python numbers: numpy.float64 is a subclass of float)
        return obj.item()
    if isinstance(obj, (str, int, float, bool)) or obj is None or torch.is_tensor(obj):
        return obj
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(obj))
    if strict:
        raise TypeError(f"{where}: cannot store a value of type {type(obj).__name__} in a checkpoint")
    return str(obj)


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """checkpoint.py:235-260: `{meta, state_dict (CPU), optimizer}`; meta gets a time stamp as
    Trainer.save_checkpoint adds one (trainer.py:342-356)."""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError("meta must be a dict or None, but got {}".format(type(meta)))
    meta = dict(_plain(meta), time=time.asctime())
    d = osp.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    if hasattr(model, "module"):
        model = model.module
    checkpoint = {"meta": meta, "state_dict": weights_to_cpu(model.state_dict())}
    if optimizer is not None:
        checkpoint["optimizer"] = _plain(optimizer.state_dict(), strict=True, where="optimizer state")
    torch.save(checkpoint, filename)


def save_epoch_checkpoint(model, out_dir, epoch, iteration, optimizer=None, meta=None,
                          filename_tmpl="epoch_{}.pth"):
    """Trainer.save_checkpoint (trainer.py:342-356): epoch_N.pth + a `latest.pth` symlink."""
    meta = dict(meta or {}, epoch=epoch, iter=iteration)
    name = filename_tmpl.format(epoch)
    path = osp.join(out_dir, name)
    save_checkpoint(model, path, optimizer=optimizer, meta=meta)
    link = osp.join(out_dir, "latest.pth")
    if osp.lexists(link):
        os.remove(link)
    os.symlink(name, link)
    return path


def resume(model, filename, optimizer=None, map_location="cpu", logger=None):
    """Trainer.resume (trainer.py:496-509): weights, then epoch / iter and the optimiser state.
    Returns (epoch, iter)."""
    checkpoint = load_checkpoint(model, filename, map_location=map_location, strict=False, logger=logger)
    meta = checkpoint.get("meta", {}) if isinstance(checkpoint, dict) else {}
    if optimizer is not None and "optimizer" in checkpoint:
        optimizer.load_state_dict(checkpoint["optimizer"])
    return int(meta.get("epoch", 0)), int(meta.get("iter", 0))
