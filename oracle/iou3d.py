"""ctypes front end of the C oracle (oracle/iou3d_oracle.c) and, when built, of the compiled
reference (oracle/_ref/libref_iou3d_cpu.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "iou3d_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle_iou3d.so")

_F = ctypes.POINTER(ctypes.c_float)
_lib = None
_ref = None


def build(verbose=False):
    if os.path.exists(OUT) and os.path.getmtime(OUT) > os.path.getmtime(SRC):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", SRC, "-o", OUT, "-lm"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def lib():
    global _lib
    if _lib is None:
        h = ctypes.CDLL(build())
        h.pn_oracle_box_overlap.restype = ctypes.c_float
        h.pn_oracle_box_overlap.argtypes = [_F, _F]
        h.pn_oracle_iou_bev.restype = ctypes.c_float
        h.pn_oracle_iou_bev.argtypes = [_F, _F]
        h.pn_oracle_iou_matrix.restype = None
        h.pn_oracle_iou_matrix.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, _F, ctypes.c_int]
        h.pn_oracle_nms.restype = ctypes.c_int
        h.pn_oracle_nms.argtypes = [_F, ctypes.c_int, ctypes.c_float,
                                    ctypes.POINTER(ctypes.c_int64), _F]
        _lib = h
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 7, a.shape
    return a


def iou_matrix(a, b, overlap_only=False):
    a, b = _f32(a), _f32(b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().pn_oracle_iou_matrix(a.ctypes.data_as(_F), a.shape[0], b.ctypes.data_as(_F), b.shape[0],
                               out.ctypes.data_as(_F), int(overlap_only))
    return out


def nms(boxes, thresh):
    """boxes sorted by score; returns (keep positions int64, min |IoU - thresh| consulted)."""
    boxes = _f32(boxes) if len(boxes) else np.zeros((0, 7), np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    gap = ctypes.c_float(0)
    k = lib().pn_oracle_nms(boxes.ctypes.data_as(_F), n, float(thresh),
                            keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(gap))
    return keep[:k].copy(), float(gap.value)


def ref_lib():
    """The compiled reference, or None (not built / reference absent)."""
    global _ref
    if _ref is None:
        from . import build_ref
        path = build_ref.OUT if os.path.exists(build_ref.OUT) else None
        if path is None:
            return None
        import torch  # noqa: F401  (libtorch must be loaded first)
        h = ctypes.CDLL(path)
        h.ref_boxes_iou_bev_cpu.restype = ctypes.c_int
        h.ref_boxes_iou_bev_cpu.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, _F]
        h.ref_boxes_aligned_iou_bev_cpu.restype = ctypes.c_int
        h.ref_boxes_aligned_iou_bev_cpu.argtypes = [_F, _F, ctypes.c_int, _F]
        _ref = h
    return _ref


def ref_iou_matrix(a, b):
    h = ref_lib()
    if h is None:
        return None
    a, b = _f32(a), _f32(b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    h.ref_boxes_iou_bev_cpu(a.ctypes.data_as(_F), a.shape[0], b.ctypes.data_as(_F), b.shape[0],
                            out.ctypes.data_as(_F))
    return out
