"""ctypes binding of libpillarnet_hip.so (the C ABI in include/pillarnet_hip.h).

There is no CPU fallback: if the library is missing, `lib()` raises, and every
operator in this package goes through `lib()`.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpillarnet_hip.so")
LIB_PATH = os.environ.get("PN_LIB_PATH", LIB_PATH)   # tuning: A/B two builds inside one GPU job

PN_MAX_BATCH = 32
PN_DTYPE_F32, PN_DTYPE_BF16 = 0, 1


class PillarnetHipError(RuntimeError):
    pass


class pnPillarGrid(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("height", c_int32), ("width", c_int32),
        ("point_feats", c_int32), ("out_channels", c_int32),
        ("x_min", c_float), ("y_min", c_float), ("inv_pillar", c_float),
        ("pillar_size", c_float), ("x_offset", c_float), ("y_offset", c_float),
        ("true_divide", c_int32),
    ]


class pnDecodeGroup(ctypes.Structure):
    _fields_ = [
        ("maps", c_void_p), ("height", c_int32), ("width", c_int32), ("channels", c_int32),
        ("num_classes", c_int32), ("row_stride", c_int32), ("has_vel", c_int32), ("has_iou", c_int32),
        ("class_filter", c_int32), ("label_offset", c_int32), ("stride", c_int32),
        ("score_threshold", c_float), ("rectifier", c_float), ("use_rectify", c_int32),
        ("nms_threshold", c_float), ("pre_max", c_int32), ("post_max", c_int32),
    ]


class pnDecodeCommon(ctypes.Structure):
    _fields_ = [
        ("batch", c_int32), ("pillar_size", c_float), ("x_min", c_float), ("y_min", c_float),
        ("has_range", c_int32), ("range", c_float * 6),
    ]


# name -> (restype, argtypes); every symbol declared in include/pillarnet_hip.h
SIGNATURES = {
    "pn_abi_version": (c_int, []),
    "pn_last_error": (c_char_p, []),
    "pn_pillarize_workspace_bytes": (c_size_t, [POINTER(pnPillarGrid), c_int64]),
    "pn_pillarize": (c_int, [POINTER(pnPillarGrid), c_void_p, c_int64, POINTER(c_int32),
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_size_t, POINTER(c_void_p),
                             POINTER(c_void_p), c_void_p]),
    "pn_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pn_nms_normal": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_nms_rotated": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_float,
                               c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "pn_nms_rotated_static": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_float,
                                      c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pn_boxes_pairwise_bev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                      c_void_p]),
    "pn_spconv_subm_rules": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "pn_spconv_down_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pn_spconv_down_rules": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "pn_spconv_forward": (c_int, [c_int, c_void_p, c_int, ctypes.c_longlong, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p]),
    "pn_sparse_to_dense": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pn_occupancy_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pn_occupancy_from_indices": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_decode_workspace_bytes": (c_size_t, [POINTER(pnDecodeGroup), c_int, c_int, c_int]),
    "pn_centerhead_detect": (c_int, [POINTER(pnDecodeGroup), c_int, POINTER(pnDecodeCommon), c_int,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "pn_centerhead_select_capacity": (c_int, [POINTER(pnDecodeGroup), c_int]),
    "pn_centerhead_select": (c_int, [POINTER(pnDecodeGroup), c_int, POINTER(pnDecodeCommon), c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_centerhead_detect_selected": (c_int, [POINTER(pnDecodeGroup), c_int, POINTER(pnDecodeCommon), c_int,
                                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_void_p]),
    "pn_gather_gemm_blocks": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong, c_void_p, c_int,
                                      ctypes.c_longlong, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                      c_int, c_void_p, c_int, ctypes.c_longlong, c_int, c_void_p]),
    "pn_gather_gemm": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                               c_void_p, c_void_p]),
    "pn_gg_set_tile_policy": (c_int, [c_int]),
    "pn_gather_gemm_grouped": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                       c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                       c_void_p, ctypes.c_longlong, c_void_p]),
    "pn_tap_sum": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                           c_void_p, c_int, ctypes.c_longlong, c_void_p]),
    "pn_tap_sum_blocks": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_int, ctypes.c_longlong, c_int, c_int,
                                  c_int, c_void_p, c_void_p, c_int, ctypes.c_longlong, c_void_p]),
    "pn_create_point_pillar_index_stack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                                   c_void_p, c_void_p]),
    "pn_create_pillar_indices": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_gather_indice": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "pn_gather_feature": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "pn_gather_feature_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pn_scatter_max_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pn_scatter_max": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "pn_scatter_max_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pn_spconv_invert_rules": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "pn_spconv_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pn_spconv_wgrad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "pn_roi_grid_pool": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                 c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pn_roi_grid_pool_grad": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float,
                                      c_float, c_float, c_void_p, c_void_p]),
    "pn_points_in_boxes_2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pn_points_to_voxel": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "pn_merge_sweeps_workspace_bytes": (c_size_t, [c_int]),
    "pn_merge_sweeps": (c_int, [c_void_p, c_int, c_int, POINTER(c_int32), c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PillarnetHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc, gfx950). This package has no CPU fallback."
            )
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pn_last_error().decode("utf-8", "replace")
        raise PillarnetHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


# Flop accounting (bench.py's counting pass, off the clock): while this is a list, every matrix launch of the inference
# path appends (tag, flops per active (row, tap) pair = 2 * cin * cout, pairs) -- pairs an int or a 0-d device tensor
# counted on the launch's own rulebook, so that what is summed is the work this path issues on this frame.
FLOP_LOG = None


def log_flops(tag, cin, cout, pairs, useful=1.0):
    if FLOP_LOG is not None:
        FLOP_LOG.append((tag, 2.0 * cin * cout * useful, pairs))
