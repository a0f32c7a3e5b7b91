"""The C-ABI library loads without a GPU and exports exactly what include/pillarnet_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

from pillarnet_lts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pillarnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 14
    h = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_error_string():
    L = _lib.lib()
    assert L.pn_abi_version() == 1
    assert isinstance(L.pn_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before anything touches the device."""
    L = _lib.lib()
    g = _lib.pnPillarGrid()
    g.batch, g.height, g.width, g.point_feats, g.out_channels = 0, 10, 10, 5, 32
    offs = (ctypes.c_int32 * 2)(0, 0)
    rc = L.pn_pillarize(ctypes.byref(g), None, 0, offs, None, None, None, None, None, None, None, None, 0,
                        None, None, None)
    assert rc == -1 and b"null" in L.pn_last_error()
    assert L.pn_nms_workspace_bytes(0, 10) == 0
    assert L.pn_nms_workspace_bytes(2, 1000) >= 2 * 1000 * 16 * 8
    assert L.pn_spconv_forward(7, None, 32, 0, None, 0, None, 0, None, None, None, None, 0, None, 32, None) == -1


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.pnPillarGrid) == 12 * 4
    assert ctypes.sizeof(_lib.pnDecodeGroup) == 8 + 16 * 4       # pointer + 16 x 4-byte fields
    assert ctypes.sizeof(_lib.pnDecodeCommon) == 11 * 4
