"""pillarnet-lts_amd: the MI355X-native pillar-detection forward path behind the det3d
registry / operator API of VISION-SJTU/PillarNet-LTS."""
from . import _lib  # noqa: F401
from .spconv import SparseConvTensor  # noqa: F401

__all__ = ["SparseConvTensor"]
