"""pillarnet-lts_amd: the MI355X-native pillar-detection forward path behind the det3d
registry / operator API of VISION-SJTU/PillarNet-LTS."""
from . import _lib  # noqa: F401
from .spconv import SparseConvTensor  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .registry import (BACKBONES, DETECTORS, HEADS, NECKS, READERS, Registry,  # noqa: F401
                       build_from_cfg)
from . import readers, backbones, necks, bbox_heads, detectors, second_stage, loading, legacy  # noqa: F401  (fill the registries)
from .builder import (build_backbone, build_detector, build_head, build_neck,  # noqa: F401
                      build_reader)
from .checkpoint import (load_checkpoint, load_state_dict, resume, save_checkpoint,  # noqa: F401
                         save_epoch_checkpoint)
from .wire import collate_kitti  # noqa: F401
from . import solver, wire  # noqa: F401
from .frame_graph import FrameGraphs  # noqa: F401
