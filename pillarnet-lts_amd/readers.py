"""READERS registry: `DynamicPFE` (det3d/models/readers/dynamic_pillar_encoder.py:9-50)."""
import torch.nn as nn

from .pillar_ops import PillarMaxPooling
from .registry import READERS


@READERS.register_module
class DynamicPFE(nn.Module):
    """Dynamic pillar feature encoder.  The reference computes per-sample cell coordinates and a
    range mask with ~10 torch kernels per sample, concatenates, and hands int coordinates to
    PillarMaxPooling; here the raw point tensors go straight into the fused HIP reader, which does
    all of that in its first pass over the points."""

    def __init__(self, in_channels=5, num_filters=(32,), pillar_size=0.1,
                 pc_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        assert len(num_filters) > 0
        self.pillar_size = pillar_size
        self.pc_range = pc_range
        # the two extra inputs are the point's offset from its pillar centre
        self.pfn_layers = PillarMaxPooling(mlps=[2 + in_channels] + list(num_filters),
                                           pillar_size=pillar_size, point_cloud_range=pc_range)
        self.height, self.width = self.pfn_layers.height, self.pfn_layers.width

    def forward(self, data, **kwargs):
        if not self.training:
            return self.pfn_layers(list(data["points"]))
        # training: the reference's own sequence (dynamic_pillar_encoder.py:30-50) on torch ops, then the
        # unfused, differentiable PillarMaxPooling
        import torch
        xy, cnt, feats = [], [], []
        for points in data["points"]:
            cx = ((points[:, 0] - self.pc_range[0]) / self.pillar_size).floor().int()
            cy = ((points[:, 1] - self.pc_range[1]) / self.pillar_size).floor().int()
            m = (cx >= 0) & (cx < self.width) & (cy >= 0) & (cy < self.height)
            xy.append(torch.stack((cx[m], cy[m]), dim=1))
            feats.append(points[m])
            cnt.append(int(m.sum()))
        xy = torch.cat(xy)
        cnt = xy.new_tensor(cnt, dtype=torch.int32)
        return self.pfn_layers.forward_unfused(xy, cnt, torch.cat(feats))
