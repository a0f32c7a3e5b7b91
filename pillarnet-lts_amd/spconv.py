"""Our stand-in for the subset of `spconv.pytorch` (spconv 2.x, third-party, not vendored by the
reference) that the hot path uses: `SparseConvTensor` here; the conv modules live in
`sparse_conv.py`.  Call sites being served: det3d/ops/pillar_ops/pillar_modules.py:74,
det3d/models/backbones/base.py:9-10,139-140, det3d/models/backbones/PillarResNet.py:139,
det3d/models/necks/rpn.py:196-199.

MI355X-first difference: the number of active sites is a *device* int32 (`num_active`), tensors
are allocated at a host-known capacity, and the first `num_active` rows are valid.  Nothing on the
inference path needs the count on the host, so a frame queues without a sync; `.features` /
`.indices` (the spconv attributes, exact row count) sync lazily only if somebody asks.
"""
import torch


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, num_active=None,
                 occ_bits=None, occ_rank=None, keepalive=None):
        """features [cap, C], indices [cap, 3] int32 (b, y, x) sorted by flat index.
        num_active: device int32[1] (None: every row is active, as in spconv)."""
        self.features_cap = features
        self.indices_cap = indices
        self.spatial_shape = list(spatial_shape)
        self.batch_size = int(batch_size)
        if num_active is None:
            num_active = torch.tensor([features.shape[0]], dtype=torch.int32, device=features.device)
        self.num_active = num_active
        self._occ_bits = occ_bits     # uint64 words viewed as int64 [ceil(B*H*W/64)]
        self._occ_rank = occ_rank     # int32 [words]
        self.indice_dict = {}         # indice_key -> rulebook (spconv's cache, base.py indice_key)
        self._keepalive = keepalive
        self._n_host = None

    def _build_occupancy(self):
        """Occupancy index for a tensor built spconv-style from (features, indices) only."""
        from . import _lib
        idx = self.indices_cap
        if idx.device.type != "cuda":
            raise _lib.PillarnetHipError("SparseConvTensor: indices must live on the GPU")
        if idx.dtype != torch.int32 or not idx.is_contiguous():
            self.indices_cap = idx = idx.to(torch.int32).contiguous()
        H, W = self.spatial_shape
        B = self.batch_size
        words = (B * H * W + 63) // 64
        L = _lib.lib()
        bits = torch.empty(words, dtype=torch.int64, device=idx.device)
        rank = torch.empty(words, dtype=torch.int32, device=idx.device)
        cells = torch.zeros(1, dtype=torch.int32, device=idx.device)
        sb = L.pn_occupancy_scratch_bytes(B, H, W)
        scratch = torch.empty(sb, dtype=torch.uint8, device=idx.device)
        rc = L.pn_occupancy_from_indices(_lib.ptr(idx), _lib.ptr(self.num_active), self.capacity, B,
                                         H, W, _lib.ptr(bits), _lib.ptr(rank), _lib.ptr(cells),
                                         _lib.ptr(scratch), sb, _lib.current_stream(idx.device))
        _lib.check(rc, "pn_occupancy_from_indices")
        self._occ_bits, self._occ_rank = bits, rank

    @property
    def occ_bits(self):
        if self._occ_bits is None:
            self._build_occupancy()
        return self._occ_bits

    @property
    def occ_rank(self):
        if self._occ_rank is None:
            self._build_occupancy()
        return self._occ_rank

    # -- spconv API ---------------------------------------------------------------------------
    @property
    def capacity(self):
        # (index-only tensors -- features None -- exist while the rulebooks are prefetched)
        return (self.features_cap if self.features_cap is not None else self.indices_cap).shape[0]

    def num_active_host(self):
        if self._n_host is None:
            self._n_host = int(self.num_active.item())  # the one sync, only on request
        return self._n_host

    @property
    def features(self):
        return self.features_cap[: self.num_active_host()]

    @property
    def indices(self):
        return self.indices_cap[: self.num_active_host()]

    def exact(self):
        """Same tensor with capacity == active rows (one host sync); the training path works on
        exact tensors so that BatchNorm statistics and autograd see only real rows."""
        n = self.num_active_host()
        if n == self.capacity:
            return self
        out = SparseConvTensor(self.features_cap[:n], self.indices_cap[:n].contiguous(), self.spatial_shape,
                               self.batch_size, self.num_active, self._occ_bits, self._occ_rank, self._keepalive)
        out._n_host = n
        return out

    def replace_feature(self, new_features):
        """spconv 2.x `replace_feature` (used at base.py:9-10): same sites, new feature rows."""
        if new_features.shape[0] != self.capacity:
            assert new_features.shape[0] == self.num_active_host()
            full = self.features_cap.new_empty((self.capacity, new_features.shape[1]),
                                               dtype=new_features.dtype)
            full[: new_features.shape[0]] = new_features
            new_features = full
        out = SparseConvTensor(new_features, self.indices_cap, self.spatial_shape, self.batch_size,
                               self.num_active, self._occ_bits, self._occ_rank, self._keepalive)
        out.indice_dict = self.indice_dict
        out._n_host = self._n_host
        return out

    def dense(self, channels_last=True, out_channels=None):
        """[B, C, H, W]; zero where inactive (spconv `.dense()`, PillarResNet.py:139).  The memory
        format is channels-last (each active row lands as one contiguous C-vector, and the MIOpen
        convs downstream want NHWC)."""
        from . import sparse_conv
        if self.features_cap.requires_grad:
            assert out_channels is None
            x = self.exact()
            return sparse_conv.ToDenseFunction.apply(x.features_cap, x)
        return sparse_conv.to_dense(self, channels_last=channels_last, out_channels=out_channels)
