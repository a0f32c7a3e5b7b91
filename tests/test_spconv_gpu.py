"""GPU parity of the sparse-conv rulebooks + implicit GEMM (through the C ABI) against the CPU
oracle (oracle/sparse_oracle.py; parity unpinned by the reference -- spconv is third-party).

Bars: active sets / (b,y,x) rows bit-exact.  fp32 features: |err| <= 1e-4 + 1e-4|ref| (north_star:
BEV maps within 1e-3 fp32).  bf16 features: relative L2 error <= 1e-2 vs the fp32 oracle fed the
same bf16-rounded inputs and weights."""
import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so
from pillarnet_lts_amd import sparse_conv as sc
from pillarnet_lts_amd.spconv import SparseConvTensor

pytestmark = pytest.mark.gpu


def _random_sparse(rng, B, H, W, C, n, extra_cap=0):
    cells = np.sort(rng.choice(B * H * W, n, replace=False))
    idx = np.stack((cells // (H * W), (cells // W) % H, cells % W), axis=1).astype(np.int32)
    feats = rng.normal(size=(n, C)).astype(np.float32)
    return idx, feats


def _to_dev(idx, feats, B, H, W, cuda, dtype, extra_cap=7):
    n = idx.shape[0]
    cap = n + extra_cap  # capacity > active rows, tail is garbage on purpose
    fi = torch.full((cap, 3), -12345, dtype=torch.int32, device=cuda)
    fi[:n] = torch.from_numpy(idx).to(cuda)
    ff = torch.full((cap, feats.shape[1]), float("nan"), dtype=dtype, device=cuda)
    ff[:n] = torch.from_numpy(feats).to(cuda).to(dtype)
    num = torch.tensor([n], dtype=torch.int32, device=cuda)
    return SparseConvTensor(ff, fi, (H, W), B, num)


def _conv(cls, cin, cout, bias, cuda, seed, **kw):
    torch.manual_seed(seed)
    m = cls(cin, cout, 3, bias=bias, **kw).to(cuda).eval()
    return m


def _bn(c, cuda, seed):
    g = torch.Generator().manual_seed(seed)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) * 0.4 + 0.8)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return bn.to(cuda).eval()


def _check(got, want, dtype):
    if dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    else:
        rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-12)
        assert rel <= 1e-2, rel


def _round(a, dtype):
    return torch.from_numpy(a).to(dtype).float().numpy()


CHANNELS = [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [c for c in CHANNELS if c[0] == c[1]])
def test_subm_conv(cuda, cin, cout, dtype):
    rng = np.random.default_rng(cin)
    B, H, W, n = 2, 40, 37, 900
    idx, feats = _random_sparse(rng, B, H, W, cin, n)
    feats = _round(feats, dtype)
    x = _to_dev(idx, feats, B, H, W, cuda, dtype)
    conv = _conv(sc.SubMConv2d, cin, cout, True, cuda, 1, padding=1, indice_key="k")
    y = conv(x)
    w = _round(conv.weight.detach().cpu().numpy(), dtype)
    want = so.subm_conv2d(feats, idx, (H, W), w, conv.bias.detach().cpu().numpy())
    got = y.features.float().cpu().numpy()
    assert got.shape == want.shape
    _check(got, want, dtype)
    assert "k" in x.indice_dict  # rulebook cached under the indice_key


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [c for c in CHANNELS if c[0] != c[1]])
def test_strided_conv(cuda, cin, cout, dtype):
    rng = np.random.default_rng(cout)
    B, H, W, n = 2, 41, 36, 700   # odd H exercises the floor in the output size
    idx, feats = _random_sparse(rng, B, H, W, cin, n)
    feats = _round(feats, dtype)
    x = _to_dev(idx, feats, B, H, W, cuda, dtype)
    conv = _conv(sc.SparseConv2d, cin, cout, False, cuda, 2, stride=2, padding=1)
    y = conv(x)
    w = _round(conv.weight.detach().cpu().numpy(), dtype)
    want, widx, shape = so.sparse_conv2d(feats, idx, (H, W), B, w, None, 2, 1)
    assert list(shape) == y.spatial_shape
    assert y.num_active_host() == widx.shape[0]
    np.testing.assert_array_equal(y.indices.cpu().numpy(), widx)
    _check(y.features.float().cpu().numpy(), want, dtype)
    # the new level carries a working occupancy index: a SubM conv on top of it must agree too
    conv2 = _conv(sc.SubMConv2d, cout, cout, True, cuda, 3, padding=1)
    z = conv2(y)
    yin = _round(y.features.float().cpu().numpy(), dtype)
    want2 = so.subm_conv2d(yin, widx, shape, _round(conv2.weight.detach().cpu().numpy(), dtype),
                           conv2.bias.detach().cpu().numpy())
    _check(z.features.float().cpu().numpy(), want2, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_bn_relu_residual(cuda, dtype):
    rng = np.random.default_rng(5)
    B, H, W, C, n = 1, 50, 50, 64, 1200
    idx, feats = _random_sparse(rng, B, H, W, C, n)
    feats = _round(feats, dtype)
    x = _to_dev(idx, feats, B, H, W, cuda, dtype)
    conv = _conv(sc.SubMConv2d, C, C, True, cuda, 4, padding=1)
    bn = _bn(C, cuda, 9)
    y = sc.conv_forward(x, conv, bn, relu=True, residual=x)
    w = _round(conv.weight.detach().cpu().numpy(), dtype)
    ref = so.subm_conv2d(feats, idx, (H, W), w, conv.bias.detach().cpu().numpy())
    ref = so.batchnorm_eval(ref, bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(),
                            bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), 1e-3)
    ref = np.maximum(ref + feats, 0)
    _check(y.features.float().cpu().numpy(), ref, dtype)
    # SparseSequential fuses the same chain
    seq = sc.SparseSequential(conv, bn, sc.SparseReLU()).eval()
    ref2 = so.batchnorm_eval(so.subm_conv2d(feats, idx, (H, W), w, conv.bias.detach().cpu().numpy()),
                             bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(),
                             bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), 1e-3)
    _check(seq(x).features.float().cpu().numpy(), np.maximum(ref2, 0), dtype)


def test_edge_cases(cuda):
    # empty tensor, single site, full grid, sites on the border
    for n, H, W in [(0, 8, 8), (1, 8, 8), (64, 8, 8), (5, 1, 9)]:
        rng = np.random.default_rng(n)
        idx, feats = _random_sparse(rng, 1, H, W, 32, n)
        x = _to_dev(idx, feats, 1, H, W, cuda, torch.float32)
        conv = _conv(sc.SubMConv2d, 32, 32, True, cuda, 1, padding=1)
        y = conv(x)
        assert y.num_active_host() == n
        if n:
            want = so.subm_conv2d(feats, idx, (H, W), conv.weight.detach().cpu().numpy(),
                                  conv.bias.detach().cpu().numpy())
            _check(y.features.cpu().numpy(), want, torch.float32)
        down = _conv(sc.SparseConv2d, 32, 64, False, cuda, 2, stride=2, padding=1)
        z = down(x)
        want, widx, shape = so.sparse_conv2d(feats, idx, (H, W), 1, down.weight.detach().cpu().numpy())
        assert z.num_active_host() == widx.shape[0]
        if widx.shape[0]:
            np.testing.assert_array_equal(z.indices.cpu().numpy(), widx)
            _check(z.features.cpu().numpy(), want, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("channels_last", [True, False])
def test_to_dense(cuda, dtype, channels_last):
    rng = np.random.default_rng(3)
    B, H, W, C, n = 2, 23, 19, 256, 300
    idx, feats = _random_sparse(rng, B, H, W, C, n)
    feats = _round(feats, dtype)
    x = _to_dev(idx, feats, B, H, W, cuda, dtype)
    d = x.dense(channels_last=channels_last)
    assert tuple(d.shape) == (B, C, H, W)
    np.testing.assert_array_equal(d.float().cpu().numpy(), so.to_dense(feats, idx, (H, W), B))


def test_full_size_level1_properties(cuda):
    """nuScenes-size level (1440^2, 32 ch): linearity and locality hold at any size."""
    from pillarnet_lts_amd import pillar_ops, synth
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(cuda)
    w = torch.randn(32, 7, device=cuda) * 0.5
    sp, _ = pillar_ops.pillarize([pts], w, torch.ones(32, device=cuda), torch.zeros(32, device=cuda),
                                 0.075, [-54, -54, -5.0, 54, 54, 3.0])
    conv = _conv(sc.SubMConv2d, 32, 32, False, cuda, 1, padding=1, indice_key="res1")
    y1 = conv(sp).features_cap
    M = sp.num_active_host()
    sp2 = sp.replace_feature(sp.features_cap * 2.0)
    y2 = conv(sp2).features_cap
    torch.testing.assert_close(y2[:M], 2.0 * y1[:M], rtol=1e-5, atol=1e-5)
    # centre tap only => identity map on the features
    with torch.no_grad():
        conv.weight.zero_()
        conv.weight[:, 1, 1, :] = torch.eye(32, device=cuda)
    y3 = conv(sp).features_cap
    torch.testing.assert_close(y3[:M], sp.features_cap[:M], rtol=0, atol=0)
    # spot-check 2000 rows against the oracle
    idx = sp.indices.cpu().numpy()
    feats = sp.features.cpu().numpy()
    torch.manual_seed(0)
    with torch.no_grad():
        conv.weight.copy_(torch.randn_like(conv.weight) * 0.1)
    got = conv(sp).features.cpu().numpy()
    want = so.subm_conv2d(feats, idx, (1440, 1440), conv.weight.detach().cpu().numpy())
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
