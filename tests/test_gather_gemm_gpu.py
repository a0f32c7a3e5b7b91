"""pn_gather_gemm (both kernels: window-staged for 3x3 rulebooks, per-tap for the rest) against the
fp32 definition out[m] = sum_tap W[tap] . in[nbr[tap][m]] on arbitrary rulebooks."""
import numpy as np
import pytest
import torch

from pillarnet_lts_amd import _lib, dense_hip

pytestmark = pytest.mark.gpu


def reference(x, nbr, w, scale, shift, residual, relu, M):
    """x [R,cin] bf16, nbr [T,cap] int32, w [T,cout,cin] bf16 -> fp32 [M,cout]."""
    xf, wf = x.float(), w.float()
    out = torch.zeros((M, w.shape[1]), dtype=torch.float32, device=x.device)
    for t in range(w.shape[0]):
        idx = nbr[t, :M].long()
        g = xf[idx.clamp(min=0)] * (idx >= 0).unsqueeze(1)
        out += g @ wf[t].t()
    out = out * scale + shift
    if residual is not None:
        out = out + residual[:M].float()
    return torch.relu(out) if relu else out


def call(x, nbr, w, scale, shift, rows, count=None, residual=None, relu=True, fp32_out=False, out=None,
         out_rows=None):
    cout, cin = w.shape[1], w.shape[2]
    if out is None:
        out = torch.full((rows, cout), float("nan"), dtype=torch.float32 if fp32_out else torch.bfloat16,
                         device=x.device)
    rc = _lib.lib().pn_gather_gemm(_lib.ptr(x), cin, x.stride(0), x.shape[0], _lib.ptr(nbr), nbr.shape[1], w.shape[0],
                                   _lib.ptr(count), rows, _lib.ptr(w), cout, _lib.ptr(scale), _lib.ptr(shift),
                                   _lib.ptr(residual), 0 if residual is None else residual.stride(0), int(relu),
                                   _lib.ptr(out), out.stride(0), int(fp32_out), _lib.ptr(out_rows),
                                   _lib.current_stream(x.device))
    _lib.check(rc, "pn_gather_gemm")
    return out


def sorted_sparse_rules(H, W, density, seed, dev):
    """SubM rulebook of a random sparse H x W map with rows in flat-index order (what the backbone has)."""
    g = torch.Generator().manual_seed(seed)
    occ = torch.rand(H, W, generator=g) < density
    # a dense stripe next to a sparse one: runs of neighbouring image rows longer than the tile's
    occ[H // 2] = True
    occ[H // 2 + 1, : W // 8] = True
    pos = torch.full((H, W), -1, dtype=torch.int64)
    M = int(occ.sum())
    pos[occ] = torch.arange(M)
    ys, xs = torch.nonzero(occ, as_tuple=True)
    nbr = torch.full((9, M), -1, dtype=torch.int32)
    for k in range(9):
        yy, xx = ys + k // 3 - 1, xs + k % 3 - 1
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = pos[yy.clamp(0, H - 1), xx.clamp(0, W - 1)]
        nbr[k] = torch.where(ok, v, torch.full_like(v, -1)).int()
    return nbr.to(dev), M


def _mk(cin, cout, R, dev, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(9, cout, cin, generator=g) / (3 * cin ** 0.5)).to(dev).bfloat16()
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
    return x, w, scale, shift


def _check(got, want, M):
    got = got[:M].float()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= 1e-2 * ref + 2e-2, (err, ref)       # bf16 output rounding of O(1) values


CASES = [(32, 32), (32, 64), (64, 64), (64, 32), (128, 128), (256, 256), (512, 256), (256, 64), (64, 320), (128, 256)]


@pytest.mark.parametrize("cin,cout", CASES)
def test_dense_grid_rulebook(cuda, cin, cout):
    B, H, W = 2, 37, 41
    nbr, Ho, Wo = dense_hip.grid_rules(B, H, W, 1, cuda)
    rows = B * Ho * Wo
    x, w, scale, shift = _mk(cin, cout, rows, cuda, cin * 7 + cout)
    res = torch.randn(rows, cout, device=cuda).bfloat16()
    got = call(x, nbr, w, scale, shift, rows, residual=res)
    _check(got, reference(x, nbr, w, scale, shift, res, True, rows), rows)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (128, 128), (256, 256)])
def test_sorted_sparse_rulebook_with_device_count(cuda, cin, cout):
    nbr, M = sorted_sparse_rules(90, 700, 0.25, 3, cuda)
    cap = M + 300                                  # capacity > active rows; the count lives on the device
    nbr_cap = torch.full((9, cap), 123456, dtype=torch.int32, device=cuda)   # garbage past the count is never read
    nbr_cap[:, :M] = nbr
    count = torch.tensor([M], dtype=torch.int32, device=cuda)
    x, w, scale, shift = _mk(cin, cout, cap, cuda, 11)
    got = call(x, nbr_cap, w, scale, shift, cap, count=count, relu=False)
    _check(got, reference(x, nbr_cap, w, scale, shift, None, False, M), M)
    assert torch.isnan(got[M:].float()).all()      # rows past the count are not written


@pytest.mark.parametrize("cin,cout", [(64, 64), (256, 256), (128, 64), (32, 32), (32, 64)])
def test_unsorted_rulebook_takes_gather_mode(cuda, cin, cout):
    """A rulebook without any row locality (random input rows, 30 % missing): every tile's runs exceed the
    window, so the kernel's gather mode must produce the same numbers."""
    g = torch.Generator().manual_seed(5)
    rows, R = 1000, 5000
    nbr = torch.randint(0, R, (9, rows), generator=g).int()
    nbr[torch.rand(9, rows, generator=g) < 0.3] = -1
    nbr = nbr.to(cuda)
    x, w, scale, shift = _mk(cin, cout, R, cuda, 13)
    got = call(x, nbr, w, scale, shift, rows, fp32_out=True)
    want = reference(x, nbr, w, scale, shift, None, True, rows)
    assert (got - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())


def test_strided_rulebook_scatter_rows_and_row_stride(cuda):
    """stride-2 grid rulebook, input rows wider than cin (channel slice of a concat buffer), fp32 output
    scattered through out_rows into a wider output."""
    B, H, W, cin, cout = 1, 40, 52, 64, 128
    nbr, Ho, Wo = dense_hip.grid_rules(B, H, W, 2, cuda)
    rows = B * Ho * Wo
    g = torch.Generator().manual_seed(9)
    wide = torch.randn(B * H * W, 96, generator=g).to(cuda).bfloat16()
    x = wide[:, 32:96]
    _, w, scale, shift = _mk(cin, cout, 8, cuda, 17)
    perm = torch.randperm(rows, generator=g).int().to(cuda)
    out = torch.zeros((rows, 160), dtype=torch.float32, device=cuda)
    call(x, nbr, w, scale, shift, rows, fp32_out=True, out=out[:, 16:144], out_rows=perm)
    want = reference(x.contiguous(), nbr, w, scale, shift, None, True, rows)
    got = out[perm.long(), 16:144]
    assert (got - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
    assert out[:, :16].abs().max().item() == 0 and out[:, 144:].abs().max().item() == 0


def test_all_neighbours_missing_and_tiny_problems(cuda):
    x, w, scale, shift = _mk(64, 64, 16, cuda, 19)
    nbr = torch.full((9, 5), -1, dtype=torch.int32, device=cuda)
    got = call(x, nbr, w, scale, shift, 5, relu=False, fp32_out=True)
    np.testing.assert_allclose(got.cpu().numpy(), np.broadcast_to(shift.cpu().numpy(), (5, 64)), atol=1e-6)
    nbr[4, :] = torch.arange(5, dtype=torch.int32)          # centre tap only
    got = call(x, nbr, w, scale, shift, 5, relu=False, fp32_out=True)
    want = reference(x, nbr, w, scale, shift, None, False, 5)
    assert (got - want).abs().max().item() <= 1e-3
    count0 = torch.zeros(1, dtype=torch.int32, device=cuda)
    got = call(x, nbr, w, scale, shift, 5, count=count0)
    assert torch.isnan(got.float()).all()                    # zero active rows: nothing written, no fault


_MODE_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import torch
from pillarnet_lts_amd import dense_hip
from test_gather_gemm_gpu import call, _mk, sorted_sparse_rules
dev = torch.device("cuda:0")
h = hashlib.sha256()
for (cin, cout) in [(64, 64), (128, 128), (256, 256)]:
    nbr, Ho, Wo = dense_hip.grid_rules(2, 33, 47, 1, dev)
    x, w, scale, shift = _mk(cin, cout, 2 * 33 * 47, dev, 5)
    h.update(call(x, nbr, w, scale, shift, 2 * Ho * Wo).cpu().view(torch.int16).numpy().tobytes())
    nbr, M = sorted_sparse_rules(60, 500, 0.3, 7, dev)
    x, w, scale, shift = _mk(cin, cout, M, dev, 6)
    h.update(call(x, nbr, w, scale, shift, M, relu=False).cpu().view(torch.int16).numpy().tobytes())
print(h.hexdigest())
"""


def test_window_and_gather_modes_are_bit_identical(cuda):
    """The window kernel's two staging modes walk (kernel row, channel chunk, kx) in the same order, so forcing
    every tile into gather mode must not change a single output bit (fresh process per mode: the switch is read
    once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _MODE_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    digests = {}
    # (and a second tile shape / weight-ring depth: the accumulation order does not depend on the tiling)
    for name, env in (("window", {}), ("gather", {"PN_GG_WIN_GATHER": "1"}), ("tiles128", {"PN_GG_WIN_CFG": "T"})):
        out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[name] = out.stdout.strip().splitlines()[-1]
    assert digests["window"] == digests["gather"] == digests["tiles128"], digests


@pytest.mark.parametrize("cin,cout,fp32_out", [(256, 256, False), (384, 112, True), (64, 128, False), (128, 320, False)])
def test_one_tap_gemm_identity_rulebook_and_scatter(cuda, cin, cout, fp32_out):
    """ntaps == 1: a plain GEMM over the rows (rulebook NULL = identity, or an explicit row map), optionally
    scattered through out_rows -- the head's per-tap products and the transposed convs' parity GEMMs."""
    g = torch.Generator().manual_seed(cin + cout)
    R, rows = 3000, 2900
    x = torch.randn(R, cin, generator=g).to(cuda).bfloat16()
    w = (torch.randn(1, cout, cin, generator=g) / cin ** 0.5).to(cuda).bfloat16()
    scale = (torch.rand(cout, generator=g) + 0.5).to(cuda)
    shift = (torch.randn(cout, generator=g) * 0.1).to(cuda)
    ident = torch.arange(rows, dtype=torch.int32, device=cuda).view(1, rows)
    want = reference(x, ident, w, scale, shift, None, True, rows)
    tol = 2e-3 if fp32_out else 2e-2
    out = torch.full((rows, cout), float("nan"), dtype=torch.float32 if fp32_out else torch.bfloat16, device=cuda)
    rc = _lib.lib().pn_gather_gemm(_lib.ptr(x), cin, cin, R, None, 0, 1, None, rows, _lib.ptr(w), cout, _lib.ptr(scale),
                                   _lib.ptr(shift), None, 0, 1, _lib.ptr(out), cout, int(fp32_out), None,
                                   _lib.current_stream(cuda))
    _lib.check(rc, "pn_gather_gemm")
    assert (out.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    # explicit (shuffled) rulebook + scattered output rows
    perm = torch.randperm(R, generator=g)[:rows].int().to(cuda).view(1, rows)
    dest = torch.randperm(rows, generator=g).int().to(cuda)
    got = call(x, perm, w, scale, shift, rows, fp32_out=fp32_out, out_rows=dest)
    want = reference(x, perm, w, scale, shift, None, True, rows)
    assert (got[dest.long()].float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
