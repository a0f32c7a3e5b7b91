"""GPU: training path (a13) -- reference-shaped pillar_cuda operators, scatter-max forward/backward,
sparse-conv data/weight gradients against torch autograd on the masked dense definition, and a
whole-detector training step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import pillar_oracle as po
from pillarnet_lts_amd import pillar_ops, presets, solver, sparse_conv as sc, synth, targets
from pillarnet_lts_amd.spconv import SparseConvTensor

pytestmark = pytest.mark.gpu


def test_pillar_cuda_surface_matches_oracle(cuda):
    """The seven reference-shaped operators, chained exactly as PillarQueryAndGroup.forward chains them."""
    grid = dict(pillar_size=0.5, pc_range=[-8, -8, -3, 8, 8, 1])
    rng = np.random.default_rng(0)
    pts = [rng.uniform(-9, 9, (1500, 5)).astype(np.float32), rng.uniform(-9, 9, (700, 5)).astype(np.float32)]
    xy, cnt, feats = po.dynamic_pfe_select(pts, grid["pc_range"], grid["pillar_size"])
    want_idx, want_ppi, want_f = po.pillar_query_and_group(xy, cnt, feats, grid["pillar_size"], grid["pc_range"])
    g = pillar_ops.PillarQueryAndGroup(grid["pillar_size"], grid["pc_range"])
    t = lambda a: torch.from_numpy(a).to(cuda)
    idx, ppi, f = pillar_ops.query_and_group(g, t(xy), t(cnt), t(feats))
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)
    np.testing.assert_array_equal(ppi.cpu().numpy(), want_ppi)
    np.testing.assert_allclose(f.cpu().numpy(), want_f, rtol=0, atol=1e-6)


def test_gather_feature_backward(cuda):
    """GatherFeature (group_utils.py:16-36) and pillar_cuda.gather_feature_grad_wrapper: rows gathered with
    repeats, gradient = scatter-add of the incoming rows"""
    from pillarnet_lts_amd import pillar_ops
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(37, 6, generator=g).to(cuda).requires_grad_()
    index = torch.randint(0, 37, (500,), generator=g).int().to(cuda)
    w = torch.randn(500, 6, generator=g).to(cuda)
    out = pillar_ops.gather_feature(feats, index)
    assert torch.equal(out, feats.detach()[index.long()])
    (out * w).sum().backward()
    want = torch.zeros(37, 6, device=cuda).index_add_(0, index.long(), w)
    np.testing.assert_allclose(feats.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    acc = torch.ones(37, 6, device=cuda)               # the raw wrapper accumulates into what it is given
    assert pillar_ops.pillar_cuda.gather_feature_grad_wrapper(w, index, acc) == 1
    np.testing.assert_allclose(acc.cpu().numpy(), (want + 1).cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_scatter_max_forward_backward(cuda):
    rng = np.random.default_rng(1)
    L, C, M = 5000, 32, 700
    src = rng.normal(size=(L, C)).astype(np.float32)
    idx = rng.integers(0, M - 50, L).astype(np.int32)          # the last 50 pillars stay empty
    want = po.scatter_max(src, idx, M)
    s = torch.from_numpy(src).to(cuda).requires_grad_(True)
    out = pillar_ops.scatter_max(s, torch.from_numpy(idx).to(cuda), M)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    go = torch.from_numpy(rng.normal(size=(M, C)).astype(np.float32)).to(cuda)
    out.backward(go)
    g = s.grad.cpu().numpy()
    # gradient lands on exactly one maximal point per (pillar, channel) with out > 0, nothing elsewhere
    gsum = np.zeros((M, C), np.float32)
    np.add.at(gsum, idx, g)
    np.testing.assert_allclose(gsum, np.where(want > 0, go.cpu().numpy(), 0), rtol=0, atol=0)
    nz = np.nonzero(g)
    assert np.all(src[nz] == want[idx[nz[0]], nz[1]])
    # deterministic argmax: lowest point index among ties
    src2 = np.ones((6, 32), np.float32)
    o2 = pillar_ops.scatter_max(torch.from_numpy(src2).to(cuda).requires_grad_(True),
                                torch.zeros(6, dtype=torch.int32, device=cuda), 1)
    s2 = torch.from_numpy(src2).to(cuda).requires_grad_(True)
    pillar_ops.scatter_max(s2, torch.zeros(6, dtype=torch.int32, device=cuda), 1).sum().backward()
    assert float(s2.grad[0].sum()) == 32 and float(s2.grad[1:].abs().sum()) == 0 and float(o2.detach().sum()) == 32


def _dense_reference(feats, idx, B, H, W, weight, bias, subm):
    """torch-CPU double autograd on the masked dense definition (oracle/sparse_oracle.dense_definition)."""
    f = torch.from_numpy(feats).double().requires_grad_(True)
    w = torch.from_numpy(weight).double().requires_grad_(True)
    b = torch.from_numpy(bias).double().requires_grad_(True) if bias is not None else None
    i = torch.from_numpy(idx).long()
    dense = torch.zeros((B, H, W, feats.shape[1]), dtype=torch.float64)
    dense = dense.index_put((i[:, 0], i[:, 1], i[:, 2]), f).permute(0, 3, 1, 2)
    mask = torch.zeros((B, 1, H, W), dtype=torch.float64)
    mask[i[:, 0], 0, i[:, 1], i[:, 2]] = 1
    wk = w.permute(0, 3, 1, 2)
    if subm:
        y = Fn.conv2d(dense, wk, b, 1, 1) * mask
        mo = mask
    else:
        mo = (Fn.max_pool2d(mask, 3, 2, 1) > 0).double()
        y = Fn.conv2d(dense, wk, b, 2, 1) * mo
    return f, w, b, y, mo


@pytest.mark.parametrize("subm,cin,cout", [(True, 32, 32), (True, 64, 64), (False, 32, 64), (False, 64, 128)])
def test_sparse_conv_gradients(cuda, subm, cin, cout):
    rng = np.random.default_rng(cin + cout)
    B, H, W, n = 2, 21, 18, 260
    cells = np.sort(rng.choice(B * H * W, n, replace=False))
    idx = np.stack((cells // (H * W), (cells // W) % H, cells % W), 1).astype(np.int32)
    feats = rng.normal(size=(n, cin)).astype(np.float32)
    cls = sc.SubMConv2d if subm else sc.SparseConv2d
    torch.manual_seed(1)
    conv = cls(cin, cout, 3, stride=1 if subm else 2, padding=1, bias=subm).to(cuda).train()
    x = SparseConvTensor(torch.from_numpy(feats).to(cuda).requires_grad_(True), torch.from_numpy(idx).to(cuda),
                         (H, W), B)
    y = conv(x)
    f, w, b, yref, mo = _dense_reference(feats, idx, B, H, W, conv.weight.detach().cpu().numpy(),
                                         conv.bias.detach().cpu().numpy() if subm else None, subm)
    yi = y.indices.cpu().long()
    got = y.features
    want = yref[yi[:, 0], :, yi[:, 1], yi[:, 2]]
    assert got.shape[0] == int(mo.sum())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-4)
    go = torch.from_numpy(rng.normal(size=tuple(got.shape)).astype(np.float32))
    got.backward(go.to(cuda))
    want.backward(go.double())
    np.testing.assert_allclose(x.features_cap.grad.cpu().numpy(), f.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-3)
    if subm:
        np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-3)


def _small_train_setup(cuda):
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    rng = [-12, -12, -5.0, 12, 12, 3.0]
    cfg.model.reader.pc_range = rng
    cfg.model.bbox_head.point_cloud_range = rng
    cfg.train_cfg.assigner.pc_range = rng
    model = presets.build(cfg).to(cuda).train()
    samples = [targets.synthetic_example(synth.nusc_sweep(5000, seed=s, r_max=14.0), cfg.model.bbox_head.tasks, rng,
                                         cfg.pillar_size, n_boxes=25, seed=s, train_cfg=cfg.train_cfg)
               for s in (1, 2)]
    return cfg, model, targets.collate(samples, cuda)


def test_detector_train_step(cuda):
    cfg, model, ex = _small_train_setup(cuda)
    opt = solver.TrueWdAdam(model.parameters(), wd=0.01)
    sch = solver.OneCycle(20, 1e-3)
    first, losses = solver.train_step(model, ex, opt, sch, 0)
    assert np.isfinite(first)
    assert set(losses) >= {"loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive", "reg_iou_loss"}
    assert len(losses["loss"]) == 6
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    nonzero = sum(float(p.grad.abs().sum()) > 0 for p in model.parameters())
    assert nonzero > 0.9 * sum(1 for _ in model.parameters())
    vals = [first]
    for step in range(1, 8):
        v, _ = solver.train_step(model, ex, opt, sch, step)
        vals.append(v)
    assert all(np.isfinite(vals))
    assert min(vals[-3:]) < vals[0], vals       # the same batch gets easier
    # and the trained weights still run through the inference path
    model.to_inference(torch.float32)
    with torch.no_grad():
        out = model({"points": ex["points"], "metadata": [{}, {}]}, return_loss=False)
    assert len(out) == 2 and out[0]["box3d_lidar"].shape[1] == 9


def test_reader_gradients_match_torch(cuda):
    """unfused reader: d(loss)/d(PFN weights) equals torch autograd on the oracle's grouping."""
    cfg, model, ex = _small_train_setup(cuda)
    reader = model.reader
    sp = reader({"points": ex["points"]})
    go = torch.randn_like(sp.features_cap)
    (sp.features_cap * go).sum().backward()
    lin = reader.pfn_layers.shared_mlps[0]
    got = lin.weight.grad.clone()
    # reference: same grouping from the oracle, MLP + segment max in torch
    pts = [p.cpu().numpy() for p in ex["points"]]
    xy, cnt, feats = po.dynamic_pfe_select(pts, cfg.model.reader.pc_range, cfg.pillar_size)
    idx, ppi, f7 = po.pillar_query_and_group(xy, cnt, feats, cfg.pillar_size, cfg.model.reader.pc_range)
    np.testing.assert_array_equal(sp.indices.cpu().numpy(), idx)
    lin.weight.grad = None
    y = reader.pfn_layers.shared_mlps(torch.from_numpy(f7).to(cuda))
    M = idx.shape[0]
    ref = torch.zeros((M, 32), device=cuda).scatter_reduce(0, torch.from_numpy(ppi).long().to(cuda)[:, None].expand(-1, 32),
                                                          y, "amax", include_self=True)
    (ref * go).sum().backward()
    torch.testing.assert_close(got, lin.weight.grad, rtol=1e-3, atol=1e-4)


# ---- configs[3] at full size: batch 4 x 300 000 points on the 1440^2 grid ---------------------------
def test_full_size_train_step_batch4(cuda):
    """One data-parallel training step at the configs[3] per-GPU shape (samples_per_gpu=4, 10 sweeps x
    30 000 points each): finite losses for all six tasks, a gradient on every parameter tensor, the loss
    falls on the repeated batch, and the step leaves BN statistics finite."""
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg).to(cuda).train()
    rng = cfg.model.reader.pc_range
    samples = [targets.synthetic_example(synth.nusc_frame(10, 30000, seed=200 + s), cfg.model.bbox_head.tasks, rng,
                                         cfg.pillar_size, n_boxes=40, seed=200 + s, train_cfg=cfg.train_cfg)
               for s in range(4)]
    ex = targets.collate(samples, cuda)
    assert len(ex["points"]) == 4 and all(p.shape == (300000, 5) for p in ex["points"])
    trainer = solver.DataParallelTrainer(model, total_steps=20)
    assert trainer.sync_bn is False                  # one process: plain BN
    n_tensors = sum(1 for _ in model.parameters())
    assert n_tensors == 348 and len(model.state_dict()) == 570
    vals = []
    for step in range(3):
        v, losses = trainer.step(ex, step)
        vals.append(float(v))
        if step == 0:
            assert len(losses["loss"]) == 6 and all(torch.isfinite(l).all() for l in losses["loss"])
            zero = [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
            assert not zero, zero[:5]
            nonzero = sum(float(p.grad.abs().sum()) > 0 for p in model.parameters())
            assert nonzero > 0.9 * n_tensors, (nonzero, n_tensors)
    assert all(np.isfinite(vals)) and vals[-1] < vals[0], vals
    assert all(torch.isfinite(b).all() for b in model.buffers())
    assert torch.cuda.max_memory_allocated() < 40 * 2 ** 30


def test_full_size_sparse_conv_gradient_on_a_sampled_window(cuda):
    """The sparse-conv data gradient on the full-size level-2 active set (the 720^2 grid of one 10-sweep
    frame, ~245 000 sites, 64 -> 64 SubM conv) against torch autograd on the masked dense definition,
    evaluated on a 48 x 48 window (plus a one-cell halo, which is all a 3x3 gradient can see)."""
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg).to(cuda).train()
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=200)).to(cuda)
    with torch.no_grad():
        sp = model.reader({"points": [pts]})
        lvl2 = model.backbone.conv2[0](model.backbone.conv1(sp)).exact()     # the strided conv's output sites
    idx = lvl2.indices
    n = idx.shape[0]
    assert n > 200000
    g = torch.Generator(device="cpu").manual_seed(9)
    feats = torch.randn(n, 64, generator=g).to(cuda).requires_grad_(True)
    conv = sc.SubMConv2d(64, 64, 3, padding=1, bias=True, indice_key="probe").to(cuda).train()
    x = SparseConvTensor(feats, idx.contiguous(), lvl2.spatial_shape, 1)
    y = conv(x)
    go = torch.randn(n, 64, generator=g).to(cuda)
    y.features.backward(go)
    # window around the ego vehicle, where the level is densest
    H, W = lvl2.spatial_shape
    y0, x0, S = H // 2 - 24, W // 2 - 24, 48
    iy, ix = idx[:, 1].long(), idx[:, 2].long()
    halo = ((iy >= y0 - 1) & (iy < y0 + S + 1) & (ix >= x0 - 1) & (ix < x0 + S + 1)).nonzero()[:, 0]
    inner = ((iy[halo] >= y0) & (iy[halo] < y0 + S) & (ix[halo] >= x0) & (ix[halo] < x0 + S)).cpu().numpy()
    assert inner.sum() > 1000
    sub_idx = idx[halo].cpu().numpy().copy()
    sub_idx[:, 1] -= y0 - 1
    sub_idx[:, 2] -= x0 - 1
    f, w, b, yref, mo = _dense_reference(feats.detach()[halo].cpu().numpy(), sub_idx, 1, S + 2, S + 2,
                                         conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(), True)
    si = torch.from_numpy(sub_idx).long()
    want_rows = yref[si[:, 0], :, si[:, 1], si[:, 2]]
    np.testing.assert_allclose(y.features.detach()[halo].cpu().numpy()[inner], want_rows.detach().numpy()[inner],
                               rtol=1e-4, atol=1e-4)
    # gradient: only the inner sites' outputs feed the reference loss through in-window taps, so restrict
    # grad_out to the halo'd patch and compare the inner sites' input gradients
    want_rows.backward(go[halo].cpu().double())
    got = feats.grad[halo].cpu().numpy()[inner]
    # inner sites see all their 3x3 neighbours inside the halo'd patch, so both sides sum the same terms
    np.testing.assert_allclose(got, f.grad.numpy()[inner], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cin,cout,rows,n_in,pad", [(32, 32, 5003, 4000, 32), (64, 32, 777, 900, 32), (32, 64, 1, 5, 32),
                                                    (64, 64, 20001, 20001, 32), (128, 256, 3000, 2500, 32),
                                                    (256, 256, 4097, 4097, 32), (128, 128, 9999, 7000, 32),
                                                    (64, 128, 2049, 2049, 32), (256, 128, 1500, 1500, 32),
                                                    (128, 128, 5000, 5000, 33), (256, 256, 3001, 3001, 1)])
def test_rulebook_weight_gradient_kernel(cuda, cin, cout, rows, n_in, pad):
    """pn_spconv_wgrad against the definition in float64: dW[co][tap][ci] = sum_m g[m][co] * x[nbr[tap][m]][ci],
    missing neighbours (-1) skipped, rows not a multiple of the chunking, row strides wider than the channels (`pad`
    odd: rows that are not 16-byte aligned, which the LDS-staged kernel of the wide layers hands to the per-tile one);
    two runs are bit-identical (fixed summation order)."""
    from pillarnet_lts_amd import sparse_conv
    gen = torch.Generator().manual_seed(cin * 1000 + cout + rows)
    x_full = torch.randn((n_in, cin + pad), generator=gen).to(cuda)
    g_full = torch.randn((rows, cout + 2 * pad), generator=gen).to(cuda)
    x, g = x_full[:, :cin], g_full[:, :cout]
    nbr = torch.randint(-n_in // 3, n_in, (9, rows + 11), generator=gen).clamp_(min=-1).int().to(cuda)
    got = sparse_conv.rulebook_wgrad(x, nbr, g, rows)
    assert got.shape == (cout, 9, cin)
    want = torch.zeros((cout, 9, cin), dtype=torch.float64, device=cuda)
    for k in range(9):
        idx = nbr[k, :rows].long()
        ok = idx >= 0
        want[:, k, :] = g[ok].double().t() @ x[idx[ok]].double()
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-5 * max(scale, 1.0)
    assert torch.equal(got, sparse_conv.rulebook_wgrad(x, nbr, g, rows))
    one = sparse_conv.rulebook_wgrad(x[:rows] if rows <= n_in else x_full[:0, :cin], None, g, rows) if rows <= n_in else None
    if one is not None:
        np.testing.assert_allclose(one[:, 0].cpu().numpy(), (g.double().t() @ x[:rows].double()).cpu().numpy(),
                                   rtol=0, atol=2e-5 * max(scale, 1.0))
