"""Micro-benchmark of pn_gather_gemm at the layer shapes of PillarNet-18 / nuScenes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pillarnet_lts_amd import _lib, presets, sparse_conv, synth  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def grid_rules(H, W, stride=1):
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    oy, ox = torch.meshgrid(torch.arange(Ho), torch.arange(Wo), indexing="ij")
    nbr = torch.full((9, Ho * Wo), -1, dtype=torch.int32)
    for k in range(9):
        iy, ix = oy * stride - 1 + k // 3, ox * stride - 1 + k % 3
        ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
        nbr[k] = torch.where(ok, iy * W + ix, torch.full_like(iy, -1)).reshape(-1).int()
    return nbr.to(dev), Ho * Wo


def run(name, x, nbr, rows, count, cin, cout, iters=20, fp32_out=False):
    w = (torch.randn(9, cout, cin, device=dev) * 0.05).bfloat16()
    if os.environ.get("GG_BENCH_ZEROS"):     # data-independent timing (power / clock effects of real operands)
        w.zero_()
        x = torch.zeros_like(x)
    scale = torch.ones(cout, device=dev)
    shift = torch.zeros(cout, device=dev)
    out = torch.empty((rows, cout), dtype=torch.float32 if fp32_out else torch.bfloat16, device=dev)

    def call():
        rc = L.pn_gather_gemm(_lib.ptr(x), cin, cin, x.shape[0], _lib.ptr(nbr), nbr.shape[1], 9, _lib.ptr(count), rows,
                              _lib.ptr(w), cout, _lib.ptr(scale), _lib.ptr(shift), None, 0, 1, _lib.ptr(out),
                              cout, int(fp32_out), None, _lib.current_stream(dev))
        _lib.check(rc, "gg")

    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    M = int(count.item()) if count is not None else rows
    pairs = int((nbr[:, :M] >= 0).sum().item())
    tf = 2.0 * cin * cout * pairs / us / 1e6
    tf_all = 2.0 * cin * cout * 9 * M / us / 1e6
    print(f"{name:34s} rows {M:7d} cin {cin:4d} cout {cout:4d} pairs/row {pairs / max(M, 1):4.2f}  "
          f"{us:8.1f} us  {tf:7.1f} TF/s (issued {tf_all:7.1f})", flush=True)


def main():
    torch.manual_seed(0)
    model = presets.build(presets.nusc_pillarnet18())
    synth.randomize_bn_(model, 1)
    model.to_inference(torch.bfloat16)
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)
    with torch.no_grad():
        sp = model.reader({"points": [pts]})
        sp = sp.replace_feature(sp.features_cap.bfloat16())
        feats = model.backbone(sp)
    lv = {"L1": (sp, 32), "L2": (feats["conv2"], 64), "L3": (feats["conv3"], 128)}
    for name, (t, c) in lv.items():
        nbr = sparse_conv._subm_rules(t, None)
        x = torch.randn((t.capacity, c), device=dev).bfloat16()
        run(f"sparse {name} subm {c}->{c}", x, nbr, t.capacity, t.num_active, c, c)
    for name, src, cin, cout in (("L1->L2", sp, 32, 64), ("L2->L3", feats["conv2"], 64, 128), ("L3->L4", feats["conv3"], 128, 256)):
        out_idx, out_num, out_bits, out_rank, nbr, shape, out_cap = sparse_conv._down_rules(src, 2, 1)
        x = torch.randn((src.capacity, cin), device=dev).bfloat16()
        run(f"sparse {name} stride 2 {cin}->{cout}", x, nbr, out_cap, out_num, cin, cout)
    c4 = model.backbone.conv4
    with torch.no_grad():
        y = sparse_conv.conv_forward(feats["conv3"], c4[0], c4[1], True)
    nbr = sparse_conv._subm_rules(y, None)
    run("sparse L4 subm 256->256", torch.randn((y.capacity, 256), device=dev).bfloat16(), nbr, y.capacity,
        y.num_active, 256, 256)
    for (name, H, cin, cout, stride) in [("dense 180^2 256->256", 180, 256, 256, 1),
                                         ("dense 180^2 512->256", 180, 512, 256, 1),
                                         ("dense 90^2 256->256", 90, 256, 256, 1),
                                         ("dense 180->90 s2 256->256", 180, 256, 256, 2),
                                         ("dense 180^2 256->64", 180, 256, 64, 1),
                                         ("dense 180^2 64->2304", 180, 64, 2304, 1),
                                         ("dense 180^2 384->16 fp32out", 180, 384, 16, 1)]:
        nbr, rows = grid_rules(H, H, stride)
        x = torch.randn((H * H, cin), device=dev).bfloat16()
        run(name, x, nbr, rows, None, cin, cout, fp32_out=name.endswith("fp32out"))


if __name__ == "__main__":
    main()
