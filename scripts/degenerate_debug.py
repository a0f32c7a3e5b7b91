import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from pillarnet_lts_amd import presets, synth, dense_hip
from test_detector_gpu import small_preset
cuda = torch.device("cuda:0")
torch.manual_seed(0)
model = presets.build(small_preset()); synth.randomize_bn_(model, seed=1); model.to_inference(torch.bfloat16)
frames = [np.zeros((0, 5), np.float32), np.full((50, 5), 400.0, np.float32), np.array([[0.1, 0.2, -1.0, 3.0, 0.0]], np.float32)]
ex = {"points": [torch.from_numpy(f).to(cuda) for f in frames], "metadata": [None] * 3}
calls = []
orig = dense_hip._launch
def spy(x, cin, nbr, ntaps, rows, w, scale, shift, relu, out, out_ld, fp32_out=False, out_rows=None, residual=None, res_ld=0, in_ld=0):
    orig(x, cin, nbr, ntaps, rows, w, scale, shift, relu, out, out_ld, fp32_out, out_rows, residual, res_ld, in_ld)
    calls.append((cin, w.shape[1], ntaps, rows, out, out_ld, out_rows is not None))
dense_hip._launch = spy
with torch.no_grad():
    bev, feats = model.extract_feat(ex)
    preds = model.bbox_head(bev)
torch.cuda.synchronize()
for k, (cin, cout, ntaps, rows, out, out_ld, scat) in enumerate(calls):
    o = torch.as_strided(out, (rows, cout), (out_ld, 1)).float()
    per = rows // 3
    d = (o[:per] != o[per:2 * per]).any(dim=1).nonzero().view(-1)
    print(k, f"cin {cin} cout {cout} taps {ntaps} rows {rows} scatter {scat}: frame0 vs frame1 differing rows {d.numel()} {d[:8].tolist()}")
dense_hip._launch = orig
for dtype in (torch.bfloat16, torch.float32):
    model.to_inference(dtype)
    for rep in range(3):
        with torch.no_grad():
            out = model(ex, return_loss=False)
        n = [o["scores"].shape[0] for o in out]
        k = min(n[0], n[1])
        same = torch.equal(out[0]["scores"][:k], out[1]["scores"][:k])
        print(dtype, "rep", rep, "counts", n, "prefix equal", same, flush=True)
    if dtype == torch.bfloat16:
        with torch.no_grad():
            bev, _ = model.extract_feat(ex)
            preds = model.bbox_head(bev)
        for t, p in enumerate(preds):
            pk = p.packed
            print("task", t, "packed", tuple(pk.shape), "frame0==frame1", torch.equal(pk[0], pk[1]))
