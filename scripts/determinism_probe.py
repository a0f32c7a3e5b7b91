import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import presets, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = presets.build(presets.nusc_pillarnet18())
synth.randomize_bn_(model, seed=1)
model.to_inference(torch.bfloat16)
pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)
def run():
    with torch.no_grad():
        sp = model.reader({"points": [pts]})
        M = sp.num_active_host()
        a = sp.features_cap[:M].clone()
        sp = sp.replace_feature(sp.features_cap.to(torch.bfloat16))
        f = model.backbone(sp)
        r = [a]
        for k in ("conv1", "conv2", "conv3"):
            r.append(f[k].features.clone())
        r.append(f["conv4"].clone()); r.append(f["conv5"].clone())
        bev = model.neck(f); r.append(bev[0].clone())
        preds = model.bbox_head(bev)
        r.append(preds[0]["hm"].clone()); r.append(preds[0]["reg"].clone())
        raw = model.bbox_head.predict({}, preds, model.test_cfg, raw=True)
        r += [raw[0].clone(), raw[3].clone()]
    return r
x = run(); run(); y = run()
names = ["pfn", "conv1", "conv2", "conv3", "conv4", "conv5", "bev", "hm0", "reg0", "boxes", "num"]
for n, a, b in zip(names, x, y):
    print(n, tuple(a.shape), "equal" if torch.equal(a, b) else f"DIFF max {(a.float()-b.float()).abs().max().item():g} cnt {(a!=b).sum().item()}")
print(x[-1].tolist(), y[-1].tolist())
