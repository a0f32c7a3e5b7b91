"""Runs ONE gather-GEMM shape repeatedly (for rocprofv3 --pmc passes).
   gg_one.py dense H CIN COUT [iters]   |   gg_one.py l4 [iters]  (sparse level-4 SubM of a real frame)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.gg_bench import grid_rules, run, dev  # noqa
if sys.argv[1] == "l4":
    from pillarnet_lts_amd import presets, sparse_conv, synth
    torch.manual_seed(0)
    model = presets.build(presets.nusc_pillarnet18()); synth.randomize_bn_(model, 1); model.to_inference(torch.bfloat16)
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)
    with torch.no_grad():
        sp = model.reader({"points": [pts]}); sp = sp.replace_feature(sp.features_cap.bfloat16())
        x = model.backbone.conv1(sp); x = model.backbone.conv2(x); x = model.backbone.conv3(x)
        c4 = model.backbone.conv4
        y = sparse_conv.conv_forward(x, c4[0], c4[1], True)
    nbr = sparse_conv._subm_rules(y, None)
    torch.cuda.synchronize()
    print("MARK: timed launches follow", flush=True)
    run("sparse L4 subm 256->256", torch.randn((y.capacity, 256), device=dev).bfloat16(), nbr, y.capacity,
        y.num_active, 256, 256, iters=int(sys.argv[2]) if len(sys.argv) > 2 else 10)
else:
    H, cin, cout = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    nbr, rows = grid_rules(H, H, 1)
    x = torch.randn((H * H, cin), device=dev).bfloat16()
    run(f"dense {H}^2 {cin}->{cout}", x, nbr, rows, None, cin, cout, iters=int(sys.argv[5]) if len(sys.argv) > 5 else 10)
