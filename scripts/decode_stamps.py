"""Tuning aid (needs a build with PN_EXTRA_HIPCC_FLAGS=-DPN_DECODE_STAMPS): phase times of decode_select_kernel
(set 0) on the benchmark frame."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import _lib, presets, synth
torch.manual_seed(0)
model = presets.build(presets.nusc_pillarnet18())
synth.randomize_bn_(model, seed=1)
model.to_inference(torch.bfloat16)
ex = {"points": [torch.from_numpy(synth.nusc_frame(10, 30000, seed=100)).cuda()], "metadata": [None]}
with torch.no_grad():
    for _ in range(3):
        model(ex, return_loss=False)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
h.pn_decode_debug_stamps(buf, 16)
t = list(buf)
names = ["hist scan + cut", "refine", "collect", "overflow pass", "sort"]
for i, n in enumerate(names):
    print(f"{n:20s} {t[i + 1] - t[i]:8d} cycles")
print("n, p2, refined:", t[8], t[9], t[10])
