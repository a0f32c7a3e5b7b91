"""Checkpoint contract on the CPU (det3d/torchie/trainer/checkpoint.py:49-64,67-137,166-217,
235-260; trainer.py:342-356,496-509): save -> load round trip, DDP `module.` prefix, spconv 1.x
weight layout, non-strict reporting, bare state dicts, resume, the detector's `pretrained=` hook.
All files here are written by this test; they are read back with weights_only=True."""
import logging
import os
from collections import OrderedDict

import pytest
import torch

import pillarnet_lts_amd as P
from pillarnet_lts_amd import checkpoint as C
from pillarnet_lts_amd import presets, solver


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return presets.build(presets.nusc_pillarnet18())


def _fresh():
    torch.manual_seed(123)
    return presets.build(presets.nusc_pillarnet18())


def _equal(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert sa.keys() == sb.keys()
    return all(torch.equal(sa[k], sb[k]) for k in sa)


def test_save_load_round_trip(model, tmp_path):
    f = str(tmp_path / "work" / "epoch_3.pth")
    C.save_checkpoint(model, f, meta=dict(epoch=3, iter=17, config="text", CLASSES=("car", "bus")))
    raw = torch.load(f, weights_only=True)
    assert set(raw) == {"meta", "state_dict"} and raw["meta"]["epoch"] == 3 and "time" in raw["meta"]
    assert all(v.device.type == "cpu" for v in raw["state_dict"].values())
    other = _fresh()
    assert not _equal(model, other)
    ck = P.load_checkpoint(other, f)
    assert ck["meta"]["iter"] == 17 and ck["meta"]["CLASSES"] == ["car", "bus"]
    assert _equal(model, other)


def test_spconv_keys_are_the_sparse_conv_weights(model):
    keys = C.find_all_spconv_keys(model)
    sd = model.state_dict()
    assert "backbone.conv2.0.weight" in keys and "backbone.conv1.0.conv0.0.weight" in keys
    assert all(k.startswith("backbone.") and sd[k].dim() == 4 for k in keys)
    assert len(keys) == 20                       # PillarNet-18: 5 + 3 x (1 + 4) sparse convs
    assert not any(k.startswith("backbone.conv5") for k in keys)   # conv5 is dense


def test_ddp_prefix_and_spconv1_layout(model, tmp_path):
    """A file as DDP + spconv 1.x would have written it: `module.` keys, sparse weights stored
    [kh, kw, Cin, Cout]; square convs (Cin == Cout) cannot be told apart by shape and are taken as
    they are, exactly as the reference's shape test does."""
    keys = C.find_all_spconv_keys(model)
    sd = OrderedDict()
    for k, v in model.state_dict().items():
        if k in keys and v.shape[0] != v.shape[-1]:
            v = v.permute(1, 2, 3, 0).contiguous()          # [Cout,kh,kw,Cin] -> [kh,kw,Cin,Cout]
        sd["module." + k] = v.clone()
    assert sd["module.backbone.conv2.0.weight"].shape == (3, 3, 32, 64)
    f = str(tmp_path / "old.pth")
    torch.save({"state_dict": sd, "meta": {}}, f)
    other = _fresh()
    P.load_checkpoint(other, f, strict=True)      # strict: nothing missing / unexpected / mismatched
    assert _equal(model, other)


def test_non_strict_report_and_strict_error(model, tmp_path, caplog):
    sd = OrderedDict((k, v.clone()) for k, v in model.state_dict().items())
    del sd["neck.block_5.1.weight"]
    sd["bbox_head.extra.weight"] = torch.zeros(3)
    sd["reader.pfn_layers.shared_mlps.0.weight"] = torch.zeros(32, 9)
    del sd["reader.pfn_layers.shared_mlps.1.num_batches_tracked"]      # tolerated
    f = str(tmp_path / "partial.pth")
    torch.save(sd, f)                                                 # bare state dict
    other = _fresh()
    before = other.state_dict()["neck.block_5.1.weight"].clone()
    with caplog.at_level(logging.WARNING):
        P.load_checkpoint(other, f, logger=logging.getLogger("ckpt"))
    text = caplog.text
    assert "unexpected key in source state_dict: bbox_head.extra.weight" in text
    assert "missing keys in source state_dict: neck.block_5.1.weight" in text
    assert "reader.pfn_layers.shared_mlps.0.weight" in text and "(32, 9)" in text
    assert "num_batches_tracked" not in text
    assert torch.equal(other.state_dict()["neck.block_5.1.weight"], before)
    assert torch.equal(other.state_dict()["backbone.conv2.0.weight"], model.state_dict()["backbone.conv2.0.weight"])
    with pytest.raises(RuntimeError, match="do not match exactly"):
        P.load_checkpoint(_fresh(), f, strict=True)


def test_errors(model, tmp_path):
    with pytest.raises(IOError, match="is not a checkpoint file"):
        P.load_checkpoint(model, str(tmp_path / "nope.pth"))
    with pytest.raises(IOError, match="URL"):
        P.load_checkpoint(model, "https://example.invalid/x.pth")
    f = str(tmp_path / "junk.pth")
    torch.save({"weights": 3}, f)
    with pytest.raises(RuntimeError, match="No state_dict found"):
        P.load_checkpoint(model, f)
    with pytest.raises(TypeError):
        C.save_checkpoint(model, str(tmp_path / "x.pth"), meta=[1])


def test_epoch_checkpoint_and_resume(tmp_path):
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    opt = solver.TrueWdAdam(net.parameters())
    net(torch.randn(8, 4)).sum().backward()
    opt.step()
    out = str(tmp_path / "work_dir")
    path = C.save_epoch_checkpoint(net, out, epoch=2, iteration=40, optimizer=opt.opt, meta=dict(CLASSES=["a"]))
    assert os.path.basename(path) == "epoch_2.pth"
    assert os.path.islink(os.path.join(out, "latest.pth")) and os.readlink(os.path.join(out, "latest.pth")) == "epoch_2.pth"
    C.save_epoch_checkpoint(net, out, epoch=3, iteration=60, optimizer=opt.opt)
    assert os.readlink(os.path.join(out, "latest.pth")) == "epoch_3.pth"

    torch.manual_seed(2)
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    opt2 = solver.TrueWdAdam(net2.parameters())
    epoch, it = C.resume(net2, os.path.join(out, "latest.pth"), optimizer=opt2.opt)
    assert (epoch, it) == (3, 60)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    s1, s2 = opt.opt.state_dict()["state"], opt2.opt.state_dict()["state"]
    assert s1.keys() == s2.keys()
    for k in s1:
        assert torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) and torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"])


def test_detector_pretrained_hook(model, tmp_path, capsys):
    f = str(tmp_path / "pre.pth")
    C.save_checkpoint(model, f)
    cfg = presets.nusc_pillarnet18()
    cfg.model.pretrained = f
    torch.manual_seed(77)
    loaded = presets.build(cfg)
    assert "init weight from" in capsys.readouterr().out
    assert _equal(model, loaded)
    cfg.model.pretrained = str(tmp_path / "absent.pth")      # single_stage.py:31-38: soft failure
    presets.build(cfg)
    assert "no pretrained model at" in capsys.readouterr().out


def test_resume_with_numpy_typed_hyper_parameters(tmp_path):
    """an optimiser whose param_groups hold numpy scalars (a schedule that computes lr with numpy) must come back from
    resume() with numbers, not their text; values that cannot be stored raise at save time instead of at the next step"""
    import numpy as np
    torch.manual_seed(1)
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.param_groups[0]["lr"] = np.float64(2.5e-4)
    opt.param_groups[0]["weight_decay"] = np.float32(0.5)
    net(torch.randn(4, 3)).sum().backward()
    opt.step()
    f = str(tmp_path / "np.pth")
    C.save_checkpoint(net, f, optimizer=opt, meta=dict(CLASSES=np.array([1, 2]), note=object()))
    net2 = torch.nn.Linear(3, 2)
    opt2 = torch.optim.Adam(net2.parameters(), lr=1.0)
    C.resume(net2, f, optimizer=opt2)
    g = opt2.param_groups[0]
    assert isinstance(g["lr"], float) and g["lr"] == 2.5e-4 and isinstance(g["weight_decay"], float) and g["weight_decay"] == 0.5
    net2(torch.randn(4, 3)).sum().backward()
    opt2.step()                                            # the optimiser works with what came back
    opt.param_groups[0]["hook"] = object()
    with pytest.raises(TypeError, match="optimizer state"):
        C.save_checkpoint(net, str(tmp_path / "bad.pth"), optimizer=opt)


def test_bare_ordered_dict_with_a_non_tensor_entry(model, tmp_path):
    """the reference takes any OrderedDict as a state dict (checkpoint.py:203-204)"""
    from collections import OrderedDict
    sd = OrderedDict(model.state_dict())
    f = str(tmp_path / "bare.pth")
    torch.save(OrderedDict(list(sd.items())[:4]), f)
    P.load_checkpoint(model, f)
