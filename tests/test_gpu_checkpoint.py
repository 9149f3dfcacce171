"""Checkpoint / wire-format compatibility and the int8-weight ingestion branch.

State dicts carry the reference's processed layout (sm80): what CUDA-EETQ writes loads here, what is written here loads
there (python/eetq/models/base.py:108-146, README.md:62-68).  The oracle's sm80 encoder -- pinned from both the writer and
the reader side of the reference -- stands in for "a checkpoint written on an NVIDIA GPU"."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _linear(K, N, bias, seed):
    torch.manual_seed(seed)
    return nn.Linear(K, N, bias=bias, dtype=torch.float16, device=DEV)


@pytest.mark.parametrize("cls_name", ["W8A16Linear", "EetqLinear"])
def test_state_dict_is_the_reference_wire_format(oracle, cls_name):
    """state_dict() bytes == the reference's processed bytes for the same weight; loading them back (into a fresh
    init-only module, the reference's load path) reproduces the outputs bit for bit; the in-memory buffer is gfx950."""
    from eetq_amd.modules.qlinear import EetqLinear, W8A16Linear
    from eetq_amd import ops
    K, N = 256, 192
    lin = _linear(K, N, True, 3)
    q, s = oracle.quantize(lin.weight.detach().t().contiguous().cpu().numpy())
    wname = "qweight" if cls_name == "W8A16Linear" else "weight"
    if cls_name == "W8A16Linear":
        mod = W8A16Linear.from_torch(lin)
    else:
        mod = EetqLinear(K, N, bias=True, device=DEV)
        mod.register_scale(DEV)
        w, sc = ops.quant_weights(lin.weight.detach().t().contiguous(), torch.int8, False)
        mod.weight, mod.weight_scales, mod.bias = w, sc, lin.bias.detach().clone()
        mod.eval()
    x = torch.rand(5, K, dtype=torch.float16, device=DEV)
    y = mod(x)
    sd = mod.state_dict()
    assert set(sd) == {wname, "weight_scales", "bias"}                      # no extra keys: the reference's schema
    assert np.array_equal(sd[wname].cpu().numpy(), oracle.sm80_pack(q))     # the bytes CUDA-EETQ would have written
    assert np.array_equal(getattr(mod, wname).cpu().numpy(), oracle.gfx950_pack(q))  # memory stays native
    assert sd["weight_scales"].cpu().numpy().tobytes() == s.tobytes()
    # "NVIDIA-written checkpoint": sm80 bytes straight from the oracle, loaded through load_state_dict
    fresh = W8A16Linear(K, N, bias=True, dev=DEV) if cls_name == "W8A16Linear" else EetqLinear(K, N, bias=True, device=DEV)
    if cls_name == "EetqLinear":
        fresh.register_scale(DEV)
        fresh.eval()
    nv = {wname: torch.from_numpy(oracle.sm80_pack(q)), "weight_scales": torch.from_numpy(s), "bias": lin.bias.detach().cpu()}
    fresh.load_state_dict(nv)
    assert torch.equal(getattr(fresh, wname), getattr(mod, wname))
    assert torch.equal(fresh(x), y)
    # and a state dict taken here loads here
    again = type(fresh)(K, N, True, DEV)
    if cls_name == "EetqLinear":
        again.register_scale(DEV)
        again.eval()
    again.load_state_dict(sd)
    assert torch.equal(again(x), y)


def test_wire_layout_switch_and_shapes_the_reference_cannot_hold(oracle):
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils import get_wire_layout, wire_layout
    lin = _linear(128, 64, False, 4)
    mod = W8A16Linear.from_torch(lin)
    assert get_wire_layout() == "sm80"
    with wire_layout("gfx950"):
        sd = mod.state_dict()
        assert torch.equal(sd["qweight"], mod.qweight)                    # native bytes on request
        other = W8A16Linear(128, 64, bias=False, dev=DEV)
        other.load_state_dict(sd)
        assert torch.equal(other.qweight, mod.qweight)
    assert get_wire_layout() == "sm80"
    pinned = W8A16Linear(128, 64, bias=False, dev=DEV)
    pinned.checkpoint_layout = "gfx950"                                   # this module's checkpoint is known to be native
    pinned.load_state_dict(sd)
    assert torch.equal(pinned.qweight, mod.qweight)
    with pytest.raises(ValueError):
        with wire_layout("sm90"):
            pass
    # N = 80 is not a multiple of 64: the reference has no layout for it; the bytes pass through both hooks unchanged
    odd = W8A16Linear.from_torch(_linear(128, 80, False, 5))
    sd = odd.state_dict()
    assert torch.equal(sd["qweight"], odd.qweight)
    back = W8A16Linear(128, 80, bias=False, dev=DEV)
    back.load_state_dict(sd)
    x = torch.rand(2, 128, dtype=torch.float16, device=DEV)
    assert torch.equal(back(x), odd(x))


def test_in_band_layout_tag_pickles_and_state_dict_device(tmp_path):
    """(a) torch.save(state_dict) keeps the layout tag: a dict saved under wire_layout("gfx950") loads correctly under the
    default setting (the tag, not the setting at load time, says what the bytes are); (b) torch.save(model) of a whole
    quantised module round-trips; (c) set_state_dict_device("cpu") offloads the re-encoded tensors one by one."""
    from eetq_amd.checkpoint import set_state_dict_device
    from eetq_amd.modules.qlinear import W8A16Linear
    from eetq_amd.utils import wire_layout
    mod = W8A16Linear.from_torch(_linear(128, 64, True, 7))
    x = torch.rand(3, 128, dtype=torch.float16, device=DEV)
    y = mod(x)
    with wire_layout("gfx950"):
        torch.save(mod.state_dict(), tmp_path / "native.pt")
    torch.save(mod.state_dict(), tmp_path / "wire.pt")
    for name in ("native.pt", "wire.pt"):
        sd = torch.load(tmp_path / name)
        assert sd._metadata[""]["eetq_layout"] == ("gfx950" if name == "native.pt" else "sm80")
        fresh = W8A16Linear(128, 64, bias=True, dev=DEV)
        fresh.load_state_dict(sd)                       # default process-wide setting (sm80) in both cases
        assert torch.equal(fresh.qweight, mod.qweight) and torch.equal(fresh(x), y)
    torch.save(mod, tmp_path / "whole.pt")
    whole = torch.load(tmp_path / "whole.pt", weights_only=False)
    assert torch.equal(whole.qweight, mod.qweight) and torch.equal(whole(x), y)
    again = W8A16Linear(128, 64, bias=True, dev=DEV)
    again.load_state_dict(whole.state_dict())           # the hooks survived the pickle
    assert torch.equal(again(x), y)
    set_state_dict_device("cpu")
    try:
        sd = mod.state_dict()
        assert sd["qweight"].device.type == "cpu" and sd["weight_scales"].device.type == "cuda"
        fresh = W8A16Linear(128, 64, bias=True, dev=DEV)
        fresh.load_state_dict(sd)
        assert torch.equal(fresh(x), y)
    finally:
        set_state_dict_device(None)


def test_model_round_trip_through_safetensors_and_convert_checkpoint(tmp_path, oracle):
    """eet_quantize -> save (safetensors, sm80 on disk + tagged config) -> init_only model + load -> same logits bit for
    bit; convert_checkpoint rewrites the directory to native bytes and back."""
    from safetensors.torch import load_file, save_file
    from eetq_amd.utils import (checkpoint_layout, convert_checkpoint, convert_model_layout_, eet_quantize,
                                quantization_config, wire_layout)

    def make():
        torch.manual_seed(0)
        return nn.Sequential(nn.Linear(128, 256, dtype=torch.float16), nn.ReLU(), nn.Linear(256, 64, dtype=torch.float16)).to(DEV)
    model = eet_quantize(make())
    x = torch.rand(3, 128, dtype=torch.float16, device=DEV)
    y = model(x)
    d = tmp_path / "ckpt"
    d.mkdir()
    save_file({k: v.contiguous().cpu() for k, v in model.state_dict().items()}, str(d / "model.safetensors"))
    json.dump({"quantization_config": quantization_config()}, open(d / "config.json", "w"))
    assert checkpoint_layout(json.load(open(d / "config.json"))) == "sm80"
    assert checkpoint_layout({"quantization_config": {"quant_method": "eetq", "bits": 8}}) == "sm80"   # reference-written
    disk = load_file(str(d / "model.safetensors"))
    q0, _ = oracle.quantize(make()[0].weight.detach().t().contiguous().cpu().numpy())
    assert np.array_equal(disk["0.qweight"].numpy(), oracle.sm80_pack(q0))
    fresh = eet_quantize(make(), init_only=True)
    fresh.load_state_dict(disk)
    assert torch.equal(fresh(x), y)
    # offline conversion sm80 -> gfx950 (in a second directory), load with the matching wire layout
    n = convert_checkpoint(str(d), str(tmp_path / "native"), dst="gfx950")
    assert n == 2 and checkpoint_layout(json.load(open(tmp_path / "native" / "config.json"))) == "gfx950"
    native = load_file(str(tmp_path / "native" / "model.safetensors"))
    assert torch.equal(native["0.qweight"].to(DEV), model[0].qweight) and torch.equal(native["0.bias"], disk["0.bias"])
    with wire_layout("gfx950"):
        fresh2 = eet_quantize(make(), init_only=True)
        fresh2.load_state_dict(native)
    assert torch.equal(fresh2(x), y)
    # ... and back, in place
    assert convert_checkpoint(str(tmp_path / "native"), dst="sm80") == 2
    assert torch.equal(load_file(str(tmp_path / "native" / "model.safetensors"))["2.qweight"], disk["2.qweight"])
    # a loader that copies tensors straight into the buffers (no load_state_dict): convert the live model afterwards
    raw = eet_quantize(make(), init_only=True)
    for name, buf in raw.named_buffers():
        buf.data.copy_(disk[name].to(DEV))
    assert not torch.equal(raw(x), y)
    assert convert_model_layout_(raw, "sm80") == 2
    assert torch.equal(raw(x), y)


class FakeLinear8bitLt(nn.Linear):
    """Shape of bitsandbytes' Linear8bitLt as the reference reads it (python/eetq/utils/quantizer.py:44-48): int8 weight
    [out, in] quantised per output row, and state_dict()["SCB"] = the per-row absmax."""

    def __init__(self, src):
        super().__init__(src.in_features, src.out_features, bias=src.bias is not None, device=src.weight.device,
                         dtype=torch.float16)
        w = src.weight.detach().float()
        scb = w.abs().amax(dim=1)
        q = torch.round(w / (scb[:, None] / 127.0)).clamp(-127, 127).to(torch.int8)
        self.weight = nn.Parameter(q, requires_grad=False)
        self.register_buffer("SCB", scb.half())
        if src.bias is not None:
            self.bias = nn.Parameter(src.bias.detach().clone(), requires_grad=False)


def test_int8_weight_ingestion_branch(oracle):
    """eet_quantize / quantize_and_preprocess_weights on already-int8 weights (bitsandbytes): only re-laid-out, scales =
    SCB / 127 (quantizer.py:44-48, qlinear.py:17-19)."""
    from eetq_amd.modules.qlinear import W8A16Linear, quantize_and_preprocess_weights
    from eetq_amd.utils import eet_quantize
    src = _linear(256, 128, True, 9)
    model = nn.Sequential(FakeLinear8bitLt(src))
    q_rows = model[0].weight.detach().clone()                      # [out, in] int8
    scb = model[0].SCB.detach().clone()
    eet_quantize(model, include=[FakeLinear8bitLt])
    mod = model[0]
    assert isinstance(mod, W8A16Linear)
    want_scales = torch.div(scb, 127.0).half()
    assert torch.equal(mod.weight_scales, want_scales)
    assert np.array_equal(mod.qweight.cpu().numpy(), oracle.gfx950_pack(q_rows.t().contiguous().cpu().numpy()))
    x = torch.rand(4, 256, dtype=torch.float16, device=DEV)
    y = mod(x).cpu().numpy()
    ref = oracle.w8a16_gemm(x.cpu().numpy(), q_rows.t().contiguous().cpu().numpy(), want_scales.cpu().numpy())
    ref = (torch.from_numpy(ref) + src.bias.detach().cpu()).numpy()
    assert np.abs(y.astype(np.float32) - ref.astype(np.float32)).max() <= 2e-3 * np.abs(ref).max() + 1e-3
    # and close to the fp16 layer it came from (8-bit row-wise quantisation error)
    with torch.no_grad():
        assert (mod(x) - src(x)).abs().max().item() < 2e-2
    pw, sc = quantize_and_preprocess_weights(q_rows, want_scales)
    assert torch.equal(pw, mod.qweight) and sc is want_scales
    with pytest.raises(AssertionError):
        quantize_and_preprocess_weights(q_rows)                     # int8 weights need their scales
    with pytest.raises(ValueError):
        quantize_and_preprocess_weights(q_rows.float())
