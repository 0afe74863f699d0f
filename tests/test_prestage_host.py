"""CPU checks of the pre-stage nets' HOST side (dvd_amd/prestage.py): the op-list builder and the weight packer are
validated by interpreting the op list with plain torch ops (this interpreter is test infrastructure - the product runs
the list on the HIP executor) and comparing with the oracle, which is itself pinned to the real reference (golden G8)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dvd_amd import prestage, synth
from oracle import prestage_oracle as PO


def interpret(program, weights, x):
    """Run a prestage.Program on the CPU: x [1,C,H,W] -> {slot: tensor}."""
    slots = {0: x}
    for op, (src, cin, cout, ks, kp) in zip([o for o in program.ops if o["op"] == prestage.CONV], program.convs):
        op["_meta"] = (cin, cout, ks, kp)
    for o in program.ops:
        a = slots[o["a"]]
        if o["op"] == prestage.CONV:
            cin, cout, ks, kp = o["_meta"]
            off = o["w_off"]
            w = weights[off:off + cout * kp].reshape(cout, kp)[:, :ks * ks * cin].reshape(cout, ks, ks, cin).permute(0, 3, 1, 2)
            b = weights[off + cout * kp:off + cout * kp + cout]
            inp = a if o.get("b", -1) < 0 else torch.cat((a, slots[o["b"]]), 1)
            y = F.conv2d(inp, w.contiguous(), b, padding=o["dil"] * (ks // 2), dilation=o["dil"])
            slots[o["dst"]] = F.relu(y) if o["act"] == 2 else y
        elif o["op"] == prestage.POOL:
            slots[o["dst"]] = F.max_pool2d(a, 2, stride=2, ceil_mode=bool(o["flag"]))
        elif o["op"] == prestage.RESIZE:
            slots[o["dst"]] = F.interpolate(a, size=slots[o["b"]].shape[2:], mode="bilinear", align_corners=bool(o["flag"]))
        elif o["op"] == prestage.ADD:
            slots[o["dst"]] = a + slots[o["b"]]
        elif o["op"] == prestage.SIGMOID:
            slots[o["dst"]] = torch.sigmoid(a)
    return slots


@pytest.fixture(scope="module")
def src288():
    src = torch.from_numpy(synth.smooth_image("g8/src", 512, 512, 1234))[None]
    return F.interpolate(src, size=96, mode="bilinear", align_corners=True)      # a small size keeps this test fast


def test_u2netp_program_equals_oracle(src288):
    sd = synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")
    P, outs = prestage.build_u2netp("msk.")
    with torch.no_grad():
        got = interpret(P, P.pack(sd), src288)
        want = PO.u2netp(sd, src288, "msk.")
    for slot, ref in zip(outs, want):
        assert got[slot].shape == ref.shape
        assert float((got[slot] - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))


def test_unet_program_equals_oracle(src288):
    sd = synth.synth_convnet_state_dict("unet", 13)
    P, outs = prestage.build_unet()
    with torch.no_grad():
        got = interpret(P, P.pack(sd), src288)
        want = PO.unet(sd, src288)
    for slot, ref in zip(outs, want):
        assert float((got[slot] - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))


def test_reference_named_modules_round_trip():
    """Seg / GeoTr_Seg_Inf / UNet hold the reference's state_dict keys: checkpoints saved by the reference load with
    strict=True, and reload_segmodel's 6-character prefix stripping works (geotr_core.py:1090-1112)."""
    seg = prestage.Seg()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.").items()}
    assert list(seg.state_dict().keys()) == list(sd.keys())
    seg.load_state_dict(sd, strict=True)
    assert torch.equal(seg.state_dict()["msk.stage3.rebnconv2.conv_s1.weight"], sd["msk.stage3.rebnconv2.conv_s1.weight"])
    with pytest.raises(RuntimeError):
        seg.load_state_dict({"bogus.weight": torch.zeros(1), **sd}, strict=True)
    line = prestage.UNet(n_channels=3, n_classes=1)
    sdl = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_convnet_state_dict("unet", 13).items()}
    assert list(line.state_dict().keys()) == list(sdl.keys())
    line.load_state_dict(sdl, strict=True)
    dewarp = prestage.GeoTr_Seg_Inf()
    plain = synth.synth_convnet_state_dict("u2netp", 11)
    import io
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pth") as f:
        torch.save({"model." + k: torch.from_numpy(np.asarray(v)) for k, v in plain.items()}, f.name)
        prestage.reload_segmodel(dewarp.msk, f.name)
    assert torch.equal(dewarp.msk.state_dict()["stage1.rebnconvin.conv_s1.weight"],
                       torch.from_numpy(plain["stage1.rebnconvin.conv_s1.weight"]))
