"""GPU parity of the pre-stage conditioning nets (U2NETP x2, line UNet) and of the image ingest
(rank 2) against their CPU oracles and the golden vectors made from the real reference (G8)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dvd_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def tt(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-6))


@pytest.fixture(scope="module")
def models():
    from dvd_amd import prestage
    dewarp, seg, line = prestage.GeoTr_Seg_Inf(), prestage.Seg(), prestage.UNet(n_channels=3, n_classes=1)
    sds = (synth.synth_convnet_state_dict("u2netp", 11), synth.synth_convnet_state_dict("u2netp", 22, prefix="msk."),
           synth.synth_convnet_state_dict("unet", 13))
    dewarp.msk.load_state_dict(tt(sds[0]), strict=True)
    seg.load_state_dict(tt(sds[1]), strict=True)
    line.load_state_dict(tt(sds[2]), strict=True)
    for m in (dewarp, seg, line):
        m.to("cuda")
        m.eval()
    return dewarp, seg, line, sds


@pytest.mark.parametrize("size", [(288, 288), (96, 160), (70, 50)])
def test_u2netp_vs_oracle(models, size):
    """All seven outputs, at the production size, a non-square one and one whose pooled sizes are ODD (ceil-mode pools,
    35 -> 18 -> 9 -> 5 -> 3 -> 2)."""
    from oracle import prestage_oracle as PO
    dewarp, seg, line, sds = models
    src = torch.from_numpy(synth.smooth_image("u2/src", 512, 512, 7))[None]
    x = F.interpolate(src, size=size, mode="bilinear", align_corners=True)
    with torch.no_grad():
        want = PO.u2netp(sds[1], x, "msk.")
    got = seg.msk(x.cuda())
    for name, g, w in zip(("d0", "hx6", "hx5d", "hx4d", "hx3d", "hx2d", "hx1d"), got, want):
        assert tuple(g.shape) == tuple(w.shape), name
        assert relerr(g.cpu(), w) < 2e-5, (name, relerr(g.cpu(), w))


def test_unet_vs_oracle(models):
    from oracle import prestage_oracle as PO
    dewarp, seg, line, sds = models
    x = torch.from_numpy(synth.smooth_image("un/src", 288, 288, 9))[None]
    with torch.no_grad():
        want = PO.unet(sds[2], x)
    got = line(x.cuda())
    for g, w in zip(got, want):
        assert relerr(g.cpu(), w) < 2e-5, relerr(g.cpu(), w)


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("shape", [(9, 9, 16, 16), (288, 288, 64, 64), (37, 53, 288, 200), (288, 288, 512, 512)])
def test_resize_bilinear_vs_aten(align, shape):
    from dvd_amd import prestage
    hin, win, hout, wout = shape
    x = torch.from_numpy(synth.uniform("rs/x", (2, 3, hin, win), -1, 1, 3))
    ref = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=align)
    got = prestage.resize_bilinear(x.cuda(), (hout, wout), align).cpu()
    assert float((got - ref).abs().max()) < 2e-6


def test_conditioning_vs_reference_golden(models):
    """The whole pre-stage (evaluation.py:162-216) against the real reference's output (golden G8).  The document mask is
    a hard threshold of a probability, so a pixel within rounding of 0.5 may flip; flips are counted (must be rare and
    near the threshold) and everything downstream is compared with the oracle run on the DEVICE's mask decision."""
    from dvd_amd import prestage
    from oracle import prestage_oracle as PO
    g = np.load(os.path.join(GOLD, "prestage_g16.npz"))
    grid = int(g["grid"])
    dewarp, seg, line, sds = models
    src = torch.from_numpy(synth.smooth_image("g8/src", 512, 512, 1234))[None]
    got = prestage.conditioning(dewarp, seg, line, src.cuda(), grid)
    np.testing.assert_allclose(got["mask_cat"][0, 0, ::8, ::8].cpu().numpy(), g["mask_cat_sub"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(got["mask_y512"][0].cpu().numpy(), g["mask_y512"], rtol=0, atol=2e-4)
    _, dev_mask = prestage.threshold_mask_mul(prestage.resize_bilinear(got["d0"], 288, True), torch.zeros(1, 3, 288, 288, device="cuda"))
    src288 = prestage.resize_bilinear(src.cuda(), 288, True)
    d0_288 = seg.msk(src288)[0]
    bits = np.packbits((d0_288 > 0.5)[0, 0].cpu().numpy())
    flips = int(np.unpackbits(bits ^ g["mask_bits"]).sum())
    print("document-mask pixels that differ from the reference's:", flips, "of", 288 * 288)
    assert flips <= 8
    with torch.no_grad():
        want = PO.prestage(sds[0], sds[1], sds[2], src, grid, mask_override=(d0_288 > 0.5).float().cpu())
    assert relerr(got["line_msk"].cpu(), want["line_msk"]) < 5e-5
    if flips == 0:
        np.testing.assert_allclose(got["line_msk"][0].cpu().numpy(), g["line_msk"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("hw", [(1024, 768), (3508, 2480), (300, 700), (512, 512), (97, 131), (1024, 1024)])
@pytest.mark.parametrize("swap", [False, True])
def test_ingest_bit_exact_vs_oracle(hw, swap):
    """cv2.resize(INTER_LINEAR, uint8) / 255 restated in integer arithmetic: the kernel equals the oracle bit for bit."""
    from dvd_amd import ops
    from oracle import ingest_oracle as IO
    h, w = hw
    img = synth.synth_document(0, 8, 11, full_res=(h, w))["src_u8"]
    y_ref, rgb_ref = IO.ingest(img, swap, 512)
    y, rgb = ops.ingest_u8(torch.from_numpy(img).cuda(), swap_rb=swap, out_size=512, want_rgb=True)
    assert np.array_equal(y.cpu().numpy(), y_ref)
    assert np.array_equal(rgb.cpu().numpy(), rgb_ref)


def test_plugin_run_from_images(tmp_path, monkeypatch):
    """val_TDiff.run on synthetic page IMAGES: ingest -> pre-stage nets -> sampler -> batched unwarp."""
    monkeypatch.chdir(tmp_path)
    import admin.settings as ws
    from dvd_amd import val_TDiff
    s = ws.Settings()
    s.env.grid_size, s.env.diffusion_steps = 16, 3
    s.env.num_synthetic_docs, s.env.batch_docs, s.env.full_res = 3, 2, (320, 240)
    s.env.visualize, s.env.use_prestage_nets = False, True
    s.name, s.seed, s.severity, s.corruption_number = "pytest_img", 0, 0, 0
    results = val_TDiff.run(s)
    assert len(results) == 3
    for path, img in results:
        assert img.dtype == torch.uint8 and tuple(img.shape) == (320, 240, 3)


@pytest.mark.parametrize("size", [(288, 288), (70, 50)])
def test_batched_nets_equal_per_image(models, size):
    """A batch of images goes through each net's op list ONCE (every conv is one GEMM over batch * H * W rows): every
    output equals the per-image run bit for bit (kernel choices and the split-K factor depend on the per-image shape only),
    at the production size and at one whose pooled sizes are odd."""
    dewarp, seg, line, sds = models
    imgs = torch.stack([F.interpolate(torch.from_numpy(synth.smooth_image(f"bt/src{i}", 512, 512, 7 + i))[None], size=size,
                                      mode="bilinear", align_corners=True)[0] for i in range(3)]).cuda()
    for net in (seg.msk, line):
        both = net(imgs)
        for i in range(3):
            one = net(imgs[i:i + 1].contiguous())
            for b, o in zip(both, one):
                assert torch.equal(b[i:i + 1], o)


def test_batched_conditioning_equals_per_document(models):
    """prestage.conditioning for 3 documents at once == each document alone, bit for bit (mask threshold included)."""
    from dvd_amd import prestage
    dewarp, seg, line, sds = models
    src = torch.stack([torch.from_numpy(synth.smooth_image(f"bt/doc{i}", 512, 512, 100 + i)) for i in range(3)]).cuda()
    both = prestage.conditioning(dewarp, seg, line, src, 16)
    for i in range(3):
        one = prestage.conditioning(dewarp, seg, line, src[i:i + 1].contiguous(), 16)
        for k in ("mask_cat", "mask_y512", "line_msk"):
            assert torch.equal(both[k][i:i + 1], one[k]), (i, k)


def test_executor_cache_keeps_two_batch_sizes_per_shape(models):
    """a caller that alternates batch sizes (a single document, then a batch, then a single one again ...) must
    not rebuild the executor - workspace re-allocation and weight re-bind - on every call: two executors per (h, w, device)
    stay alive, the least recently used one goes when a third batch size arrives."""
    dewarp, seg, line, sds = models
    net = line
    img = torch.from_numpy(synth.smooth_image("lru/src", 64, 64, 5))[None].cuda()
    n1 = net._net(64, 64, 1)
    n3 = net._net(64, 64, 3)
    assert net._net(64, 64, 1) is n1 and net._net(64, 64, 3) is n3          # both alive, nothing rebuilt
    n2 = net._net(64, 64, 2)                                                 # third size: the LRU one (batch 1) goes
    assert net._net(64, 64, 3) is n3 and net._net(64, 64, 2) is n2
    assert len([k for k in net._nets if k[1:3] == (64, 64)]) == net.MAX_EXECUTORS_PER_SHAPE
    assert net._net(64, 64, 1) is not n1
    assert torch.equal(net(img)[0], net(img.repeat(2, 1, 1, 1))[0][:1])      # and the results do not care
