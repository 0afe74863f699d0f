"""Property checks of the ingest oracle (a restatement of OpenCV's 8-bit INTER_LINEAR resize; cv2 is absent here, so
parity with a real cv2 is unpinned - these pin the restatement's internal consistency)."""
import numpy as np

from oracle import ingest_oracle as IO


def test_identity_constant_and_ramp():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (512, 512, 3)).astype(np.uint8)
    assert np.array_equal(IO.cv2_resize_linear_u8(img, 512), img)               # same size: every coefficient is (2048, 0)
    const = np.full((300, 777, 3), 137, np.uint8)
    assert np.array_equal(IO.cv2_resize_linear_u8(const, 512), np.full((512, 512, 3), 137, np.uint8))
    ramp = np.tile(np.linspace(0, 255, 2048).astype(np.uint8)[None, :, None], (64, 1, 3))
    out = IO.cv2_resize_linear_u8(ramp, 512)
    assert (np.diff(out[0, :, 0].astype(int)) >= 0).all() and out[0, 0, 0] <= 1 and out[0, -1, 0] >= 254
    up = IO.cv2_resize_linear_u8(np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8).repeat(2, 0), 8)
    assert up[0, 0, 0] == 0 and up[0, -1, 0] == 255 and (np.diff(up[0, :, 0].astype(int)) >= 0).all()


def test_ingest_layout_and_channel_swap():
    img = np.zeros((40, 60, 3), np.uint8)
    img[..., 0], img[..., 2] = 10, 200                       # "BGR": B = 10, R = 200
    y, rgb = IO.ingest(img, swap_rb=True, out_size=16)
    assert y.shape == (3, 16, 16) and y.dtype == np.float32
    assert np.allclose(y[0], 200 / 255.0) and np.allclose(y[2], 10 / 255.0)
    assert rgb[0, 0, 0] == 200 and rgb[0, 0, 2] == 10


def test_exact_2x_downscale_takes_the_area_mean():
    """cv::resize switches INTER_LINEAR to INTER_AREA when both scales are exactly 2: rounded mean of each 2x2 block."""
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    out = IO.cv2_resize_linear_u8(img, 32)
    s = img.astype(np.int32)
    assert np.array_equal(out, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # one axis at 2x only: the bilinear path (no switch)
    out2 = IO.cv2_resize_linear_u8(img[:, :48], 32)
    assert out2.shape == (32, 32, 3)
