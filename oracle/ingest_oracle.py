"""CPU oracle of the image ingest - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates what datasets/doc_dataset/doc_benchmark.py:75-97 computes after the file is decoded:
    img_ori = cv2.imread(path)[:, :, ::-1];  img = cv2.resize(img_ori, (512, 512));  source_image = ToTensor(img) / 255.
`cv2` (opencv-python, un-pinned in the reference's requirements.txt) is ABSENT from this image and from the reference
tree, so cv2.resize is restated from OpenCV's published 8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp:
resizeGeneric_ / HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast>, INTER_RESIZE_COEF_BITS = 11).
PARITY UNPINNED: there is no cv2 here to generate golden vectors with; the HIP kernel is checked bit-exactly against
this restatement, and the restatement against properties (identity at equal size, constant images, monotone ramps).
"""
import numpy as np


def _axis(ssize: int, dsize: int):
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * (ssize / dsize) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    f = np.where(lo | hi, np.float32(0), f)
    s = np.where(lo, 0, np.where(hi, ssize - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)      # cvRound: nearest, ties to even
    a1 = np.rint(f * np.float32(2048)).astype(np.int32)
    return s, a0, a1


def cv2_resize_linear_u8(img: np.ndarray, out_size: int) -> np.ndarray:
    """img [H,W,3] uint8 -> [out,out,3] uint8, OpenCV INTER_LINEAR fixed-point arithmetic.  cv::resize switches
    INTER_LINEAR to INTER_AREA when both scale factors are EXACTLY 2 (resize.cpp: `if (interpolation == INTER_LINEAR &&
    is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`), whose 8-bit fast path is the rounded
    2x2 mean (S00 + S01 + S10 + S11 + 2) >> 2 (resizeAreaFast_, ResizeAreaFastVec)."""
    h, w, _ = img.shape
    if h == 2 * out_size and w == 2 * out_size:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, ax0, ax1 = _axis(w, out_size)
    sy, ay0, ay1 = _axis(h, out_size)
    x1 = np.minimum(sx + 1, w - 1)
    y1 = np.minimum(sy + 1, h - 1)
    src = img.astype(np.int32)
    rows0, rows1 = src[sy], src[y1]                                   # [out, W, 3]
    S0 = rows0[:, sx] * ax0[None, :, None] + rows0[:, x1] * ax1[None, :, None]
    S1 = rows1[:, sx] * ax0[None, :, None] + rows1[:, x1] * ax1[None, :, None]
    v = (((ay0[:, None, None] * (S0 >> 4)) >> 16) + ((ay1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def ingest(img_hwc_u8: np.ndarray, swap_rb: bool, out_size: int = 512):
    """-> (source_image [3,out,out] float32 in 0..1, img_ori RGB [H,W,3] uint8)."""
    rgb = img_hwc_u8[:, :, ::-1] if swap_rb else img_hwc_u8
    small = cv2_resize_linear_u8(np.ascontiguousarray(rgb), out_size)
    return (small.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32), np.ascontiguousarray(rgb)
