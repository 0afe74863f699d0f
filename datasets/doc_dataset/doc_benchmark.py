"""`Doc_benchmark` - the benchmark-directory dataset of the sampling plug-in (datasets/doc_dataset/doc_benchmark.py:49-97):
one item per file of `data_root`, as a dict with the reference's keys.

What differs, deliberately: the reference's worker processes decode with cv2 AND resize to 512 x 512 on the CPU; here a
worker only decodes (PIL, EXIF orientation applied as `cv2.imread` applies it, RGB) and the resize + `/255` run on the
GPU in `dvd_ingest_u8` (ingest.hip, OpenCV's 8-bit INTER_LINEAR restated in integer arithmetic).  An item therefore
carries

    source_image_ori  [3,H,W]  input_transform(RGB uint8 HWC)   - as the reference (float, 0..255, with ArrayToTensor)
    path              str                                        - as the reference (the file's full path)

and NOT `source_image`: `run_evaluation_docunet` computes it from `source_image_ori` on the device when a loader item
lacks it, and takes it as given when an item has it (a loader built on the reference's own cv2 dataset)."""
from __future__ import annotations

import os

import numpy as np
from torch.utils.data import Dataset

_IMAGE_EXTS = (".jpg", ".jpeg", ".png", ".bmp", ".tif", ".tiff", ".webp")


class Doc_benchmark(Dataset):
    def __init__(self, data_root, input_transform) -> None:
        self.data_root = data_root
        self.input_transform = input_transform
        self.init_img_parms()

    def init_img_parms(self):
        # the reference takes os.listdir as it comes (:60-62) and cv2.imread fails on a stray file; here: the image files,
        # case-insensitively, in sorted order so that every rank of a sharded run sees the same index -> file map
        self.data_paths = sorted(f for f in os.listdir(self.data_root) if f.lower().endswith(_IMAGE_EXTS))

    def __len__(self):
        return len(self.data_paths)

    def load_rgb_u8(self, sample_path):
        from PIL import Image, ImageOps
        im = ImageOps.exif_transpose(Image.open(sample_path))
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))

    def __getitem__(self, idx):
        sample_path = os.path.join(self.data_root, self.data_paths[idx])
        img_ori = self.load_rgb_u8(sample_path)
        return {"source_image_ori": self.input_transform(img_ori), "path": sample_path}
