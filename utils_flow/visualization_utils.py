"""`visualize_dewarping` - the egress of the sampling path (utils_flow/visualization_utils.py:64-78): warp the
full-resolution source by the full-resolution sampling grid, truncate to uint8, write
`vis_hp/<eval_dataset_name>/<settings.name>/dewarped_pred/warped_<file stem>.png`.

Called as the reference calls it (evaluation.py:311-312)

    visualize_dewarping(settings, sample, data, i, source_vis, data_path, ref_flow)

`sample` is the grid `((interp(flow) + base)*2 - 1)*0.987` [1,2,H,W] and the warp is `reg_model_bilin([source_vis, sample])`
on the drop-in HIP kernel (32 B/px).  `dvd_amd.evaluation.run_evaluation_docunet` instead passes `warped_u8=`: the bytes its
fused tail kernel (`dvd_unwarp_u8[_batch]`, 6 B/px: up-sampling, base grid, affine, gather and truncation in one launch) already
produced from the COARSE flow - bit-identical to the long way (tests/test_gpu_ops.py::test_unwarp_golden, golden G5)."""
from __future__ import annotations

import os

import numpy as np
from PIL import Image

from datasets.utils.warping import register_model2

reg_model_bilin = register_model2((512, 512), "bilinear")


def _stem(data_path):
    name = data_path[0] if isinstance(data_path, (list, tuple)) else data_path
    return name.split("/")[-1][:-4]              # the reference's rule (:78): basename minus a 4-character extension


def visualize_dewarping(settings, sample, data, i, source_vis, data_path, ref_flow=None, *, warped_u8=None):
    """Returns the uint8 [H,W,3] image it wrote (the reference returns None)."""
    out_dir = f"vis_hp/{settings.env.eval_dataset_name}/{settings.name}"
    os.makedirs(f"{out_dir}/pred_flow", exist_ok=True)
    os.makedirs(f"{out_dir}/dewarped_pred", exist_ok=True)
    if warped_u8 is None:
        warped = reg_model_bilin([source_vis.to(sample.device).float(), sample])
        warped_u8 = warped[0].permute(1, 2, 0).detach().cpu().numpy().astype(np.uint8)
    else:
        warped_u8 = warped_u8.detach().cpu().numpy() if hasattr(warped_u8, "detach") else np.asarray(warped_u8)
    Image.fromarray(warped_u8).save(f"{out_dir}/dewarped_pred/warped_{_stem(data_path)}.png")
    if ref_flow is not None:
        os.makedirs(f"{out_dir}/pred_flow_ref", exist_ok=True)
        os.makedirs(f"{out_dir}/dewarped_pred_ref", exist_ok=True)
        ref = reg_model_bilin([source_vis.to(ref_flow.device).float(), ref_flow])
        ref = ref[0].permute(1, 2, 0).detach().cpu().numpy().astype(np.uint8)
        name = data_path[0] if isinstance(data_path, (list, tuple)) else data_path
        Image.fromarray(ref).save(f"{out_dir}/dewarped_pred_ref/warped_{name.split('/')[-1]}")   # with its extension (:92)
    return warped_u8
