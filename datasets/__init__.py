"""The reference's `datasets` package as far as the sampling path imports it (`val_TDiff.py:8,97`:
`datasets.Doc_benchmark`; `visualization_utils.py:10` / `gaussian_diffusion.py:12`: `datasets.utils.warping`).
With the repository root as the working directory this package shadows the unrelated `datasets` wheel of the
image, exactly as the reference's own `datasets/` does."""
from datasets.doc_dataset.doc_benchmark import Doc_benchmark  # noqa: F401
