"""Same import paths as the reference package; the implementations live in dvd_amd/."""
from dvd_amd import dist_util, gaussian_diffusion, logger, respace, script_util  # noqa: F401
