from dvd_amd.gaussian_diffusion import *  # noqa: F401,F403
