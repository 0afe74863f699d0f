from dvd_amd.respace import *  # noqa: F401,F403
