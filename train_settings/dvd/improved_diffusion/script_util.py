from dvd_amd.script_util import *  # noqa: F401,F403
