from dvd_amd.evaluation import *  # noqa: F401,F403
from dvd_amd.evaluation import run_evaluation_docunet, run_sample_lr_dewarping  # noqa: F401
