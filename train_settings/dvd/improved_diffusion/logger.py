from dvd_amd.logger import configure, log, info, warn, error, get_dir  # noqa: F401
