import importlib


def env_settings():
    """Instantiate admin.local.EnvironmentSettings (reference admin/environment.py:98-109)."""
    try:
        return importlib.import_module("admin.local").EnvironmentSettings()
    except ImportError as exc:
        raise RuntimeError("admin/local.py is missing: copy the template from the repository and set your paths") from exc
