from admin.environment import env_settings


class Settings:
    """Run settings: `.env` holds admin/local.py's attributes; the launcher adds module_name, script_name,
    project_path, seed, name, severity, corruption_number (reference run_sampling.py:34-39,60-61)."""

    def __init__(self):
        self.set_default()

    def set_default(self):
        self.env = env_settings()
        self.use_gpu = True
