#!/usr/bin/env python3
"""Launcher with the reference's command line (reference run_sampling.py:66-87):

    python run_sampling.py --train_module dvd --train_name val_TDiff --name <run>

dispatches to train_settings/<module>/<name>.run(settings)."""
import argparse
import importlib
import os
import random
import shutil
from datetime import date

import numpy as np
import torch

import admin.settings as ws_settings


def run_sampling(train_module, train_name, seed, name, cudnn_benchmark=True, corruption=False):
    print(f"Sampling:  {train_module}  {train_name}\nDate: {date.today().strftime('%d/%m/%Y')}")
    settings = ws_settings.Settings()
    settings.module_name, settings.script_name = train_module, train_name
    settings.project_path = f"train_settings/{train_module}/{train_name}"
    settings.seed, settings.name = seed, name
    save_dir = os.path.join(settings.env.workspace_dir, settings.project_path)
    os.makedirs(save_dir, exist_ok=True)
    shutil.copyfile(settings.project_path + ".py", os.path.join(save_dir, settings.script_name + ".py"))
    module = importlib.import_module(f"train_settings.{train_module.replace('/', '.')}.{train_name.replace('/', '.')}")
    run = getattr(module, "run")
    rounds = [(5, c) for c in range(15)] if corruption else [(0, 0)]
    for severity, number in rounds:
        settings.severity, settings.corruption_number = severity, number
        run(settings)


def main():
    p = argparse.ArgumentParser(description="Run a sampling script in train_settings.")
    p.add_argument("--train_module", type=str, required=True)
    p.add_argument("--train_name", type=str, required=True)
    p.add_argument("--cudnn_benchmark", type=bool, default=True)
    p.add_argument("--seed", type=int, default=1992)
    p.add_argument("--name", type=str, default="Default")
    p.add_argument("--corruption", action="store_true")
    a = p.parse_args()
    a.seed = torch.initial_seed() & (2 ** 32 - 1)      # as the reference: the CLI value is overwritten (:77-78)
    print(f"Seed is {a.seed}")
    random.seed(int(a.seed))
    np.random.seed(a.seed)
    run_sampling(a.train_module, a.train_name, seed=a.seed, name=a.name, cudnn_benchmark=a.cudnn_benchmark,
                 corruption=a.corruption)


if __name__ == "__main__":
    main()
