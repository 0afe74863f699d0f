"""Minimal logger with the call surface the sampling plug-in uses (configure / log / info / get_dir).
The reference carries the OpenAI-baselines logger (idf/logger.py:443-478); only these entry points are
reached from val_TDiff.run and evaluation.py on the sampling path."""
from __future__ import annotations

import datetime
import os
import sys

_state = {"dir": None, "file": None}


def configure(dir=None, format_strs=None, comm=None, log_suffix=""):
    """Creates ./checkpoints/<dir>_<timestamp>/log.txt like the reference (idf/logger.py:457-461)."""
    stamp = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S-%f")
    base = os.path.join("checkpoints", f"{dir or 'dvd'}_{stamp}")
    os.makedirs(base, exist_ok=True)
    _state["dir"] = base
    _state["file"] = open(os.path.join(base, f"log{log_suffix}.txt"), "a")
    log(f"Logging to {base}")


def get_dir():
    return _state["dir"]


def log(*args):
    msg = " ".join(str(a) for a in args)
    print(msg, file=sys.stdout, flush=True)
    if _state["file"] is not None:
        _state["file"].write(msg + "\n")
        _state["file"].flush()


info = log
warn = log
error = log
