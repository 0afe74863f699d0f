"""Process-group helpers with the reference's names (idf/dist_util.py:21-72), MI355X-first:
one process per GPU launched by torchrun (env:// rendezvous, no MPI), backend "nccl" == RCCL over xGMI,
and ONE flat broadcast of the packed weight blob instead of an MPI byte-broadcast of the pickled
checkpoint plus one dist.broadcast per parameter."""
from __future__ import annotations

import io
import os

import torch as th
import torch.distributed as dist

GPUS_PER_NODE = 8


def _env_int(name, default):
    return int(os.environ.get(name, default))


def setup_dist(backend=None):
    """Initialise torch.distributed from torchrun's environment (single process: a 1-rank group)."""
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if th.cuda.is_available() else "gloo"
    if backend == "nccl":
        th.cuda.set_device(_env_int("LOCAL_RANK", 0) % max(1, th.cuda.device_count()))
    dist.init_process_group(backend=backend, init_method="env://")


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def dev():
    if th.cuda.is_available():
        return th.device(f"cuda:{_env_int('LOCAL_RANK', rank()) % GPUS_PER_NODE}")
    return th.device("cpu")


def load_state_dict(path, **kwargs):
    """Every rank gets the checkpoint's state_dict; only rank 0 touches the file system."""
    payload = [None]
    if rank() == 0:
        with open(path, "rb") as f:
            payload[0] = f.read()
    if world_size() > 1:
        dist.broadcast_object_list(payload, src=0)
    return th.load(io.BytesIO(payload[0]), **kwargs)


def broadcast_blob(blob: th.Tensor, src: int = 0) -> th.Tensor:
    """The sampling path's only collective: one flat broadcast of the packed weights."""
    if world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def sync_params(params):
    """Reference API (one broadcast per tensor); kept for compatibility, the engine uses broadcast_blob."""
    for p in params:
        with th.no_grad():
            if world_size() > 1:
                dist.broadcast(p, 0)


def shard_documents(n_docs: int, r: int = None, w: int = None):
    """Document indices owned by rank r: d -> rank d mod world (documents are independent units)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    return list(range(r, n_docs, w))
