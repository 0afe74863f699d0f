"""Multi-process (gloo, world_size 2, CPU) test of the sampling path's distributed plumbing: rank-0 packs the
weight blob, ONE flat broadcast, both ranks hold identical bytes; documents shard disjointly with no collective."""
import hashlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dvd_amd import dist_util, synth, weights
    dist_util.setup_dist(backend="gloo")
    assert dist_util.rank() == rank and dist_util.world_size() == world
    grid = 16
    # blob layout must be computable on every rank without the weights
    if rank == 0:
        packed = weights.pack(synth.synth_state_dict(grid, 7, blocks=[11]), grid)
        names = sorted(packed)
        blob = torch.cat([packed[n].contiguous().view(-1).view(torch.uint8) for n in names])
        size = torch.tensor([blob.numel()])
    else:
        size = torch.tensor([0])
    dist.broadcast(size, 0)
    if rank != 0:
        blob = torch.empty(int(size), dtype=torch.uint8)
    dist_util.broadcast_blob(blob, src=0)
    digest = hashlib.sha256(blob.numpy().tobytes()).hexdigest()
    docs = dist_util.shard_documents(7)
    dist.barrier()
    q.put((rank, digest, docs))
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "ranks disagree on the broadcast weight blob"
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
