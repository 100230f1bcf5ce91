"""CPU, world_size 2, gloo: the N>1 path of the batched mode (frame sharding with one overlap frame and
the gather to rank 0) is correct by construction -- the stitched result equals the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rover_slam_amd import sharding

OWNED, KMAX = 4, 8


def _fake_results(shard):
    """Deterministic stand-in for per-frame extraction / per-pair matching results keyed by GLOBAL index."""
    g = torch.arange(shard.start, shard.start + shard.frames)
    n = (g % 5 + 3).to(torch.int32)
    kxy = (g[:, None, None] * 100 + torch.arange(KMAX)[None, :, None] * 2 + torch.arange(2)[None, None, :]).to(torch.int32)
    gp = g[:-1]
    S = (gp % 3 + 1).to(torch.int32)
    pairs = (gp[:, None, None] * 1000 + torch.arange(KMAX)[None, :, None] + torch.arange(2)[None, None, :]).to(torch.int32)
    return n, kxy, S, pairs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = sharding.shard_frames(OWNED, world, rank)
    res = _fake_results(shard)
    g = sharding.gather_to_root(list(res), world, rank)
    if rank == 0:
        out = sharding.assemble(g, OWNED)
        q.put([t.numpy() for t in out])
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_layout():
    s0, s1 = sharding.shard_frames(32, 8, 0), sharding.shard_frames(32, 8, 7)
    assert (s0.start, s0.frames, s0.owned) == (0, 33, 32)
    assert (s1.start, s1.frames, s1.owned) == (224, 33, 32)
    # consecutive shards overlap by exactly one frame; pairs tile [0, world*owned) without gaps
    a, b = sharding.shard_frames(32, 8, 3), sharding.shard_frames(32, 8, 4)
    assert a.start + a.frames - 1 == b.start


def test_gather_world2_gloo():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: one "rank" owning everything
    whole = sharding.Shard(start=0, frames=world * OWNED + 1, owned=world * OWNED)
    ref = _fake_results(whole)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b.numpy())
