"""CPU, world_size 2, gloo: the N>1 path of the batched mode (frame sharding with one overlap frame and
the ONE-collective gather of the packed results to rank 0) is correct by construction -- the stitched result
equals the single-process one -- and `bench.py --gpus 2` really starts two ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rover_slam_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWNED, KMAX = 4, 8


def _fill(pack, shard):
    """Deterministic stand-in for per-frame extraction / per-pair matching results keyed by GLOBAL index."""
    g = torch.arange(shard.start, shard.start + shard.frames)
    pack.n.copy_((g % 5 + 3).to(torch.int32))
    pack.kxy.copy_((g[:, None, None] * 100 + torch.arange(KMAX)[None, :, None] * 2 + torch.arange(2)[None, None, :]).to(torch.int32))
    gp = g[:-1]
    pack.S.copy_((gp % 3 + 1).to(torch.int32))
    pack.pairs.copy_((gp[:, None, None] * 1000 + torch.arange(KMAX)[None, :, None] + torch.arange(2)[None, None, :]).to(torch.int32))
    pack.ms.copy_((gp[:, None] + torch.arange(KMAX)[None, :] / 16.0).to(torch.float32))
    pack.desc.fill_(float(shard.start))


def _worker(rank, world, port, q, with_desc):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = sharding.shard_frames(OWNED, world, rank)
    pack = sharding.ResultPack(shard.frames, KMAX, torch.device("cpu"))
    _fill(pack, shard)
    g = sharding.RootGather(pack, world, rank, with_desc=with_desc)
    for _ in range(2):                       # the receive buffer is reused from step to step
        recv = g()
    if rank == 0:
        assert recv.data_ptr() == g.recv.data_ptr() and recv.shape == (world, g.nbytes)
        out = [t.numpy().copy() for t in sharding.assemble(g, OWNED)]
        if with_desc:
            out.append(np.array([float(g.rank_view(r, "desc")[0, 0, 0]) for r in range(world)]))
        q.put(out)
    else:
        assert recv is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_layout():
    s0, s1 = sharding.shard_frames(32, 8, 0), sharding.shard_frames(32, 8, 7)
    assert (s0.start, s0.frames, s0.owned) == (0, 33, 32)
    assert (s1.start, s1.frames, s1.owned) == (224, 33, 32)
    # consecutive shards overlap by exactly one frame; pairs tile [0, world*owned) without gaps
    a, b = sharding.shard_frames(32, 8, 3), sharding.shard_frames(32, 8, 4)
    assert a.start + a.frames - 1 == b.start
    # strong scaling: 256 frames in all
    for world in (1, 2, 4, 8):
        sh = [sharding.shard_frames_strong(256, world, r) for r in range(world)]
        assert sum(s.owned for s in sh) == 256 and sh[-1].start + sh[-1].frames == 257
    with pytest.raises(ValueError):
        sharding.shard_frames_strong(256, 3, 0)


def test_result_pack_layout():
    p = sharding.ResultPack(33, 1024, torch.device("cpu"))
    # compact prefix = counts + keypoints + matches; bulk = scores + descriptors
    assert p.compact_bytes < 1 << 20 and p.total_bytes > 33 * 1024 * 1024
    offs = [p.layout[k][0] for k in p.FIELDS]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert p.layout["ms"][0] + p.layout["ms"][1] <= p.compact_bytes <= p.layout["score"][0]
    p.kxy[3, 5, 1] = 77
    assert p.view_of(p.buf, "kxy")[3, 5, 1] == 77 and p.payload().numel() == p.compact_bytes and p.payload(True).numel() == p.total_bytes


@pytest.mark.parametrize("with_desc", [False, True])
def test_gather_world2_gloo(with_desc):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, with_desc)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: one "rank" owning everything
    whole = sharding.Shard(start=0, frames=world * OWNED + 1, owned=world * OWNED)
    ref = sharding.ResultPack(whole.frames, KMAX, torch.device("cpu"))
    _fill(ref, whole)
    for a, b in zip(got, (ref.n, ref.kxy, ref.S, ref.pairs, ref.ms)):
        assert np.array_equal(a, b.numpy())
    if with_desc:
        assert got[5].tolist() == [0.0, float(OWNED)]


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_gpus2_spawns_two_ranks(scaling):
    """`bench.py --gpus 2` without a launcher starts two rank processes that form one process group (here: gloo, CPU
    tensors, --check-launch = everything but the GPU work) and rank 0 reports n_gpus = 2."""
    r = _bench("--gpus", "2", "--check-launch", "--scaling", scaling)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["scaling"] == scaling
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["allreduce_sum_of_ones"] == 2
    devs = line["rccl"]["devices"]
    assert sorted(d["rank"] for d in devs) == [0, 1] and len({d["pid"] for d in devs}) == 2
    assert "child processes" in line["launcher"]
    owned = 4 if scaling == "weak" else 8
    assert line["gathered_frame_ids"] == list(range(2 * owned + 1))


def test_bench_refuses_mislabelled_runs():
    # --gpus that disagrees with the launcher's WORLD_SIZE
    r = _bench("--gpus", "4", "--check-launch", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "disagrees" in r.stderr
    # more ranks than GPUs on the RCCL backend (this container has none)
    if torch.cuda.device_count() < 2:
        r = _bench("--gpus", "2", "--steps", "1")
        assert r.returncode != 0 and "visible" in r.stderr
