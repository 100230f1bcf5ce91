"""Frame-level data parallelism for the batched mode (BASELINE config 4).

The reference is strictly single GPU, batch 1 (device_id = 0, src/Extractors/superpoint_onnx.cc:19).
Here independent frames are sharded over the GPUs of one node, one process per GPU.  A stream of
consecutive-pair matches (i, i+1) needs frame i+1 on the rank that owns frame i, so every rank
extracts ONE overlapping frame: rank r owns frames [r*n, r*n + n] and reports n frames / n pairs.
No inter-GPU dependency exists on the data path; the only collective is the trivial gather of the
compact results to rank 0 (torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class Shard:
    start: int      # first global frame index extracted by this rank
    frames: int     # frames extracted (owned + 1 overlap)
    owned: int      # frames / pairs this rank reports


def shard_frames(frames_per_rank: int, world: int, rank: int) -> Shard:
    """Weak-scaling layout: every rank owns `frames_per_rank` frames and pairs."""
    assert 0 <= rank < world
    return Shard(start=rank * frames_per_rank, frames=frames_per_rank + 1, owned=frames_per_rank)


def gather_to_root(tensors, world: int, rank: int):
    """Gather a list of equally-shaped per-rank tensors to rank 0.  Returns on rank 0 a list (one
    entry per input tensor) of lists (one tensor per rank); None elsewhere."""
    if world == 1:
        return [[t] for t in tensors]
    out = []
    for t in tensors:
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)
        out.append(bufs)
    return out if rank == 0 else None


def assemble(gathered, owned: int):
    """Rank 0: stitch per-rank [frames, ...] arrays into the global frame order, dropping every
    rank's overlap frame except the last rank's (so global frame count = world*owned + 1), and
    per-rank [pairs, ...] arrays into global pair order."""
    n_list, kxy_list, S_list, pairs_list = gathered
    world = len(n_list)
    n = torch.cat([t[:owned] for t in n_list[:-1]] + [n_list[-1]]) if world > 1 else n_list[0]
    kxy = torch.cat([t[:owned] for t in kxy_list[:-1]] + [kxy_list[-1]]) if world > 1 else kxy_list[0]
    S = torch.cat(S_list)
    pairs = torch.cat(pairs_list)
    return n, kxy, S, pairs
