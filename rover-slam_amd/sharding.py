"""Frame-level data parallelism for the batched mode (BASELINE config 4).

The reference is strictly single GPU, batch 1 (device_id = 0, src/Extractors/superpoint_onnx.cc:19,
src/Matchers/lightglue_onnx.cpp:24).  Here independent frames are sharded over the GPUs of one node, one
process per GPU.  A stream of consecutive-pair matches (i, i+1) needs frame i+1 on the rank that owns
frame i, so every rank extracts ONE overlapping frame: rank r owns frames [r*n, r*n + n] and reports
n frames / n pairs.  No inter-GPU dependency exists on the data path; the only collective is the trivial
gather of the compact results to rank 0 (torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests).

All outputs of a rank live in ONE contiguous device buffer (ResultPack); the gather is ONE collective per
step on a prefix of that buffer into a receive buffer that rank 0 allocates once.
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


def shard_frames_strong(total_frames: int, world: int, rank: int) -> Shard:
    """Strong-scaling layout: `total_frames` frames (and pairs) in all, split evenly over the ranks."""
    assert 0 <= rank < world
    if total_frames % world:
        raise ValueError(f"strong scaling needs total_frames ({total_frames}) divisible by the number of ranks ({world})")
    return shard_frames(total_frames // world, world, rank)


_ALIGN = 256


class ResultPack:
    """Per-rank outputs of one step, carved out of one contiguous byte buffer:
    [ n | S | kxy | pairs | ms ]  (compact: what a tracker consumes, ~0.7 MB for 33 frames at Kmax 1024)
    [ score | desc ]              (bulk: 34 MB, gathered only on request)."""

    FIELDS = ("n", "S", "kxy", "pairs", "ms", "score", "desc")

    def __init__(self, frames: int, kmax: int, device):
        npairs = max(frames - 1, 1)
        spec = [("n", torch.int32, (frames,)), ("S", torch.int32, (npairs,)), ("kxy", torch.int32, (frames, kmax, 2)),
                ("pairs", torch.int32, (npairs, kmax, 2)), ("ms", torch.float32, (npairs, kmax)),
                ("score", torch.float32, (frames, kmax)), ("desc", torch.float32, (frames, kmax, 256))]
        self.frames, self.kmax, self.layout = frames, kmax, {}
        off = 0
        for name, dt, shape in spec:
            nbytes = int(torch.tensor([], dtype=dt).element_size())
            for d in shape:
                nbytes *= d
            self.layout[name] = (off, nbytes, dt, shape)
            off = (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
            if name == "ms":
                self.compact_bytes = off
        self.total_bytes = off
        self.buf = torch.zeros(self.total_bytes, dtype=torch.uint8, device=device)
        for name in self.FIELDS:
            setattr(self, name, self.view_of(self.buf, name))

    def view_of(self, flat_u8, name):
        off, nbytes, dt, shape = self.layout[name]
        return flat_u8[off:off + nbytes].view(dt).view(shape)

    def payload(self, with_desc=False):
        """The contiguous prefix a gather moves."""
        return self.buf if with_desc else self.buf[:self.compact_bytes]


class RootGather:
    """ONE dist.gather per step into a receive buffer allocated once on rank 0."""

    def __init__(self, pack: ResultPack, world: int, rank: int, with_desc=False, always_collective=False):
        self.pack, self.world, self.rank, self.with_desc = pack, world, rank, with_desc
        self.always_collective = always_collective     # world size 1: still go through dist.gather (RCCL path check on one GPU)
        self.nbytes = pack.total_bytes if with_desc else pack.compact_bytes
        # gloo has no device-memory gather: the functional check of the N-rank flow on a 1-GPU box stages through
        # (pinned) host memory; RCCL ("nccl") gathers device to device
        self.via_host = (pack.buf.device.type == "cuda" and (world > 1 or always_collective) and dist.is_initialized() and dist.get_backend() != "nccl")
        rdev = torch.device("cpu") if self.via_host else pack.buf.device
        self.stage = torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() if self.via_host else None
        self.recv = torch.empty((world, self.nbytes), dtype=torch.uint8, device=rdev) if rank == 0 else None
        self.recv_list = list(self.recv.unbind(0)) if rank == 0 else None

    def __call__(self):
        src = self.pack.payload(self.with_desc)
        if self.world == 1 and not self.always_collective:
            if self.recv is not None:
                self.recv[0].copy_(src)
            return self.recv
        if self.via_host:
            self.stage.copy_(src)        # synchronises with the producing stream
            src = self.stage
        dist.gather(src, self.recv_list, dst=0)
        return self.recv

    def rank_view(self, r: int, name: str):
        """Rank 0: field `name` of rank r's last gathered payload."""
        return self.pack.view_of(self.recv[r], name)


def assemble(gather: RootGather, owned: int):
    """Rank 0: stitch the gathered per-rank [frames, ...] arrays into the global frame order, dropping every
    rank's overlap frame except the last rank's (global frame count = world*owned + 1), and the per-rank
    [pairs, ...] arrays into global pair order."""
    world = gather.world
    per_frame = lambda name: torch.cat([gather.rank_view(r, name)[:owned] for r in range(world - 1)]
                                       + [gather.rank_view(world - 1, name)])
    per_pair = lambda name: torch.cat([gather.rank_view(r, name)[:owned] for r in range(world)])
    return per_frame("n"), per_frame("kxy"), per_pair("S"), per_pair("pairs"), per_pair("ms")
