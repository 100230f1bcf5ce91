#!/usr/bin/env python3
"""In-kernel timeline of lg_attention_lat_kernel (tuning build, RFE_ALAT_ABL=4) at one-pair size through rfe_k_attention.
usage (GPU box): RFE_LIBRARY=rover-slam_amd/librover_fe_tuning.so RFE_ALAT_ABL=4 python tools/kbench/lat_timeline_att.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rover_slam_amd import capi  # noqa: E402

ctx = capi.Context(0)
nseq, L = 2, 1024
rng = np.random.default_rng(0)
qkv = ctx.alloc(nseq * L * 768 * 4).upload(rng.standard_normal((nseq * L, 768)).astype(np.float32))
o = ctx.alloc(nseq * L * 1024)
for _ in range(5):
    ctx._chk(capi.lib.rfe_k_attention(ctx.h, qkv.ptr, qkv.ptr + 1024, qkv.ptr + 2048, 768, o.ptr, nseq, L, L, None, None, None, None))
nb = 256
buf = (C.c_ulonglong * (nb * 8))()
capi.lib.rfe_k_dbg_timeline_att.argtypes = [C.c_void_p, C.c_int]
rc = capi.lib.rfe_k_dbg_timeline_att(buf, nb * 8)
t = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
clk, wall = t[:, :4], t[:, 4:]
d = np.diff(clk, axis=1)
print("rc", rc, "blocks", int((clk[:, 0] > 0).sum()))
print("shader-clock cycles per phase (entry->first tile landed, key loop of wave 0, merge + store): mean", d.mean(0).round(0), "min", d.min(0), "max", d.max(0))
w0 = wall.min()
print("wall (10 ns): entry min/max", wall[:, 0].min() - w0, wall[:, 0].max() - w0, "| exit min/max", wall[:, 3].min() - w0, wall[:, 3].max() - w0)
print("wall ticks per phase: mean", np.diff(wall, axis=1).mean(0).round(1))
