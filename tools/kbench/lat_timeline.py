#!/usr/bin/env python3
"""In-kernel timeline of gemm_lat_kernel (tuning build, RFE_GLAT_ABL=4): one LightGlue FFN block at one-pair size (2048 rows) through
rfe_k_lightglue_ffn -> ffn.0 (stats) + ffn.3 (LayerNorm + GELU on the fragment); the LAST kernel's per-workgroup timestamps are read back.
usage (GPU box): RFE_LIBRARY=rover-slam_amd/librover_fe_tuning.so RFE_GLAT_ABL=4 python tools/kbench/lat_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rover_slam_amd import capi, weights as Wt  # noqa: E402

ctx = capi.Context(0)
ctx.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=11))
ctx.set_option(capi.OPT_LG_FOLD_WO, 0)
rows = int(os.environ.get("ROWS", "2048"))
rng = np.random.default_rng(0)
x = ctx.alloc(rows * 1024).upload(rng.standard_normal((rows, 256)).astype(np.float32))
m = ctx.alloc(rows * 1024).upload(rng.standard_normal((rows, 256)).astype(np.float32))
o = ctx.alloc(rows * 1024)
for _ in range(5):
    ctx._chk(capi.lib.rfe_k_lightglue_ffn(ctx.h, 0, 0, x.ptr, m.ptr, rows, o.ptr))
nb = 256
buf = (C.c_ulonglong * (nb * 8))()
capi.lib.rfe_k_dbg_timeline.argtypes = [C.c_void_p, C.c_int]
rc = capi.lib.rfe_k_dbg_timeline(buf, nb * 8)
t = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
clk, wall = t[:, :4], t[:, 4:]
print("rc", rc, "blocks with data", int((clk[:, 0] > 0).sum()))
d = np.diff(clk, axis=1)
print("shader-clock cycles per phase (entry->first stage landed, K loop, epilogue): mean", d.mean(0).round(0), "min", d.min(0), "max", d.max(0))
w0 = wall.min()
print("wall clock (10 ns ticks) relative to the earliest entry: entry min/max", wall[:, 0].min() - w0, wall[:, 0].max() - w0,
      "| exit min/max", wall[:, 3].min() - w0, wall[:, 3].max() - w0)
dw = np.diff(wall, axis=1)
print("wall ticks per phase: mean", dw.mean(0).round(1), "max", dw.max(0))
