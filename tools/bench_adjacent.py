#!/usr/bin/env python3
"""Measurement of the adjacent rows SURVEY.md 8(f) N2 / N3 / N4 at Rover-SLAM sizes (GPU box): the device-resident entry points
  rfe_stereo_match_dev            Frame::ComputeStereoMatches            src/Frame.cc:1159-1446
  rfe_search_candidates_dev       SearchByProjection1's descriptor scan  src/Matchers/SPmatcher.cc:1218-1248
  rfe_distinctive_descriptors_dev MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:438-530
  rfe_l2_distance_matrix_dev      DescriptorDistance_sp, all pairs        src/Matchers/SPmatcher.cc:2184-2189
  rfe_binarize_descriptors_dev    Frame::binarize_descriptors            src/Frame.cc:1034-1043
timed with HIP events (rfe_profile_* is per stage; here: wall time of `reps` back-to-back asynchronous calls between two synchronisations), their ALGORITHMIC bytes
(descriptor rows the reference's loop touches, 1 KB each, plus outputs) as GB/s against the 8 TB/s HBM roofline -- these kernels are bandwidth / latency work, no matrix
instructions -- and the same call through the CPU oracle (oracle/rfe_oracle.c, the port of the reference's loops; all usable cores) next to it.  Results are checked
against the oracle before anything is timed (this tool is test / measurement infrastructure like tools/fuzz_parity.py, not product code).  Writes a markdown table to stdout (-> profiles/rNN_adjacent_rows.md).
usage: python tools/bench_adjacent.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(reps):
    from rover_slam_amd import capi, weights as Wt, synth
    from oracle import oracle   # test infrastructure: the checker of every result below, and the timed CPU port next to it (as bench.py's cpu_baseline leg)
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7, desc_center="auto"))
    H, W, K = 480, 752, 1024
    # a stereo pair: the right view is the left one shifted by a disparity (synth's stream gives horizontally shifted frames with shift_step = 8)
    frames, _ = synth.make_frames(2, H, W, seed=31, max_shift=16, shift_step=8)
    dev = lambda a: ctx.alloc(np.ascontiguousarray(a).nbytes).upload(np.ascontiguousarray(a))
    dimg = dev(frames)
    dn, dk, ds, dd = ctx.alloc(2 * 4), ctx.alloc(2 * K * 8), ctx.alloc(2 * K * 4), ctx.alloc(2 * K * 1024)
    ctx._chk(capi.lib.rfe_extract_u8_dev(ctx.h, dimg.ptr, H, W, W, 2, K, 0.0005, dn.ptr, dk.ptr, ds.ptr, dd.ptr))
    n = dn.download((2,), np.int32)
    kxy = dk.download((2, K, 2), np.int32).astype(np.float32)
    desc = dd.download((2, K, 256), np.float32)
    N, Nr = int(n[0]), int(n[1])
    rows = []

    def timed(fn):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    def cpu(fn, min_s=1.0):
        fn()
        t0 = time.perf_counter(); it = 0
        while time.perf_counter() - t0 < min_s:
            fn(); it += 1
        return (time.perf_counter() - t0) / it * 1e6

    # ---- N2: stereo
    dkl, dkr = dev(kxy[0, :N]), dev(kxy[1, :Nr])
    du, dz = ctx.alloc(K * 4), ctx.alloc(K * 4)
    mb, mbf = 0.11, 0.11 * 435.0
    call = lambda: ctx._chk(capi.lib.rfe_stereo_match_dev(ctx.h, dimg.ptr, dimg.ptr + H * W, H, W, W, dkl.ptr, N, dkr.ptr, Nr, dd.ptr, dd.ptr + K * 1024, mb, mbf, du.ptr, dz.ptr))
    call(); ctx.synchronize()
    u_ref, z_ref = oracle.stereo_match(frames[0], frames[1], kxy[0, :N], kxy[1, :Nr], desc[0, :N], desc[1, :Nr], mb, mbf)
    assert np.array_equal(du.download((N,), np.float32), u_ref) and np.array_equal(dz.download((N,), np.float32), z_ref)
    # algorithmic bytes: per left keypoint its own descriptor + the right descriptors of its +-2-row band inside the disparity range (counted on the host), + 2 x 11 x 21 image bytes
    ys = kxy[1, :Nr, 1]
    cands = 0
    for i in range(N):
        x, y = kxy[0, i]
        m = (np.abs(ys - y) <= 2) & (kxy[1, :Nr, 0] <= x) & (kxy[1, :Nr, 0] >= x - mbf / mb)
        cands += int(m.sum())
    bytes_n2 = N * 1024 + cands * 1024 + N * (11 * 11 + 11 * 21) + N * 8
    t = timed(call)
    tc = cpu(lambda: oracle.stereo_match(frames[0], frames[1], kxy[0, :N], kxy[1, :Nr], desc[0, :N], desc[1, :Nr], mb, mbf))
    rows.append((f"N2 `rfe_stereo_match_dev` ({N} x {Nr} keypoints, {W}x{H}, {cands} band candidates, {int((u_ref >= 0).sum())} stereo matches)", t, bytes_n2, tc))

    # ---- N3: search over CSR candidate lists (a local map of 2000 points projected into the frame, ~20 features in each search window)
    rng = np.random.default_rng(5)
    Nq = 2000
    q = desc[0, rng.integers(0, N, Nq)] + 0.05 * rng.standard_normal((Nq, 256)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    lens = rng.integers(5, 36, Nq)
    off = np.zeros(Nq + 1, np.int32); off[1:] = np.cumsum(lens)
    cand = rng.integers(0, Nr, off[-1]).astype(np.int32)
    skip = (rng.random(Nr) < 0.3).astype(np.uint8)
    dq, doff, dc, dsk = dev(q.astype(np.float32)), dev(off), dev(cand), dev(skip)
    dbi, dbd, dsd = ctx.alloc(Nq * 4), ctx.alloc(Nq * 4), ctx.alloc(Nq * 4)
    call = lambda: ctx._chk(capi.lib.rfe_search_candidates_dev(ctx.h, dq.ptr, Nq, dd.ptr + K * 1024, Nr, doff.ptr, dc.ptr, dsk.ptr, dbi.ptr, dbd.ptr, dsd.ptr))
    call(); ctx.synchronize()
    rbi, rbd, rsd = oracle.search_candidates(q.astype(np.float32), desc[1, :Nr], off, cand, skip)
    assert np.array_equal(dbi.download((Nq,), np.int32), rbi) and np.array_equal(dbd.download((Nq,), np.float32), rbd) and np.array_equal(dsd.download((Nq,), np.float32), rsd)
    scanned = int((skip[cand] == 0).sum())
    bytes_s = Nq * 1024 + scanned * 1024 + int(off[-1]) * 5 + Nq * 12
    t = timed(call)
    tc = cpu(lambda: oracle.search_candidates(q.astype(np.float32), desc[1, :Nr], off, cand, skip))
    rows.append((f"N3 `rfe_search_candidates_dev` ({Nq} map points, {int(off[-1])} candidates, {scanned} scanned)", t, bytes_s, tc))

    # ---- N3: distinctive descriptors (1000 map points, 3-15 observations each)
    lens2 = rng.integers(3, 16, 1000).astype(np.int32)
    off2 = np.zeros(len(lens2) + 1, np.int32); off2[1:] = np.cumsum(lens2)
    total = int(off2[-1])
    obs = desc[0, rng.integers(0, N, total)] + 0.1 * rng.standard_normal((total, 256)).astype(np.float32)
    obs = (obs / np.linalg.norm(obs, axis=1, keepdims=True)).astype(np.float32)
    dobs, doff2 = dev(obs), dev(off2)
    dbest, dmed = ctx.alloc(len(lens2) * 4), ctx.alloc(len(lens2) * 4)
    call = lambda: ctx._chk(capi.lib.rfe_distinctive_descriptors_dev(ctx.h, dobs.ptr, doff2.ptr, len(lens2), total, int(lens2.max()), dbest.ptr, dmed.ptr))
    call(); ctx.synchronize()
    rb, rm = oracle.distinctive_descriptors(obs, off2)
    assert np.array_equal(dbest.download((len(lens2),), np.int32), rb) and np.array_equal(dmed.download((len(lens2),), np.float32), rm)
    bytes_d = int((lens2.astype(np.int64) ** 2).sum()) * 1024 + len(lens2) * 8     # the reference computes the full n x n distance table of every point
    t = timed(call)
    tc = cpu(lambda: oracle.distinctive_descriptors(obs, off2))
    rows.append((f"N3 `rfe_distinctive_descriptors_dev` ({len(lens2)} map points, {total} observations)", t, bytes_d, tc))

    # ---- all-pairs distances and binarisation of one frame's descriptors
    dout = ctx.alloc(N * Nr * 4)
    call = lambda: ctx._chk(capi.lib.rfe_l2_distance_matrix_dev(ctx.h, dd.ptr, N, dd.ptr + K * 1024, Nr, dout.ptr))
    t = timed(call)
    a64, b64 = desc[0, :N].astype(np.float64), desc[1, :Nr].astype(np.float64)
    tc = cpu(lambda: np.sqrt(np.maximum((a64 * a64).sum(1)[:, None] + (b64 * b64).sum(1)[None] - 2 * a64 @ b64.T, 0)))
    rows.append((f"N3 `rfe_l2_distance_matrix_dev` ({N} x {Nr}; CPU column: numpy float64 BLAS form, not the reference's loop)", t, (N + Nr) * 1024 + N * Nr * 4, tc))
    dbits = ctx.alloc(N * 256)
    call = lambda: ctx._chk(capi.lib.rfe_binarize_descriptors_dev(ctx.h, dd.ptr, N, dbits.ptr))
    t = timed(call)
    tc = cpu(lambda: (desc[0, :N] > 0).astype(np.uint8))
    rows.append((f"N4 `rfe_binarize_descriptors_dev` ({N} rows; inside the extractor it is a second output of `desc_sample_kernel`, no extra pass)", t, N * 1280, tc))

    print(f"| entry point (workload) | GPU us / call ({reps} back-to-back calls) | algorithmic bytes | GB/s | of 8 TB/s | CPU oracle us / call ({oracle.threads()} threads) | GPU / CPU |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, t, b, tc in rows:
        gbs = b / t * 1e-3
        print(f"| {name} | {t:.1f} | {b / 1e6:.2f} MB | {gbs:.0f} | {gbs / 8000:.3f} | {tc:.0f} | {tc / t:.0f}x |")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200))
