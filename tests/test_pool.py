"""The multi-device pool of the C ABI (rfe_pool_*, rover-slam_amd/csrc/rfe_pool.hip): configs[3] for a C / C++ host.

CPU: the sharding rule (rfe_pool_shard) against rover-slam_amd/sharding.py and its own invariants; creation fails loudly without a GPU.
GPU: a pool of one through RCCL (self send / receive into the root buffer) equals the single-ctx stream call bit for bit; three
members sharing the one device of the box through the COPY transport equal (a) single-ctx calls on each member's shard bit for bit
(the stitching is right) and (b) the one-ctx call on the whole stream (SuperPoint bit-exact, match lists identical up to fp32
borderline flips: smaller shards select other tilings).  RCCL between DIFFERENT devices needs an N-GPU node: unmeasured here.
Reference contrast: one device, src/Extractors/superpoint_onnx.cc:19."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from rover_slam_amd import sharding, synth, weights as Wt


def test_pool_shard_matches_python_sharding_and_covers_the_stream():
    from rover_slam_amd import capi
    for per, world in [(32, 1), (32, 8), (4, 3), (1, 5)]:
        F = per * world + 1
        for r in range(world):
            sh = sharding.shard_frames(per, world, r)
            assert capi.pool_shard(F, world, r) == (sh.start, sh.frames, sh.owned)
    for F in range(1, 40):
        for n in range(1, 9):
            pairs_seen, nxt = 0, 0
            for r in range(n):
                first, frames, own = capi.pool_shard(F, n, r)
                if own:
                    assert first == nxt and frames == own + 1      # shards are consecutive, one overlap frame each
                    nxt = first + own
                else:
                    assert frames == (1 if (F == 1 and r == 0) else 0)
                pairs_seen += own
            assert pairs_seen == F - 1 and (nxt == F - 1 or F == 1)
            owns = [capi.pool_shard(F, n, r)[2] for r in range(n)]
            assert max(owns) - min(owns) <= 1                      # as even as possible
    a = C.c_int()
    assert capi.lib.rfe_pool_shard(0, 1, 0, C.byref(a), None, None) < 0 and capi.lib.rfe_pool_shard(5, 2, 2, None, None, None) < 0


def test_pool_fails_loudly_without_gpu_and_on_bad_arguments():
    import torch
    from rover_slam_amd import capi
    h = C.c_void_p()
    assert capi.lib.rfe_pool_create(None, 1, C.byref(h)) == -1 and h.value is None
    assert capi.lib.rfe_pool_create((C.c_int * 1)(0), 0, C.byref(h)) == -1
    assert capi.lib.rfe_pool_size(None) == 0 and capi.lib.rfe_pool_ctx(None, 0) is None
    assert capi.lib.rfe_pool_extract_match_stream(None, None, 8, 8, 8, 1, 1, 0.0, 0.0, 0, None, None, None, None, None, None, None) < 0
    if torch.cuda.device_count() == 0:
        with pytest.raises(capi.RfeError) as e:
            capi.Pool([0])
        assert "no HIP device" in str(e.value)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_pool_driver(tmp_path):
    exe = str(tmp_path / "pool_driver")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pool_driver.c"),
           "-o", exe, "-L" + os.path.join(ROOT, "rover-slam_amd"), "-lrover_fe", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "rover-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_plain_c_host_of_the_pool_compiles_and_fails_loudly_without_gpu(tmp_path):
    """include/rover_fe.h is C-clean (C99, -Wall -Werror) and a plain-C host links against librover_fe.so alone (no RCCL, no torch)."""
    import torch
    exe = _build_pool_driver(tmp_path)
    (tmp_path / "f.u8").write_bytes(bytes(2 * 64 * 64))
    r = subprocess.run([exe, str(tmp_path / "f.u8"), "2", "64", "64", "32", "1", "0", "none", "none", str(tmp_path / "o.bin")], capture_output=True, text=True)
    if torch.cuda.device_count() == 0:
        assert r.returncode == 1 and "no HIP device" in r.stderr


def _same_matches(a, b, lo=0, hi=None, b_lo=0):
    """S equal and, pair by pair, the first S entries of pairs / ms equal (entries past S are unspecified)."""
    hi = len(a["S"]) if hi is None else hi
    for p in range(lo, hi):
        q = p - lo + b_lo
        s = int(a["S"][p])
        if s != int(b["S"][q]) or not np.array_equal(a["pairs"][p, :s], b["pairs"][q, :s]) or not np.array_equal(a["ms"][p, :s], b["ms"][q, :s]):
            return False
    return True


def _single_ctx_stream(ctx, frames, kmax):
    from rover_slam_amd import capi
    B, H, W = frames.shape
    P = max(B - 1, 1)
    spec = [("n", np.int32, (B,)), ("kxy", np.int32, (B, kmax, 2)), ("score", np.float32, (B, kmax)), ("desc", np.float32, (B, kmax, 256)),
            ("S", np.int32, (P,)), ("pairs", np.int32, (P, kmax, 2)), ("ms", np.float32, (P, kmax))]
    dimg = ctx.alloc(frames.nbytes).upload(frames)
    bufs = {nm: ctx.alloc(int(np.prod(sh)) * np.dtype(dt).itemsize) for nm, dt, sh in spec}
    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, kmax, 0.0005, 0.1, bufs["n"].ptr, bufs["kxy"].ptr, bufs["score"].ptr,
                                                   bufs["desc"].ptr, bufs["S"].ptr, bufs["pairs"].ptr, bufs["ms"].ptr))
    ctx.synchronize()
    out = {nm: bufs[nm].download(sh, dt) for nm, dt, sh in spec}
    for b in list(bufs.values()) + [dimg]:
        b.free()
    return out


@pytest.fixture(scope="module")
def wsets():
    return Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)


@pytest.mark.gpu
def test_pool_of_one_through_rccl_equals_single_ctx(wsets):
    from rover_slam_amd import capi
    frames, _ = synth.make_frames(5, 240, 320, seed=5)
    kmax = 512
    pool = capi.Pool([0])
    try:
        assert pool.size == 1
        assert pool.has_rccl, "librccl could not be opened / ncclCommInitAll failed on the GPU box"
        pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
        got = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)
        again = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_COPY)
    finally:
        pool.close()
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsets[0]); ctx.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
    ref = _single_ctx_stream(ctx, frames, kmax)
    ctx.close()
    assert ref["n"].min() > 50 and ref["S"].sum() > 0
    for k in ("n", "kxy", "score", "desc"):
        assert np.array_equal(got[k], ref[k]), f"RCCL transport: {k} differs from the single-ctx call"
        assert np.array_equal(again[k], ref[k]), f"COPY transport: {k} differs from the single-ctx call"
    assert _same_matches(got, ref), "RCCL transport: matches differ from the single-ctx call"
    assert _same_matches(again, ref), "COPY transport: matches differ from the single-ctx call"


@pytest.mark.gpu
def test_pool_three_members_on_one_device_copy_transport(wsets, oracle):
    from rover_slam_amd import capi
    from tolerances import LG_SCORE_TOL, lists_agree
    F, kmax = 11, 256                                      # 10 pairs over 3 members: 4 + 3 + 3
    frames, _ = synth.make_frames(F, 240, 320, seed=9)
    pool = capi.Pool([0, 0, 0])
    try:
        assert pool.size == 3 and not pool.has_rccl          # members share a device: no communicator
        pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
        ids = {capi.lib.rfe_weights_id(capi.lib.rfe_pool_ctx(pool.h, r), capi.KIND_LIGHTGLUE) for r in range(3)}
        assert len(ids) == 1 and 0 not in ids                # one shared device copy of the weights
        with pytest.raises(capi.RfeError) as e:
            pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)
        assert "share a device" in str(e.value)
        got = pool.extract_match_stream(frames, kmax=kmax)   # AUTO -> COPY
        got2 = pool.extract_match_stream(frames, kmax=kmax, with_desc=False)
    finally:
        pool.close()
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsets[0]); ctx.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
    whole = _single_ctx_stream(ctx, frames, kmax)
    # (a) each member's rows = a single-ctx call on its shard, bit for bit
    for r in range(3):
        first, nfr, own = capi.pool_shard(F, 3, r)
        part = _single_ctx_stream(ctx, np.ascontiguousarray(frames[first:first + nfr]), kmax)
        rows = nfr if r == 2 else own
        for k in ("n", "kxy", "score", "desc"):
            assert np.array_equal(got[k][first:first + rows], part[k][:rows]), (r, k)
        assert _same_matches(got, part, first, first + own), r
    ctx.close()
    for k in ("n", "kxy"):
        assert np.array_equal(got2[k], got[k])
    assert _same_matches(got2, got)
    # (b) against the one-ctx call on the whole stream: SuperPoint bit-exact, matches up to fp32 borderline flips
    for k in ("n", "kxy", "score", "desc"):
        assert np.array_equal(got[k], whole[k]), k
    assert whole["S"].sum() > 0
    for p in range(F - 1):
        a, b = int(got["S"][p]), int(whole["S"][p])
        # two fp32 evaluations (4- / 3-pair shards against the 10-pair batch: other GEMM / attention tilings).  Measured on this very stream
        # (tools/diag_pool_pair.py -> profiles/r03_pool_pair_diag.md): nine pairs agree to <= 8.5e-5, pair 5 to 2.0e-4 -- there the fp32 CPU
        # oracle itself sits 2.65e-4 from a float64 evaluation, the whole-stream call 9e-5 from the oracle and the shard call 1.4e-4 from
        # float64: the fp32 noise floor of a match score is not smaller at 256 keypoints than at 1024, so the stated LG_SCORE_TOL applies
        ok, dev = lists_agree(got["pairs"][p, :a], got["ms"][p, :a], whole["pairs"][p, :b], whole["ms"][p, :b], slack=LG_SCORE_TOL)
        assert ok and dev < LG_SCORE_TOL, (p, a, b, dev)


@pytest.mark.gpu
def test_pool_applies_hyper_parameters_to_every_member(wsets, oracle):
    """rfe_pool_set_hparams (a graph exported with another NMS radius / border / top-k rule, include/rover_fe.h rfe_hparams): every member must
    extract with the new values -- checked against the oracle on a frame of each member's shard -- and an invalid block must be refused."""
    from rover_slam_amd import capi
    F, kmax = 7, 300
    frames, _ = synth.make_frames(F, 160, 208, seed=21)
    pool = capi.Pool([0, 0, 0])
    try:
        pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
        base = pool.extract_match_stream(frames, kmax=kmax)
        pool.set_hparams(sp_nms_radius=3, sp_remove_borders=2, sp_topk_always=1)
        for r in range(3):
            h = capi.HParams()
            assert capi.lib.rfe_get_hparams(capi.lib.rfe_pool_ctx(pool.h, r), h) == 0
            assert (h.sp_nms_radius, h.sp_remove_borders, h.sp_topk_always) == (3, 2, 1)
        got = pool.extract_match_stream(frames, kmax=kmax)
        with pytest.raises(capi.RfeError):
            pool.set_hparams(sp_nms_radius=9)
    finally:
        pool.close()
    assert not np.array_equal(base["kxy"], got["kxy"])
    for r in range(3):
        first, nfr, own = capi.pool_shard(F, 3, r)
        i = first + (nfr - 1 if r == 2 else own - 1)          # a frame this member owns
        ref = oracle.superpoint(wsets[0], frames[i], kmax=kmax, nms_radius=3, border=2, topk_always=True)
        assert got["n"][i] == ref["n"] > 20
        assert np.array_equal(got["kxy"][i], ref["kxy"]) and np.array_equal(got["score"][i], ref["score"]) and np.array_equal(got["desc"][i], ref["desc"])


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_plain_c_host_runs_the_pool(tmp_path, wsets):
    """tests/cpp/pool_driver.c (C99, no Python in the process): two members on the box's one device, weights from RFEW files, results
    equal to the ctypes pool call."""
    from rover_slam_amd import capi
    F, H, W, kmax = 7, 240, 320, 256
    frames, _ = synth.make_frames(F, H, W, seed=21)
    frames.tofile(str(tmp_path / "frames.u8"))
    Wt.save(str(tmp_path / "sp.rfew"), wsets[0], 1); Wt.save(str(tmp_path / "lg.rfew"), wsets[1], 2)
    exe = _build_pool_driver(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "frames.u8"), str(F), str(H), str(W), str(kmax), "2", "0", str(tmp_path / "sp.rfew"),
                        str(tmp_path / "lg.rfew"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "member 1: frames [3, 7), 3 pairs" in r.stdout
    raw = np.fromfile(str(tmp_path / "out.bin"), np.int32)
    P = F - 1
    o = 0
    got = {}
    for nm, cnt, shape in (("n", F, (F,)), ("S", P, (P,)), ("kxy", F * kmax * 2, (F, kmax, 2)), ("pairs", P * kmax * 2, (P, kmax, 2)), ("ms", P * kmax, (P, kmax))):
        got[nm] = raw[o:o + cnt].reshape(shape); o += cnt
    got["ms"] = got["ms"].view(np.float32)
    assert o == raw.size
    pool = capi.Pool([0, 0])
    try:
        pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
        ref = pool.extract_match_stream(frames, kmax=kmax, with_desc=False)
    finally:
        pool.close()
    assert ref["S"].sum() > 0
    assert np.array_equal(got["n"], ref["n"]) and np.array_equal(got["kxy"], ref["kxy"]) and _same_matches(got, ref)


@pytest.mark.gpu
def test_pool_gather_failure_falls_back_to_copy(wsets):
    """a member whose gather fails inside its ncclGroup aborts EVERY member's communicator, so that no healthy member is left in
    hipStreamSynchronize on transfers that wait for the failed peer; the call then delivers through COPY (AUTO) or reports the failure (RCCL
    requested) and later calls stay on COPY.  One member on a 1-GPU box; with >= 2 GPUs the partial failure (member 1 of 2) runs as well."""
    import torch
    from rover_slam_amd import capi
    frames, _ = synth.make_frames(5, 160, 208, seed=5)
    kmax = 256
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsets[0]); ctx.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
    ref = _single_ctx_stream(ctx, frames, kmax)
    ctx.close()
    layouts = [([0], 0)] + ([([0, 1], 1)] if torch.cuda.device_count() >= 2 else [])
    for devices, victim in layouts:
        pool = capi.Pool(devices)
        try:
            assert pool.has_rccl
            pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
            ok = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)
            assert capi.lib.rfe_k_pool_inject_gather_failure(pool.h, victim) == 0
            with pytest.raises(capi.RfeError, match="injected"):
                pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)      # explicit RCCL: the failure is reported, nothing hangs
            assert not pool.has_rccl                                                         # communicators are gone for good
            after = pool.extract_match_stream(frames, kmax=kmax)                             # AUTO -> COPY
        finally:
            pool.close()
        for got in (ok, after):
            for k in ("n", "kxy", "score", "desc"):
                assert np.array_equal(got[k], ref[k]), k
            assert _same_matches(got, ref)
    # AUTO with more than one member takes RCCL by itself: the injected failure must be absorbed inside the call
    if torch.cuda.device_count() >= 2:
        pool = capi.Pool([0, 1])
        try:
            pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
            capi.lib.rfe_k_pool_inject_gather_failure(pool.h, 1)
            got = pool.extract_match_stream(frames, kmax=kmax)
            assert not pool.has_rccl and _same_matches(got, ref)
        finally:
            pool.close()


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_pool_over_every_visible_device_through_rccl(wsets):
    """RCCL between DISTINCT devices (ncclCommInitAll over all of them, grouped ncclSend / ncclRecv into member 0's root buffer over xGMI): runs by
    itself the day the suite lands on a multi-GPU box -- on the 1-GPU pool of this build it is skipped, and says so.  Every device count from 2 up to
    the visible one; shards that are ragged (11 frames over 2 .. 8 members) so that the per-member row counts and offsets differ."""
    from rover_slam_amd import capi
    n = _n_gpus()
    if n < 2:
        pytest.skip(f"{n} GPU visible: RCCL between distinct devices needs at least two (the world-size-1 RCCL path is test_pool_of_one_through_rccl_equals_single_ctx)")
    frames, _ = synth.make_frames(11, 160, 208, seed=5)
    kmax = 256
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsets[0]); ctx.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
    ref = _single_ctx_stream(ctx, frames, kmax)
    ctx.close()
    for m in sorted({2, n}):
        pool = capi.Pool(list(range(m)))
        try:
            assert pool.size == m and pool.has_rccl
            pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
            first = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)     # the communicators' first collective (connection set-up)
            again = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)
            copy = pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_COPY)
        finally:
            pool.close()
        for got in (first, again, copy):
            for k in ("n", "kxy", "score", "desc"):
                assert np.array_equal(got[k], ref[k]), (m, k)
            assert _same_matches(got, ref), m


@pytest.mark.gpu
@pytest.mark.parametrize("members", [1, 2])
def test_pool_failure_on_the_very_first_gather(wsets, members):
    """the failing member aborts its OWN communicator at once, before it asks for the exclusive lock -- a peer whose ncclGroupEnd is
    still inside first-use connection set-up with the failed member holds the shared lock until that abort releases it.  The failure is therefore
    injected into the FIRST gather a fresh pool ever runs (no connection exists yet); the call must come back -- through COPY under AUTO -- and the
    pool must keep working.  Two members need two GPUs (skipped otherwise)."""
    from rover_slam_amd import capi
    if _n_gpus() < members:
        pytest.skip(f"{members} members on distinct devices need {members} GPUs")
    frames, _ = synth.make_frames(5, 160, 208, seed=5)
    kmax = 256
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsets[0]); ctx.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
    ref = _single_ctx_stream(ctx, frames, kmax)
    ctx.close()
    pool = capi.Pool(list(range(members)))
    try:
        assert pool.has_rccl
        pool.set_weights(capi.KIND_SUPERPOINT, wsets[0]); pool.set_weights(capi.KIND_LIGHTGLUE, wsets[1])
        assert capi.lib.rfe_k_pool_inject_gather_failure(pool.h, members - 1) == 0
        if members == 1:
            with pytest.raises(capi.RfeError, match="injected"):
                pool.extract_match_stream(frames, kmax=kmax, transport=capi.POOL_RCCL)
        else:
            got = pool.extract_match_stream(frames, kmax=kmax)          # AUTO takes RCCL with two members: the failure is absorbed inside the call
            assert _same_matches(got, ref)
        assert not pool.has_rccl
        after = pool.extract_match_stream(frames, kmax=kmax)
    finally:
        pool.close()
    for k in ("n", "kxy", "score", "desc"):
        assert np.array_equal(after[k], ref[k]), k
    assert _same_matches(after, ref)
