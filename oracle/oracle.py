"""ctypes wrapper over oracle/librfe_oracle.so -- CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED w.r.t. the true reference (see oracle/rfe_oracle.h header).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librfe_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "rfe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librfe_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def usable_cpus():
    """CPUs this process can really run on: min(logical CPUs, affinity mask, cgroup v2/v1 CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp, ip, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.rfo_sp_weight_count.restype = C.c_int64
        L.rfo_lg_weight_count.restype = C.c_int64
        L.rfo_sp_layer_offset.restype = C.c_int64
        L.rfo_sp_layer_offset.argtypes = [C.c_int, C.c_int]
        L.rfo_expf.restype = C.c_float
        L.rfo_expf.argtypes = [C.c_float]
        L.rfo_conv3x3.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.rfo_linear.argtypes = [fp, C.c_int, C.c_int, fp, fp, C.c_int, fp]
        L.rfo_softmax65_d2s.argtypes = [fp, C.c_int, C.c_int, fp]
        L.rfo_nms.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.rfo_sumsq256.restype = C.c_float
        L.rfo_sumsq256.argtypes = [fp]
        L.rfo_l2norm256.argtypes = [fp, fp]
        L.rfo_superpoint.restype = C.c_int
        L.rfo_superpoint.argtypes = [fp, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                     ip, fp, fp, fp, fp, fp, fp]
        L.rfo_superpoint_ex.restype = C.c_int
        L.rfo_superpoint_ex.argtypes = [fp, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                        ip, fp, fp, fp, fp, fp, fp]
        L.rfo_superpoint_f32.restype = C.c_int
        L.rfo_superpoint_f32.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                         ip, fp, fp, fp, fp, fp, fp]
        L.rfo_lightglue.restype = C.c_int
        L.rfo_lightglue.argtypes = [fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_float, ip, fp, fp, fp, fp]
        L.rfo_normalize_keypoints.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.rfo_postprocess_fused.restype = C.c_int
        L.rfo_postprocess_fused.argtypes = [ip, fp, C.c_int, C.c_float, ip, C.c_int]
        L.rfo_stereo_match.argtypes = [u8p, u8p, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp, C.c_float, C.c_float, fp, fp]
        L.rfo_set_num_threads.argtypes = [C.c_int]
        L.rfo_get_max_threads.restype = C.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            L.rfo_set_num_threads(usable_cpus())
        L.rfo_search_candidates.argtypes = [fp, C.c_int, fp, ip, ip, u8p, ip, fp, fp]
        L.rfo_distinctive_descriptors.argtypes = [fp, ip, C.c_int, ip, fp]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _opt(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def sp_weight_count():
    return int(lib().rfo_sp_weight_count())


def lg_weight_count():
    return int(lib().rfo_lg_weight_count())


def expf(x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.ravel(), out.ravel()
    for i in range(flat_in.size):
        flat_out[i] = L.rfo_expf(float(flat_in[i]))
    return out


def conv3x3(x_nhwc, w_oihw, bias, relu=True, pool=False):
    """x: [H,W,Cin] f32; w: [Cout,Cin,3,3]; returns [H',W',Cout]."""
    x, xp = _f(x_nhwc)
    w, wp = _f(w_oihw)
    b, bp = _f(bias)
    H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = np.empty((Ho, Wo, Cout), np.float32)
    lib().rfo_conv3x3(xp, H, W, Cin, wp, bp, Cout, int(relu), int(pool), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def linear(a, w, bias=None):
    a, ap = _f(a)
    w, wp = _f(w)
    M, K = a.shape
    N = w.shape[0]
    out = np.empty((M, N), np.float32)
    if bias is not None:
        b, bp = _f(bias)
    else:
        bp = None
    lib().rfo_linear(ap, M, K, wp, bp, N, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def softmax65_d2s(logits, Hc, Wc):
    l, lp = _f(logits)
    out = np.empty((Hc * 8, Wc * 8), np.float32)
    lib().rfo_softmax65_d2s(lp, Hc, Wc, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def nms(score, radius=4):
    s, sp = _f(score)
    out = np.empty_like(s)
    lib().rfo_nms(sp, s.shape[0], s.shape[1], radius, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def l2norm256(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    flat = x.reshape(-1, 256)
    o = out.reshape(-1, 256)
    for i in range(flat.shape[0]):
        lib().rfo_l2norm256(flat[i].ctypes.data_as(C.POINTER(C.c_float)), o[i].ctypes.data_as(C.POINTER(C.c_float)))
    return out


def superpoint(weights, img_u8, kmax=1024, thr=0.0005, nms_radius=4, border=4, debug=False, topk_always=False):
    """One frame.  Returns dict(n, kxy[Kmax,2] i32, score[Kmax], desc[Kmax,256] (+ debug taps)).
    A float32 image is taken as already normalised (the reference's float entry, superpoint_onnx.cc:88-118); anything else as u8."""
    w, wp = _f(weights)
    assert w.size == sp_weight_count()
    is_f32 = isinstance(img_u8, np.ndarray) and img_u8.dtype == np.float32
    img = np.ascontiguousarray(img_u8, dtype=np.float32 if is_f32 else np.uint8)
    H, W = img.shape
    Hs, Ws = H // 8 * 8, W // 8 * 8          # score-map frame (= the image when H, W are multiples of 8)
    kxy = np.zeros((kmax, 2), np.int32)
    score = np.zeros((kmax,), np.float32)
    desc = np.zeros((kmax, 256), np.float32)
    dbg = {}
    if debug:
        dbg = dict(scoremap=np.empty((Hs, Ws), np.float32), nms=np.empty((Hs, Ws), np.float32),
                   descmap=np.empty((H // 8, W // 8, 256), np.float32),
                   feat=np.empty((H // 8, W // 8, 128), np.float32))
    entry = lib().rfo_superpoint_f32 if is_f32 else lib().rfo_superpoint_ex
    n = entry(wp, img.ctypes.data_as(C.POINTER(C.c_float if is_f32 else C.c_uint8)), H, W, kmax, thr, nms_radius, border, int(topk_always),
                             kxy.ctypes.data_as(C.POINTER(C.c_int32)), _opt(score), _opt(desc),
                             _opt(dbg.get("scoremap")), _opt(dbg.get("nms")), _opt(dbg.get("descmap")),
                             _opt(dbg.get("feat")))
    out = dict(n=int(n), kxy=kxy, score=score, desc=desc)
    out.update(dbg)
    return out


def lightglue(weights, k0n, k1n, d0, d1, filter_thr=0.1, debug=False):
    """One pair.  k*n normalised keypoints [M,2]; d* [M,256].  Returns dict(S, pairs[S,2], ms[S])."""
    w, wp = _f(weights)
    assert w.size == lg_weight_count()
    k0, k0p = _f(k0n)
    k1, k1p = _f(k1n)
    a0, a0p = _f(d0)
    a1, a1p = _f(d1)
    M, N = k0.shape[0], k1.shape[0]
    cap = max(1, min(M, N))
    pairs = np.zeros((cap, 2), np.int32)
    ms = np.zeros((cap,), np.float32)
    dbg = {}
    if debug:
        dbg = dict(x0=np.empty((M, 256), np.float32), x1=np.empty((N, 256), np.float32),
                   scores=np.empty((M, N), np.float32))
    S = lib().rfo_lightglue(wp, k0p, k1p, a0p, a1p, M, N, filter_thr,
                            pairs.ctypes.data_as(C.POINTER(C.c_int32)), _opt(ms),
                            _opt(dbg.get("x0")), _opt(dbg.get("x1")), _opt(dbg.get("scores")))
    out = dict(S=int(S), pairs=pairs[:S].copy(), ms=ms[:S].copy())
    out.update(dbg)
    return out


def normalize_keypoints(kxy, h, w):
    k, kp = _f(kxy)
    out = np.empty_like(k)
    lib().rfo_normalize_keypoints(kp, k.shape[0], h, w, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def postprocess_fused(pairs, ms, match_thresh, M):
    pairs = np.ascontiguousarray(pairs, np.int32)
    ms = np.ascontiguousarray(ms, np.float32)
    vn = np.full((M,), -1, np.int32)
    size = lib().rfo_postprocess_fused(pairs.ctypes.data_as(C.POINTER(C.c_int32)), _opt(ms), int(ms.shape[0]),
                                       match_thresh, vn.ctypes.data_as(C.POINTER(C.c_int32)), M)
    return int(size), vn


def stereo_match(img_l, img_r, k_l, k_r, d_l, d_r, mb, mbf):
    """Frame::ComputeStereoMatches restatement.  k_*: [N,2] pixel keypoints; returns (uRight[N], depth[N])."""
    il = np.ascontiguousarray(img_l, np.uint8); ir = np.ascontiguousarray(img_r, np.uint8)
    H, W = il.shape
    kl, klp = _f(k_l); kr, krp = _f(k_r)
    dl, dlp = _f(d_l); dr, drp = _f(d_r)
    N, Nr = kl.shape[0], kr.shape[0]
    u = np.empty((max(N, 1),), np.float32); d = np.empty((max(N, 1),), np.float32)
    lib().rfo_stereo_match(il.ctypes.data_as(C.POINTER(C.c_uint8)), ir.ctypes.data_as(C.POINTER(C.c_uint8)), H, W,
                           klp, N, krp, Nr, dlp, drp, mb, mbf, _opt(u), _opt(d))
    return u[:N], d[:N]


def search_candidates(q, f, offsets, cand, skip=None):
    """Best / second-best candidate scan (SPmatcher.cc:1218-1248).  Returns (best_idx, best_dist, second_dist)."""
    qa, qp = _f(q); fa, fpp = _f(f)
    off = np.ascontiguousarray(offsets, np.int32); cd = np.ascontiguousarray(cand, np.int32)
    Nq = qa.shape[0]
    sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    bi = np.empty((max(Nq, 1),), np.int32); bd = np.empty((max(Nq, 1),), np.float32); sd = np.empty((max(Nq, 1),), np.float32)
    ip = C.POINTER(C.c_int32)
    lib().rfo_search_candidates(qp, Nq, fpp, off.ctypes.data_as(ip), cd.ctypes.data_as(ip),
                                None if sk is None else sk.ctypes.data_as(C.POINTER(C.c_uint8)),
                                bi.ctypes.data_as(ip), _opt(bd), _opt(sd))
    return bi[:Nq], bd[:Nq], sd[:Nq]


def distinctive_descriptors(desc, offsets):
    """MapPoint::ComputeDistinctiveDescriptors batched (MapPoint.cc:438-530).  Returns (best[Np], median[Np])."""
    da, dp = _f(desc)
    off = np.ascontiguousarray(offsets, np.int32)
    Np = off.shape[0] - 1
    b = np.empty((max(Np, 1),), np.int32); m = np.empty((max(Np, 1),), np.float32)
    ip = C.POINTER(C.c_int32)
    lib().rfo_distinctive_descriptors(dp, off.ctypes.data_as(ip), Np, b.ctypes.data_as(ip), _opt(m))
    return b[:Np], m[:Np]


def threads():
    """OpenMP threads the oracle runs on."""
    return int(lib().rfo_get_max_threads())
