"""ctypes binding of librover_fe.so (include/rover_fe.h).  The HIP library is the product's only
compute path: importing this module raises ImportError when the library has not been built, and
`Context()` raises RuntimeError when no gfx950 device can be opened -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFE_LIBRARY selects another build of the same ABI (tools/ use the -DRFE_TUNING build for A/B measurements)
LIB_PATH = os.environ.get("RFE_LIBRARY") or os.path.join(_HERE, "librover_fe.so")

KIND_SUPERPOINT, KIND_LIGHTGLUE = 1, 2
OPT_LG_FOLD_WO = 1
OPT_LG_FP16X2 = 2    # LightGlue Linears + attention as split products on the f16 matrix pipe (default off)
OPT_HOST_GRAPH = 3   # host entries replay a captured hipGraph per repeated call shape (default off; helps fixed-capacity callers only)

EXPORTS = [
    "rfe_init", "rfe_destroy", "rfe_last_error", "rfe_version", "rfe_load_weights", "rfe_load_onnx", "rfe_set_weights",
    "rfe_weight_count", "rfe_weights_id", "rfe_get_hparams", "rfe_set_hparams", "rfe_set_option", "rfe_get_option", "rfe_set_stream", "rfe_synchronize", "rfe_malloc", "rfe_free", "rfe_host_malloc", "rfe_host_free", "rfe_workspace_bytes", "rfe_memcpy_h2d",
    "rfe_memcpy_d2h", "rfe_extract_u8", "rfe_extract_u8_dev", "rfe_extract_f32", "rfe_extract_f32_dev", "rfe_extract_u8_bin", "rfe_extract_u8_bin_dev", "rfe_match", "rfe_match_dev", "rfe_match_fused",
    "rfe_extract_match_stream_dev", "rfe_stereo_match", "rfe_stereo_match_dev", "rfe_stereo_frame_dev", "rfe_l2_distance_matrix", "rfe_binarize_descriptors",
    "rfe_search_candidates", "rfe_distinctive_descriptors",
    "rfe_l2_distance_matrix_dev", "rfe_binarize_descriptors_dev", "rfe_search_candidates_dev", "rfe_distinctive_descriptors_dev",
    "rfe_pool_create", "rfe_pool_destroy", "rfe_pool_last_error", "rfe_pool_size", "rfe_pool_ctx", "rfe_pool_has_rccl", "rfe_pool_set_weights",
    "rfe_pool_load_weights", "rfe_pool_set_option", "rfe_pool_set_hparams", "rfe_pool_shard", "rfe_pool_extract_match_stream",
    "rfe_profile_enable", "rfe_profile_filter", "rfe_profile_reset", "rfe_profile_read",
    "rfe_k_conv3x3", "rfe_k_linear", "rfe_k_scoremap", "rfe_k_select", "rfe_k_select_keys", "rfe_k_lightglue_taps", "rfe_k_set_lightglue_tap", "rfe_k_lightglue_ffn", "rfe_k_attention", "rfe_k_lightglue_self_attention", "rfe_k_pool_inject_gather_failure", "rfe_k_onnx_convert",
]

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(rover-slam_amd/csrc/Makefile). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

_vp, _fp, _ip, _u8p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses (host or device)
lib.rfe_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
lib.rfe_destroy.argtypes = [C.c_void_p]
lib.rfe_destroy.restype = None
lib.rfe_last_error.argtypes = [C.c_void_p]
lib.rfe_last_error.restype = C.c_char_p
lib.rfe_version.restype = C.c_char_p
lib.rfe_load_weights.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
lib.rfe_set_weights.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int64]
lib.rfe_weight_count.argtypes = [C.c_int]
lib.rfe_weight_count.restype = C.c_int64
lib.rfe_weights_id.restype = C.c_uint64
lib.rfe_weights_id.argtypes = [C.c_void_p, C.c_int]


class HParams(C.Structure):
    """include/rover_fe.h: rfe_hparams -- the constants baked into the reference's two .onnx graphs (RFEW v2 header)."""
    _fields_ = [("sp_max_keypoints", C.c_int32), ("sp_detection_threshold", C.c_float), ("sp_nms_radius", C.c_int32),
                ("sp_remove_borders", C.c_int32), ("sp_topk_always", C.c_int32), ("lg_layers", C.c_int32), ("lg_heads", C.c_int32), ("lg_filter_threshold", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


lib.rfe_get_hparams.argtypes = [C.c_void_p, C.POINTER(HParams)]
lib.rfe_set_hparams.argtypes = [C.c_void_p, C.POINTER(HParams)]
lib.rfe_pool_set_hparams.argtypes = [C.c_void_p, C.POINTER(HParams)]
lib.rfe_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.rfe_get_option.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
lib.rfe_set_stream.argtypes = [C.c_void_p, C.c_void_p]
lib.rfe_synchronize.argtypes = [C.c_void_p]
lib.rfe_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.rfe_free.argtypes = [C.c_void_p, C.c_void_p]
lib.rfe_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.rfe_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
_ext = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _ip, _ip, _fp, _fp]
lib.rfe_extract_u8.argtypes = _ext
lib.rfe_extract_f32.argtypes = _ext
lib.rfe_extract_f32_dev.argtypes = _ext
lib.rfe_extract_u8_dev.argtypes = _ext
lib.rfe_extract_u8_bin.argtypes = _ext + [_u8p]
lib.rfe_extract_u8_bin_dev.argtypes = _ext + [_u8p]
_mt = [C.c_void_p, _fp, _fp, _fp, _fp, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_float, _ip, _ip, _fp]
lib.rfe_match.argtypes = _mt
lib.rfe_match_dev.argtypes = _mt
lib.rfe_match_fused.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_float,
                                C.c_float, _ip]
lib.rfe_extract_match_stream_dev.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.c_float, _ip, _ip, _fp, _fp, _ip, _ip, _fp]
_st = [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_float, C.c_float, _fp, _fp]
lib.rfe_stereo_match.argtypes = _st
lib.rfe_stereo_match_dev.argtypes = _st
lib.rfe_stereo_frame_dev.argtypes = [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_int, _ip, _ip, _fp, _fp, _fp, _fp, _ip, _ip, _fp]
lib.rfe_l2_distance_matrix.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _fp]
lib.rfe_binarize_descriptors.argtypes = [C.c_void_p, _fp, C.c_int, _u8p]
lib.rfe_search_candidates.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _ip, _ip, _u8p, _ip, _fp, _fp]
lib.rfe_distinctive_descriptors.argtypes = [C.c_void_p, _fp, _ip, C.c_int, _ip, _fp]
lib.rfe_l2_distance_matrix_dev.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _fp]
lib.rfe_binarize_descriptors_dev.argtypes = [C.c_void_p, _fp, C.c_int, _u8p]
lib.rfe_search_candidates_dev.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int, _ip, _ip, _u8p, _ip, _fp, _fp]
lib.rfe_distinctive_descriptors_dev.argtypes = [C.c_void_p, _fp, _ip, C.c_int, C.c_int, C.c_int, _ip, _fp]
lib.rfe_profile_enable.argtypes = [C.c_void_p, C.c_int]
lib.rfe_profile_filter.argtypes = [C.c_void_p, C.c_char_p]
lib.rfe_profile_reset.argtypes = [C.c_void_p]
lib.rfe_profile_read.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
lib.rfe_k_conv3x3.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]
lib.rfe_k_linear.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, _fp]
lib.rfe_k_scoremap.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]
lib.rfe_k_select.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _ip, _ip, _fp]
lib.rfe_k_select_keys.argtypes = lib.rfe_k_select.argtypes
lib.rfe_k_lightglue_taps.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]
lib.rfe_k_set_lightglue_tap.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp]
lib.rfe_k_lightglue_ffn.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp]
lib.rfe_k_lightglue_self_attention.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.c_void_p, C.c_int, C.c_int, _fp, _fp, C.POINTER(C.c_int32)]
lib.rfe_k_attention.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _fp]


POOL_AUTO, POOL_RCCL, POOL_COPY = 0, 1, 2
lib.rfe_pool_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
lib.rfe_pool_destroy.argtypes = [C.c_void_p]
lib.rfe_pool_destroy.restype = None
lib.rfe_pool_last_error.argtypes = [C.c_void_p]
lib.rfe_pool_last_error.restype = C.c_char_p
lib.rfe_pool_size.argtypes = [C.c_void_p]
lib.rfe_pool_ctx.argtypes = [C.c_void_p, C.c_int]
lib.rfe_pool_ctx.restype = C.c_void_p
lib.rfe_pool_has_rccl.argtypes = [C.c_void_p]
lib.rfe_k_pool_inject_gather_failure.argtypes = [C.c_void_p, C.c_int]
lib.rfe_load_onnx.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
lib.rfe_k_onnx_convert.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_char_p, C.c_int]
lib.rfe_host_malloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
lib.rfe_host_free.argtypes = [C.c_void_p]
lib.rfe_host_free.restype = None
lib.rfe_workspace_bytes.argtypes = [C.c_void_p]
lib.rfe_workspace_bytes.restype = C.c_int64
lib.rfe_pool_set_weights.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int64]
lib.rfe_pool_load_weights.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
lib.rfe_pool_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.rfe_pool_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.rfe_pool_extract_match_stream.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                              _ip, _ip, _fp, _fp, _ip, _ip, _fp]


class RfeError(RuntimeError):
    pass


def _addr(a):
    """address of a numpy array (host) / int (device pointer) / object with data_ptr() (torch)."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(type(a))


class DevBuf:
    """Device allocation owned by a Context (rfe_malloc / rfe_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._chk(lib.rfe_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(lib.rfe_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._chk(lib.rfe_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib.rfe_free(self.ctx.h, self.ptr)
            self.ptr = None


class StereoStream:
    """Device-resident stereo stream (rfe_stereo_frame_dev): owns the output buffers, push() enqueues one stereo frame
    (device image pointers) without any host synchronisation, results() downloads the last frame's outputs."""

    def __init__(self, ctx, H, W, kmax=1024, mb=0.11, mbf=0.11 * 435.0, thr=0.0005, filter_thr=0.1):
        self.ctx, self.H, self.W, self.K = ctx, H, W, kmax
        self.mb, self.mbf, self.thr, self.filter_thr = mb, mbf, thr, filter_thr
        K = kmax
        self._spec = [("n", np.int32, (2,)), ("kxy", np.int32, (2, K, 2)), ("score", np.float32, (2, K)), ("desc", np.float32, (2, K, 256)),
                      ("u_right", np.float32, (K,)), ("depth", np.float32, (K,)), ("S", np.int32, (1,)), ("pairs", np.int32, (K, 2)),
                      ("ms", np.float32, (K,))]
        self.bufs = {name: ctx.alloc(int(np.prod(shape)) * np.dtype(dt).itemsize) for name, dt, shape in self._spec}
        self.first = True

    def push(self, img_l_dev, img_r_dev, stride=None, reset=False):
        b = self.bufs
        self.ctx._chk(lib.rfe_stereo_frame_dev(self.ctx.h, _addr(img_l_dev), _addr(img_r_dev), self.H, self.W, stride or self.W, self.K,
                                               self.thr, self.filter_thr, self.mb, self.mbf, int(reset or self.first), b["n"].ptr,
                                               b["kxy"].ptr, b["score"].ptr, b["desc"].ptr, b["u_right"].ptr, b["depth"].ptr,
                                               b["S"].ptr, b["pairs"].ptr, b["ms"].ptr))
        self.first = False

    def results(self):
        self.ctx.synchronize()
        out = {name: self.bufs[name].download(shape, dt) for name, dt, shape in self._spec}
        out["S"] = int(out["S"][0])
        return out

    def close(self):
        for b in self.bufs.values():
            b.free()
        self.bufs = {}


def pool_shard(F, n, member):
    """rfe_pool_shard: (first_frame, frames, pairs) of `member` among n for an F-frame stream (pure function, no device)."""
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.rfe_pool_shard(F, n, member, C.byref(a), C.byref(b), C.byref(c))
    if rc < 0:
        raise RfeError(f"rfe_pool_shard({F}, {n}, {member}) failed ({rc})")
    return a.value, b.value, c.value


class Pool:
    """rfe_pool: one ctx + host thread per member device; a stream of F frames sharded with one overlap frame, results
    gathered into member 0's device (RCCL or copies) and returned as host arrays in global order."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib.rfe_pool_create(devs, len(devices), C.byref(h))
        if rc != 0:
            raise RfeError(f"rfe_pool_create failed ({rc}): {lib.rfe_pool_last_error(None).decode()}")
        self.h = h

    def close(self):
        if self.h:
            lib.rfe_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RfeError(f"librover_fe pool error {rc}: {lib.rfe_pool_last_error(self.h).decode()}")
        return rc

    @property
    def size(self):
        return lib.rfe_pool_size(self.h)

    @property
    def has_rccl(self):
        return bool(lib.rfe_pool_has_rccl(self.h))

    def set_weights(self, kind, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        self._chk(lib.rfe_pool_set_weights(self.h, kind, blob.ctypes.data, blob.size))

    def set_option(self, option, value):
        self._chk(lib.rfe_pool_set_option(self.h, option, int(value)))

    def set_hparams(self, **kw):
        """Change some graph hyper-parameters (keys of rfe_hparams) on EVERY member; the others keep member 0's current values."""
        h = HParams()
        rc = lib.rfe_get_hparams(lib.rfe_pool_ctx(self.h, 0), C.byref(h))
        if rc < 0:
            raise RfeError(f"rfe_get_hparams on pool member 0 failed ({rc})")
        for k, v in kw.items():
            if not hasattr(h, k):
                raise KeyError(k)
            setattr(h, k, v)
        self._chk(lib.rfe_pool_set_hparams(self.h, C.byref(h)))

    def extract_match_stream(self, frames_u8, kmax=1024, thr=0.0005, filter_thr=0.1, transport=POOL_AUTO, with_desc=True):
        """frames_u8: host [F,H,W] uint8.  Returns dict n [F], kxy [F,K,2], score / desc (with_desc), S [F-1], pairs [F-1,K,2], ms [F-1,K]."""
        img = np.ascontiguousarray(frames_u8, np.uint8)
        F, H, W = img.shape
        P = max(F - 1, 1)
        out = {"n": np.zeros((F,), np.int32), "kxy": np.zeros((F, kmax, 2), np.int32), "S": np.zeros((P,), np.int32),
               "pairs": np.zeros((P, kmax, 2), np.int32), "ms": np.zeros((P, kmax), np.float32)}
        if with_desc:
            out["score"] = np.zeros((F, kmax), np.float32)
            out["desc"] = np.zeros((F, kmax, 256), np.float32)
        self._chk(lib.rfe_pool_extract_match_stream(self.h, img.ctypes.data, H, W, W, F, kmax, thr, filter_thr, transport, out["n"].ctypes.data,
                                                    out["kxy"].ctypes.data, out["score"].ctypes.data if with_desc else None,
                                                    out["desc"].ctypes.data if with_desc else None, out["S"].ctypes.data,
                                                    out["pairs"].ctypes.data, out["ms"].ctypes.data))
        return out


class Context:
    """One rfe_ctx (own HIP stream + workspaces).  Single caller, like one ORT session of the reference."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib.rfe_init(device, C.byref(h))
        if rc != 0:
            raise RfeError(f"rfe_init failed ({rc}): {lib.rfe_last_error(None).decode()}")
        self.h = h

    def close(self):
        if self.h:
            lib.rfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RfeError(f"librover_fe error {rc}: {lib.rfe_last_error(self.h).decode()}")
        return rc

    # ---- weights
    def set_weights(self, kind, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        self._chk(lib.rfe_set_weights(self.h, kind, blob.ctypes.data, blob.size))

    def load_weights(self, sp_path=None, lg_path=None):
        self._chk(lib.rfe_load_weights(self.h, sp_path.encode() if sp_path else None, lg_path.encode() if lg_path else None))

    def get_hparams(self):
        h = HParams()
        self._chk(lib.rfe_get_hparams(self.h, C.byref(h)))
        return h.as_dict()

    def set_hparams(self, **kw):
        """Change some of the graph hyper-parameters (keys of rfe_hparams); the others keep their current values."""
        h = HParams()
        self._chk(lib.rfe_get_hparams(self.h, C.byref(h)))
        for k, v in kw.items():
            if not hasattr(h, k):
                raise KeyError(k)
            setattr(h, k, v)
        self._chk(lib.rfe_set_hparams(self.h, C.byref(h)))

    def set_option(self, option, value):
        self._chk(lib.rfe_set_option(self.h, option, int(value)))

    def get_option(self, option):
        v = C.c_int()
        self._chk(lib.rfe_get_option(self.h, option, C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        self._chk(lib.rfe_set_stream(self.h, stream_ptr))

    def workspace_bytes(self):
        return int(lib.rfe_workspace_bytes(self.h))

    def synchronize(self):
        self._chk(lib.rfe_synchronize(self.h))

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    # ---- host-buffer entry points
    def extract(self, img_u8, kmax=1024, thr=0.0005, pad_cols=0, binarized=False):
        """img_u8: [B,H,W] or [H,W] uint8 (host).  Returns n[B], kxy[B,Kmax,2], score[B,Kmax], desc[B,Kmax,256]
        (+ desc_bin u8 [B,Kmax,256] when binarized=True: Frame::binarize_descriptors written by the sampling kernel).
        pad_cols > 0 passes the frames with a row stride of W + pad_cols bytes (like a cv::Mat ROI)."""
        img = np.ascontiguousarray(img_u8, np.uint8)
        if img.ndim == 2:
            img = img[None]
        B, H, W = img.shape
        stride = W + pad_cols
        if pad_cols:
            wide = np.full((B, H, stride), 255, np.uint8)
            wide[:, :, :W] = img
            img = wide
        n = np.zeros((B,), np.int32)
        kxy = np.zeros((B, kmax, 2), np.int32)
        score = np.zeros((B, kmax), np.float32)
        desc = np.zeros((B, kmax, 256), np.float32)
        if binarized:
            dbin = np.zeros((B, kmax, 256), np.uint8)
            self._chk(lib.rfe_extract_u8_bin(self.h, img.ctypes.data, H, W, stride, B, kmax, thr, n.ctypes.data, kxy.ctypes.data,
                                             score.ctypes.data, desc.ctypes.data, dbin.ctypes.data))
            return n, kxy, score, desc, dbin
        self._chk(lib.rfe_extract_u8(self.h, img.ctypes.data, H, W, stride, B, kmax, thr, n.ctypes.data, kxy.ctypes.data,
                                     score.ctypes.data, desc.ctypes.data))
        return n, kxy, score, desc

    def extract_f32(self, img_f32, kmax=1024, thr=0.0005):
        """The float entry: img_f32 [B,H,W] or [H,W] float32, already normalised (what the reference's Extractor_Inference is handed).
        Same outputs as extract()."""
        img = np.ascontiguousarray(img_f32, np.float32)
        if img.ndim == 2:
            img = img[None]
        B, H, W = img.shape
        n = np.zeros((B,), np.int32)
        kxy = np.zeros((B, kmax, 2), np.int32)
        score = np.zeros((B, kmax), np.float32)
        desc = np.zeros((B, kmax, 256), np.float32)
        self._chk(lib.rfe_extract_f32(self.h, img.ctypes.data, H, W, W, B, kmax, thr, n.ctypes.data, kxy.ctypes.data,
                                      score.ctypes.data, desc.ctypes.data))
        return n, kxy, score, desc

    def match(self, k0n, k1n, d0, d1, m, n, filter_thr=0.1):
        """Batched pairs, host arrays: k0n [P,Mmax,2], d0 [P,Mmax,256], m [P] ...  Returns S[P], pairs, ms."""
        k0n = np.ascontiguousarray(k0n, np.float32); k1n = np.ascontiguousarray(k1n, np.float32)
        d0 = np.ascontiguousarray(d0, np.float32); d1 = np.ascontiguousarray(d1, np.float32)
        m = np.ascontiguousarray(m, np.int32); n = np.ascontiguousarray(n, np.int32)
        P, Mmax = k0n.shape[0], k0n.shape[1]
        Nmax = k1n.shape[1]
        cap = min(Mmax, Nmax)
        S = np.zeros((P,), np.int32)
        pairs = np.zeros((P, cap, 2), np.int32)
        ms = np.zeros((P, cap), np.float32)
        self._chk(lib.rfe_match(self.h, k0n.ctypes.data, k1n.ctypes.data, d0.ctypes.data, d1.ctypes.data, m.ctypes.data,
                                n.ctypes.data, P, Mmax, Nmax, filter_thr, S.ctypes.data, pairs.ctypes.data, ms.ctypes.data))
        return S, pairs, ms

    def match_fused(self, kpts0, kpts1, desc0, desc1, rows, cols, filter_thr=0.1, match_thresh=0.0):
        kpts0 = np.ascontiguousarray(kpts0, np.float32).reshape(-1, 2)
        kpts1 = np.ascontiguousarray(kpts1, np.float32).reshape(-1, 2)
        desc0 = np.ascontiguousarray(desc0, np.float32); desc1 = np.ascontiguousarray(desc1, np.float32)
        M, N = kpts0.shape[0], kpts1.shape[0]
        vn = np.full((max(M, 1),), -1, np.int32)
        size = self._chk(lib.rfe_match_fused(self.h, kpts0.ctypes.data, M, kpts1.ctypes.data, N, desc0.ctypes.data,
                                             desc1.ctypes.data, rows, cols, filter_thr, match_thresh, vn.ctypes.data))
        return size, vn[:M]

    def stereo_match(self, img_l, img_r, k_l, k_r, d_l, d_r, mb, mbf):
        """Frame::ComputeStereoMatches on host arrays; returns (uRight[N], depth[N])."""
        il = np.ascontiguousarray(img_l, np.uint8); ir = np.ascontiguousarray(img_r, np.uint8)
        kl = np.ascontiguousarray(k_l, np.float32).reshape(-1, 2); kr = np.ascontiguousarray(k_r, np.float32).reshape(-1, 2)
        dl = np.ascontiguousarray(d_l, np.float32); dr = np.ascontiguousarray(d_r, np.float32)
        H, W = il.shape
        N, Nr = kl.shape[0], kr.shape[0]
        u = np.full((max(N, 1),), -1, np.float32); z = np.full((max(N, 1),), -1, np.float32)
        self._chk(lib.rfe_stereo_match(self.h, il.ctypes.data, ir.ctypes.data, H, W, W, kl.ctypes.data, N, kr.ctypes.data, Nr,
                                       dl.ctypes.data, dr.ctypes.data, mb, mbf, u.ctypes.data, z.ctypes.data))
        return u[:N], z[:N]

    def search_candidates(self, q, f, offsets, cand, skip=None):
        """Best / second-best scan of SearchByProjection1 (SPmatcher.cc:1218-1248) over CSR candidate lists."""
        qa = np.ascontiguousarray(q, np.float32).reshape(-1, 256); fa = np.ascontiguousarray(f, np.float32).reshape(-1, 256)
        off = np.ascontiguousarray(offsets, np.int32); cd = np.ascontiguousarray(cand, np.int32)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        Nq = qa.shape[0]
        bi = np.empty((max(Nq, 1),), np.int32); bd = np.empty((max(Nq, 1),), np.float32); sd = np.empty((max(Nq, 1),), np.float32)
        self._chk(lib.rfe_search_candidates(self.h, qa.ctypes.data, Nq, fa.ctypes.data, fa.shape[0], off.ctypes.data,
                                            cd.ctypes.data, None if sk is None else sk.ctypes.data, bi.ctypes.data,
                                            bd.ctypes.data, sd.ctypes.data))
        return bi[:Nq], bd[:Nq], sd[:Nq]

    def distinctive_descriptors(self, desc, offsets):
        """MapPoint::ComputeDistinctiveDescriptors for many map points (MapPoint.cc:438-530)."""
        da = np.ascontiguousarray(desc, np.float32).reshape(-1, 256)
        off = np.ascontiguousarray(offsets, np.int32)
        Np = off.shape[0] - 1
        b = np.empty((max(Np, 1),), np.int32); m = np.empty((max(Np, 1),), np.float32)
        self._chk(lib.rfe_distinctive_descriptors(self.h, da.ctypes.data, off.ctypes.data, Np, b.ctypes.data, m.ctypes.data))
        return b[:Np], m[:Np]

    # ---- profiling
    def profile(self, on=True):
        self._chk(lib.rfe_profile_enable(self.h, int(on)))

    def profile_filter(self, stage=None):
        """Record events for one stage only (None = all stages)."""
        self._chk(lib.rfe_profile_filter(self.h, stage.encode() if stage else None))

    def profile_reset(self):
        self._chk(lib.rfe_profile_reset(self.h))

    def profile_read(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)()
        calls = (C.c_int64 * 64)()
        k = self._chk(lib.rfe_profile_read(self.h, names, 4096, ms, calls, 64))
        nm = names.value.decode().split(";") if k else []
        return {nm[i]: (ms[i], calls[i]) for i in range(k)}
