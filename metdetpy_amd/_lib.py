"""ctypes binding of libmetdet_hip.so (the C-ABI in include/metdet_hip.h).

There is no CPU fallback: if the library is missing or no AMD GPU is visible, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDH_LIB") or os.path.join(_HERE, "libmetdet_hip.so")  # MDH_LIB: A/B builds of the same library

MDH_MAX_LINES = 512
MDH_MAX_WINDOW = 255
MDH_NUM_KERNELS = 8
MDH_ABI_VERSION = 4
MDH_HOUGH_EXACT, MDH_HOUGH_ACCUMULATOR = 0, 1
MDH_DET_M3, MDH_DET_CLASSIC = 0, 1
MDH_ROLE_FULL, MDH_ROLE_BAND, MDH_ROLE_HOUGH = 0, 1, 2
MDH_BAND_HALO = 4
MDH_STREAM_NOISE, MDH_STREAM_POST_LAST, MDH_STREAM_POST_NEXT = 0, 1, 2
MDH_MAX_LOCAL_BANDS = 16
SENSITIVITY_ID = {"low": 0, "normal": 1, "high": 2}


class MdhCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("detector", C.c_int32), ("adaptive_bi_thre", C.c_int32), ("init_value", C.c_int32),
                ("sensitivity", C.c_int32), ("interval", C.c_int32), ("roi", C.c_int32 * 4),
                ("hough_threshold", C.c_int32), ("hough_min_len", C.c_int32), ("hough_max_gap", C.c_int32),
                ("dy_mask", C.c_int32), ("max_batch", C.c_int32), ("role", C.c_int32), ("mask_area", C.c_double),
                ("hough_mode", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self, *a, **kw):  # the ABI guard (include/metdet_hip.h): the library refuses any other size
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(MdhCfg)


class MdhFrameInfo(C.Structure):
    _fields_ = [("n_raw", C.c_int32), ("n_nms", C.c_int32), ("lit_count", C.c_int32), ("bi_threshold", C.c_int32),
                ("bi_threshold_float", C.c_double), ("snr", C.c_double), ("sigma", C.c_double), ("dst_sum", C.c_double),
                ("hough_gap", C.c_double)]


class MdhCollectorCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("min_len", C.c_double), ("max_interval", C.c_double),
                ("time_range", C.c_double * 2),
                ("speed_range", C.c_double * 2), ("drct_range", C.c_double * 2), ("det_thre", C.c_double), ("thre2", C.c_double),
                ("fps", C.c_double), ("exp_frame", C.c_int32), ("runtime_w", C.c_int32), ("runtime_h", C.c_int32),
                ("num_cls", C.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(MdhCollectorCfg)


TARGET_DTYPE = np.dtype([("flag", "<i4"), ("group", "<i4"), ("start_frame", "<i8"), ("end_frame", "<i8"),
                         ("last_activate_frame", "<i8"), ("duration", "<i8"), ("speed", "<f8"), ("dist", "<f8"),
                         ("fix_dist", "<f8"), ("fix_speed", "<f8"), ("fix_motion_duration", "<f8"), ("fix_duration", "<f8"),
                         ("drct_loss", "<f8"), ("score", "<f8"), ("num_pts", "<i4"), ("category", "<i4"), ("pt1", "<i4", (2, )),
                         ("pt2", "<i4", (2, )), ("center_off", "<i4"), ("n_center", "<i4"), ("score_is_int", "<i4"),
                         ("reserved", "<i4")])
assert TARGET_DTYPE.itemsize == 144

FRAME_INFO_DTYPE = np.dtype([("n_raw", "<i4"), ("n_nms", "<i4"), ("lit_count", "<i4"), ("bi_threshold", "<i4"),
                             ("bi_threshold_float", "<f8"), ("snr", "<f8"), ("sigma", "<f8"), ("dst_sum", "<f8"),
                             ("hough_gap", "<f8")])
assert FRAME_INFO_DTYPE.itemsize == C.sizeof(MdhFrameInfo)

EXPORTS = [
    "mdh_create", "mdh_destroy", "mdh_last_error", "mdh_update", "mdh_detect", "mdh_process_batch", "mdh_get_binary",
    "mdh_lineset_nms", "mdh_device_alloc", "mdh_device_free", "mdh_device_upload", "mdh_profile_enable",
    "mdh_profile_read", "mdh_kernel_name", "mdh_frames_seen", "mdh_pre_create", "mdh_pre_run", "mdh_pre_destroy",
    "mdh_max_stack", "mdh_submit_batch", "mdh_collect_batch", "mdh_in_flight", "mdh_max_in_flight", "mdh_debug_hough",
    "mdh_band_noise", "mdh_band_pixels", "mdh_hough_planes", "mdh_band_skip", "mdh_seek", "mdh_frame_sums",
    "mdh_inject_sums", "mdh_mute_hough_until", "mdh_hough_planes_submit",
    "mdh_collector_create", "mdh_collector_destroy", "mdh_collector_update", "mdh_collector_update_batch", "mdh_collector_events",
    "mdh_collector_active", "mdh_pre_run_batch", "mdh_host_alloc", "mdh_host_free", "mdh_device_copy", "mdh_abi_version", "mdh_hw_queues_hint", "mdh_dense_passes", "mdh_last_density", "mdh_dense_now",
    "mdh_stream_handle", "mdh_band_submit", "mdh_band_sums_device", "mdh_band_pixels_submit", "mdh_band_collect", "mdh_chain",
    "mdh_device_copy_on", "mdh_bands_allreduce_local", "mdh_hough_input_planes",
    "mdh_host_register", "mdh_host_unregister", "mdh_set_auto_register", "mdh_update_stats", "mdh_set_device",
    "mdh_set_wait_timeout_ms", "mdh_pre_traffic_model", "mdh_device_reset", "mdh_share_streams", "mdh_band_thresholds_submit", "mdh_band_pixels_submit_to",
    "mdh_device_count", "mdh_stream_create", "mdh_stream_destroy", "mdh_stream_sync", "mdh_dev_malloc", "mdh_dev_free", "mdh_dev_copy",
]

_lib = None
_hw_queues = 12


def prefer_hw_queues(n: int) -> bool:
    """How many hardware queues the HIP runtime should map the library's streams onto (GPU_MAX_HW_QUEUES; HIP's default is 4).
    The runtime reads it ONCE, when it initialises, so this only has an effect before the library is loaded; a value the host
    exported itself always wins.  The binding asks for 12 (one detector: ~10 active streams, full overlap); a process that runs
    MANY contexts on one device (LocalBands: 8 bands + a Hough context) asks for 4 -- the command processor time-slices
    oversubscribed queues, measured 54 k frames/s with 3-5 queues against 5 k with 32.  Returns whether the
    request can still take effect.  Results never depend on it."""
    global _hw_queues
    if _lib is not None:
        return False
    _hw_queues = int(n)
    return "GPU_MAX_HW_QUEUES" not in os.environ


class MetDetHipError(RuntimeError):
    pass


class _HostOnlyLibrary:
    """MDH_LIB pointing at the host-only sanitizer build (`make -C metdetpy_amd/csrc san`): lineset_nms and the native collector are
    there, every entry point that needs a GPU is not -- and says so when called (it never computes anything instead)."""

    class _Absent:
        def __init__(self, name):
            self.name = name

        def __call__(self, *a, **k):
            raise MetDetHipError(f"{self.name}: {LIB_PATH} is the host-only (sanitizer) build; it has no GPU entry points")

    def __init__(self, L):
        self.__dict__["_L"] = L

    def __getattr__(self, name):
        try:
            return getattr(self.__dict__["_L"], name)
        except AttributeError:
            if not name.startswith("mdh_"):
                raise
            stub = self._Absent(name)
            self.__dict__[name] = stub
            return stub


def load():
    """Load libmetdet_hip.so and declare the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MetDetHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C metdetpy_amd/csrc` (there is no CPU fallback)")
    # host side of the boundary: the library's streams want their own hardware queues (HIP's default is 4; read once when the
    # runtime initialises, i.e. at the process's first HIP call).  The C library itself never touches the environment; this
    # binding does, once, here, and only when the host has not decided: an unchanged MetDetPy.py gets the tuned value.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(_hw_queues))
    # a job of several ranks (WORLD_SIZE > 1): RCCL's buffers cross process boundaries through dmabuf IPC -- the only kind the
    # MI355X hosts' driver supports; with the legacy mode the communicator fails with `hipIpcGetMemHandle: invalid argument`.
    # Read when the runtime initialises as well, so it is set here, not next to the communicator (rccl.py), and only if unset.
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = C.CDLL(LIB_PATH)
    if not hasattr(L, "mdh_create") and hasattr(L, "mdh_collector_create"):
        L = _HostOnlyLibrary(L)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
    L.mdh_create.argtypes = [i32, i32, i32, i32, C.POINTER(MdhCfg)]
    L.mdh_create.restype = vp
    L.mdh_destroy.argtypes = [vp]
    L.mdh_destroy.restype = None
    L.mdh_last_error.restype = C.c_char_p
    L.mdh_update.argtypes = [vp, vp]
    L.mdh_detect.argtypes = [vp, vp, vp, vp, vp]
    L.mdh_process_batch.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.mdh_submit_batch.argtypes = [vp, vp, i32, i32]
    L.mdh_collect_batch.argtypes = [vp, vp, vp, vp, vp]
    L.mdh_in_flight.argtypes = [vp]
    L.mdh_max_in_flight.argtypes = [vp]
    L.mdh_get_binary.argtypes = [vp, i32, i32, vp]
    L.mdh_lineset_nms.argtypes = [vp, i32, vp, vp]
    L.mdh_device_alloc.argtypes = [vp, u64]
    L.mdh_device_alloc.restype = vp
    L.mdh_device_free.argtypes = [vp, vp]
    L.mdh_host_alloc.argtypes = [vp, u64]
    L.mdh_host_alloc.restype = vp
    L.mdh_host_free.argtypes = [vp, vp]
    L.mdh_share_streams.argtypes = [vp, vp]
    L.mdh_set_device.argtypes = [i32]
    L.mdh_device_count.argtypes = []
    L.mdh_stream_create.argtypes = [i32]
    L.mdh_stream_create.restype = vp
    L.mdh_stream_destroy.argtypes = [i32, vp]
    L.mdh_stream_sync.argtypes = [i32, vp, dbl]
    L.mdh_dev_malloc.argtypes = [i32, u64]
    L.mdh_dev_malloc.restype = vp
    L.mdh_dev_free.argtypes = [i32, vp]
    L.mdh_dev_copy.argtypes = [i32, vp, vp, u64, i32, vp]
    L.mdh_device_reset.argtypes = [i32]
    L.mdh_set_wait_timeout_ms.argtypes = [vp, dbl]
    L.mdh_host_register.argtypes = [vp, vp, u64]
    L.mdh_host_unregister.argtypes = [vp, vp]
    L.mdh_set_auto_register.argtypes = [vp, i32]
    L.mdh_update_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    L.mdh_device_upload.argtypes = [vp, vp, vp, u64]
    L.mdh_device_copy.argtypes = [vp, vp, vp, u64]
    L.mdh_profile_enable.argtypes = [vp, i32]
    L.mdh_profile_read.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), i32]
    L.mdh_kernel_name.argtypes = [i32]
    L.mdh_kernel_name.restype = C.c_char_p
    L.mdh_frames_seen.argtypes = [vp]
    L.mdh_frames_seen.restype = i64
    L.mdh_pre_create.argtypes = [i32, i32, i32, i32, i32, vp]
    L.mdh_pre_create.restype = vp
    L.mdh_pre_run.argtypes = [vp, vp, i32, i32, vp, i32]
    L.mdh_pre_run_batch.argtypes = [vp, vp, i32, i32, i32, vp, i32]
    L.mdh_pre_traffic_model.argtypes = [vp, i32]
    L.mdh_pre_traffic_model.restype = u64
    L.mdh_pre_destroy.argtypes = [vp]
    L.mdh_pre_destroy.restype = None
    L.mdh_max_stack.argtypes = [i32, vp, i32, u64, i32, vp, i32]
    L.mdh_band_noise.argtypes = [vp, vp, i32, i32, vp, C.POINTER(i32)]
    L.mdh_band_pixels.argtypes = [vp, vp, dbl, i32, i32, i32, i32, vp, i32, vp]
    L.mdh_hough_planes.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.mdh_hough_planes_submit.argtypes = [vp, vp, i32, i32, i32, i32]
    L.mdh_band_skip.argtypes = [vp]
    L.mdh_collector_create.argtypes = [C.POINTER(MdhCollectorCfg)]
    L.mdh_collector_create.restype = vp
    L.mdh_collector_destroy.argtypes = [vp]
    L.mdh_collector_destroy.restype = None
    L.mdh_collector_update.argtypes = [vp, i64, vp, vp, i32]
    L.mdh_collector_update_batch.argtypes = [vp, i32, i64, i64, i32, vp, vp, vp]
    L.mdh_collector_events.argtypes = [vp, vp, i32, vp, i32, C.POINTER(i32)]
    L.mdh_collector_active.argtypes = [vp]
    L.mdh_frame_sums.argtypes = [i32, vp, i32, u64, i32, vp]
    L.mdh_seek.argtypes = [vp, i64, i64, dbl, dbl, dbl, i32]
    L.mdh_inject_sums.argtypes = [vp, i64, i64, i64]
    L.mdh_mute_hough_until.argtypes = [vp, i64]
    L.mdh_dense_passes.argtypes = [vp]
    L.mdh_dense_passes.restype = i64
    L.mdh_dense_now.argtypes = [vp]
    L.mdh_stream_handle.argtypes = [vp, i32]
    L.mdh_stream_handle.restype = vp
    L.mdh_band_submit.argtypes = [vp, vp, i32, i32, C.POINTER(i32)]
    L.mdh_band_sums_device.argtypes = [vp]
    L.mdh_band_sums_device.restype = vp
    L.mdh_band_pixels_submit.argtypes = [vp, vp, dbl, i32, i32, i32, i32, vp, i32]
    L.mdh_band_thresholds_submit.argtypes = [vp, vp, dbl]
    L.mdh_band_pixels_submit_to.argtypes = [vp, vp, dbl, i32, i32, vp, i32]
    L.mdh_hough_input_planes.argtypes = [vp]
    L.mdh_hough_input_planes.restype = vp
    L.mdh_band_collect.argtypes = [vp, vp]
    L.mdh_chain.argtypes = [vp, i32, vp, i32]
    L.mdh_device_copy_on.argtypes = [vp, i32, vp, vp, u64]
    L.mdh_bands_allreduce_local.argtypes = [C.POINTER(vp), i32]
    L.mdh_last_density.argtypes = [vp]
    L.mdh_last_density.restype = dbl
    if not hasattr(L, "mdh_abi_version") or L.mdh_abi_version() != MDH_ABI_VERSION:
        raise MetDetHipError(f"{LIB_PATH} implements another ABI version than this binding ({MDH_ABI_VERSION}): rebuild it "
                             "(`make -C metdetpy_amd/csrc -B`)")
    _lib = L
    return L


def bands_allreduce_local(contexts):
    """Sum of the staged passes' noise sums over band contexts of ONE process and device, on the device (no host wait)."""
    L = load()
    arr = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
    check(L.mdh_bands_allreduce_local(arr, len(contexts)), "mdh_bands_allreduce_local")


def last_error() -> str:
    return load().mdh_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise MetDetHipError(f"{what} failed (code {rc}): {last_error()}")


def device_reset(device: int = 0):
    """hipDeviceReset: releases what this process holds on the device (all contexts must be closed)."""
    check(load().mdh_device_reset(int(device)), "mdh_device_reset")


def set_device(device: int):
    """hipSetDevice for this thread (the RCCL communicator of rccl.py binds to the current device)."""
    check(load().mdh_set_device(int(device)), "mdh_set_device")


def lineset_nms(lines: np.ndarray):
    """utils.py:780-839 on the host side of the library (no GPU needed)."""
    lines = np.ascontiguousarray(lines, dtype=np.int32).reshape(-1, 4)
    n = len(lines)
    out = np.zeros((n, 4), np.int32)
    prob = np.zeros(n, np.float64)
    k = load().mdh_lineset_nms(lines.ctypes.data, n, out.ctypes.data, prob.ctypes.data)
    return out[:k], prob[:k]


class Context:
    """Owns one mdh_ctx (one detector on one GPU)."""

    def __init__(self, H, W, n, cfg: MdhCfg, device: int = 0):
        self.L = load()
        self.H, self.W, self.n = int(H), int(W), int(n)
        self.handle = self.L.mdh_create(int(device), self.H, self.W, self.n, C.byref(cfg))
        if not self.handle:
            raise MetDetHipError(f"mdh_create failed: {last_error()}")
        self.max_batch = cfg.max_batch if cfg.max_batch > 0 else 64
        self._info1 = np.zeros(1, FRAME_INFO_DTYPE)
        self._raw1 = np.zeros((MDH_MAX_LINES, 4), np.int32)
        self._nms1 = np.zeros((MDH_MAX_LINES, 4), np.int32)
        self._prob1 = np.zeros(MDH_MAX_LINES, np.float64)
        # the per-frame protocol is latency-bound: addresses are taken once, not per call
        self._p1 = (self._info1.ctypes.data, self._raw1.ctypes.data, self._nms1.ctypes.data, self._prob1.ctypes.data)
        self._empty4 = np.zeros((0, 4), np.int32)
        self._empty1 = np.zeros(0, np.float64)

    def close(self):
        if getattr(self, "handle", None):
            self.L.mdh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, frame: np.ndarray):
        if frame.shape != (self.H, self.W) or frame.dtype != np.uint8:
            raise ValueError(f"frame must be uint8 {(self.H, self.W)}, got {frame.dtype} {frame.shape}")
        f = frame if frame.flags.c_contiguous else np.ascontiguousarray(frame)
        rc = self.L.mdh_update(self.handle, f.__array_interface__["data"][0])
        if rc:
            check(rc, "mdh_update")

    def detect(self):
        rc = self.L.mdh_detect(self.handle, *self._p1)
        if rc:
            check(rc, "mdh_detect")
        info = self._info1[0].copy()
        nr = min(int(info["n_raw"]), MDH_MAX_LINES)
        if nr == 0:
            return info, self._empty4, self._empty4, self._empty1
        nn = int(info["n_nms"])
        return info, self._raw1[:nr].copy(), self._nms1[:nn].copy(), self._prob1[:nn].copy()

    def detect_fields(self):
        """detect() for the per-frame loop, which is latency-bound: the frame's scalars as a plain tuple in FRAME_INFO_DTYPE's
        field order (one tolist() instead of a record copy and a field lookup per attribute: ~5 us a frame)."""
        rc = self.L.mdh_detect(self.handle, *self._p1)
        if rc:
            check(rc, "mdh_detect")
        fields = self._info1.tolist()[0]
        nr = min(fields[0], MDH_MAX_LINES)
        if nr == 0:
            return fields, self._empty4, self._empty4, self._empty1
        nn = fields[1]
        return fields, self._raw1[:nr].copy(), self._nms1[:nn].copy(), self._prob1[:nn].copy()

    def process_batch(self, frames, K=None, on_device=False):
        """frames: (K,H,W) uint8 ndarray (host) or an integer device pointer with K given."""
        if on_device:
            ptr = int(frames)
            assert K is not None
        else:
            if frames.ndim != 3 or frames.shape[1:] != (self.H, self.W) or frames.dtype != np.uint8:
                raise ValueError("frames must be uint8 (K,H,W)")
            frames = np.ascontiguousarray(frames)
            K = len(frames)
            ptr = frames.ctypes.data
        info = np.zeros(K, FRAME_INFO_DTYPE)
        raw = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        nms = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        prob = np.zeros((K, MDH_MAX_LINES), np.float64)
        check(self.L.mdh_process_batch(self.handle, ptr, int(K), int(bool(on_device)), info.ctypes.data, raw.ctypes.data,
                                       nms.ctypes.data, prob.ctypes.data), "mdh_process_batch")
        return info, raw, nms, prob

    def process_batch_into(self, ptr, K, on_device, info, raw, nms, prob):
        check(self.L.mdh_process_batch(self.handle, int(ptr), int(K), int(bool(on_device)), info.ctypes.data,
                                       raw.ctypes.data if raw is not None else None,
                                       nms.ctypes.data if nms is not None else None,
                                       prob.ctypes.data if prob is not None else None), "mdh_process_batch")

    def submit(self, ptr, K, on_device):
        """Queue a pass of K <= max_batch frames and return at once (at most self.max_in_flight passes may be queued).  The
        frames must stay valid until the pass has been collected."""
        check(self.L.mdh_submit_batch(self.handle, int(ptr), int(K), int(bool(on_device))), "mdh_submit_batch")

    def collect_into(self, info, raw, nms, prob) -> int:
        """Wait for the oldest pass in flight and fetch its results; returns its frame count."""
        rc = self.L.mdh_collect_batch(self.handle, info.ctypes.data, raw.ctypes.data if raw is not None else None,
                                      nms.ctypes.data if nms is not None else None,
                                      prob.ctypes.data if prob is not None else None)
        if rc < 0:
            check(rc, "mdh_collect_batch")
        return rc

    @property
    def in_flight(self) -> int:
        return self.L.mdh_in_flight(self.handle)

    @property
    def max_in_flight(self) -> int:
        return self.L.mdh_max_in_flight(self.handle)

    def get_binary(self, index=-1, stage=0):
        out = np.empty((self.H, self.W), np.uint8)
        check(self.L.mdh_get_binary(self.handle, int(index), int(stage), out.ctypes.data), "mdh_get_binary")
        return out

    def device_alloc(self, nbytes: int) -> int:
        p = self.L.mdh_device_alloc(self.handle, int(nbytes))
        if not p:
            raise MetDetHipError(f"mdh_device_alloc({nbytes}) failed: {last_error()}")
        return p

    def host_alloc(self, nbytes: int) -> int:
        """Pinned host memory (for frames submitted with on_device=False); returns the address."""
        p = self.L.mdh_host_alloc(self.handle, int(nbytes))
        if not p:
            raise MetDetHipError(f"mdh_host_alloc({nbytes}) failed: {last_error()}")
        return p

    def host_free(self, ptr: int):
        check(self.L.mdh_host_free(self.handle, ptr), "mdh_host_free")

    def host_register(self, arr) -> None:
        """Register a loader's frame buffers (a NumPy array or (address, nbytes)) so that update() reads frames lying in them in place."""
        ptr, nbytes = arr if isinstance(arr, tuple) else (arr.__array_interface__["data"][0], arr.nbytes)
        check(self.L.mdh_host_register(self.handle, ptr, int(nbytes)), "mdh_host_register")

    def host_unregister(self, arr) -> None:
        ptr = arr[0] if isinstance(arr, tuple) else arr.__array_interface__["data"][0]
        check(self.L.mdh_host_unregister(self.handle, ptr), "mdh_host_unregister")

    def set_wait_timeout(self, seconds: float) -> None:
        """Collects give up (MetDetHipError, the context is poisoned) when their pass has not completed after `seconds` (0: never)."""
        check(self.L.mdh_set_wait_timeout_ms(self.handle, float(seconds) * 1e3), "mdh_set_wait_timeout_ms")

    def set_auto_register(self, on: bool) -> None:
        check(self.L.mdh_set_auto_register(self.handle, 1 if on else 0), "mdh_set_auto_register")

    def update_stats(self) -> dict:
        d, c, a = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(self.L.mdh_update_stats(self.handle, C.byref(d), C.byref(c), C.byref(a)), "mdh_update_stats")
        return {"direct_frames": d.value, "copied_frames": c.value, "auto_registered": a.value}

    def device_free(self, ptr: int):
        check(self.L.mdh_device_free(self.handle, ptr), "mdh_device_free")

    def device_upload(self, dptr: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        check(self.L.mdh_device_upload(self.handle, dptr, a.ctypes.data, a.nbytes), "mdh_device_upload")

    def device_copy(self, dst: int, src: int, nbytes: int):
        check(self.L.mdh_device_copy(self.handle, int(dst), int(src), int(nbytes)), "mdh_device_copy")

    def profile_enable(self, on=True):
        check(self.L.mdh_profile_enable(self.handle, int(on)), "mdh_profile_enable")

    def profile_read(self, reset=True):
        ms = (C.c_double * MDH_NUM_KERNELS)()
        nl = (C.c_int64 * MDH_NUM_KERNELS)()
        check(self.L.mdh_profile_read(self.handle, ms, nl, int(reset)), "mdh_profile_read")
        return {self.L.mdh_kernel_name(k).decode(): (ms[k], nl[k]) for k in range(MDH_NUM_KERNELS)}

    # ---- row-band tiling (include/metdet_hip.h, "Row-band tiling of ONE stream across GPUs") ----
    def band_noise(self, frames, K=None, on_device=False):
        """Step 1: stage K sub-image frames, return this band's share of the noise sums (2*n_jobs int64)."""
        if on_device:
            ptr = int(frames)
            assert K is not None
        else:
            if frames.ndim != 3 or frames.shape[1:] != (self.H, self.W) or frames.dtype != np.uint8:
                raise ValueError(f"frames must be uint8 (K,{self.H},{self.W})")
            frames = np.ascontiguousarray(frames)
            K, ptr = len(frames), frames.ctypes.data
        sums = np.zeros(2 * int(K), np.int64)
        nj = C.c_int(0)
        check(self.L.mdh_band_noise(self.handle, ptr, int(K), int(bool(on_device)), sums.ctypes.data, C.byref(nj)),
              "mdh_band_noise")
        self._band_K = int(K)
        return sums[:2 * nj.value]

    def band_pixels(self, sums_total, roi_px_total, own_row0, own_rows, rows_stride, nsplit, out=None):
        """Step 2: thresholds from the global sums, pixel stages, owned dst rows in all-to-all send order.
        out: None (a host array is returned) or an integer device pointer of nsplit*ceil(K/nsplit)*rows_stride*WW u64."""
        K = getattr(self, "_band_K", 0) or 1  # without a staged pass the library reports the sequence error
        WW = (self.W + 63) // 64
        Kp = -(-K // nsplit)
        info = np.zeros(K, FRAME_INFO_DTYPE)
        tot = np.ascontiguousarray(sums_total, dtype=np.int64)
        tot = tot if len(tot) else np.zeros(2, np.int64)
        host = None
        if out is None:
            host = np.zeros((nsplit, Kp, rows_stride, WW), np.uint64)
        check(self.L.mdh_band_pixels(self.handle, tot.ctypes.data, float(roi_px_total), int(own_row0), int(own_rows),
                                     int(rows_stride), int(nsplit), host.ctypes.data if host is not None else int(out),
                                     int(host is None), info.ctypes.data), "mdh_band_pixels")
        return host, info

    # ---- the same steps without host round trips (include/metdet_hip.h): everything is queued, the sums stay on the device ----
    def band_submit(self, frames, K=None, on_device=False) -> int:
        """Stage K sub-image frames and queue this band's share of the noise sums; returns the number of due frames."""
        if on_device:
            ptr = int(frames)
            assert K is not None
        else:
            if frames.ndim != 3 or frames.shape[1:] != (self.H, self.W) or frames.dtype != np.uint8:
                raise ValueError(f"frames must be uint8 (K,{self.H},{self.W})")
            frames = np.ascontiguousarray(frames)
            K, ptr = len(frames), frames.ctypes.data
        nj = C.c_int(0)
        check(self.L.mdh_band_submit(self.handle, ptr, int(K), int(bool(on_device)), C.byref(nj)), "mdh_band_submit")
        self._band_K = int(K)
        return nj.value

    @property
    def band_sums_device(self) -> int:
        return int(self.L.mdh_band_sums_device(self.handle) or 0)

    def stream_handle(self, which: int) -> int:
        """hipStream_t (as an integer) of one of the context's streams: MDH_STREAM_*; for collectives the host enqueues (RCCL)."""
        return int(self.L.mdh_stream_handle(self.handle, int(which)) or 0)

    def band_thresholds_submit(self, roi_px_total, sums_total_device: int = 0):
        """The staged pass's thresholds on their own, ahead of band_pixels_submit (several bands of one process: all bands' chains
        before the first band's window kernel)."""
        check(self.L.mdh_band_thresholds_submit(self.handle, int(sums_total_device) or None, float(roi_px_total)), "mdh_band_thresholds_submit")

    def band_pixels_submit(self, roi_px_total, own_row0, own_rows, rows_stride, nsplit, out: int, sums_total_device: int = 0,
                           nofill: bool = False):
        check(self.L.mdh_band_pixels_submit(self.handle, int(sums_total_device) or None, float(roi_px_total), int(own_row0), int(own_rows),
                                            int(rows_stride), int(nsplit), int(out), 1 if nofill else 0), "mdh_band_pixels_submit")

    def band_pixels_submit_to(self, roi_px_total, own_row0, own_rows, hough: "Context", full_row0, sums_total_device: int = 0):
        """Owned rows AND their lit counts / bounding box straight into the pass `hough` (a MDH_ROLE_HOUGH context of this process
        and device) queues next: nothing left for that context to assemble."""
        check(self.L.mdh_band_pixels_submit_to(self.handle, int(sums_total_device) or None, float(roi_px_total), int(own_row0), int(own_rows),
                                               hough.handle, int(full_row0)), "mdh_band_pixels_submit_to")

    @property
    def hough_input_planes(self) -> int:
        """Device pointer of the dst planes the next pass queued on this (Hough) context reads (bands export straight into it)."""
        return int(self.L.mdh_hough_input_planes(self.handle) or 0)

    def band_collect(self):
        """Thresholds (bi_threshold, bi_threshold_float, snr, sigma) of the oldest queued band pass."""
        info = np.zeros(self.max_batch, FRAME_INFO_DTYPE)
        K = self.L.mdh_band_collect(self.handle, info.ctypes.data)
        if K < 0:
            check(K, "mdh_band_collect")
        return info[:K]

    def share_streams_of(self, donor: "Context"):
        """Queue this context's stages on `donor`'s streams (several contexts of one process on one device); before the first pass."""
        check(self.L.mdh_share_streams(self.handle, donor.handle), "mdh_share_streams")
        self._stream_donor = donor  # (keeps the donor alive as long as this context)

    def chain_to(self, my_stream: int, consumer: "Context", consumer_stream: int):
        """Everything queued on my stream so far precedes what is queued on the consumer's stream from now on."""
        check(self.L.mdh_chain(self.handle, int(my_stream), consumer.handle, int(consumer_stream)), "mdh_chain")

    def device_copy_on(self, which: int, dst: int, src: int, nbytes: int):
        check(self.L.mdh_device_copy_on(self.handle, int(which), int(dst), int(src), int(nbytes)), "mdh_device_copy_on")

    def band_skip(self):
        """Drop the pass staged by band_noise (its frames stay in the window)."""
        check(self.L.mdh_band_skip(self.handle), "mdh_band_skip")

    def seek(self, t0, ema_updates=0, snr=0.0, sigma=0.0, thr_float=0.0, thr=0):
        """Reposition the context: the next frame is number t0 + 1; noise EMA / threshold state as given."""
        check(self.L.mdh_seek(self.handle, int(t0), int(ema_updates), float(snr), float(sigma), float(thr_float), int(thr)),
              "mdh_seek")

    def hough_planes_submit(self, pieces, K, nbands, band_rows, on_device=False):
        """hough_planes without the wait: queue the pass; `hough_planes_collect(K)` (or collect_into) fetches the oldest one."""
        if on_device:
            ptr = int(pieces)
        else:
            pieces = np.ascontiguousarray(pieces, dtype=np.uint64)
            assert pieces.size == nbands * K * band_rows * ((self.W + 63) // 64), pieces.shape
            ptr = pieces.ctypes.data
        check(self.L.mdh_hough_planes_submit(self.handle, ptr, int(K), int(bool(on_device)), int(nbands), int(band_rows)),
              "mdh_hough_planes_submit")

    def hough_planes_collect(self, K):
        info = np.zeros(K, FRAME_INFO_DTYPE)
        raw = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        nms = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        prob = np.zeros((K, MDH_MAX_LINES), np.float64)
        self.collect_into(info, raw, nms, prob)
        return info, raw, nms, prob

    def inject_sums(self, t, s1, s2):
        """Noise sums (S1, S2) of due frame number t (1-based, not yet submitted), used instead of the context's own."""
        check(self.L.mdh_inject_sums(self.handle, int(t), int(s1), int(s2)), "mdh_inject_sums")

    def mute_hough_until(self, t):
        """Frames numbered <= t skip the Hough stage (time-window warm-up)."""
        check(self.L.mdh_mute_hough_until(self.handle, int(t)), "mdh_mute_hough_until")

    def hough_planes(self, pieces, K, nbands, band_rows, on_device=False):
        """Step 3 (MDH_ROLE_HOUGH context): pieces = (nbands, K, band_rows, WW) u64 host array or a device pointer."""
        if on_device:
            ptr = int(pieces)
        else:
            pieces = np.ascontiguousarray(pieces, dtype=np.uint64)
            assert pieces.size == nbands * K * band_rows * ((self.W + 63) // 64), pieces.shape
            ptr = pieces.ctypes.data
        info = np.zeros(K, FRAME_INFO_DTYPE)
        raw = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        nms = np.zeros((K, MDH_MAX_LINES, 4), np.int32)
        prob = np.zeros((K, MDH_MAX_LINES), np.float64)
        check(self.L.mdh_hough_planes(self.handle, ptr, int(K), int(bool(on_device)), int(nbands), int(band_rows),
                                      info.ctypes.data, raw.ctypes.data, nms.ctypes.data, prob.ctypes.data), "mdh_hough_planes")
        return info, raw, nms, prob

    @property
    def frames_seen(self) -> int:
        return self.L.mdh_frames_seen(self.handle)

    @property
    def dense_passes(self) -> int:
        """Batched passes queued in the dense output format of the window kernel so far (diagnostics; results do not depend on it)."""
        return self.L.mdh_dense_passes(self.handle)

    @property
    def dense_now(self) -> bool:
        return self.L.mdh_dense_now(self.handle) == 1

    @property
    def last_density(self) -> float:
        return self.L.mdh_last_density(self.handle)
