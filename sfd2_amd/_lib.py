"""ctypes binding of libsfd2hip.so (C-ABI in include/sfd2_hip.h).

There is NO fallback: if the shared object is missing or no MI355X is visible the
product path raises.  (oracle/ is test infrastructure and is never imported here.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsfd2hip.so")


def use_library(path):
    """Kernel A/B experiments (tools/): bind a differently built libsfd2hip before the first load().  An explicit call,
    not an environment variable, so nothing outside the process can redirect the product to another binary."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is first loaded")
    LIB_PATH = os.path.abspath(path)

FLAG_ASYNC = 1
FLAG_NO_STABILITY = 2
FLAG_IMG_NORMALISED = 4
FLAG_IMG_U8_HWC = 8
FLAG_IMG_BGR = 16
FLAG_IMG_U8_X = 32
FLAG_MATCH_OUT16 = 64
FLAG_DESC_STORE64 = 128
MATCH_HLOC, MATCH_ITLOC_NNM, MATCH_ITLOC_NNR = 0, 1, 2
DT_F32, DT_F64, DT_F16 = 0, 1, 2
LAYOUT_ND, LAYOUT_DN = 0, 1
SIM_F16, SIM_F16X2 = 0, 1


class Sfd2Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 4)]


class MatchConf(ctypes.Structure):
    _fields_ = [("flavour", ctypes.c_int32), ("do_mutual_check", ctypes.c_int32),
                ("ratio_threshold", ctypes.c_float), ("distance_threshold", ctypes.c_float),
                ("sim_mode", ctypes.c_int32)]


class Timings(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_float), ("ms_backbone", ctypes.c_float), ("ms_post", ctypes.c_float),
                ("ms_match", ctypes.c_float), ("n_candidates", ctypes.c_int64)]


class DescSet(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("n", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("layout", ctypes.c_int32), ("on_device", ctypes.c_int32),
                ("rows", ctypes.c_void_p), ("n_rows", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class LayerTiming(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("kernel", ctypes.c_char * 48), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double), ("ms_total", ctypes.c_double), ("launches", ctypes.c_int32)]


class ExtractRecord(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("n_candidates", ctypes.c_int32), ("saturated", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


RANGE_TENSORS, RANGE_GROUPS = 17, 14


class RangeStatus(ctypes.Structure):
    _fields_ = [("n_tensors", ctypes.c_int32), ("max_stored", ctypes.c_float * RANGE_TENSORS),
                ("max_value", ctypes.c_float * RANGE_TENSORS), ("exponent", ctypes.c_int32 * RANGE_TENSORS),
                ("saturated", ctypes.c_uint32), ("low", ctypes.c_uint32), ("fallbacks", ctypes.c_int32)]


# every symbol include/sfd2_hip.h declares (tests/test_abi.py checks the two lists agree)
EXPORTS = [
    "sfd2_version", "sfd2_last_error", "sfd2_ctx_create", "sfd2_ctx_destroy", "sfd2_get_stream",
    "sfd2_load_weights", "sfd2_det", "sfd2_extract", "sfd2_extract_count", "sfd2_simple_nms",
    "sfd2_select_keypoints", "sfd2_sample_descriptors", "sfd2_heatmap", "sfd2_debug_activation",
    "sfd2_match", "sfd2_match_batch", "sfd2_get_timings", "sfd2_sync", "sfd2_set_profiling",
    "sfd2_get_layer_timings", "sfd2_set_precision", "sfd2_extract_spp", "sfd2_nms_fast",
    "sfd2_set_profile_filter", "sfd2_extract_multiscale", "sfd2_set_option", "sfd2_extract_match", "sfd2_preprocess", "sfd2_extract_spp_levels", "sfd2_match_segments",
    "sfd2_get_range_status", "sfd2_range_tensor_name", "sfd2_calibrate_range", "sfd2_get_act_exponents", "sfd2_set_act_exponents",
    "sfd2_extract_record_async", "sfd2_desc_pack", "sfd2_get_margin_status", "sfd2_get_relax_status", "sfd2_get_option", "sfd2_device_pci_bus_id",
]

_lib = None


def load():
    """Loads the shared object (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if LIB_PATH == os.path.join(_HERE, "libsfd2hip.so"):
        # (re)build when the shared object is missing or older than its sources and a hipcc is at hand; on a box
        # without hipcc the shipped .so is used as it is
        # One process per GPU means several ranks may get here at once: the build runs under an exclusive
        # file lock, the loser(s) re-check and find the library up to date; build_lib links to a temporary name and
        # renames, so a concurrent CDLL never sees a half-written file.
        from . import build as _build
        if _build.needs_build() and _build.have_hipcc():
            import fcntl
            with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    if _build.needs_build():
                        _build.build_lib()
                finally:
                    fcntl.flock(lk, fcntl.LOCK_UN)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing and no hipcc was found to build it "
                           "(hipcc --offload-arch=gfx950; `python -c 'import __graft_entry__ as g; g.build()'`). "
                           "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
    pi = ctypes.POINTER(ctypes.c_int)
    lib.sfd2_version.restype = ci
    lib.sfd2_last_error.restype = ctypes.c_char_p
    lib.sfd2_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.sfd2_ctx_destroy.argtypes = [vp]
    lib.sfd2_ctx_destroy.restype = None
    lib.sfd2_get_stream.argtypes = [vp]
    lib.sfd2_get_stream.restype = vp
    lib.sfd2_load_weights.argtypes = [vp, ctypes.POINTER(Sfd2Tensor), ci]
    lib.sfd2_det.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, pi, pi, pi, pi]
    lib.sfd2_extract.argtypes = [vp, vp, ci, ci, ci, cf, ci, ci, vp, vp, vp, ci, i64, pi]
    lib.sfd2_extract_count.argtypes = [vp, pi]
    lib.sfd2_extract_multiscale.argtypes = [vp, vp, ci, ci, ci, vp, ci, cf, ci, ci, vp, vp, vp, ci, i64, pi]
    lib.sfd2_extract_spp.argtypes = [vp, vp, ci, ci, ci, cf, ci, vp, vp, vp, i64, pi, vp, vp]
    lib.sfd2_nms_fast.argtypes = [vp, vp, ci, ci, cf, ci, vp]
    lib.sfd2_extract_spp_levels.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, cf, ci, vp, vp, vp, i64, vp]
    lib.sfd2_simple_nms.argtypes = [vp, vp, ci, ci, ci, vp]
    lib.sfd2_select_keypoints.argtypes = [vp, vp, ci, ci, cf, ci, ci, ci, vp, vp, i64, pi]
    lib.sfd2_sample_descriptors.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci, vp]
    lib.sfd2_heatmap.argtypes = [vp, vp, ci, ci, vp, ci, ci, ci, ci, vp]
    lib.sfd2_debug_activation.argtypes = [vp, ctypes.c_char_p, vp, i64, pi, pi, pi]
    lib.sfd2_match.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci, ctypes.POINTER(MatchConf), vp, vp, ci]
    lib.sfd2_match_segments.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, ctypes.POINTER(MatchConf), vp, vp, ci]
    lib.sfd2_match_batch.argtypes = [vp, ctypes.POINTER(DescSet), ctypes.POINTER(DescSet), ci, ci,
                                     ctypes.POINTER(MatchConf), vp, vp, ci, ci]
    lib.sfd2_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    lib.sfd2_sync.argtypes = [vp]
    lib.sfd2_set_precision.argtypes = [vp, ci]
    lib.sfd2_set_option.argtypes = [vp, ctypes.c_char_p, ci]
    lib.sfd2_get_option.argtypes = [vp, ctypes.c_char_p, pi]
    lib.sfd2_device_pci_bus_id.argtypes = [ci, ctypes.c_char_p, ci, pi]
    lib.sfd2_preprocess.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.sfd2_extract_match.argtypes = [vp, vp, ci, ci, cf, ci, ci, vp, vp, vp, ctypes.POINTER(DescSet), ci, ci,
                                       ctypes.POINTER(MatchConf), vp, vp]
    lib.sfd2_set_profiling.argtypes = [vp, ci]
    lib.sfd2_set_profile_filter.argtypes = [vp, ctypes.c_char_p]
    lib.sfd2_get_layer_timings.argtypes = [vp, ctypes.POINTER(LayerTiming), ci, pi]
    lib.sfd2_get_range_status.argtypes = [vp, ctypes.POINTER(RangeStatus), ci]
    lib.sfd2_range_tensor_name.argtypes = [ci]
    lib.sfd2_range_tensor_name.restype = ctypes.c_char_p
    lib.sfd2_calibrate_range.argtypes = [vp, vp, ci, ci, ci, ci]
    lib.sfd2_get_act_exponents.argtypes = [vp, vp, vp, ci, pi]
    lib.sfd2_get_margin_status.argtypes = [vp, vp, pi, ctypes.POINTER(ctypes.c_float)]
    lib.sfd2_get_relax_status.restype = ci
    lib.sfd2_get_relax_status.argtypes = [vp, ctypes.POINTER(ctypes.c_float), pi]
    lib.sfd2_set_act_exponents.argtypes = [vp, vp, ci]
    lib.sfd2_extract_record_async.argtypes = [vp, vp, ci]
    lib.sfd2_desc_pack.argtypes = [vp, ctypes.POINTER(DescSet), ci, vp, ci]
    for name in EXPORTS:
        getattr(lib, name)  # raises AttributeError if the .so lacks a declared symbol
    if lib.sfd2_version() < 108:
        raise RuntimeError(f"{LIB_PATH} is version {lib.sfd2_version()}, this binding needs >= 108 (rebuild: __graft_entry__.build())")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libsfd2hip: " + load().sfd2_last_error().decode("utf-8", "replace"))


def ptr(a):
    """Raw address of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


class Context:
    """One sfd2_ctx: one GPU, one HIP stream, one set of packed weights."""

    def __init__(self, device=0):
        self.lib = load()
        h = ctypes.c_void_p()
        check(self.lib.sfd2_ctx_create(int(device), ctypes.byref(h)))
        self.h = h
        self.device = int(device)
        self.options = {}           # what set_option was asked for, in order (ResSegNetV2.replica replays it on the other lanes of a pipelined driver)

    def close(self):
        if getattr(self, "h", None):
            self.lib.sfd2_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self.lib.sfd2_get_stream(self.h)

    def load_weights(self, sd):
        """sd: {name: array-like}; fp32 tensors of the reference state_dict."""
        keep, arr = [], (Sfd2Tensor * len(sd))()
        n = 0
        for name, v in sd.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if a.dtype.kind != "f":
                continue  # num_batches_tracked
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.ndim > 4:
                continue
            keep.append(a)
            arr[n].name = name.encode()
            arr[n].data = a.ctypes.data
            arr[n].ndim = a.ndim
            for i, s in enumerate(a.shape):
                arr[n].shape[i] = s
            n += 1
        check(self.lib.sfd2_load_weights(self.h, arr, n))

    def timings(self):
        t = Timings()
        check(self.lib.sfd2_get_timings(self.h, ctypes.byref(t)))
        return {"ms_total": t.ms_total, "ms_backbone": t.ms_backbone, "ms_post": t.ms_post,
                "ms_match": t.ms_match, "n_candidates": t.n_candidates}

    def set_precision(self, mode):
        """'f16', 'f32', 'f16x3', 'f16c' (include/sfd2_hip.h SFD2_PREC_*), or 'f16x3d' = SFD2_PREC_F16X3 with option "x3_desc16": that mode's
        key points, the descriptor branch in plain fp16 (descriptors within 1e-3 instead of 2e-5)."""
        check(self.lib.sfd2_set_precision(self.h, {'f16': 0, 'f32': 1, 'f16x3': 2, 'f16c': 3, 'f16x3d': 2}[mode]))
        check(self.lib.sfd2_set_option(self.h, b"x3_desc16", 1 if mode == 'f16x3d' else 0))

    def set_option(self, key, value):
        """'fuse', 'fuse_det', 'alias', 'graphs', 'fuse_post', 'sparse_desc', 'branches' (include/sfd2_hip.h sfd2_set_option)."""
        check(self.lib.sfd2_set_option(self.h, key.encode(), int(value)))
        self.options.pop(key, None)
        self.options[key] = int(value)

    def get_option(self, key):
        """What the context runs with (sfd2_get_option): the value last set or, for rb_inner / comp_heads / c3b_plain, the load-time self-check's choice."""
        v = ctypes.c_int(0)
        check(self.lib.sfd2_get_option(self.h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def sync(self):
        check(self.lib.sfd2_sync(self.h))

    def set_profiling(self, max_steps, kernel_filter=None):
        check(self.lib.sfd2_set_profile_filter(self.h, (kernel_filter or "").encode()))
        check(self.lib.sfd2_set_profiling(self.h, int(max_steps)))

    def layer_timings(self):
        """[{name, kernel, flops, bytes, ms_total, launches}] accumulated since set_profiling()."""
        n = ctypes.c_int(0)
        arr = (LayerTiming * 64)()
        check(self.lib.sfd2_get_layer_timings(self.h, arr, 64, ctypes.byref(n)))
        return [{"name": arr[i].name.decode(), "kernel": arr[i].kernel.decode(), "flops": arr[i].flops,
                 "bytes": arr[i].bytes, "ms_total": arr[i].ms_total, "launches": arr[i].launches}
                for i in range(min(n.value, 64))]

    def range_status(self, reset=False):
        """Largest value every stored tensor of the compensated mode reached since the last reset (include/sfd2_hip.h,
        sfd2_range_status): {'tensors': {name: {'max_stored', 'max_value', 'exponent'}}, 'saturated': [names], 'low': [names],
        'fallbacks': n}."""
        rs = RangeStatus()
        check(self.lib.sfd2_get_range_status(self.h, ctypes.byref(rs), 1 if reset else 0))
        names = [self.lib.sfd2_range_tensor_name(i).decode() for i in range(rs.n_tensors)]
        return {"tensors": {n: {"max_stored": rs.max_stored[i], "max_value": rs.max_value[i], "exponent": rs.exponent[i]}
                            for i, n in enumerate(names)},
                "saturated": [n for i, n in enumerate(names) if rs.saturated >> i & 1],
                "low": [n for i, n in enumerate(names) if rs.low >> i & 1], "fallbacks": rs.fallbacks}

    def calibrate_range(self, img, normalised=False):
        """img: [3,H,W] float32 (numpy: host; torch cuda tensor: device).  Sets the activation exponents of the fp16 family from
        this image's fp32 activations (sfd2_calibrate_range)."""
        on_dev = 0 if isinstance(img, np.ndarray) else 1
        if isinstance(img, np.ndarray):
            img = np.ascontiguousarray(img, dtype=np.float32)
        check(self.lib.sfd2_calibrate_range(self.h, ptr(img), on_dev, int(img.shape[-2]), int(img.shape[-1]),
                                            FLAG_IMG_NORMALISED if normalised else 0))

    def act_exponents(self):
        e = np.zeros(RANGE_GROUPS, dtype=np.int32)
        m = np.zeros(RANGE_GROUPS, dtype=np.float32)
        n = ctypes.c_int(0)
        check(self.lib.sfd2_get_act_exponents(self.h, e.ctypes.data, m.ctypes.data, RANGE_GROUPS, ctypes.byref(n)))
        return e, m

    def margin_status(self):
        """The load-time self-check of SFD2_PREC_F16C (include/sfd2_hip.h sfd2_get_margin_status): probe errors of the four option sets
        (-1 = not measured), which one runs, the target."""
        e = np.zeros(4, dtype=np.float32)
        ch, tg = ctypes.c_int(0), ctypes.c_float(0)
        check(self.lib.sfd2_get_margin_status(self.h, e.ctypes.data, ctypes.byref(ch), ctypes.byref(tg)))
        names = ["as set", "rb_inner=0", "comp_heads=1", "rb_inner=0 comp_heads=1"]
        ep, on = ctypes.c_float(0), ctypes.c_int(0)
        check(self.lib.sfd2_get_relax_status(self.h, ctypes.byref(ep), ctypes.byref(on)))
        return {"errors": {n: float(v) for n, v in zip(names, e)}, "choice": int(ch.value),
                "running": names[ch.value] if ch.value >= 0 else None, "target": float(tg.value),
                "c3b_plain": bool(on.value), "error_with_c3b_plain": float(ep.value)}

    def set_act_exponents(self, exps=None):
        if exps is None:
            check(self.lib.sfd2_set_act_exponents(self.h, None, 0))
        else:
            e = np.ascontiguousarray(exps, dtype=np.int32)
            check(self.lib.sfd2_set_act_exponents(self.h, e.ctypes.data, int(e.size)))

    def debug_activation(self, name):
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self.lib.sfd2_debug_activation(self.h, name.encode(), None, 0, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
        out = np.empty((c.value, h.value, w.value), dtype=np.float32)
        check(self.lib.sfd2_debug_activation(self.h, name.encode(), out.ctypes.data, out.size, ctypes.byref(c),
                                             ctypes.byref(h), ctypes.byref(w)))
        return out


_default_ctx = {}


def default_context(device=0):
    """Shared per-device context (used by the matchers)."""
    d = int(device)
    if d not in _default_ctx:
        _default_ctx[d] = Context(d)
    return _default_ctx[d]
