"""ctypes binding of libprcore.so (include/prcore.h) -- the only way the package computes.

There is no CPU fallback: if the shared library is missing or no MI355X is visible the
functions raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C passiveradar_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRCORE_LIB", os.path.join(_HERE, "libprcore.so"))   # override: A/B kernel builds

PRC_OK, PRC_EINVAL, PRC_ESHAPE, PRC_EHIP, PRC_EROCFFT, PRC_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
CAF_AUTO, CAF_DIRECT, CAF_FFT, CAF_FFT4096 = 0, 1, 2, 3
DOPPLER_AUTO, DOPPLER_ROCFFT, DOPPLER_COLUMN = 0, 1, 2
CAF_MULTI_AUTO, CAF_MULTI_TURNS, CAF_MULTI_SHARED, CAF_MULTI_PAIRS = 0, 1, 2, 3
CAF_MULTI_MODES = {"auto": CAF_MULTI_AUTO, "turns": CAF_MULTI_TURNS, "shared": CAF_MULTI_SHARED, "pairs": CAF_MULTI_PAIRS}
# prc_option
(OPT_CAF_MULTI_MODE, OPT_CAF_GROUP_MB, OPT_LS_TEAM_PIECES, OPT_LS_TEAM_ALIGN, OPT_NLMS_WAVES, OPT_LS_CACHE_LIMIT_MB,
 OPT_NLMS_WG_WAVES, OPT_CAF_XCD_CONTIG, OPT_CAF_PAIR_FRAMES, OPT_FE_METHOD,
 OPT_CFAR_METHOD, OPT_MARKERS, OPT_FE_BALANCE, OPT_CAF_TEAM8, OPT_FE_FOLD) = range(15)
CAF_MAX_REFS = 8
COMM_ID_BYTES = 128


class PrcoreError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libprcore error {code}: {msg}")
        self.code = code


DESC_MAGIC = 0x36435250      # PRC_DESC_MAGIC, "PRC6"


class _Desc(C.Structure):
    """Descriptor structs of include/prcore.h: the first field says how large the HOST's struct is (version 600), so the
    library never reads past it and takes fields the host does not know as 0."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))
        self.magic = DESC_MAGIC


class CafDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("magic", C.c_uint32),
                ("n", C.c_int64), ("range_bins", C.c_int32), ("freq_bins", C.c_int32),
                ("max_frames", C.c_int32), ("method", C.c_int32), ("doppler", C.c_int32),
                ("ntaps", C.c_int32), ("taps_host", C.POINTER(C.c_float)), ("multi", C.c_int32)]


class LsDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("magic", C.c_uint32),
                ("n", C.c_int64), ("filter_len", C.c_int32), ("peek", C.c_int32),
                ("circular", C.c_int32), ("max_blocks", C.c_int32), ("method", C.c_int32)]


class FrontendDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("magic", C.c_uint32),
                ("n_in", C.c_int64), ("raw_dtype", C.c_int32), ("up", C.c_int32), ("down", C.c_int32),
                ("ntaps", C.c_int32), ("n_pre_remove", C.c_int32), ("max_blocks", C.c_int32),
                ("taps_host", C.POINTER(C.c_float))]


class IirDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("magic", C.c_uint32),
                ("q", C.c_int32), ("padlen", C.c_int32), ("settle", C.c_int32), ("nzeros", C.c_int32),
                ("npoles", C.c_int32), ("zeros_host", C.POINTER(C.c_double)),
                ("poles_host", C.POINTER(C.c_double)), ("gain", C.c_double)]


MIN_LIB_VERSION = 600      # PRC_VERSION of include/prcore.h these ctypes declarations mirror

RAW_DTYPES = {"int8": 0, "uint8": 1, "int16": 2, "float32": 3, "complex64": 4}

_lib = None
_lib_lock = threading.Lock()

_SIGNATURES = {
    "prc_version": (C.c_int, []),
    "prc_last_error": (C.c_char_p, []),
    "prc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "prc_set_device": (C.c_int, [C.c_int]),
    "prc_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "prc_mem_info": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "prc_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "prc_free": (C.c_int, [C.c_void_p]),
    "prc_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "prc_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "prc_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "prc_stream_sync": (C.c_int, [C.c_void_p]),
    "prc_set_option": (C.c_int, [C.c_int32, C.c_int64]),
    "prc_get_option": (C.c_int, [C.c_int32, C.POINTER(C.c_int64)]),
    "prc_caf_plan_multi_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "prc_caf_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CafDesc)]),
    "prc_caf_plan_destroy": (C.c_int, [C.c_void_p]),
    "prc_caf_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int64)]),
    "prc_caf_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_caf_execute_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "prc_caf_execute_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_caf_execute_doppler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_ls_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(LsDesc)]),
    "prc_ls_plan_destroy": (C.c_int, [C.c_void_p]),
    "prc_ls_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                 C.c_int32, C.c_double, C.POINTER(C.c_double), C.c_int32, C.c_double,
                                 C.c_void_p, C.c_void_p]),
    "prc_ls_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "prc_ls_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "prc_nlms_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                   C.c_void_p]),
    "prc_frontend_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(FrontendDesc)]),
    "prc_frontend_plan_destroy": (C.c_int, [C.c_void_p]),
    "prc_frontend_out_len": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "prc_frontend_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                       C.POINTER(C.c_double), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "prc_frontend_execute2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                        C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "prc_deinterleave": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "prc_frequency_shift_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                            C.c_double, C.c_void_p]),
    "prc_frequency_shift_phases": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                             C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_cfar2d": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                             C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_cfar2d_c64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                 C.c_void_p, C.c_int32, C.c_void_p]),
    "prc_decimate_iir": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(IirDesc), C.c_void_p, C.c_void_p]),
    "prc_channel_offset": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(IirDesc), C.c_int64,
                                     C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "prc_xcorr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                            C.c_void_p]),
    "prc_frequency_shift": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                      C.c_double, C.c_void_p]),
    "prc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "prc_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32]),
    "prc_comm_destroy": (C.c_int, [C.c_void_p]),
    "prc_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int32)]),
    "prc_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "prc_comm_loopback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "prc_gather_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_void_p,
                                    C.c_int32, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under
    torch/lib (same SONAMEs as /opt/rocm's).  Whichever copy is loaded first serves every later user of that
    SONAME: if libprcore pulled in /opt/rocm's first, a later ``import torch`` would run on a runtime it was not
    built with and report "No HIP GPUs are available".  So when torch is installed its copy is loaded first (without
    importing torch) -- the configuration every GPU test runs in.  Found by the fuzz run with two caller threads: one thread
    loading this library while another is half way through `import torch` used to take the "torch is already there" shortcut."""
    import importlib.util
    import sys
    # (no shortcut for "torch is in sys.modules": another thread may be in the middle of `import torch` -- the module is
    # registered before its libraries are loaded -- and loading an already loaded library again is a no-op)
    try:
        mod = sys.modules.get("torch")
        spec = getattr(mod, "__spec__", None) if mod is not None else importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libprcore.so once; raise (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is not built: run `make -C passiveradar_amd/csrc` (hipcc, gfx950). "
                    "passiveradar_amd has no CPU fallback.")
            _share_hip_runtime_with_torch()
            handle = C.CDLL(LIB_PATH)
            # a stale build next to newer Python would fail below with an AttributeError on the first missing symbol, or --
            # worse -- read descriptors of another layout: say what to do instead
            handle.prc_version.restype = C.c_int
            built = handle.prc_version()
            if built < MIN_LIB_VERSION:
                raise ImportError(f"{LIB_PATH} was built from prcore.h version {built}, this package needs >= "
                                  f"{MIN_LIB_VERSION} (descriptor layout with struct_size): rebuild it with "
                                  f"`make -C passiveradar_amd/csrc`")
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(rc):
    """Map a prc_status to the reference's error convention (PRC_ESHAPE -> ValueError)."""
    if rc == PRC_OK:
        return
    msg = lib().prc_last_error().decode("utf-8", "replace")
    if rc == PRC_ESHAPE:
        raise ValueError(msg)
    if rc == PRC_EINVAL:
        raise ValueError(msg)
    if rc == PRC_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise PrcoreError(rc, msg)


def device_count():
    n = C.c_int(0)
    rc = lib().prc_device_count(C.byref(n))
    return n.value if rc == PRC_OK else 0


def current_device():
    """the calling thread's current HIP device index (-1 without a GPU)"""
    d = C.c_int(-1)
    return d.value if lib().prc_get_device(C.byref(d)) == PRC_OK else -1


def set_option(option, value):
    """prc_set_option: process-wide tuning option (the library reads no environment variables); plans copy the options
    that concern them when they are created.  Returns the previous value."""
    old = get_option(option)
    check(lib().prc_set_option(int(option), int(value)))
    return old


def get_option(option):
    v = C.c_int64(0)
    check(lib().prc_get_option(int(option), C.byref(v)))
    return int(v.value)


def mem_info():
    """(free, total) bytes of the current device's memory (hipMemGetInfo)"""
    f, t = C.c_size_t(0), C.c_size_t(0)
    check(lib().prc_mem_info(C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def rccl_version():
    """ncclGetVersion of the RCCL bound by the gather (None when RCCL cannot be loaded)"""
    v = C.c_int32(0)
    return int(v.value) if lib().prc_comm_rccl_version(C.byref(v)) == PRC_OK else None


def require_gpu():
    if device_count() < 1:
        raise PrcoreError(PRC_EHIP, "no ROCm device visible: " +
                          lib().prc_last_error().decode("utf-8", "replace"))


class DeviceBuffer:
    """A hipMalloc'ed region owned by Python (library-side allocator, no torch needed)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().prc_malloc(C.byref(p), max(self.nbytes, 8)))
        self.ptr = p.value

    def upload(self, arr, offset_bytes=0, stream=None):
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes
        check(lib().prc_memcpy_h2d(self.ptr + offset_bytes, arr.ctypes.data, arr.nbytes, stream))
        check(lib().prc_stream_sync(stream))   # arr may be a temporary

    def zero(self, offset_bytes=0, nbytes=None, stream=None):
        nbytes = self.nbytes - offset_bytes if nbytes is None else nbytes
        check(lib().prc_memset(self.ptr + offset_bytes, 0, nbytes, stream))

    def download(self, shape, dtype, offset_bytes=0, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        check(lib().prc_memcpy_d2h(out.ctypes.data, self.ptr + offset_bytes, out.nbytes, stream))
        check(lib().prc_stream_sync(stream))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            lib().prc_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def is_device_tensor(x):
    """torch CUDA/HIP tensor (zero-copy path) vs host array (staged through DeviceBuffer)."""
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def torch_stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` (default: the current device)"""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
