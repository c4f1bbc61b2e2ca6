"""The C-ABI library loads without a GPU and exports every symbol include/prcore.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "prcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("prc_caf_plan_create", "prc_caf_execute", "prc_ls_plan_create", "prc_ls_execute",
                 "prc_nlms_execute", "prc_xcorr", "prc_frequency_shift", "prc_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from passiveradar_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(handle, s)]
    assert not missing, missing
    assert handle.prc_version() >= 100


def test_python_binding_matches_header():
    from passiveradar_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    _lib.lib()      # argtypes/restype set for every symbol


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must raise, never compute on the CPU."""
    import numpy as np
    from passiveradar_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    from passiveradar_amd.clutter_removal import LS_Filter_Toeplitz, NLMS_filter
    from passiveradar_amd.range_doppler_processing import fast_xambg
    x = np.ones(4096, np.complex64)
    with pytest.raises(_lib.PrcoreError):
        fast_xambg(x, x, 7, 64)
    with pytest.raises(_lib.PrcoreError):
        LS_Filter_Toeplitz(x, x, 8)
    with pytest.raises(_lib.PrcoreError):
        NLMS_filter(x, x, 8, 0.1)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under passiveradar_amd/ may reference it."""
    pkg = os.path.join(REPO, "passiveradar_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, fn), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "np_oracle" not in src and "liboracle" not in src, fn


def test_header_is_plain_c(tmp_path):
    """include/prcore.h must stay a C header (the boundary a cgo / JNI / ctypes binding consumes): C99 and C++11"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "include/prcore.h"\nint main(void) { prc_caf_desc d; prc_ls_desc l; prc_iir_desc i; prc_frontend_desc f;\n'
                   '  (void)d; (void)l; (void)i; (void)f; return prc_version() > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", repo, str(src)])
    if shutil.which("g++"):
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", repo, "-x", "c++", str(src)])


def test_options_api_without_a_gpu():
    """prc_set_option / prc_get_option: defaults, round trip, range checks -- no device needed"""
    from passiveradar_amd import _lib
    assert _lib.get_option(_lib.OPT_CAF_MULTI_MODE) == _lib.CAF_MULTI_AUTO
    assert _lib.get_option(_lib.OPT_LS_TEAM_PIECES) == 32 and _lib.get_option(_lib.OPT_LS_TEAM_ALIGN) == 1
    old = _lib.set_option(_lib.OPT_LS_TEAM_PIECES, 8)
    try:
        assert old == 32 and _lib.get_option(_lib.OPT_LS_TEAM_PIECES) == 8
    finally:
        _lib.set_option(_lib.OPT_LS_TEAM_PIECES, old)
    for opt, bad in ((_lib.OPT_CAF_MULTI_MODE, 9), (_lib.OPT_NLMS_WAVES, 3), (_lib.OPT_LS_TEAM_ALIGN, 2), (99, 0)):
        with pytest.raises(ValueError):
            _lib.set_option(opt, bad)


def test_library_reads_no_environment_variables():
    """kernel-selection knobs are descriptor fields or prc_set_option: libprcore.so must not even import getenv"""
    import shutil
    import subprocess
    from passiveradar_amd import _lib
    if shutil.which("nm") is None:
        pytest.skip("no nm")
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    assert not re.search(r"\b(secure_)?getenv\b", undefined)
    src_dir = os.path.join(REPO, "passiveradar_amd", "csrc")
    for fn in os.listdir(src_dir):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(src_dir, fn)).read(), fn


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / offsetof of the descriptor structs as gcc lays them out == the ctypes mirrors in _lib.py"""
    import shutil
    import subprocess
    from passiveradar_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "include/prcore.h"\nint main(void) {\n'
                   '  prc_caf_desc d; PRC_DESC_INIT(d);\n'
                   '  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(prc_caf_desc), offsetof(prc_caf_desc, taps_host), offsetof(prc_caf_desc, multi),\n'
                   '         sizeof(prc_ls_desc), sizeof(prc_frontend_desc), sizeof(prc_iir_desc));\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(prc_caf_desc, struct_size), offsetof(prc_caf_desc, magic),\n'
                   '         offsetof(prc_ls_desc, struct_size), offsetof(prc_ls_desc, magic), offsetof(prc_frontend_desc, struct_size),\n'
                   '         offsetof(prc_frontend_desc, magic), offsetof(prc_iir_desc, struct_size), offsetof(prc_iir_desc, magic));\n'
                   '  printf("%u %u %u %u %u %u %d\\n", PRC_CAF_DESC_SIZE_600, PRC_LS_DESC_SIZE_600, PRC_FRONTEND_DESC_SIZE_600, PRC_IIR_DESC_SIZE_600,\n'
                   '         d.struct_size, d.magic, PRC_VERSION); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", REPO, str(src), "-o", str(exe)])
    rows = [[int(v) for v in ln.split()] for ln in subprocess.check_output([str(exe)], text=True).splitlines()]
    want = [ctypes.sizeof(_lib.CafDesc), _lib.CafDesc.taps_host.offset, _lib.CafDesc.multi.offset,
            ctypes.sizeof(_lib.LsDesc), ctypes.sizeof(_lib.FrontendDesc), ctypes.sizeof(_lib.IirDesc)]
    assert rows[0] == want
    # struct_size and magic lead every descriptor (the library reads them before it knows anything else about the layout)
    assert rows[1] == [0, 4] * 4
    # the version-600 floor is this header's own sizeof; PRC_DESC_INIT fills the header of a descriptor as _lib._Desc does
    assert rows[2] == [want[0], want[3], want[4], want[5], want[0], _lib.DESC_MAGIC, _lib.MIN_LIB_VERSION]
    d = _lib.LsDesc()
    assert (d.struct_size, d.magic) == (ctypes.sizeof(_lib.LsDesc), _lib.DESC_MAGIC)


def _old_layout_caf_desc():
    """prc_caf_desc as header version 500 laid it out (no struct_size / magic; `n` leads)"""
    class Old(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int64), ("range_bins", ctypes.c_int32), ("freq_bins", ctypes.c_int32),
                    ("max_frames", ctypes.c_int32), ("method", ctypes.c_int32), ("doppler", ctypes.c_int32),
                    ("ntaps", ctypes.c_int32), ("taps_host", ctypes.POINTER(ctypes.c_float)), ("multi", ctypes.c_int32),
                    ("reserved", ctypes.c_int32)]
    return Old(4096, 7, 64, 1, 0, 0, 0, None, 0, 0)


def test_descriptor_of_another_layout_is_refused_before_anything_is_read():
    """descriptors grew between rounds with nothing telling the library how large the HOST's struct is.
    From version 600 every descriptor starts with struct_size + magic: a pre-600 host (old layout), a struct shorter than the
    600 layout and a size that is not a multiple of 4 are PRC_EINVAL (ValueError) with a message that says to rebuild --
    checked before the library touches a device, so this runs without a GPU."""
    from passiveradar_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()

    def create(fn, desc):
        return fn(ctypes.byref(h), ctypes.cast(ctypes.byref(desc), fn.argtypes[1]))

    old = _old_layout_caf_desc()
    assert create(lib.prc_caf_plan_create, old) == _lib.PRC_EINVAL
    msg = lib.prc_last_error().decode()
    assert "magic" in msg and "older than version 600" in msg and "rebuilt" in msg, msg
    for cls, fn, floor in ((_lib.CafDesc, lib.prc_caf_plan_create, 56), (_lib.LsDesc, lib.prc_ls_plan_create, 40),
                           (_lib.FrontendDesc, lib.prc_frontend_plan_create, 48)):
        assert ctypes.sizeof(cls) == floor
        for bad in (floor - 8, floor + 2, 0, 1 << 20):
            d = cls()
            d.struct_size = bad
            assert create(fn, d) == _lib.PRC_EINVAL, (cls.__name__, bad)
            msg = lib.prc_last_error().decode()
            assert f"struct_size = {bad}" in msg and f"at least {floor}" in msg, msg
        with pytest.raises(ValueError):
            d = cls()
            d.magic = 0
            _lib.check(create(fn, d))
    # the IIR descriptor travels by pointer into two entry points
    iir = _lib.IirDesc()
    iir.struct_size = 48
    rc = lib.prc_decimate_iir(None, 100, ctypes.byref(iir), None, None)
    assert rc == _lib.PRC_EINVAL and "prc_iir_desc.struct_size = 48" in lib.prc_last_error().decode()


def test_descriptor_from_a_newer_host_is_read_up_to_what_the_library_knows():
    """a host built against a LATER header passes a larger struct: the library reads its own sizeof, the size check passes
    and the call proceeds to the ordinary argument checks (here: a non-positive size, reported as such)"""
    from passiveradar_amd import _lib
    lib = _lib.lib()

    class Newer(ctypes.Structure):
        _fields_ = _lib.LsDesc._fields_ + [("knob_of_version_700", ctypes.c_int64)]
    d = Newer()
    d.struct_size, d.magic = ctypes.sizeof(Newer), _lib.DESC_MAGIC
    d.n, d.filter_len, d.peek, d.max_blocks, d.knob_of_version_700 = 0, 16, 10, 1, 12345
    h = ctypes.c_void_p()
    rc = lib.prc_ls_plan_create(ctypes.byref(h), ctypes.cast(ctypes.byref(d), lib.prc_ls_plan_create.argtypes[1]))
    assert rc == _lib.PRC_EINVAL and "non-positive size" in lib.prc_last_error().decode(), lib.prc_last_error()
