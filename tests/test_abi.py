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
