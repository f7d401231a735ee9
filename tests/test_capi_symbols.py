"""CPU test: the C-ABI library loads and exports every symbol include/sid_pm.h and include/sid_ft.h declare.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

from sea_ice_drift_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'sid_pm.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sid_pm_[a-z_0-9]+)\s*\(', src)))


def ft_header_functions():
    src = open(os.path.join(ROOT, 'include', 'sid_ft.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sid_ft_[a-z_0-9]+)\s*\(', src)))


def stage_header_functions():
    src = open(os.path.join(ROOT, 'include', 'sid_stage.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sid_stage_[a-z_0-9]+)\s*\(', src)))


def orb_header_functions():
    src = open(os.path.join(ROOT, 'include', 'sid_orb.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sid_orb_[a-z_0-9]+)\s*\(', src)))


def fg_header_functions():
    src = open(os.path.join(ROOT, 'include', 'sid_fg.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sid_fg_[a-z_0-9]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_capi.SYMBOLS)
    assert ft_header_functions() == sorted(_capi.FT_SYMBOLS)
    assert stage_header_functions() == sorted(_capi.STAGE_SYMBOLS)
    assert orb_header_functions() == sorted(_capi.ORB_SYMBOLS)
    assert fg_header_functions() == sorted(_capi.FG_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_capi.LIB_PATH), 'build with __graft_entry__.build() first'
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in header_functions() + ft_header_functions() + stage_header_functions() + orb_header_functions() + fg_header_functions():
        assert hasattr(lib, name), name
    assert lib.sid_pm_abi_version() == _capi.ABI_VERSION
    lib.sid_pm_strerror.restype = ctypes.c_char_p
    assert lib.sid_pm_strerror(-4) == b'unsupported option or size'


def test_library_carries_gfx950_code_object():
    blob = open(_capi.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob
    assert b'pm_kernel' in blob


def test_argument_errors_without_a_device():
    """Argument validation that happens before any device work."""
    lib = _capi.lib()
    assert lib.sid_pm_create(0, None) == -1
    assert lib.sid_pm_run(None) == -1
    assert b'null ctx' in lib.sid_pm_last_error()


def test_matcher_argument_errors_without_a_device():
    lib = _capi.lib()
    assert lib.sid_ft_knn2(0, None, 4, None, 4, None, None) == -1
    assert b'null pointer' in lib.sid_ft_last_error()
    assert lib.sid_ft_knn2(0, None, 0, None, 0, None, None) == 0          # nothing to do
    assert lib.sid_ft_workspace_bytes(1000, 1000) >= 8 * 1000
