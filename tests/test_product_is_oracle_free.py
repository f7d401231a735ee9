"""The shipped path must not import, call or link the oracle (or any CPU fallback)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_package_never_mentions_the_oracle():
    pkg = os.path.join(ROOT, 'sea_ice_drift_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), (f, 'imports oracle')
                assert 'pm_oracle' not in text and 'libsid_pm_oracle' not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sea_ice_drift_amd import _capi
    monkeypatch.setattr(_capi, '_lib', None)
    monkeypatch.setattr(_capi, 'LIB_PATH', str(tmp_path / 'absent.so'))
    try:
        _capi.lib()
    except ImportError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('expected ImportError')
