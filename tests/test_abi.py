"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/impdar_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'impdar_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = re.findall(r'\b((?:impdar_|mig_kirch_)\w+)\s*\(', text)
    return sorted(set(names))


def test_header_declares_the_reference_hook():
    syms = header_symbols()
    assert 'mig_kirch_loop' in syms            # mig_cython.h:11
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from impdar_amd import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_table_covers_header():
    from impdar_amd import _hip
    assert sorted(_hip.SIGNATURES) == header_symbols()


def test_no_device_fails_loudly():
    from impdar_amd import _hip
    _hip.load()
    if _hip.device_count() > 0:
        pytest.skip('a GPU is present')
    from impdar_amd.lib.NoInitRadarData import NoInitRadarData
    d = NoInitRadarData(big=True)
    with pytest.raises(_hip.HipUnavailableError):
        d.migrate('kirch')
    assert 'no HIP device' in _hip.last_error()


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, 'impdar_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(base, f)).read()
                assert 'oracle' not in src.replace('the oracle', ''), os.path.join(base, f)
