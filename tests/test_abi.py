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


def test_every_knob_is_documented_and_tested():
    """Every environment variable read inside impdar_amd/csrc/ has a row in INTEGRATION.md section 4 and is set by at
    least one GPU test; at most 20 of them (round 3 had 40)."""
    import glob
    src = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'impdar_amd', 'csrc', '*.h*')))
    knobs = sorted(set(re.findall(r'getenv\("(IMPDAR_[A-Z0-9_]+)"\)', src)))
    assert 0 < len(knobs) <= 20, knobs
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    table = doc[doc.index('## 4. Environment variables'):doc.index('## 5. CLI')]
    tests = ''.join(open(f).read() for f in glob.glob(os.path.join(ROOT, 'tests', 'test_*_gpu.py')))
    for k in knobs:
        assert '`%s`' % k in table, '%s is read by the library but has no row in INTEGRATION.md' % k
        assert "'%s'" % k in tests, '%s is not set by any GPU test' % k
    documented = set(re.findall(r'^\| `(IMPDAR_[A-Z0-9_]+)`', table[:table.index('Read by the Python host side')], flags=re.M))
    assert documented == set(knobs), sorted(documented ^ set(knobs))


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


def test_band_pass_kernels_stay_in_architectural_vgprs(tmp_path):
    """The band-pass kernels prefetch with inline-asm loads that the compiler cannot see (csrc/preproc.hip): if a
    kernel needed more than the 256 architectural VGPRs the compiler would park live values in AccVGPRs, and a
    copy of a register whose load has not landed copies garbage.  Guard the register budget at build level."""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('hipcc not available')
    from impdar_amd import build
    out = str(tmp_path / 'preproc.s')
    flags = [f for f in build.FLAGS if f != '-fPIC']
    subprocess.check_call([hipcc] + flags + ['--cuda-device-only', '-S', os.path.join(build.CSRC, 'preproc.hip'), '-o', out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    rows = re.findall(r'\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)',
                      text, flags=re.S)
    ff = [(name, int(agpr), int(scratch), int(vgpr)) for agpr, name, scratch, vgpr in rows if 'ff_' in name]
    assert len(ff) == 16, [r[0] for r in ff]
    for name, agpr, scratch, vgpr in ff:
        assert agpr == 0 and scratch == 0 and vgpr <= 256, (name, agpr, scratch, vgpr)
