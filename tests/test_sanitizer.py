"""CPU sanitizer job: the HOST side of the C-ABI library built with AddressSanitizer and
UBSan (device code is compiled without instrumentation: GPU sanitizers are not available on this pool) and a C
driver that walks its argument-error and no-device paths.  Runs on the CPU container only."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SOURCES = ['api.hip', 'comm.hip', 'kirchhoff.hip', 'stolt.hip', 'phaseshift.hip', 'preproc.hip']


def _gpu_present():
    return os.path.exists('/dev/kfd')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
@pytest.mark.skipif(_gpu_present(), reason='sanitizer job is for the CPU container (never on the GPU box)')
def test_host_shim_under_asan_and_ubsan(tmp_path):
    san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined']
    lib = str(tmp_path / 'libimpdar_hip_san.so')
    srcs = [os.path.join(ROOT, 'impdar_amd', 'csrc', s) for s in SOURCES]
    cmd = [HIPCC, '--offload-arch=gfx950', '-O1', '-g', '-fPIC', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize',
           '-Wno-unused-function', '-shared'] + san + srcs + ['-o', lib, '-L/opt/rocm/lib', '-lrocfft', '-lrccl',
                                                              '-Wl,-rpath,/opt/rocm/lib']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-4000:]
    exe = str(tmp_path / 'san_driver')
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), '..', 'lib', 'llvm', 'bin', 'clang')
    clang = clang if os.path.exists(clang) else (shutil.which('clang') or HIPCC)
    out = subprocess.run([clang, '-g', '-O0'] + san[:1] + san[2:] + [os.path.join(ROOT, 'tests', 'san', 'san_driver.c'), lib, '-o', exe,
                                                                  '-Wl,-rpath,' + str(tmp_path), '-Wl,-rpath,/opt/rocm/lib'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=99', UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert 'AddressSanitizer' not in run.stderr and 'runtime error' not in run.stderr, run.stderr[-6000:]
    assert run.returncode == 0 and 'san_driver ok' in run.stdout, (run.returncode, run.stdout, run.stderr[-4000:])
