"""CPU sanitizer job: the HOST side of the C-ABI library built with AddressSanitizer and
UBSan (device code is compiled without instrumentation: GPU sanitizers are not available on this pool) and a C
driver that walks its argument-error and no-device paths.  Runs on the CPU container only."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SOURCES = ['api.hip', 'comm.hip', 'kirchhoff.hip', 'kirch_gen.hip', 'stolt.hip', 'phaseshift.hip', 'preproc.hip']


def _gpu_present():
    return os.path.exists('/dev/kfd')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
@pytest.mark.skipif(_gpu_present(), reason='sanitizer job is for the CPU container (never on the GPU box)')
def test_host_shim_under_asan_and_ubsan(tmp_path):
    san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined']
    lib = str(tmp_path / 'libimpdar_hip_san.so')
    # host pass only (--offload-host-only): the kernels' device code is what build() compiles; here it is the shim
    # around them that runs, and no kernel can be launched without a device anyway.  Seconds instead of minutes.
    objs, procs = [], []
    for src in SOURCES:
        o = str(tmp_path / (src[:-4] + '.o'))
        objs.append(o)
        procs.append(subprocess.Popen([HIPCC, '--offload-arch=gfx950', '--offload-host-only', '-O1', '-g', '-fPIC', '-std=c++17',
                                       '-ffp-contract=off', '-Wno-unused-function', '-Wno-unused-command-line-argument'] + san +
                                      ['-c', os.path.join(ROOT, 'impdar_amd', 'csrc', src), '-o', o],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        log = pr.communicate(timeout=1800)[0]
        assert pr.returncode == 0, log[-4000:]
    # a host-only object refers to its translation unit's device image by symbol; give each an empty one (it is only
    # registered by address at load time, never opened without a device)
    undef = subprocess.run(['nm', '-u'] + objs, capture_output=True, text=True).stdout
    fat = sorted({w for w in undef.split() if w.startswith('__hip_fatbin_')})
    stub = str(tmp_path / 'fatbin_stub.c')
    with open(stub, 'w') as fo:
        for sym in fat:
            fo.write('__attribute__((aligned(4096))) const char %s[4096] = {0};\n' % sym)
    out = subprocess.run([HIPCC, '-shared', '-fPIC'] + san + objs + ['-x', 'c', stub, '-o', lib, '-L/opt/rocm/lib', '-lrocfft', '-lrccl',
                                                                     '-Wl,-rpath,/opt/rocm/lib'],
                         capture_output=True, text=True, timeout=1800)
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
