"""Build the HIP shared library in-tree (``impdar_amd/csrc/libimpdar_hip.so``).

hipcc cross-compiles for gfx950 without a GPU present.  The library links
rocFFT (Stolt / phase-shift transforms) and RCCL (multi-GPU all-gather).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libimpdar_hip.so')
SOURCES = ['api.hip', 'comm.hip', 'kirchhoff.hip', 'stolt.hip', 'phaseshift.hip', 'preproc.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off',
         '-fno-slp-vectorize', '-Wall', '-Wno-unused-function']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')) + \
        [os.path.join(HERE, '..', 'include', 'impdar_hip.h')]
    objs, jobs = [], []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps[len(srcs):]):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))      # the sources compile side by side (kirchhoff.hip is minutes)
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + \
              ['-o', LIB, '-L/opt/rocm/lib', '-lrocfft', '-lrccl', '-Wl,-rpath,/opt/rocm/lib']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
