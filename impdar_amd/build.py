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
SOURCES = ['api.hip', 'comm.hip', 'kirchhoff.hip', 'kirch_gen.hip', 'stolt.hip', 'phaseshift.hip', 'preproc.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off',
         '-fno-slp-vectorize', '-Wall', '-Wno-unused-function']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _headers(path, seen=None):
    """The local headers a source includes (transitively) + the public header: its rebuild triggers."""
    import re
    seen = set() if seen is None else seen
    with open(path) as f:
        names = re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M)
    for n in names:
        h = os.path.join(CSRC, n)
        if not os.path.exists(h):
            h = os.path.join(HERE, '..', 'include', os.path.basename(n))
        if os.path.exists(h) and h not in seen:
            seen.add(h)
            _headers(h, seen)
    return sorted(seen)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, jobs = [], []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in _headers(s)):
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
