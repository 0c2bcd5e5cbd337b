"""Time the dlopen of the HIP runtime, rocFFT, RCCL and the library itself in a fresh process (what `context_ms` of the
first-call records holds besides the context)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = time.perf_counter()
for name in ('/opt/rocm/lib/libamdhip64.so', '/opt/rocm/lib/librocfft.so.0', '/opt/rocm/lib/librccl.so.1',
             os.path.join(ROOT, 'impdar_amd', 'csrc', 'libimpdar_hip.so')):
    if len(sys.argv) > 1 and sys.argv[1] in name:
        continue
    ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
    t1 = time.perf_counter()
    print('dlopen %-45s %7.1f ms' % (os.path.basename(name), (t1 - t) * 1e3), flush=True)
    t = t1
