"""Does a rocprofv3 --pmc child process (as bench.py's counter passes run them) disturb the NEXT long kernels of this process?
usage: ps_after_pmc_repro.py [pmc] [cpu]"""
import sys, os, io, contextlib, subprocess, shutil, tempfile, time
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = 8192
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
tab = np.stack([np.linspace(1.69e8, 2.2e8, 41), np.linspace(0., 2.0 * Rp, 41)], axis=1)
def calls(tag, k=4):
    kms = []
    for i in range(k):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
        v = C.c_float(); _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v))); kms.append(round(v.value, 2))
        d._dev.free(); d._dev = None
    print(tag, kms, flush=True)
calls('warm-up')
if 'cpu' in sys.argv:
    import bench
    t0 = time.time()
    a = np.random.default_rng(1).standard_normal((2048, 2048))
    for _ in range(30): a = a @ a * 1e-3
    print('cpu leg %.1f s' % (time.time() - t0), flush=True)
if 'pmc' in sys.argv:
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    d = tempfile.mkdtemp(prefix='impdar_pmc_', dir='/tmp')
    cmd = [prof, '--pmc', 'FETCH_SIZE', '--kernel-trace', '-d', d, '-o', 'x', '--output-format', 'csv', '--',
           sys.executable, os.path.abspath('bench.py'), '--pmc-child-path', 'stolt', '--steps', '3']
    rc = subprocess.call(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print('pmc child rc', rc, flush=True)
calls('after')
time.sleep(2.0)
calls('2 s later')
