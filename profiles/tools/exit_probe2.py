"""Does a process that made its rocFFT plans on the background thread exit cleanly?  usage: exit_probe2.py"""
import subprocess, sys, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
code = r'''
import sys, os, time, io, contextlib
sys.path.insert(0, %r)
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
n = 1024
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
for i in range(int(sys.argv[2])):
    d = RadarData(None); d.data, (d.snum, d.tnum) = x, x.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate(sys.argv[1], vel=1.69e8)
    time.sleep(float(sys.argv[3]))
print('done', flush=True)
''' % R
for mtype in ('phsh', 'stolt'):
    for env, tag in (({}, 'default'), ({'IMPDAR_PS_FFT': 'own', 'IMPDAR_STOLT_FFT': 'own'}, 'own only')):
        for ncall, sleep in ((1, 0.0), (1, 3.0), (4, 1.0)):
            r = subprocess.run([sys.executable, '-c', code, mtype, str(ncall), str(sleep)], capture_output=True, text=True, env=dict(os.environ, **env))
            print(mtype, tag, 'calls', ncall, 'sleep', sleep, '-> rc', r.returncode, r.stdout.strip(), r.stderr.strip()[-150:].replace('\n', ' | '), flush=True)
