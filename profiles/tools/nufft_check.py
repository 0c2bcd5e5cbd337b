"""The transform path (ps_nufft_kernel, IMPDAR_PS_MFMA=6) against the vector kernels (=0) and the oracle on small records."""
import sys, os, io, contextlib, json
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
from oracle import mig_oracle
lib, ctx = _hip.load(), _hip.context()
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for snum, tnum in ((520, 40), (1000, 33), (2100, 16), (4200, 8)):
    geo = synth.geometry(snum, tnum)
    data = (synth.noise_radargram(snum, tnum, seed=snum) + 0.5).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    for kind, vel in (('const', 1.69e8), ('layers', np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])),
                      ('boundary', np.array([[1.68e8, 0.], [1.68e8, 0.45 * Rp], [1.8e8, 0.7 * Rp], [1.9e8, 1.2 * Rp]]))):
        want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], vel, 20, 30)
        out = {}
        for mode in ('6', '0', '1'):
            os.environ['IMPDAR_PS_MFMA'] = mode
            d = RadarData(None); d.data, d.snum, d.tnum = data.copy(), snum, tnum
            d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
            with contextlib.redirect_stdout(io.StringIO()):
                d.migrate('phsh', vel=vel, htaper=20, vtaper=30)
            buf = C.create_string_buffer(1024); lib.impdar_ctx_last_metrics(ctx, buf, len(buf))
            out[mode] = (json.loads(buf.value.decode())['kernel'], rel(d.data, want))
        print(snum, tnum, kind, out, flush=True)
