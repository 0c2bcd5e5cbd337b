"""Round 6: the series path (ps_series_kernel, IMPDAR_PS_MFMA=7) against the oracle on small records and, at a given size,
timed against the kernels it would replace.
usage: r06_series_quick.py check            small records, several profiles, float32 / float64: error against the oracle
       r06_series_quick.py time [n [profiles [dtypes]]]   device / kernel ms at n x n (default 8192), modes 7 (series) and 1 (default policy)"""
import sys, os, io, contextlib, json
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
from oracle import mig_oracle
lib, ctx = _hip.load(), _hip.context()


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def profiles(n, geo):
    u = np.linspace(0., 1., n)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab4 = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    tab41 = np.stack([np.linspace(1.69e8, 2.2e8, 41), np.linspace(0., 2.0 * Rp, 41)], axis=1)
    return {'gradient': np.ascontiguousarray(1.69e8 + 0.5e8 * u),
            'falling': np.ascontiguousarray(2.2e8 - 0.5e8 * u),
            'wavy': np.ascontiguousarray(1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u),
            'firn': np.ascontiguousarray(1.69e8 + 0.6e8 * np.exp(-np.arange(n) * geo['dt'] / 0.8e-6)),
            'vz4': np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], tab4)),
            'layers41': np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], tab41))}


def metrics():
    buf = C.create_string_buffer(2048)
    lib.impdar_ctx_last_metrics(ctx, buf, len(buf))
    return json.loads(buf.value.decode())


def oracle_ps(data, geo, vm):
    """the oracle with a per-step profile (what getVelocityProfile would have returned)"""
    snum, tnum = data.shape
    tap = mig_oracle._apply_taper(data, 20, 30, inplace_form=True)
    nt = 2 ** int(np.ceil(np.log2(snum)))
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    FK = np.fft.fft2(tap, s=(nt, tnum))
    TK = mig_oracle.phase_shift_tk(FK, vm, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
    return np.fft.ifft(TK).real


def check():
    for snum, tnum in ((300, 64), (520, 40), (1100, 24), (2100, 16), (4200, 8)):
        geo = synth.geometry(snum, tnum)
        data = (synth.noise_radargram(snum, tnum, seed=snum) + 0.5)
        for name, vm in profiles(snum, geo).items():
            want = oracle_ps(data.astype(np.float64), geo, vm)
            row = {}
            for dt in (np.float32, np.float64):
                for mode in ('7', '1', '0'):
                    os.environ['IMPDAR_PS_MFMA'] = mode
                    x = np.ascontiguousarray(data.astype(dt))
                    out = np.empty((snum, tnum), dtype=dt)
                    nt = 2 ** int(np.ceil(np.log2(snum)))
                    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
                    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
                    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
                    dp = C.POINTER(C.c_double)
                    _hip.check(lib.impdar_phaseshift(ctx, x.ctypes.data_as(C.c_void_p), _hip.dtype_code(dt), snum, tnum, nt,
                                                     kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                                     tt.ctypes.data_as(dp), C.c_double(0.0), vm.ctypes.data_as(dp), snum, C.c_double(20.),
                                                     C.c_double(30.), out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
                    e = rel(out.astype(np.float64), want)
                    row[np.dtype(dt).name + ':' + mode] = (metrics()['kernel'], '%.2e' % e[0], '%.2e' % e[1])
            print(snum, tnum, name, row, flush=True)


def timeit(n, names=None, dts=None):
    geo = synth.geometry(n, n)
    kx = 2. * np.pi * np.fft.fftfreq(n, d=1.0)
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    p = lambda a: _hip.as_dp(a)[1]
    for name, vm in profiles(n, geo).items():
        if names and name not in names:
            continue
        for dt in (np.float32, np.float64):
            if dts and np.dtype(dt).name not in dts:
                continue
            x = np.random.default_rng(0).standard_normal((n, n)).astype(dt)
            d_in = _hip.DeviceArray.from_host(ctx, x)
            d_out = _hip.DeviceArray(ctx, (n, n), dt)
            res = {}
            for mode in ('7', '1', '0'):
                os.environ['IMPDAR_PS_MFMA'] = mode
                ms, kms = [], []
                for i in range(3):
                    _hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, _hip.dtype_code(dt), n, n, n, p(kx), p(ws), geo['dt'], p(geo['travel_time']),
                                                         0.0, p(vm), n, 100.0, 1000.0, d_out.ptr), 'ps')
                    v = C.c_float(); _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v))); ms.append(round(v.value, 2))
                    _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v))); kms.append(round(v.value, 2))
                res[mode] = dict(kernel=metrics().get('kernel'), device_ms=ms, kernel_ms=kms)
                if mode == '7':
                    o7 = d_out.to_host().astype(np.float64)
                elif mode == '1':
                    o1 = d_out.to_host().astype(np.float64)
            res['7 vs 1 rel L2'] = '%.2e' % rel(o7, o1)[0]
            print(n, name, np.dtype(dt).name, json.dumps(res), flush=True)
            d_in.free(); d_out.free()


if __name__ == '__main__':
    if sys.argv[1] == 'check':
        check()
    else:
        timeit(int(sys.argv[2]) if len(sys.argv) > 2 else 8192, sys.argv[3].split(',') if len(sys.argv) > 3 else None,
               sys.argv[4].split(',') if len(sys.argv) > 4 else None)
