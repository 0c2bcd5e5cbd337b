"""ctypes wrapper of oracle/libkirch_oracle.so (plain-C Kirchhoff oracle).
TEST INFRASTRUCTURE ONLY -- see oracle/kirch_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'libkirch_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        _lib.kirch_oracle.restype = None
        _lib.kirch_oracle.argtypes = [dp, dp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int,
                                      C.POINTER(C.c_int), C.c_int, dp]
        _lib.kirch_oracle_threads.restype = C.c_int
    return _lib


def threads():
    return int(_load().kirch_oracle_threads())


def kirchhoff_from_gradient(grad, data, travel_time_us, dist_km, vel=1.69e8, nearfield=False, traces=None):
    """The diffraction sum alone (mig_python.py:35-60) on a given time-gradient array ``grad`` (snum, tnum)
    (and ``data`` for the near-field term): what a rank of the sharded migration runs on its exchanged image."""
    lib = _load()
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    snum, tnum = grad.shape
    d64 = np.ascontiguousarray(data if data is not None else grad, dtype=np.float64)
    tt = np.ascontiguousarray(np.asarray(travel_time_us) / 1.0e6, dtype=np.float64)
    dist = np.ascontiguousarray(dist_km, dtype=np.float64) * 1.0e3
    if traces is None:
        traces = np.arange(tnum)
    traces = np.ascontiguousarray(traces, dtype=np.int32)
    out = np.zeros((snum, len(traces)), dtype=np.float64)
    dp = C.POINTER(C.c_double)
    lib.kirch_oracle(grad.ctypes.data_as(dp), d64.ctypes.data_as(dp), snum, tnum, dist.ctypes.data_as(dp),
                     tt.ctypes.data_as(dp), float(vel), int(bool(nearfield)),
                     traces.ctypes.data_as(C.POINTER(C.c_int)), len(traces), out.ctypes.data_as(dp))
    return out


def kirchhoff(data, travel_time_us, dist_km, vel=1.69e8, nearfield=False, traces=None):
    """Same contract as mig_oracle.kirchhoff but returns only the requested
    output traces: array (snum, len(traces))."""
    lib = _load()
    data = np.asarray(data)
    snum, tnum = data.shape
    tt = np.ascontiguousarray(np.asarray(travel_time_us) / 1.0e6, dtype=np.float64)
    grad = np.ascontiguousarray(np.gradient(data, tt, axis=0), dtype=np.float64)
    d64 = np.ascontiguousarray(data, dtype=np.float64)
    dist = np.ascontiguousarray(dist_km, dtype=np.float64) * 1.0e3
    if traces is None:
        traces = np.arange(tnum)
    traces = np.ascontiguousarray(traces, dtype=np.int32)
    out = np.zeros((snum, len(traces)), dtype=np.float64)
    dp = C.POINTER(C.c_double)
    lib.kirch_oracle(grad.ctypes.data_as(dp), d64.ctypes.data_as(dp), snum, tnum, dist.ctypes.data_as(dp),
                     tt.ctypes.data_as(dp), float(vel), int(bool(nearfield)),
                     traces.ctypes.data_as(C.POINTER(C.c_int)), len(traces), out.ctypes.data_as(dp))
    return out
