"""ctypes binding of ``libimpdar_hip.so`` (the C ABI in ``include/impdar_hip.h``).

There is deliberately no CPU fallback: if the library is missing or no GPU
is visible the migration entry points raise.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IMPDAR_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libimpdar_hip.so')

F32, F64 = 0, 1
KIRCH_AUTO, KIRCH_EXACT, KIRCH_FAST = 0, 1, 2
UNIQUE_ID_BYTES = 128

ERR_ARG, ERR_HIP, ERR_FFT, ERR_COMM, ERR_NODEV, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6


class HipUnavailableError(RuntimeError):
    """The HIP extension could not be loaded or no GPU is usable."""


_lib = None
_lock = threading.Lock()
_ctx = {}

_p = C.c_void_p
_dp = C.POINTER(C.c_double)
_i = C.c_int
_d = C.c_double
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol of include/impdar_hip.h
SIGNATURES = {
    'impdar_last_error': (C.c_char_p, []),
    'impdar_device_count': (_i, []),
    'impdar_ctx_create': (_i, [_i, C.POINTER(_p)]),
    'impdar_ctx_destroy': (None, [_p]),
    'impdar_ctx_sync': (_i, [_p]),
    'impdar_ctx_last_ms': (_i, [_p, C.POINTER(C.c_float)]),
    'impdar_ctx_last_kernel_ms': (_i, [_p, C.POINTER(C.c_float)]),
    'impdar_ctx_last_metrics': (_i, [_p, C.c_char_p, C.c_size_t]),
    'impdar_dev_alloc': (_i, [_p, C.c_size_t, C.POINTER(_p)]),
    'impdar_dev_free': (_i, [_p, _p]),
    'impdar_dev_upload': (_i, [_p, _p, _p, C.c_size_t]),
    'impdar_dev_download': (_i, [_p, _p, _p, C.c_size_t]),
    'impdar_dev_memset': (_i, [_p, _p, _i, C.c_size_t]),
    'impdar_dev_download_f64': (_i, [_p, _dp, _p, _i, C.c_size_t]),
    'mig_kirch_loop': (None, [_dp, _i, _i, _dp, _dp, _dp, _dp, _d, _dp, _d, _i]),
    'impdar_kirchhoff': (_i, [_p, _p, _i, _i, _i, _dp, _dp, _d, _i, _i, _d, _dp, _dp, _dp, _i, _dp]),
    'impdar_kirch_plan_create': (_i, [_p, _i, _i, _i, _dp, _dp, _d, _i, _i, _d, _dp, _dp, _dp, _i, _i,
                                      C.POINTER(_p)]),
    'impdar_kirch_plan_destroy': (None, [_p]),
    'impdar_kirch_plan_mode': (_i, [_p]),
    'impdar_kirch_plan_tnum_pad': (_i, [_p]),
    'impdar_kirch_plan_xnoise': (_d, [_p]),
    'impdar_kirch_plan_kernel': (_i, [_p]),
    'impdar_kirch_prep': (_i, [_p, _p, _i, _i, _i]),
    'impdar_kirch_allgather': (_i, [_p]),
    'impdar_kirch_exchange': (_i, [_p, _i, _ip, _ip, _ip, _i, _ip, _ip, _ip]),
    'impdar_kirch_migrate': (_i, [_p, _p, _i, _i]),
    'impdar_kirch_last_ms': (_i, [_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'impdar_kirch_history_ms': (_i, [_p, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'impdar_kirch_count_pairs': (C.c_longlong, [_p, _i, _i]),
    'impdar_stolt': (_i, [_p, _p, _i, _i, _i, _dp, _dp, _d, _d, _d, _p]),
    'impdar_stolt_dev': (_i, [_p, _p, _i, _i, _i, _dp, _dp, _d, _d, _d, _p]),
    'impdar_fft_rows_dev': (_i, [_p, _i, _i, _i, _i, _p, _p, _d]),
    'impdar_phaseshift': (_i, [_p, _p, _i, _i, _i, _i, _dp, _dp, _d, _dp, _d, _dp, _i, _d, _d, _p]),
    'impdar_phaseshift_dev': (_i, [_p, _p, _i, _i, _i, _i, _dp, _dp, _d, _dp, _d, _dp, _i, _d, _d, _p]),
    'impdar_phaseshift_tk_dev': (_i, [_p, _p, _i, _i, _i, _i, _dp, _dp, _d, _dp, _d, _dp, _i, _d, _d, _i, _i, _p]),
    'impdar_ps_alltoall_dev': (_i, [_p, _p, _i, _i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _p]),
    'impdar_phaseshift_finish_dev': (_i, [_p, _p, _i, _i, _i, _p]),
    'impdar_phaseshift_ffd': (_i, [_p, _dp, _i, _i, _i, _dp, _dp, _d, _dp, _dp, _d, _d, _d, _dp]),
    'impdar_taper': (_i, [_p, _p, _i, _i, _i, _d, _d]),
    'impdar_filtfilt': (_i, [_p, _p, _i, _i, _i, _dp, _dp, _i, _dp]),
    'impdar_filtfilt_dev': (_i, [_p, _p, _i, _i, _i, _dp, _dp, _i, _dp]),
    'impdar_fir_shift': (_i, [_p, _p, _i, _i, _i, _dp, _i]),
    'impdar_fir_shift_dev': (_i, [_p, _p, _i, _i, _i, _dp, _i]),
    'impdar_trace_lerp': (_i, [_p, _p, _i, _i, _i, _ip, _ip, _dp, _dp, _i, _p]),
    'impdar_cast_dev': (_i, [_p, _p, _i, _p, _i, C.c_size_t]),
    'impdar_trace_lerp_dev': (_i, [_p, _p, _i, _i, _i, _ip, _ip, _dp, _dp, _i, _p]),
    'impdar_comm_unique_id': (_i, [C.c_char_p]),
    'impdar_comm_init': (_i, [_p, C.c_char_p, _i, _i]),
    'impdar_comm_rank': (_i, [_p]),
    'impdar_comm_size': (_i, [_p]),
    'impdar_comm_info': (_i, [_p, _ip, _ip, _ip, _ip]),
    'impdar_comm_barrier': (_i, [_p]),
}


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailableError(
                'HIP extension %s not built; run `python -c "import __graft_entry__ as g; g.build()"` '
                'or `python -m impdar_amd.build`' % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as exc:
            raise HipUnavailableError('cannot load %s: %s' % (LIB_PATH, exc))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def mapped_rccl():
    """Paths of every librccl mapped into this process (there must be one, and it must be the ROCm install's:
    a torch wheel ships its own older librccl.so under the same SONAME, and whichever is loaded first serves
    both)."""
    paths = set()
    try:
        with open('/proc/self/maps') as fi:
            for line in fi:
                if 'librccl' in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def require_system_rccl():
    load()
    foreign = [p for p in mapped_rccl() if not os.path.realpath(p).startswith(os.path.realpath('/opt/rocm') + os.sep)
               and '/opt/rocm' not in p]
    if foreign and os.environ.get('IMPDAR_ALLOW_FOREIGN_RCCL') != '1':
        raise HipUnavailableError(
            'librccl was already mapped from %s before libimpdar_hip.so was loaded (import torch after '
            'impdar_amd._hip.load(), or not at all): the library is linked against /opt/rocm\'s RCCL; set '
            'IMPDAR_ALLOW_FOREIGN_RCCL=1 to run anyway' % ', '.join(foreign))


def last_error():
    msg = load().impdar_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(rc, what=''):
    """Map a C status to the exception type the reference raises."""
    if rc == 0:
        return
    msg = '%s%s' % (what + ': ' if what else '', last_error())
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NODEV:
        raise HipUnavailableError(msg)
    raise RuntimeError('impdar HIP error %d: %s' % (rc, msg))


def device_count():
    return load().impdar_device_count()


def context(device=None):
    """Per-process context for ``device`` (default: $IMPDAR_DEVICE, else $LOCAL_RANK, else 0).  A rank whose
    device does not exist is an error -- two ranks must never end up sharing GPU 0 silently."""
    lib = load()
    if device is None:
        device = int(os.environ.get('IMPDAR_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        ndev = lib.impdar_device_count()
        if device < 0 or device >= max(ndev, 1):
            raise HipUnavailableError('rank wants GPU %d but only %d device(s) are visible (LOCAL_RANK / '
                                      'IMPDAR_DEVICE / IMPDAR_NGPUS larger than the node)' % (device, ndev))
    with _lock:
        if device in _ctx:
            return _ctx[device]
    h = _p()
    check(lib.impdar_ctx_create(device, C.byref(h)), 'impdar_ctx_create')
    with _lock:
        _ctx[device] = h
    return h


def as_dp(a):
    """float64 C-contiguous array -> (array, double*)."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError('device kernels take float32 or float64 data, got %s' % dt)


class DeviceArray(object):
    """A (rows, cols) row-major array resident in HBM."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = _p()
        check(load().impdar_dev_alloc(ctx, self.nbytes, C.byref(self.ptr)), 'impdar_dev_alloc')

    @classmethod
    def from_host(cls, ctx, a):
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.shape, a.dtype)
        check(load().impdar_dev_upload(ctx, d.ptr, a.ctypes.data_as(_p), d.nbytes), 'impdar_dev_upload')
        return d

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(load().impdar_dev_download(self.ctx, out.ctypes.data_as(_p), self.ptr, self.nbytes),
              'impdar_dev_download')
        return out

    def to_host_f64(self):
        """float64 host copy (widened on several host threads when the array is float32)."""
        out = np.empty(self.shape, dtype=np.float64)
        n = int(np.prod(self.shape))
        check(load().impdar_dev_download_f64(self.ctx, out.ctypes.data_as(_dp), self.ptr, dtype_code(self.dtype), n),
              'impdar_dev_download_f64')
        return out

    def free(self):
        if self.ptr:
            load().impdar_dev_free(self.ctx, self.ptr)
            self.ptr = _p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
