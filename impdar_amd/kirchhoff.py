"""Resident (device-side) Kirchhoff plan: what ``bench.py`` and the multi-GPU
path drive.  Thin object wrapper over the ``impdar_kirch_*`` C ABI."""
import ctypes as C

import numpy as np

from . import _hip
from .lib.migrationlib.mig_hip import gradient_coefficients

MODE_NAMES = {_hip.KIRCH_EXACT: 'exact', _hip.KIRCH_FAST: 'fast'}
KERNEL_NAMES = {0: 'kirch_exact_kernel', 1: 'kirch_exact_tab_kernel', 2: 'kirch_dquad_kernel', 3: 'kirch_quad_kernel',
                4: 'kirch_tab_kernel', 5: 'kirch_gen_kernel'}


class KirchhoffPlan(object):
    def __init__(self, ctx, dtype, snum, tnum, dist_km, travel_time_us, vel=1.69e8, nearfield=False,
                 mode='auto', nranks=1):
        self.lib = _hip.load()
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.snum, self.tnum = int(snum), int(tnum)
        tt_sec = np.ascontiguousarray(np.asarray(travel_time_us) / 1.0e6, dtype=np.float64)
        dist = np.ascontiguousarray(dist_km, dtype=np.float64) * 1.0e3
        uniform, h, ga, gb, gc = gradient_coefficients(tt_sec)
        modes = {'auto': _hip.KIRCH_AUTO, 'exact': _hip.KIRCH_EXACT, 'fast': _hip.KIRCH_FAST}
        self._keep = [tt_sec, dist, ga, gb, gc]
        self.h = _hip._p()
        rc = self.lib.impdar_kirch_plan_create(
            ctx, _hip.dtype_code(self.dtype), self.snum, self.tnum, _hip.as_dp(dist)[1], _hip.as_dp(tt_sec)[1],
            float(vel), int(bool(nearfield)), int(uniform), h, _hip.as_dp(ga)[1], _hip.as_dp(gb)[1],
            _hip.as_dp(gc)[1], modes[mode], int(nranks), C.byref(self.h))
        _hip.check(rc, 'impdar_kirch_plan_create')
        self.mode = MODE_NAMES[self.lib.impdar_kirch_plan_mode(self.h)]
        self.tnum_pad = self.lib.impdar_kirch_plan_tnum_pad(self.h)
        self.kernel = KERNEL_NAMES[self.lib.impdar_kirch_plan_kernel(self.h)]
        # position noise of the profile in units of dx; the float64 table-driven kernels meet max(1e-12, 0.1 xnoise)
        self.xnoise = float(self.lib.impdar_kirch_plan_xnoise(self.h))

    def prep(self, d_data, ld, jlo, nloc):
        """Gradient + transpose of a local column block (device array)."""
        ptr = d_data.ptr if hasattr(d_data, 'ptr') else d_data
        _hip.check(self.lib.impdar_kirch_prep(self.h, ptr, int(ld), int(jlo), int(nloc)), 'impdar_kirch_prep')

    def allgather(self):
        _hip.check(self.lib.impdar_kirch_allgather(self.h), 'impdar_kirch_allgather')

    def exchange(self, send, recv):
        """Halo exchange: ``send`` / ``recv`` are lists of (peer, row_lo, row_hi) (``parallel.plan_exchange``)."""
        def cols(lst):
            a = np.ascontiguousarray(np.asarray(lst, dtype=np.int32).reshape(-1, 3).T)
            ip = C.POINTER(C.c_int)
            return a, len(lst), a[0].ctypes.data_as(ip), a[1].ctypes.data_as(ip), a[2].ctypes.data_as(ip)
        sa, ns, sp, sl, sh = cols(send)
        ra, nr, rp, rl, rh = cols(recv)
        _hip.check(self.lib.impdar_kirch_exchange(self.h, ns, sp, sl, sh, nr, rp, rl, rh), 'impdar_kirch_exchange')

    def migrate(self, d_out, xlo, xhi):
        ptr = d_out.ptr if hasattr(d_out, 'ptr') else d_out
        _hip.check(self.lib.impdar_kirch_migrate(self.h, ptr, int(xlo), int(xhi)), 'impdar_kirch_migrate')

    def last_ms(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _hip.check(self.lib.impdar_kirch_last_ms(self.h, C.byref(a), C.byref(b), C.byref(c)),
                   'impdar_kirch_last_ms')
        return a.value, b.value, c.value

    def history_ms(self, back):
        """(prep, allgather, migrate) HIP-event ms of the step ``back`` steps ago."""
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _hip.check(self.lib.impdar_kirch_history_ms(self.h, int(back), C.byref(a), C.byref(b), C.byref(c)),
                   'impdar_kirch_history_ms')
        return a.value, b.value, c.value

    def count_pairs(self, xlo, xhi):
        return int(self.lib.impdar_kirch_count_pairs(self.h, int(xlo), int(xhi)))

    def sync(self):
        _hip.check(self.lib.impdar_ctx_sync(self.ctx), 'impdar_ctx_sync')

    def destroy(self):
        if self.h:
            self.lib.impdar_kirch_plan_destroy(self.h)
            self.h = _hip._p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def migrate_resident(ctx, data, dist_km, travel_time_us, vel=1.69e8, nearfield=False, mode='auto'):
    """Upload, migrate every trace, download (element type = data dtype)."""
    data = np.ascontiguousarray(data)
    snum, tnum = data.shape
    plan = KirchhoffPlan(ctx, data.dtype, snum, tnum, dist_km, travel_time_us, vel, nearfield, mode)
    d_in = _hip.DeviceArray.from_host(ctx, data)
    d_out = _hip.DeviceArray(ctx, (snum, tnum), data.dtype)
    plan.prep(d_in, tnum, 0, tnum)
    plan.migrate(d_out, 0, tnum)
    plan.sync()
    out = d_out.to_host()
    ms = plan.last_ms()
    plan.destroy()
    d_in.free()
    d_out.free()
    return out, plan.mode, ms
