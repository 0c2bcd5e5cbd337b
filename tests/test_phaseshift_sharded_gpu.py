"""Phase shift sharded over the wavenumbers: impdar_phaseshift_tk_dev -> impdar_ps_alltoall_dev ->
impdar_phaseshift_finish_dev.  One GPU here, so the slabs of an N-rank job are computed one after another on the
same device and the all-to-all is carried out on the host following parallel.alltoall_layout (the table the device
exchange follows; its N > 1 movement over a real transport is tests/test_parallel_gloo.py); the one-rank job runs
the device exchange itself, with and without an RCCL communicator.

Tolerances: wavenumber slabs of TK are BIT-EQUAL to the same rows of the unsharded TK (same kernels, same
arithmetic per wavenumber); the finished image differs from the unsharded one only by rocFFT's rounding of the
inverse transform over k (other batch count): float32 relative max 2e-6, float64 1e-13."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_max

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_transform_implementation(monkeypatch):
    """The first call of a power-of-two size runs on the library's own row transforms and later calls on rocFFT's plans
    (csrc/own_fft.h): the bit-for-bit comparisons of this module pin one implementation."""
    monkeypatch.setenv('IMPDAR_PS_FFT', 'own')


def _case(snum, tnum, kind, dtype, seed=3):
    from impdar_amd import synth
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum, dx=2.0)
    data = synth.noise_radargram(snum, tnum, seed=seed).astype(dtype)
    nt = int(2 ** np.ceil(np.log(snum) / np.log(2)))
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    vm = None
    if kind == 'vz':
        vm = np.interp(np.arange(snum), [0, snum // 3, snum // 3 + 6, snum], [1.69e8, 1.69e8, 2.3e8, 2.3e8])
    elif kind == 'vz_runs':            # two long runs of constant velocity: the matrix-core kernel for float32
        vm = np.where(np.arange(snum) < snum // 2, 1.69e8, 2.1e8).astype(np.float64)
    return dict(snum=snum, tnum=tnum, nt=nt, kx=kx, ws=ws, dt=float(geo['dt']), travel_time=geo['travel_time']), data, vm


def _tk(hip, ctx, d_in, g, vm, dtype, k0, nk):
    lib = hip.load()
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    d_tk = hip.DeviceArray(ctx, (max(nk, 1), g['snum']), cdt)
    rc = lib.impdar_phaseshift_tk_dev(ctx, d_in.ptr, hip.dtype_code(dtype), g['snum'], g['tnum'], g['nt'],
                                      hip.as_dp(g['kx'])[1], hip.as_dp(g['ws'])[1], g['dt'],
                                      hip.as_dp(g['travel_time'])[1], 1.69e8 if vm is None else 0.0,
                                      None if vm is None else hip.as_dp(vm)[1], 0 if vm is None else g['snum'],
                                      100.0, 1000.0, k0, nk, d_tk.ptr)
    hip.check(rc, 'impdar_phaseshift_tk_dev')
    tk = d_tk.to_host()[:nk]
    d_tk.free()
    return tk


def _unsharded(hip, ctx, d_in, g, vm, dtype):
    lib = hip.load()
    d_out = hip.DeviceArray(ctx, (g['snum'], g['tnum']), dtype)
    rc = lib.impdar_phaseshift_dev(ctx, d_in.ptr, hip.dtype_code(dtype), g['snum'], g['tnum'], g['nt'],
                                   hip.as_dp(g['kx'])[1], hip.as_dp(g['ws'])[1], g['dt'], hip.as_dp(g['travel_time'])[1],
                                   1.69e8 if vm is None else 0.0, None if vm is None else hip.as_dp(vm)[1],
                                   0 if vm is None else g['snum'], 100.0, 1000.0, d_out.ptr)
    hip.check(rc, 'impdar_phaseshift_dev')
    out = d_out.to_host()
    d_out.free()
    return out


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('snum,tnum,kind,world', [(300, 128, 'const', 3), (300, 100, 'vz', 4), (512, 600, 'vz_runs', 3),
                                                  (1024, 96, 'const', 8), (40, 2, 'vz', 3)])
def test_slabs_and_host_alltoall_give_the_unsharded_image(hip, dtype, snum, tnum, kind, world):
    from impdar_amd import parallel
    lib = hip.load()
    ctx = hip.context()
    g, data, vm = _case(snum, tnum, kind, dtype)
    d_in = hip.DeviceArray.from_host(ctx, data)
    want = _unsharded(hip, ctx, d_in, g, vm, dtype)
    full = _tk(hip, ctx, d_in, g, vm, dtype, 0, tnum)
    k_edges, tau_edges = parallel.slab_edges(tnum, world), parallel.slab_edges(snum, world)
    esz = full.itemsize
    sbufs = []
    for r in range(world):
        slab = _tk(hip, ctx, d_in, g, vm, dtype, k_edges[r], k_edges[r + 1] - k_edges[r])
        assert np.array_equal(slab.view(np.uint8), full[k_edges[r]:k_edges[r + 1]].view(np.uint8)), r
        sbufs.append(np.concatenate([slab[:, tau_edges[s]:tau_edges[s + 1]].ravel() for s in range(world)]).view(np.uint8))
    got = np.empty((snum, tnum), dtype=dtype)
    for r in range(world):
        tw = tau_edges[r + 1] - tau_edges[r]
        rbuf = np.full(tnum * tw * esz, 0xA5, dtype=np.uint8)
        _, recv = parallel.alltoall_layout(tau_edges, k_edges, r, esz)
        for s, roff, rn in recv:
            send_s, _ = parallel.alltoall_layout(tau_edges, k_edges, s, esz)
            _, soff, sn = send_s[r]
            assert sn == rn
            rbuf[roff:roff + rn] = sbufs[s][soff:soff + sn]
        if tw == 0:
            continue
        d_t2 = hip.DeviceArray.from_host(ctx, rbuf.view(full.dtype).reshape(tnum, tw))
        d_rows = hip.DeviceArray(ctx, (tw, tnum), dtype)
        hip.check(lib.impdar_phaseshift_finish_dev(ctx, d_t2.ptr, hip.dtype_code(dtype), tw, tnum, d_rows.ptr), 'finish')
        got[tau_edges[r]:tau_edges[r + 1]] = d_rows.to_host()
        d_t2.free()
        d_rows.free()
    d_in.free()
    tol = 2e-6 if dtype == np.float32 else 1e-13
    assert rel_max(got, want.astype(np.float64)) < tol, rel_max(got, want.astype(np.float64))


@pytest.mark.parametrize('with_comm', [False, True])
@pytest.mark.parametrize('dtype,kind', [(np.float32, 'vz_runs'), (np.float64, 'const')])
def test_one_rank_job_through_the_device_exchange(hip, dtype, kind, with_comm):
    """world = 1: the orchestration of parallel.migrate_phaseshift_sharded with the device all-to-all; with a
    communicator the rank's own block travels through a grouped ncclSend/ncclRecv pair."""
    from impdar_amd import parallel
    lib = hip.load()
    snum, tnum = 512, 320
    g, data, vm = _case(snum, tnum, kind, dtype)
    ctx = hip.context()
    own = None
    if with_comm:
        own = C.c_void_p()
        hip.check(lib.impdar_ctx_create(0, C.byref(own)), 'ctx')
        buf = C.create_string_buffer(hip.UNIQUE_ID_BYTES)
        hip.check(lib.impdar_comm_unique_id(buf), 'unique_id')
        hip.check(lib.impdar_comm_init(own, buf.raw, 0, 1), 'comm_init')
        ctx = own
    try:
        d_in = hip.DeviceArray.from_host(ctx, data)
        want = _unsharded(hip, ctx, d_in, g, vm, dtype)
        d_in.free()
        lo, hi, rows = parallel.migrate_phaseshift_sharded(data, g, 1.69e8 if vm is None else 0.0, vm,
                                                           rdv=parallel.Rendezvous(0, 1), ctx=ctx)
        assert (lo, hi) == (0, snum) and rows.dtype == dtype
        tol = 2e-6 if dtype == np.float32 else 1e-13
        assert rel_max(rows, want.astype(np.float64)) < tol
    finally:
        if own is not None:
            lib.impdar_ctx_destroy(own)


def test_bad_slab_arguments_are_errors(hip):
    lib = hip.load()
    ctx = hip.context()
    g, data, vm = _case(64, 16, 'const', np.float32)
    d_in = hip.DeviceArray.from_host(ctx, data)
    with pytest.raises(ValueError):
        _tk(hip, ctx, d_in, g, vm, np.float32, 10, 7)            # [10, 17) outside [0, 16)
    d_tk = hip.DeviceArray(ctx, (16, 64), np.complex64)
    d_t2 = hip.DeviceArray(ctx, (16, 64), np.complex64)
    ia = C.c_int * 3
    with pytest.raises(ValueError):                              # two ranks, no communicator
        hip.check(lib.impdar_ps_alltoall_dev(ctx, d_tk.ptr, hip.F32, 64, 16, 2, 0, ia(0, 32, 64), ia(0, 8, 16), d_t2.ptr), 'a2a')
    with pytest.raises(ValueError):                              # edges that do not span the axes
        hip.check(lib.impdar_ps_alltoall_dev(ctx, d_tk.ptr, hip.F32, 64, 16, 1, 0, (C.c_int * 2)(0, 60), (C.c_int * 2)(0, 16),
                                             d_t2.ptr), 'a2a')
    for d in (d_in, d_tk, d_t2):
        d.free()


def test_single_process_front_door(hip):
    """parallel.run_sharded_phaseshift: shared-memory job directory, one worker per GPU (one here), float64 result
    like migrationPhaseShift; and migrationPhaseShift(dat, ngpus=1) stays the ordinary path."""
    from impdar_amd import parallel, synth
    from impdar_amd.lib import migrationlib
    from conftest import make_dat
    snum, tnum = 300, 90
    geo = synth.geometry(snum, tnum, dx=2.0)
    for dtype, vel in ((np.float32, 1.69e8), (np.float64, np.array([[1.69e8, 0.0], [2.0e8, 100.0], [2.2e8, 400.0]]))):
        data = synth.noise_radargram(snum, tnum, seed=12).astype(dtype)
        gg = dict(geo, data=data)
        dat = make_dat(gg)
        migrationlib.migrationPhaseShift(dat, vel=vel, ngpus=1)
        want = dat.data
        dat = make_dat(gg)
        os.environ['IMPDAR_NGPUS'] = '0'
        nt = 512
        from impdar_amd.lib.migrationlib import mig_hip
        kx = mig_hip._kx(dat)
        ws = 2. * np.pi * np.fft.fftfreq(nt, d=dat.dt)
        vmig = migrationlib.getVelocityProfile(dat, vel)
        vconst, vm = (float(vmig), None) if not hasattr(vmig, '__len__') else (0.0, vmig)
        got = parallel.run_sharded_phaseshift(data, nt, kx, ws, dat.dt, dat.travel_time, vconst, vm, ngpus=1)
        assert got.dtype == np.float64 and got.shape == (snum, tnum)
        assert rel_max(got, want) < (2e-6 if dtype == np.float32 else 1e-13)
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('impdar_shard_')]
