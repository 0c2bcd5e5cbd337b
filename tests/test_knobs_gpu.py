"""Every environment variable the library reads has a GPU test that sets it (the table is in INTEGRATION.md,
section "Environment variables"; tests/test_abi.py::test_every_knob_is_documented_and_tested keeps the three in step).
This file holds the ones no other test sets: the slot reserve of multi-rank plans, the emulation of one rank of an
N-rank plan on a single GPU, the strided transform plans of the phase shift, the four 1-D passes of Stolt, and the
JSON metrics line of the four migration entry points."""
import ctypes as C
import json

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _dat(data, geo):
    from impdar_amd.lib.RadarData import RadarData
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = data.copy(), data.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    return d


def test_metrics_line_of_the_four_migration_entry_points(hip, monkeypatch, capfd):
    """IMPDAR_METRICS: one JSON line per migration on stderr beside the reference's 'complete in N seconds' print
    (SURVEY section 5): entry point, kernel, kernel / device milliseconds from HIP events, sizes, traces per second,
    device count."""
    from impdar_amd import synth
    snum, tnum = 300, 140
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=4).astype(np.float32)
    monkeypatch.setenv('IMPDAR_METRICS', '1')
    capfd.readouterr()
    _dat(x, geo).migrate('kirch', vel=1.69e8)
    _dat(x, geo).migrate('stolt', vel=1.69e8)
    _dat(x, geo).migrate('phsh', vel=1.69e8)
    _dat(x, geo).migrate('tk', vel=1.69e8)
    lines = [json.loads(l) for l in capfd.readouterr().err.splitlines() if l.startswith('{') and '"impdar_metrics"' in l]
    assert [m['entry'] for m in lines] == ['impdar_kirchhoff', 'impdar_stolt', 'impdar_phaseshift', 'impdar_taper']
    assert lines[0]['kernel'] == 'kirch_quad_kernel' and lines[0]['plan'] in ('new', 'cached') and lines[0]['kernel_ms'] > 0
    assert lines[1]['device_ms'] > 0 and 'stolt_stretch' in lines[1]['kernel']
    assert lines[2]['kernel'].startswith('ps_') and lines[2]['kernel_ms'] > 0 and lines[2]['device_ms'] >= lines[2]['kernel_ms']
    for m in lines:
        assert (m['snum'], m['tnum'], m['devices']) == (snum, tnum, 1) and m['traces_per_s'] > 0 and m['wall_s'] > 0
    monkeypatch.delenv('IMPDAR_METRICS')
    _dat(x, geo).migrate('stolt', vel=1.69e8)
    assert '"impdar_metrics"' not in capfd.readouterr().err


def test_trace_lines_of_a_first_call(hip, monkeypatch, capfd):
    """IMPDAR_TRACE=1: the library prints its milestones with the time since its first trace point -- plan creation
    per rocFFT plan, uploads, the kernels enqueued, the download: the breakdown behind profiles/r05_first_call.txt."""
    import re
    from impdar_amd import synth
    snum, tnum = 212, 76                       # a size no other test uses: the plans are new
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=4).astype(np.float32)
    monkeypatch.setenv('IMPDAR_TRACE', '1')
    capfd.readouterr()
    _dat(x, geo).migrate('stolt', vel=1.69e8)
    _dat(x, geo).migrate('phsh', vel=1.69e8)
    err = capfd.readouterr().err
    lines = re.findall(r'^\[impdar \+ *([0-9.]+) ms, unix [0-9.]+\] (.*)$', err, flags=re.M)
    text = [t for _, t in lines]
    assert any(t.startswith('rocfft_plan_create 2-D') and t.endswith('done') for t in text), err
    assert any(t.startswith('stolt: all kernels enqueued') for t in text) and any('phaseshift: plans ready' in t for t in text), err
    stamps = [float(ms) for ms, _ in lines]
    assert stamps and stamps[-1] >= stamps[0]
    monkeypatch.delenv('IMPDAR_TRACE')
    _dat(x, geo).migrate('stolt', vel=1.69e8)
    assert '[impdar +' not in capfd.readouterr().err


def test_phase_shift_strided_transform_plans(hip, monkeypatch):
    """IMPDAR_PS_FFT=strided: the transforms over the traces / wavenumbers as rocFFT's strided plans on the arrays as
    they lie (rounds 1-3a) instead of transpose + contiguous plan: same image to the transforms' rounding."""
    from impdar_amd import synth
    from oracle import mig_oracle
    snum, tnum = 260, 96
    geo = synth.geometry(snum, tnum)
    for dtype, tol in ((np.float32, 2e-4), (np.float64, 1e-10)):
        x = synth.noise_radargram(snum, tnum, seed=6).astype(dtype)
        want = mig_oracle.phase_shift(x.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'],
                                      1.69e8, 10, 20)
        outs = []
        for fft in (None, 'strided'):
            if fft:
                monkeypatch.setenv('IMPDAR_PS_FFT', fft)
            else:
                monkeypatch.delenv('IMPDAR_PS_FFT', raising=False)
            d = _dat(x, geo)
            d.migrate('phsh', vel=1.69e8, htaper=10, vtaper=20)
            outs.append(d.data)
            assert rel_l2(d.data, want) < tol, (dtype, fft, rel_l2(d.data, want))
        assert rel_l2(outs[0], outs[1]) < tol
    monkeypatch.delenv('IMPDAR_PS_FFT', raising=False)


def test_stolt_four_one_dimensional_passes(hip, monkeypatch):
    """IMPDAR_STOLT_FFT=1d: R2C over time, C2C over the traces, and back, as four 1-D rocFFT plans instead of the two
    2-D real plans: same image to the transforms' rounding."""
    from impdar_amd import synth
    from oracle import mig_oracle
    snum, tnum = 250, 97
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=8)
    want = mig_oracle.stolt(x, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 10, 20)
    outs = []
    for fft in (None, '1d'):
        if fft:
            monkeypatch.setenv('IMPDAR_STOLT_FFT', fft)
        else:
            monkeypatch.delenv('IMPDAR_STOLT_FFT', raising=False)
        d = _dat(x, geo)
        d.migrate('stolt', vel=1.68e8, htaper=10, vtaper=20)
        outs.append(d.data)
        assert rel_l2(d.data, want) < 1e-10, (fft, rel_l2(d.data, want))
    monkeypatch.delenv('IMPDAR_STOLT_FFT', raising=False)
    # (the plan is rebuilt when the sizes change: leave the default behind for the tests that follow)
    _dat(x[:, :60], {k: (v[:60] if np.ndim(v) == 1 and len(v) == tnum else v) for k, v in geo.items()}).migrate('stolt', vel=1.68e8)


def test_slot_reserve_and_rank_emulation(hip, monkeypatch):
    """IMPDAR_KIRCH_RESERVE=<R>: the persistent ring kernels leave R workgroup slots free (32 by default in a
    multi-rank plan, so that RCCL's kernels run underneath the diffraction sum).  IMPDAR_COMM_EMULATE=1: a plan built
    for N ranks driven over a 1-rank communicator -- one rank of an N-rank run on a single GPU, its exchange a self
    send/recv (profiles/tools/exchange_overlap.py).  The output block of the emulated rank equals the same columns
    of the one-rank run bit for bit, whatever the reserve."""
    from impdar_amd import _hip, synth
    from impdar_amd.kirchhoff import KirchhoffPlan, migrate_resident
    lib = hip.load()
    snum, tnum, nranks = 1100, 2048, 4
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=9).astype(np.float32)
    want, _, _ = migrate_resident(hip.context(), x, geo['dist'], geo['travel_time'], mode='fast')
    ctx = C.c_void_p()
    hip.check(lib.impdar_ctx_create(0, C.byref(ctx)), 'ctx')
    buf = C.create_string_buffer(hip.UNIQUE_ID_BYTES)
    hip.check(lib.impdar_comm_unique_id(buf), 'unique_id')
    hip.check(lib.impdar_comm_init(ctx, buf.raw, 0, 1), 'comm_init')
    xlo, xhi = 512, 1024
    try:
        plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], mode='fast', nranks=nranks)
        with pytest.raises(ValueError):
            plan.allgather()                   # a 4-rank plan on a 1-rank communicator: refused ...
        plan.destroy()
        monkeypatch.setenv('IMPDAR_COMM_EMULATE', '1')      # ... unless the emulation is asked for
        for reserve in ('0', '48'):
            monkeypatch.setenv('IMPDAR_KIRCH_RESERVE', reserve)
            plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], mode='fast', nranks=nranks)
            d_in = _hip.DeviceArray.from_host(ctx, x)
            d_out = _hip.DeviceArray(ctx, (snum, xhi - xlo), np.float32)
            plan.prep(d_in, tnum, 0, tnum)
            plan.allgather()
            plan.migrate(d_out, xlo, xhi)
            plan.sync()
            got = d_out.to_host()
            plan.destroy()
            d_in.free()
            d_out.free()
            # (plans of 4+ ranks sum every walk in two pieces: equal to rounding, not bit for bit)
            assert rel_l2(got, want[:, xlo:xhi]) < 1e-6, (reserve, rel_l2(got, want[:, xlo:xhi]))
    finally:
        monkeypatch.delenv('IMPDAR_COMM_EMULATE', raising=False)
        monkeypatch.delenv('IMPDAR_KIRCH_RESERVE', raising=False)
        lib.impdar_ctx_destroy(ctx)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('snum,tnum', [(256, 128), (1024, 512), (2048, 64)])
def test_own_row_transforms_carry_stolt_and_phase_shift(hip, monkeypatch, snum, tnum, dtype):
    """IMPDAR_STOLT_FFT=own / IMPDAR_PS_FFT=own: every call (not only the first of a size) runs its transforms on the
    library's own power-of-two row kernels (csrc/own_fft.h) -- what a first call does by itself while a thread makes the
    rocFFT plans.  Against the oracle at the stated bars, and against the rocFFT form of the same call."""
    import ctypes as C
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    x = (synth.noise_radargram(snum, tnum, seed=snum + tnum) + 0.25).astype(dtype)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    want_s = mig_oracle.stolt(x.astype(np.float64), geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 10, 12)
    want_p = mig_oracle.phase_shift(x.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], tab, 10, 12)
    outs = {}
    for own in (True, False):
        for k in ('IMPDAR_STOLT_FFT', 'IMPDAR_PS_FFT'):
            monkeypatch.delenv(k, raising=False)
        if own:
            monkeypatch.setenv('IMPDAR_STOLT_FFT', 'own')
            monkeypatch.setenv('IMPDAR_PS_FFT', 'own')
        else:
            monkeypatch.setenv('IMPDAR_STOLT_FFT', '1d')     # (the synchronous rocFFT plans)
            monkeypatch.setenv('IMPDAR_PS_FFT', 'strided')
        d = _dat(x, geo)
        d.migrate('stolt', vel=1.68e8, htaper=10, vtaper=12)
        buf = C.create_string_buffer(1024)
        _hip.check(hip.load().impdar_ctx_last_metrics(hip.context(), buf, len(buf)), 'metrics')
        assert ('own row transforms' in json.loads(buf.value.decode())['kernel']) == own
        outs['stolt', own] = d.data
        d = _dat(x, geo)
        d.migrate('phsh', vel=tab, htaper=10, vtaper=12)
        _hip.check(hip.load().impdar_ctx_last_metrics(hip.context(), buf, len(buf)), 'metrics')
        assert (json.loads(buf.value.decode())['transforms'] == 'own') == own
        outs['phsh', own] = d.data
    for own in (True, False):
        if dtype == np.float32:
            assert rel_l2(outs['stolt', own], want_s) < 1e-4 and rel_l2(outs['phsh', own], want_p) < 2e-4
        else:
            assert np.max(np.abs(outs['stolt', own] - want_s)) < 1e-12 * np.max(np.abs(want_s))
            assert np.max(np.abs(outs['phsh', own] - want_p)) < 1e-10 * np.max(np.abs(want_p))
    assert rel_l2(outs['stolt', True], outs['stolt', False]) < (2e-6 if dtype == np.float32 else 1e-13)
    assert rel_l2(outs['phsh', True], outs['phsh', False]) < (5e-6 if dtype == np.float32 else 1e-12)


def test_every_call_runs_on_the_own_transforms_and_rocfft_on_request(hip, monkeypatch):
    """A power-of-two size: every call runs on the library's own transforms (nothing to compile, no plan); rocFFT's plans on
    request.  Same image either way (to the transforms' rounding)."""
    import ctypes as C
    from impdar_amd import _hip, synth
    for k in ('IMPDAR_STOLT_FFT', 'IMPDAR_PS_FFT'):
        monkeypatch.delenv(k, raising=False)
    snum, tnum = 512, 1024                      # (no other test uses this size)
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=77).astype(np.float32)
    buf = C.create_string_buffer(1024)

    def call(mtype):
        d = _dat(x, geo)
        d.migrate(mtype, vel=1.69e8, htaper=10, vtaper=12)
        _hip.check(hip.load().impdar_ctx_last_metrics(hip.context(), buf, len(buf)), 'metrics')
        m = json.loads(buf.value.decode())
        return d.data, ('own' if ('own row transforms' in m['kernel'] or m.get('transforms') == 'own') else 'rocfft')

    for mtype in ('stolt', 'phsh'):
        first, how1 = call(mtype)
        second, how2 = call(mtype)
        third, how3 = call(mtype)
        # (power-of-two sizes: the own transforms on every call since round 5; IMPDAR_STOLT_FFT / IMPDAR_PS_FFT = rocfft ask for the plans)
        assert (how1, how2, how3) == ('own', 'own', 'own'), (mtype, how1, how2, how3)
        assert np.array_equal(second, first) and np.array_equal(second, third)
        knob = 'IMPDAR_STOLT_FFT' if mtype == 'stolt' else 'IMPDAR_PS_FFT'
        monkeypatch.setenv(knob, 'rocfft')
        lib_form, how = call(mtype)
        monkeypatch.delenv(knob)
        assert how == 'rocfft' and rel_l2(lib_form, third) < 5e-6, (how, rel_l2(lib_form, third))
