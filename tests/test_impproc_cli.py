"""`impproc migrate` surface (reference test/test_impproc.py:432-561, 603-604)
and the .mat round trip (reference test/test_RadarDataSaving.py).  CPU-only:
the migration call is patched; the GPU variant is in test_cli_gpu.py."""
import os
import sys
from unittest.mock import MagicMock, patch

import numpy as np
import pytest

from conftest import GOLDEN
from impdar_amd.bin import impproc
from impdar_amd.lib.NoInitRadarData import NoInitRadarData
from impdar_amd.lib.RadarData import RadarData


def run_cli(argv, loaded):
    with patch.object(sys, 'argv', ['impproc'] + argv), \
            patch('impdar_amd.bin.impproc.load', return_value=loaded) as ld:
        impproc.main()
    return ld


@pytest.mark.parametrize('mtype', ['stolt', 'kirch', 'phsh', 'tk', 'sumigtk', 'sustolt', 'sumigffd'])
def test_migrate_types(mtype):
    dat = MagicMock()
    run_cli(['migrate', '--mtype', mtype, 'dummy.mat'], [dat])
    args, kwargs = dat.migrate.call_args
    assert args == (mtype,)
    assert kwargs == dict(vel=1.69e8, vtaper=1000, htaper=100, tmig=0, verbose=1, vel_fn=None, nxpad=100,
                          nearfield=False)
    dat.save.assert_called_with('dummy_migrated.mat')


def test_migrate_option_types_and_naming(tmp_path):
    dat = MagicMock()
    run_cli(['migrate', '--mtype', 'kirch', '--vel', '1.5e8', '--nearfield', '--htaper', '7', '--vtaper', '9',
             '--nxpad', '3', '--tmig', '2', '--verbose', '0', '--vel_fn', 'v.txt', 'line_raw.mat'], [dat])
    _, kw = dat.migrate.call_args
    assert kw['vel'] == 1.5e8 and kw['nearfield'] is True and kw['vel_fn'] == 'v.txt'
    for k, v in dict(htaper=7, vtaper=9, nxpad=3, tmig=2, verbose=0).items():
        assert kw[k] == v and isinstance(kw[k], int)
    dat.save.assert_called_with('line_migrated.mat')          # _raw stripped
    a, b = MagicMock(), MagicMock()
    run_cli(['migrate', '-o', str(tmp_path) + '/', 'x_raw.mat', 'y.mat'], [a, b])
    a.save.assert_called_with(os.path.join(str(tmp_path) + '/', 'x_migrated.mat'))
    b.save.assert_called_with(os.path.join(str(tmp_path) + '/', 'y_migrated.mat'))
    c = MagicMock()
    run_cli(['migrate', '-o', 'out.mat', 'x.mat'], [c])
    c.save.assert_called_with('out.mat')
    with pytest.raises(SystemExit):
        run_cli(['migrate', '--mtype', 'bad', 'x.mat'], [MagicMock()])
    with pytest.raises(SystemExit):
        run_cli(['migrate', '--htaper', '1.5', 'x.mat'], [MagicMock()])


def test_mig_defaults():
    dat = MagicMock()
    impproc.mig(dat)
    dat.migrate.assert_called_with('stolt', vel=1.69e8, vtaper=100, htaper=100, tmig=0, verbose=0, vel_fn=None,
                                   nxpad=1, nearfield=False)


def make_saveable(dtype=np.float64):
    d = NoInitRadarData(big=True)
    rng = np.random.default_rng(0)
    d.data = (rng.standard_normal((10, 20)) * 100).astype(dtype)
    d.fn = 'x.mat'
    return d


def test_mat_round_trip(tmp_path):
    d = make_saveable()
    d.flags.mig = 'kirch'
    fn = str(tmp_path / 'a.mat')
    d.save(fn)
    r = RadarData(fn)
    assert np.array_equal(r.data, d.data) and r.data.dtype == np.float64
    assert (r.snum, r.tnum) == (10, 20)
    assert np.array_equal(r.travel_time, d.travel_time)
    assert np.array_equal(r.dist, d.dist)
    assert r.flags.mig == 'kirch' and r.flags.bpass.shape == (3,)
    assert r.data_dtype == np.float64
    r.check_attrs()


def test_save_casts_back_to_file_dtype(tmp_path):
    """_RadarDataSaving.py:60-77: float results are cast back to the dtype the
    file was loaded with; NaNs in an int16 file force float16."""
    d = make_saveable(np.int16)
    fn = str(tmp_path / 'i.mat')
    d.save(fn)
    r = RadarData(fn)
    assert r.data_dtype == np.int16
    r.data = r.data.astype(np.float64) * 0.5 + 0.25           # "migrated"
    fn2 = str(tmp_path / 'j.mat')
    r.save(fn2)
    r2 = RadarData(fn2)
    assert r2.data.dtype == np.int16
    assert np.array_equal(r2.data, (d.data.astype(np.float64) * 0.5 + 0.25).astype(np.int16))
    r.data[0, 0] = np.nan
    r.save(fn2)
    back = RadarData(fn2).data        # scipy stores float16 as double; the values went through float16
    assert np.isnan(back[0, 0]) and np.array_equal(back[1:], r.data[1:].astype(np.float16).astype(np.float64))


def test_load_rejects_other_formats():
    from impdar_amd.lib.load import load
    with pytest.raises(ValueError):
        load('gssi', ['x.DZT'])
    with pytest.raises(KeyError):
        from scipy.io import savemat
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            fn = os.path.join(td, 'bad.mat')
            savemat(fn, {'nothing': np.zeros(3)})
            RadarData(fn)


def test_process_migrate_hook_always_stolt():
    """reference test/test_process.py:243-246 and lib/process.py:190-193."""
    from impdar_amd.lib import process
    d = MagicMock()
    assert process.process([d], migrate='kirch') is True
    d.migrate.assert_called_with(mtype='stolt')
    assert process.process([MagicMock()]) is False
    with pytest.raises(NotImplementedError):
        process.process([d], hfilt=(1, 2))


def test_process_chain_order_and_argument_checks():
    """reference lib/process.py:126-135 (argument checks) and :151-193 (vbp, then interp, then migrate);
    test/test_process.py:239-241 (vbp forwards its tuple)."""
    from impdar_amd.lib import process
    d = MagicMock()
    assert process.process([d], vbp=(3, 4)) is True
    d.vertical_band_pass.assert_called_with(3, 4)
    d = MagicMock()
    assert process.process([d], interp=(2.5, None), vbp=(1., 20.), migrate='x') is True
    names = [c[0] for c in d.method_calls]
    # more than one step on float data: uploaded once, resident until the last step is done
    assert names == ['to_device', 'vertical_band_pass', 'constant_space', 'migrate', 'from_device']
    d = MagicMock()
    d.data = np.zeros((4, 4), dtype=np.int16)                 # integer data goes step by step through host buffers
    process.process([d], interp=(2.5, None), vbp=(1., 20.))
    assert [c[0] for c in d.method_calls] == ['vertical_band_pass', 'constant_space']
    d.constant_space.assert_called_with(2.5)
    with pytest.raises(TypeError):
        process.process([d], vbp=1.0)
    with pytest.raises(ValueError):
        process.process([d], interp=('a', None))
    with pytest.raises(ValueError):
        process.process([d], interp=2.0)
    with pytest.raises(NotImplementedError):
        process.process([d], interp=(2.0, 'gps.csv'))


def test_vbp_and_interp_subcommands(tmp_path):
    """reference test/test_impproc.py vbp / interp cases: argument forwarding and output naming."""
    dat = MagicMock()
    run_cli(['vbp', '10', '20', 'dummy_raw.mat'], [dat])
    dat.vertical_band_pass.assert_called_with(10.0, 20.0)
    dat.save.assert_called_with('dummy_bandpassed.mat')
    dat = MagicMock()
    run_cli(['interp', '5', 'dummy.mat'], [dat])
    dat.constant_space.assert_called_with(5.0, min_movement=1.0e-2)
    dat.save.assert_called_with('dummy_interp.mat')
    dat = MagicMock()
    run_cli(['interp', '--minmove', '0.5', '2.5', 'a.mat', '-o', 'out.mat'], [dat])
    dat.constant_space.assert_called_with(2.5, min_movement=0.5)
    dat.save.assert_called_with('out.mat')
    with pytest.raises(NotImplementedError):
        run_cli(['interp', '--gps_fn', 'gps.csv', '5', 'dummy.mat'], [MagicMock()])
    with pytest.raises(SystemExit):
        run_cli(['vbp', '10', 'dummy.mat'], [MagicMock()])


def test_impdarexec_proc_migrate(tmp_path):
    from impdar_amd.bin import impdarexec
    dat = MagicMock()
    dat.fn = str(tmp_path / 'line_raw.mat')
    with patch.object(sys, 'argv', ['impdar', 'proc', '-migrate', 'phsh', dat.fn]), \
            patch('impdar_amd.lib.process.load', return_value=[dat]):
        impdarexec.main()
    dat.migrate.assert_called_with(mtype='stolt')
    dat.save.assert_called_with(str(tmp_path / 'line_proc.mat'))


# ---- .mat interop pinned to files the REFERENCE wrote (tests/golden/M*.mat, generated by make_golden.py from the
# reference's RadarData.save(), _RadarDataSaving.py:32-78; loader RadarData/__init__.py:207-244, flags
# RadarFlags.py:64-98)
def _mat_keys(path):
    from scipy.io import loadmat
    m = loadmat(path)
    return {k: v for k, v in m.items() if not k.startswith('__')}


@pytest.mark.parametrize('name,dtype,has_picks', [('M1_ref_saved_f64', np.float64, False),
                                                  ('M2_ref_resaved_int16', np.int16, True)])
def test_loads_files_written_by_the_reference(name, dtype, has_picks, tmp_path):
    from impdar_amd import synth
    src = os.path.join(GOLDEN, name + '.mat')
    r = RadarData(src)
    snum, tnum = 24, 18
    geo = synth.geometry(snum, tnum, dx=1.5)
    data = synth.diffractor_radargram(snum, tnum, ndiff=3, dx=1.5)
    data = np.round(data * 1000).astype(dtype) if np.issubdtype(dtype, np.integer) else data.astype(dtype)
    assert r.data.dtype == dtype and r.data_dtype == dtype and np.array_equal(r.data, data)
    assert (r.snum, r.tnum) == (snum, tnum) and r.dt == geo['dt'] and r.chan == 1 and r.trig_level == 0
    for attr, want in (('travel_time', geo['travel_time']), ('dist', geo['dist']), ('trace_int', geo['trace_int']),
                       ('lat', np.arange(tnum) * 2.), ('long', np.arange(tnum) * 3.),
                       ('x_coord', np.arange(tnum) * 1.5), ('y_coord', None), ('elev', 100. + 0.01 * np.arange(tnum)),
                       ('decday', np.arange(tnum, dtype=float)), ('trace_num', np.arange(tnum) + 1.),
                       ('trig', None), ('pressure', None)):
        got = getattr(r, attr)
        if want is None:
            # all-zero vectors are vectors in the file; check_attrs keeps them (only a SCALAR 0 means None)
            assert got is not None and np.array_equal(got, np.zeros(tnum))
        else:
            assert np.array_equal(got, want), attr
    assert r.flags.mig == 'kirch' and np.array_equal(r.flags.bpass, [1., 2., 10.])
    assert not r.flags.batch and not r.flags.reverse and r.flags.crop.shape == (3,)
    assert (r._picks_struct is not None) == has_picks
    # written back by this repo: same keys, same dtypes and shapes as the reference's file
    out = str(tmp_path / 'again.mat')
    r.save(out)
    a, b = _mat_keys(src), _mat_keys(out)
    assert sorted(a) == sorted(b)
    for k in a:
        if k == 'fn':               # the path the object was loaded from (the reference's loader sets it too)
            continue
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        if a[k].dtype.names is None and a[k].dtype != object:
            assert np.array_equal(a[k], b[k]), k
    assert a['flags'].dtype.names == b['flags'].dtype.names
    for f in a['flags'].dtype.names:
        assert np.array_equal(a['flags'][f][0][0], b['flags'][f][0][0]), f
    if has_picks:
        assert a['picks'].dtype.names == b['picks'].dtype.names


def test_constant_space_accepts_the_empty_picks_struct_of_resaved_files(tmp_path):
    """Every file the reference loaded and saved again carries an (empty) picks struct; interp must not refuse
    it, and it must still be in the file written afterwards.  Real picks stay refused."""
    from impdar_amd.lib.RadarData._RadarDataProcessing import picks_struct_holds_picks
    r = RadarData(os.path.join(GOLDEN, 'M2_ref_resaved_int16.mat'))
    assert r._picks_struct is not None and not picks_struct_holds_picks(r._picks_struct)
    picked = r._picks_struct.copy()
    picked['samp1'][0][0] = np.arange(36.).reshape(2, 18)
    assert picks_struct_holds_picks(picked)
    r2 = RadarData(os.path.join(GOLDEN, 'M2_ref_resaved_int16.mat'))
    r2._picks_struct = picked
    with pytest.raises(NotImplementedError):
        r2.constant_space(3.0)
