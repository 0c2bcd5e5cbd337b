"""``RadarData.constant_space``: restack the radargram onto a constant trace spacing (reference
``src/impdar/lib/RadarData/_RadarDataProcessing.py:499-583``).  The stationary-shot bookkeeping and the
per-trace attribute vectors are host NumPy; the (snum, tnum) interpolation runs on the MI355X
(``impdar_trace_lerp``), on the resident copy when the radargram is held in HBM (``to_device``)."""
import numpy as np

from ... import preproc


def picks_struct_holds_picks(struct):
    """True when a loaded ``mat['picks']`` struct has anything in samp1 / samp2 / samp3 (the reference's
    ``Picks.to_struct`` writes a lone 0 for each of them while nothing is picked, Picks.py:344-363)."""
    if struct is None:
        return False
    names = getattr(struct.dtype, 'names', None) or ()
    for key in ('samp1', 'samp2', 'samp3'):
        if key not in names:
            continue
        val = np.asarray(struct[key][0][0])
        if val.size > 1 or (val.size == 1 and val.dtype != object and not (val.flat[0] == 0 or val.flat[0] != val.flat[0])):
            return True
    return False


def constant_space(self, spacing, min_movement=1.0e-2, show_nomove=False):
    """Interpolate data and GPS attributes onto ``spacing`` metres between traces; shots that moved less
    than ``min_movement`` metres are dropped first.  ``show_nomove`` (a plot in the reference) is refused."""
    if show_nomove:
        raise NotImplementedError('show_nomove plotting is not part of the MI355X migration engine')
    # Every file the reference has loaded and saved again carries a `picks` struct (RadarData/__init__.py:239-242,
    # _RadarDataSaving.py:51-52), empty unless somebody picked: only real picks are refused, the empty struct
    # travels on to save() as it is (the reference's constant_space leaves an unpicked Picks object alone too,
    # _RadarDataProcessing.py:555-566).
    if getattr(self, 'picks', None) is not None or picks_struct_holds_picks(getattr(self, '_picks_struct', None)):
        raise NotImplementedError('re-spacing picks is not part of the MI355X migration engine')
    plan = preproc.SpacingPlan(self.dist, spacing, min_movement)
    good_vals, temp_dist, new_dists = plan.good_vals, plan.temp_dist, plan.new_dists

    dev = getattr(self, '_dev', None)
    if dev is not None:
        new_dev = plan.apply_dev(dev)
        dev.free()
        self._dev = new_dev
        self.data = None
        snum = new_dev.shape[0]
    else:
        self.data = plan.apply_host(self.data)
        snum = self.data.shape[0]

    for attr in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'trig']:
        setattr(self, attr, preproc.interp1d_linear(temp_dist, getattr(self, attr)[good_vals], new_dists))
    for attr in ['elev']:
        if getattr(self, attr) is not None:
            setattr(self, attr, preproc.interp1d_linear(temp_dist, getattr(self, attr)[good_vals], new_dists))

    self.snum = snum if self.snum is None else self.snum
    self.tnum = plan.n_new
    self.trace_num = np.arange(self.tnum).astype(int) + 1
    self.dist = new_dists
    self.trace_int = np.hstack((np.array(np.nanmean(np.diff(self.dist))), np.diff(self.dist))) * 1000.
    try:
        self.flags.interp[0] = 1
        self.flags.interp[1] = spacing
    except (IndexError, TypeError):
        self.flags.interp = np.ones((2,))
        self.flags.interp[1] = spacing
