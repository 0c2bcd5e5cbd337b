#!/usr/bin/env python3
"""Headline benchmark: Kirchhoff migration of a 10000-trace x 4096-sample
float32 radargram (BASELINE.json config 3) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W          (starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over the radargram with the input
already resident in HBM: time gradient + transpose of the rank's own input
traces, exchange of image rows (RCCL all-gather, or grouped send/recv of the
aperture halos when those are narrower), diffraction sum of the rank's
output-trace block.  Strong scaling: the radargram is fixed, output blocks are
balanced by in-aperture pair count.

Rank 0 prints ONE JSON line (see DESIGN.md section 6 for the field meanings).
No torch anywhere: the control plane is impdar_amd.parallel.Rendezvous, all
device work goes through the C ABI.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# LDS read port: 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS section: "aggregate ~150 TB/s")
LDS_PEAK_GBS = 256 * 256 * 2.4
FP32_VECTOR_PEAK_TF = 157.3
FP64_VECTOR_PEAK_TF = 78.6   # v_fma_f64: 4 cycles per wave instruction, measured (profiles/r04_valu_rates.txt): 64 x 2 / 4 x 1024 SIMDs x 2.4 GHz
FP16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
PARITY_BAR = 1e-4         # relative L2 of the fast (float32) kernel against the float64 oracle, DESIGN.md section 2


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_threads():
    return max(1, min(64, os.cpu_count() or 1))


# ---------------------------------------------------------------------------------------------------------
# CPU legs (the oracle is the checker and the reported baseline, never the product)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(data_full, geo, vel, tnum, budget_s=15.0):
    """Time the plain-C oracle (oracle/kirch_oracle.c, OpenMP on the host cores) on a bounded sample of output
    traces of the same workload.  Returns (record, columns, oracle block) -- the block doubles as the parity
    reference of the timed GPU output."""
    from oracle import c_oracle
    cores = c_oracle.threads()
    cols = np.unique(np.linspace(0, tnum - 1, 8 * cores).round().astype(np.int32))
    t0 = time.time()
    c_oracle.kirchhoff(data_full, geo['travel_time'], geo['dist'], vel, False, traces=cols[:cores])
    per_trace = (time.time() - t0) / cores
    n = int(max(cores, min(len(cols), budget_s / max(per_trace, 1e-6))))
    n = (n // cores) * cores
    sel = np.unique(np.linspace(0, tnum - 1, n).round().astype(np.int32))
    t0 = time.time()
    want = c_oracle.kirchhoff(data_full, geo['travel_time'], geo['dist'], vel, False, traces=sel)
    el = time.time() - t0
    rec = {"value": len(sel) / el, "unit": "traces/s", "cores": cores, "kind": "port",
           "sample": "%d of %d output traces, evenly spaced, full aperture, fp64, %.1f s (plain C + OpenMP restatement "
                     "of mig_python.py:35-60; includes numpy.gradient of the radargram)" % (len(sel), tnum, el)}
    return rec, sel, want


def cpu_numpy_1core(data_full, geo, vel, tnum):
    """the closed-form NumPy restatement (oracle/mig_oracle.kirchhoff, one output
    trace = a few whole-radargram ufunc passes) on ONE core, the like-for-like stand-in for the reference's
    one-core NumPy (its own O((snum*tnum)^2) loop cannot reach this size)."""
    from oracle import mig_oracle
    a, b, c = tnum // 2, tnum // 4, (3 * tnum) // 4
    t0 = time.time()
    mig_oracle.kirchhoff(data_full, geo['travel_time'], geo['dist'], vel, traces=[a])
    t1 = time.time() - t0
    t0 = time.time()
    mig_oracle.kirchhoff(data_full, geo['travel_time'], geo['dist'], vel, traces=[a, b, c])
    t3 = time.time() - t0
    per = max((t3 - t1) / 2.0, 1e-9)
    return {"value": 1.0 / per, "unit": "traces/s", "cores": 1, "kind": "port",
            "sample": "NumPy closed form, 2 output traces of %d beyond the shared set-up (%.1f s set-up incl. "
                      "numpy.gradient, %.1f s per trace), full aperture, fp64" % (tnum, t1 - per, per)}


# ---------------------------------------------------------------------------------------------------------
# HBM-side counters of the dominant kernel, measured by this run (child processes under rocprofv3)
# ---------------------------------------------------------------------------------------------------------
def under_profiler():
    pre = os.environ.get('LD_PRELOAD', '')
    return 'rocprof' in pre or any(k.startswith('ROCPROF') or k.startswith('ROCP_') for k in os.environ)


def pmc_traffic(args, kernel_substr, timeout_s=150.0):
    """FETCH_SIZE and WRITE_SIZE of the migration kernel, one rocprofv3 --pmc pass each (they do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots"), on the same geometry with an all-zero radargram (addresses do
    not depend on the data).  Returns (bytes per launch | None, note)."""
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None, None, 'rocprofv3 not found'
    if under_profiler():
        return None, None, 'bench.py itself runs under a profiler'
    raw = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='impdar_pmc_', dir='/tmp')
        cmd = [prof, '--pmc', counter, '--kernel-trace', '-d', d, '-o', 'x', '--output-format', 'csv', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-child', '--steps', '2', '--warmup', '1',
               '--tnum', str(args.tnum), '--snum', str(args.snum), '--mode', args.mode, '--dtype', args.dtype]
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            p = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                rc = p.wait(timeout_s)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                return None, None, '%s pass timed out' % counter
            if rc:
                return None, None, '%s pass exited with %d' % (counter, rc)
            vals = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r.get('Kernel_Name', '') and r.get('Counter_Name') == counter:
                        vals.append(float(r['Counter_Value']))
            if not vals:
                return None, None, 'no %s rows for %s' % (counter, kernel_substr)
            raw[counter] = sum(vals) / len(vals)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 reports exactly half of the bytes of 16 B/lane reads
    # (MI355X_MICROARCH.md, HBM section); every read of this kernel is one (LDS-DMA staging, pick rows).
    traffic = (2.0 * raw['FETCH_SIZE'] + raw['WRITE_SIZE']) * 1024.0
    return traffic, raw, ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two child runs of this command (2 steps, zero-filled '
                          'radargram, same geometry); FETCH_SIZE doubled: all reads are 16 B/lane')


def pmc_path_totals(path, calls=3, timeout_s=150.0):
    """FETCH_SIZE / WRITE_SIZE summed over EVERY kernel of one call of a secondary path (rocFFT's included), one
    rocprofv3 --pmc pass each, in child runs of this file (--pmc-child-path).  Returns ({counter: KiB per call}, note)."""
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None, 'rocprofv3 not found'
    if under_profiler():
        return None, 'bench.py itself runs under a profiler'
    raw = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='impdar_pmc_', dir='/tmp')
        cmd = [prof, '--pmc', counter, '--kernel-trace', '-d', d, '-o', 'x', '--output-format', 'csv', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-child-path', path, '--steps', str(calls)]
        try:
            p = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL)
            try:
                rc = p.wait(timeout_s)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                return None, '%s pass timed out' % counter
            if rc:
                return None, '%s pass exited with %d' % (counter, rc)
            total = 0.0
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') == counter:
                        total += float(r['Counter_Value'])
            raw[counter] = total / calls
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return raw, ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child runs of this command, summed over every kernel of '
                 '%d calls (rocFFT\'s included) / %d' % (calls, calls))


def pmc_child_path(path, calls):
    """Child of pmc_path_totals: `calls` resident calls of one secondary path, nothing else on the GPU."""
    import contextlib
    import io
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    _hip.load()
    rng = np.random.default_rng(0)
    n = 4096
    geo = synth.geometry(n, n)
    x = rng.standard_normal((n, n)).astype(np.float32)
    for _ in range(calls):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            assert path == 'stolt'
            d.migrate('stolt', vel=1.68e8, htaper=100, vtaper=1000)
        d._dev.free()
        d._dev = None


def pmc_valu_busy(kind, kernel_substr, timeout_s=150.0):
    """Share of the vector-issue slots the phase shift's frequency-sum kernel keeps busy at 8192 x 8192: one rocprofv3 --pmc
    child pass of this file (--pmc-child-ps KIND), SQ_ACTIVE_INST_VALU (quad-cycles, summed over the chip) x 4 over
    1024 SIMDs x the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs).  Returns a dict or {"error": ...}."""
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    if under_profiler():
        return {"error": "bench.py itself runs under a profiler"}
    d = tempfile.mkdtemp(prefix='impdar_pmc_', dir='/tmp')
    cmd = [prof, '--pmc', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU', '--kernel-trace', '-d', d, '-o', 'x', '--output-format', 'csv',
           '--', sys.executable, os.path.abspath(__file__), '--pmc-child-ps', kind]
    try:
        p = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            rc = p.wait(timeout_s)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            return {"error": "counter pass timed out"}
        if rc:
            return {"error": "counter pass exited with %d" % rc}
        acc = {}
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r.get('Kernel_Name', ''):
                    acc.setdefault(r.get('Counter_Name'), []).append(float(r['Counter_Value']))
        if not acc.get('SQ_ACTIVE_INST_VALU') or not acc.get('GRBM_GUI_ACTIVE'):
            return {"error": "no counter rows for %s" % kernel_substr}
        valu = float(np.mean(acc['SQ_ACTIVE_INST_VALU'])) * 4.0
        cyc = float(np.mean(acc['GRBM_GUI_ACTIVE'])) / 8.0
        return {"valu_busy": valu / (1024.0 * cyc), "kernel_cycles": cyc, "valu_instructions": float(np.mean(acc.get('SQ_INSTS_VALU', [0.])))}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_child_ps(kind):
    """Child of pmc_valu_busy: two resident phase-shift calls of `kind` at 8192 x 8192, nothing else on the GPU."""
    import ctypes as C
    from impdar_amd import _hip, synth
    from oracle import mig_oracle as mo
    lib, ctx = _hip.load(), _hip.context()
    n = 8192
    geo = synth.geometry(n, n)
    dt = np.float64 if kind.endswith('_f64') else np.float32
    x = np.random.default_rng(0).standard_normal((n, n)).astype(dt)
    vel = gazdag_velocity(kind, geo, n)
    if np.ndim(vel) == 2:
        vel = mo.get_velocity_profile(geo['travel_time'], vel)
    kx = mo._kx(n, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    vm = None if np.ndim(vel) == 0 else np.ascontiguousarray(vel, dtype=np.float64)
    d_in = _hip.DeviceArray.from_host(ctx, x)
    d_out = _hip.DeviceArray(ctx, d_in.shape, d_in.dtype)
    for _ in range(2):
        _hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, _hip.dtype_code(dt), n, n, n, kx.ctypes.data_as(dp), ws.ctypes.data_as(dp),
                                             C.c_double(geo['dt']), tt.ctypes.data_as(dp), C.c_double(float(vel) if vm is None else 0.0),
                                             vm.ctypes.data_as(dp) if vm is not None else None, 0 if vm is None else n,
                                             C.c_double(100.), C.c_double(1000.), d_out.ptr), 'impdar_phaseshift_dev')
    d_in.free()
    d_out.free()


# ---------------------------------------------------------------------------------------------------------
# parity of the 8192 x 8192 phase-shift sub-records: the oracle on a few wavenumbers, in CPU-only child processes that run
# beside the GPU legs (wavenumbers are independent in phaseShift, mig_python.py:438-487: tests/test_phaseshift_gpu.py)
# ---------------------------------------------------------------------------------------------------------
SPOT_KS = (0, 37, 200, 4096)          # zero, low, one that holds a frequency ON the evanescent boundary of 1.69e8 m/s, Nyquist
SPOT_KINDS = ('vz4', 'vz4_f64', 'const', 'const_f64', 'layers41', 'gradient', 'gradient_f64', 'firn', 'firn_f64')


def gazdag_velocity(kind, geo, n):
    """What the sub-record `kind` migrates with: a scalar, a (v, z) table or a per-step profile of length n."""
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    kind = kind[:-4] if kind.endswith('_f64') else kind
    if kind == 'vz4':
        return np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    if kind == 'const':
        return 1.69e8
    if kind == 'layers41':
        return np.stack([np.linspace(1.69e8, 2.2e8, 41), np.linspace(0., 2.0 * Rp, 41)], axis=1)
    if kind == 'firn':                # fast near the surface, flat below: what a firn column looks like (changes at every step)
        return np.ascontiguousarray(1.69e8 + 0.6e8 * np.exp(-np.arange(n) * geo['dt'] / 0.8e-6))
    assert kind == 'gradient'
    return np.ascontiguousarray(1.69e8 + 0.5e8 * np.linspace(0., 1., n))      # changes at every step: no runs of constant velocity


def config5_radargram():
    """The 8192 x 8192 float32 white-noise radargram of path_records (second draw of its generator)."""
    rng = np.random.default_rng(0)
    rng.standard_normal((4096, 4096))
    return rng.standard_normal((8192, 8192)).astype(np.float32)


def spot_dft(n):
    ks = np.array(SPOT_KS)
    cols = np.concatenate([ks, (n - ks) % n])
    return ks, cols, np.exp(-2j * np.pi * np.outer(np.arange(n), cols) / n)


def spot_oracle_child(kind, out_path):
    """`bench.py --spot-oracle KIND OUT`: CPU only.  fft_x(Re ifft_k TK)[k] = (TK[k] + conj(TK[-k])) / 2 on SPOT_KS."""
    from impdar_amd import synth
    from oracle import mig_oracle
    n = 8192
    geo = synth.geometry(n, n)
    x = config5_radargram()
    f64 = kind.endswith('_f64')
    tap = mig_oracle._apply_taper(x.astype(np.float64) if f64 else x, 100, 1000, inplace_form=True)
    tap = np.asarray(tap, dtype=np.float64 if f64 else np.float32).astype(np.float64)
    ks, cols, E = spot_dft(n)
    FKc = np.fft.fft(tap @ E.real + 1j * (tap @ E.imag), n=n, axis=0)       # (nt, len(cols)): the 2-D spectrum's columns
    kx = mig_oracle._kx(n, geo['trace_int'], geo['dist'])[cols]
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    vel = gazdag_velocity(kind, geo, n)
    vmig = vel if np.ndim(vel) == 0 else (vel if np.ndim(vel) == 1 else mig_oracle.get_velocity_profile(geo['travel_time'], vel))
    TK = mig_oracle.phase_shift_tk(FKc, vmig, kx, ws, geo['dt'], geo['travel_time'], n, len(cols))
    np.save(out_path, 0.5 * (TK[:, :len(ks)] + np.conj(TK[:, len(ks):])))


def start_spot_oracles():
    """One child per kind (about 15 s of one core each); returns {kind: (Popen, path)}."""
    tmp = tempfile.mkdtemp(prefix='impdar_spot_', dir='/tmp')
    procs = {}
    for kind in SPOT_KINDS:
        out = os.path.join(tmp, kind + '.npy')
        procs[kind] = (subprocess.Popen([sys.executable, os.path.abspath(__file__), '--spot-oracle', kind, out],
                                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                        env=dict(os.environ, OMP_NUM_THREADS='8', OPENBLAS_NUM_THREADS='8')), out)
    return procs


def spot_parity(img, spot, kind, bar):
    """rel-L2 of the image's spot wavenumbers against the child's oracle columns (None when the child failed)."""
    if not spot or kind not in spot:
        return {"error": "no oracle child"}
    proc, path = spot[kind]
    try:
        proc.wait(timeout=120)
        want = np.load(path)
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    n = img.shape[1]
    ks, cols, E = spot_dft(n)
    a = np.asarray(img, dtype=np.float64)
    got = a @ E.real[:, :len(ks)] + 1j * (a @ E.imag[:, :len(ks)])
    err = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))
    rec = {"parity_rel_l2": err, "parity_wavenumbers": list(SPOT_KS), "parity_bar": bar,
           "parity_vs": "oracle/mig_oracle.py phase_shift_tk on the (k, -k) columns of the 2-D spectrum, CPU child process"}
    if not err <= bar:
        rec["error"] = "spot wavenumbers differ from the oracle"
    return rec



def valu_roofline(kind, kernel, kms, n, flop, no_pmc):
    """The roofline object of a phase-shift record whose frequency sums ran on a transform path (ps_nufft_kernel /
    ps_series_kernel): what binds those kernels is vector ISSUE, measured by a counter pass; the direct sum's equivalent rate
    stays as a labelled extra."""
    nf, nk = n // 2, n
    esz = 16 if kind.endswith('_f64') else 8
    floor_ms = (esz * nf * nk + esz * n * nk) / (HBM_PEAK_GBS * 1e9) * 1e3       # spectrum read once + TK written once
    r = {"bound": "valu issue", "unit": VALU_UNIT, "peak": 1.0, "achieved": None, "frac": None,
         "hbm_floor_ms": floor_ms, "kernel_ms_over_hbm_floor": kms / floor_ms,
         "x_direct_sum": {"TFLOPs_equivalent": flop / (kms * 1e-3) / 1e12, "x_fp32_vector_peak": flop / (kms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TF}}
    if not no_pmc:
        c = pmc_valu_busy(kind, kernel)
        if "error" in c:
            r["counter_error"] = c["error"]
        else:
            r["achieved"] = r["frac"] = c["valu_busy"]
            r["kernel_cycles"] = c["kernel_cycles"]
            r["valu_instructions"] = c["valu_instructions"]
    return r


# ---------------------------------------------------------------------------------------------------------
# secondary paths, driver-timed: BASELINE configs 2 (Stolt) and 5 (Gazdag v(z)), config 3 in float64
# ---------------------------------------------------------------------------------------------------------
TRANSFORM_KERNELS = ('ps_nufft_kernel', 'ps_series_kernel')


def path_records(no_cpu, no_pmc=False, full_data=None, geo3=None, spot=None):
    import contextlib
    import io
    import ctypes as C
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    lib, ctx = _hip.load(), _hip.context()
    rng = np.random.default_rng(0)
    out = {}

    def dat_of(data, geo):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = data, data.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        return d

    def device_ms(run, make, reps=3, kernel=False):
        """Resident radargram: device-event duration of the path's kernels (median) and host wall of the call."""
        ms, wall, kms = [], [], []
        for i in range(reps + 1):                 # the first call pays rocFFT plan creation
            d = make()
            d.to_device()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                run(d)
            wall.append(time.perf_counter() - t0)
            v = C.c_float()
            _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'impdar_ctx_last_ms')
            ms.append(v.value)
            if kernel:
                _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v)), 'impdar_ctx_last_kernel_ms')
                kms.append(v.value)
            if i == reps:
                buf = C.create_string_buffer(1024)
                _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
                device_ms.metrics = json.loads(buf.value.decode())
                d.from_device()
                fin = bool(np.isfinite(d.data).all())
                device_ms.image = d.data
            else:
                d._dev.free()
                d._dev = None
        if kernel:
            return float(np.median(ms[1:])), float(np.median(wall[1:])), fin, float(np.median(kms[1:]))
        return float(np.median(ms[1:])), float(np.median(wall[1:])), fin

    def host_call_ms(run, make, reps=3):
        """Host arrays in, float64 image out (what RadarData.migrate on a loaded file does): median wall of `reps`
        calls after one warm-up call, PCIe both ways and the widening to float64 included."""
        wall = []
        for _ in range(reps + 1):
            d = make()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                run(d)
            wall.append(time.perf_counter() - t0)
            assert d.data.dtype == np.float64 or d.data.dtype == np.float32
        return float(np.median(wall[1:])) * 1e3

    # ---- config 2: Stolt f-k, 4096 x 4096 float32
    n = 4096
    geo = synth.geometry(n, n)
    x = rng.standard_normal((n, n)).astype(np.float32)
    ms, wall, fin = device_ms(lambda d: d.migrate('stolt', vel=1.68e8, htaper=100, vtaper=1000), lambda: dat_of(x, geo))
    algo = 40 * n * n                                    # 5 passes x (read + write) x 4 B
    rec = {"workload": "Stolt f-k migration, 4096x4096 float32 (BASELINE config 2), resident in HBM",
           "device_ms": ms, "call_ms": wall * 1e3, "traces_per_s": n / (ms * 1e-3), "output_finite": fin,
           "roofline": {"bound": "hbm", "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo,
                        "fused_floor_bytes": 8 * n * n}}
    e2e = host_call_ms(lambda d: d.migrate('stolt', vel=1.68e8, htaper=100, vtaper=1000), lambda: dat_of(x, geo))
    rec["end_to_end"] = {"wall_ms": e2e, "value": n / (e2e * 1e-3), "unit": "traces/s",
                         "note": "RadarData.migrate('stolt') on a host float32 array: H2D + the device path + D2H (the "
                                 "reference returns float32 for float32 input under NumPy >= 2); median of 3 calls"}
    if not no_cpu:
        from oracle import mig_oracle
        t0 = time.perf_counter()
        mig_oracle.stolt(x, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 100, 1000)
        el = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": n / el, "unit": "traces/s", "cores": 1, "kind": "port",
                               "sample": "NumPy closed form (pocketfft + vectorised stretch), the same 4096x4096 "
                                         "float32 radargram, %.2f s; the reference's own loop takes 100.3 s at this "
                                         "size (BASELINE.md)" % el}
    if not no_pmc:
        raw, note = pmc_path_totals('stolt')
        rec["roofline"]["traffic_source"] = note
        if raw:
            # FETCH_SIZE halves wide (16 B/lane) reads on gfx950 and other widths are uncalibrated (MI355X_MICROARCH.md,
            # HBM): the transposes and the stretch read 4-8 B/lane, rocFFT's kernels mixed widths.  WRITE_SIZE is exact
            # for streaming stores, and every pass of this pipeline writes as many bytes as it reads, so the write
            # side calibrates the read side: traffic = 2 x WRITE_SIZE (raw FETCH_SIZE kept beside it).
            wr, fe = raw['WRITE_SIZE'] * 1024.0, raw['FETCH_SIZE'] * 1024.0
            rec["roofline"]["traffic"] = 2.0 * wr
            rec["roofline"]["traffic_x_algorithmic"] = 2.0 * wr / algo
            rec["roofline"]["fabric"] = {"write_bytes": wr, "fetch_bytes_raw": fe, "fetch_raw_over_write": fe / wr if wr else None,
                                         "fetch_bytes_if_all_16B_per_lane": 2.0 * fe,
                                         "note": "per call, all 11 kernels; the 5-pass model moves 335.5 MB each way"}
    out["stolt_config2"] = rec

    # ---- config 5: Gazdag phase shift with a 1-D v(z) table, 8192 x 8192 float32
    n = 8192
    geo = synth.geometry(n, n)
    x = rng.standard_normal((n, n)).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    ms, wall, fin, kms = device_ms(lambda d: d.migrate('phsh', vel=tab, htaper=100, vtaper=1000), lambda: dat_of(x, geo),
                                   reps=3, kernel=True)
    steps_ref = float(n) ** 3                            # nt * snum * tnum rotate-accumulates: what the reference executes
    steps = float(n // 2) * n * n                        # ... and what a real radargram needs: frequencies 1..nt/2-1 stand
    #                                                      for their mirror images too, + the Nyquist row (Hermitian walk)
    flop = steps * 8                                     # 8 flop per complex multiply-accumulate
    tf = flop / (kms * 1e-3) / 1e12
    # MFMA instructions the kernel issued, counted by the kernel itself (rounds and row blocks it skips -- all-evanescent
    # chunks -- are not in it): x 32768 flop per v_mfma_f32_32x32x16_f16
    mm = getattr(device_ms, 'metrics', {})
    mfma_flop = float(mm.get('mfma_instructions', 0)) * float(mm.get('flop_per_mfma', 0)) or None
    par5 = spot_parity(device_ms.image, spot, 'vz4', 2e-4)
    nufft_note = ("ps_nufft_kernel evaluates the frequency sum of every run as a non-uniform FFT (csrc/ps_nufft.h): achieved / frac are the "
                  "DIRECT sum's 8 flop per needed rotate-accumulate over kernel_ms -- an equivalent rate for comparison with the "
                  "matrix-core kernels of rounds 3-5, not flop the kernel executes (it executes ~3 % of them: 8 window values per "
                  "frequency and piece, one 8192-point FFT per piece)")
    rec = {"workload": "phase-shift (Gazdag) migration, 1-D v(z) table, 8192x8192 float32 (BASELINE config 5), "
                       "resident in HBM",
           "kernel": mm.get('kernel'),
           "device_ms": ms, "kernel_ms": kms, "call_ms": wall * 1e3, "traces_per_s": n / (ms * 1e-3), "output_finite": fin,
           "rotate_accumulate_steps": steps_ref, "steps_executed": steps,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": FP16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "method": "non-uniform FFT" if mm.get('kernel') == 'ps_nufft_kernel' else "direct sum",
                        "frac": tf / FP16_MFMA_PEAK_TF, "algorithmic_flop": flop,
                        "mfma_flop_executed": mfma_flop,
                        "frac_executed": (mfma_flop / (kms * 1e-3) / 1e12 / FP16_MFMA_PEAK_TF) if mfma_flop else None,
                        "x_fp32_vector_peak": tf / FP32_VECTOR_PEAK_TF,
                        "frac_device_ms": flop / (ms * 1e-3) / 1e12 / FP16_MFMA_PEAK_TF,
                        "x_fp32_vector_peak_device_ms": flop / (ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TF,
                        "note": "kernel_ms = the frequency-sum kernels (ps_mfma_kernel + boundary frequencies + zero-frequency "
                                "row); achieved = 8 flop per needed complex rotate-accumulate (half walk) over kernel_ms, "
                                "against the dense float16 MFMA peak; the kernel executes three float16 products per "
                                "float32 product (hi.hi + hi.lo + lo.hi) plus 1/16 of the rotations on the vector "
                                "units; device_ms also holds the real-to-complex and trace transforms, the inverse "
                                "transform and the transposes; kernel_ms includes ps_setup_kernel and the host-side "
                                "synchronisation inside the matrix-core path; frac_device_ms / x_fp32_vector_peak_device_ms "
                                "are the same flop over device_ms (the basis rounds 1-2 reported); mfma_flop_executed = the "
                                "kernel's own count of issued MFMAs x 32768"}}
    if mm.get('kernel') in TRANSFORM_KERNELS:
        rec["roofline"] = valu_roofline('vz4', mm.get('kernel'), kms, n, flop, no_pmc)
    rec.update(par5)
    rec["kernel"] = mm.get('kernel')
    e2e = host_call_ms(lambda d: d.migrate('phsh', vel=tab, htaper=100, vtaper=1000), lambda: dat_of(x, geo), reps=2)
    rec["end_to_end"] = {"wall_ms": e2e, "value": n / (e2e * 1e-3), "unit": "traces/s",
                         "note": "RadarData.migrate('phsh', vel=table) on a host float32 array: getVelocityProfile + H2D + "
                                 "the device path + D2H widened to the float64 the reference returns; median of 2 calls"}
    if not no_cpu:
        from oracle import mig_oracle
        m = 512
        gs = synth.geometry(m, m)
        xs = x[:m, :m].astype(np.float64)
        Rs = 1.9e8 * gs['travel_time'][-1] * 1e-6 / 2.
        tabs = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rs], [1.8e8, 0.5 * Rs], [1.9e8, 1.2 * Rs]])
        t0 = time.perf_counter()
        mig_oracle.phase_shift(xs, gs['dt'], gs['trace_int'], gs['travel_time'], gs['dist'], tabs)
        el = time.perf_counter() - t0
        scale = (n // m) ** 3
        rec["cpu_baseline"] = {"value": n / (el * scale), "unit": "traces/s", "cores": 1, "kind": "port",
                               "sample": "NumPy closed form on a 512x512 radargram, %.2f s, scaled x%d "
                                         "(work = snum*nt*tnum) to the full size" % (el, scale)}
    out["gazdag_config5"] = rec
    # ... and on float64 data (what a float64 .mat file gets: the vector runs kernel ps_vz64_kernel, no matrix cores)
    x64 = x.astype(np.float64)
    ms64, wall64, fin64, kms64 = device_ms(lambda d: d.migrate('phsh', vel=tab, htaper=100, vtaper=1000), lambda: dat_of(x64, geo),
                                           reps=3, kernel=True)
    tf64 = flop / (kms64 * 1e-3) / 1e12
    mm64 = getattr(device_ms, 'metrics', {})
    img64 = device_ms.image
    out["gazdag_f64_config5"] = {
        "workload": "phase-shift (Gazdag) migration, 1-D v(z) table, 8192x8192 float64 data (BASELINE config 5 in the "
                    "reference's own arithmetic), resident in HBM",
        "device_ms": ms64, "kernel_ms": kms64, "call_ms": wall64 * 1e3, "traces_per_s": n / (ms64 * 1e-3), "output_finite": fin64,
        "steps_executed": steps,
        "roofline": {"bound": "fp64 vector", "achieved": tf64, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                     "frac": tf64 / FP64_VECTOR_PEAK_TF, "algorithmic_flop": flop,
                     "note": "8 flop per needed complex rotate-accumulate (half walk) over kernel_ms against the float64 "
                             "vector peak; 42 % of the (kx, w) plane is evanescent and skipped pair-wise (docs/DESIGN_rounds1-4.md 11.7)"}}
    out["gazdag_f64_config5"]["kernel"] = mm64.get('kernel')
    if mm64.get('kernel') in TRANSFORM_KERNELS:
        out["gazdag_f64_config5"]["roofline"] = valu_roofline('vz4_f64', mm64.get('kernel'), kms64, n, flop, no_pmc)
    out["gazdag_f64_config5"].update(spot_parity(img64, spot, 'vz4_f64', 1e-10))
    del x64, img64
    # ---- the other velocity structures at config-5 size (figures the builder alone had measured)
    def ps_dev(data, vel, reps=4):       # (the median of four timed calls: one disturbed call -- seen once per run on a shared pod -- does not move it)
        """impdar_phaseshift_dev on a resident radargram with a scalar / per-step velocity: (device ms, kernel ms, image, metrics)."""
        from oracle import mig_oracle as mo
        kx = mo._kx(n, geo['trace_int'], geo['dist'])
        ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
        tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
        dp = C.POINTER(C.c_double)
        vm = None if np.ndim(vel) == 0 else np.ascontiguousarray(vel, dtype=np.float64)
        d_in = _hip.DeviceArray.from_host(ctx, data)
        d_out = _hip.DeviceArray(ctx, d_in.shape, d_in.dtype)
        ms_, kms_ = [], []
        for _ in range(reps + 1):
            _hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, _hip.dtype_code(data.dtype), n, n, n, kx.ctypes.data_as(dp),
                                                 ws.ctypes.data_as(dp), C.c_double(geo['dt']), tt.ctypes.data_as(dp),
                                                 C.c_double(float(vel) if vm is None else 0.0),
                                                 vm.ctypes.data_as(dp) if vm is not None else None, 0 if vm is None else n,
                                                 C.c_double(100.), C.c_double(1000.), d_out.ptr), 'impdar_phaseshift_dev')
            v = C.c_float()
            _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'impdar_ctx_last_ms')
            ms_.append(v.value)
            _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v)), 'impdar_ctx_last_kernel_ms')
            kms_.append(v.value)
        buf = C.create_string_buffer(1024)
        _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
        img = d_out.to_host()
        d_in.free()
        d_out.free()
        met = json.loads(buf.value.decode())
        met['device_ms_calls'] = [round(m, 3) for m in ms_[1:]]
        return float(np.median(ms_[1:])), float(np.median(kms_[1:])), img, met

    def gazdag_extra(name, kind, data, bound, peak, bar, what):
        from oracle import mig_oracle as mo
        vel = gazdag_velocity(kind, geo, n)
        if np.ndim(vel) == 2:
            vel = mo.get_velocity_profile(geo['travel_time'], vel)           # what getVelocityProfile hands to phaseShift (:278)
        ms_, kms_, img, met = ps_dev(data, vel)
        tf_ = flop / (kms_ * 1e-3) / 1e12
        r = {"workload": "phase-shift migration, %s, 8192x8192 %s, resident in HBM" % (what, data.dtype.name),
             "kernel": met.get('kernel'), "device_ms": ms_, "kernel_ms": kms_, "device_ms_calls": met.get('device_ms_calls'),
             "traces_per_s": n / (ms_ * 1e-3),
             "output_finite": bool(np.isfinite(img).all()), "steps_executed": steps,
             "roofline": {"bound": bound, "achieved": tf_, "peak": peak, "unit": "TFLOP/s", "frac": tf_ / peak, "algorithmic_flop": flop,
                          "note": "8 flop per needed complex rotate-accumulate (half walk, evanescent pairs included) over kernel_ms"}}
        if met.get('kernel') in TRANSFORM_KERNELS:
            r["roofline"] = valu_roofline(kind, met.get('kernel'), kms_, n, flop, no_pmc)
        if met.get('mfma_instructions') and float(met['mfma_instructions']) > 0:
            ex = float(met['mfma_instructions']) * float(met['flop_per_mfma'])
            r["roofline"]["mfma_flop_executed"] = ex
            r["roofline"]["frac_executed"] = ex / (kms_ * 1e-3) / 1e12 / peak
        r.update(spot_parity(img, spot, kind, bar))
        out[name] = r

    gazdag_extra("gazdag_const_config5", 'const', x, "mfma", FP16_MFMA_PEAK_TF, 2e-4, "constant velocity 1.69e8 m/s (mig_python.py:396-420)")
    gazdag_extra("gazdag_layers41_config5", 'layers41', x, "mfma", FP16_MFMA_PEAK_TF, 2e-4,
                 "41-row (v, z) table: 21 layers of ~420 steps, every boundary smeared over single steps")
    gazdag_extra("gazdag_smooth_config5", 'gradient', x, "fp32 vector", FP32_VECTOR_PEAK_TF, 2e-4,
                 "velocity changing at EVERY step (linear gradient 1.69e8 -> 2.19e8 m/s) through the C entry point")
    gazdag_extra("gazdag_smooth_f64_config5", 'gradient_f64', x.astype(np.float64), "fp64 vector", FP64_VECTOR_PEAK_TF, 1e-10,
                 "velocity changing at EVERY step (linear gradient), float64 data")
    gazdag_extra("gazdag_const_f64_config5", 'const_f64', x.astype(np.float64), "fp64 vector", FP64_VECTOR_PEAK_TF, 1e-10,
                 "constant velocity 1.69e8 m/s, float64 data")
    gazdag_extra("gazdag_firn_config5", 'firn', x, "fp32 vector", FP32_VECTOR_PEAK_TF, 2e-4,
                 "a firn column (velocity changing at every step, fast near the surface, flat below)")
    gazdag_extra("gazdag_firn_f64_config5", 'firn_f64', x.astype(np.float64), "fp64 vector", FP64_VECTOR_PEAK_TF, 1e-10,
                 "a firn column, float64 data")
    del x

    # ---- config 3 in the reference's own arithmetic: float64 data, kirch_dquad_kernel (mig_python.py:53,118 sum in float64)
    from impdar_amd.kirchhoff import KirchhoffPlan
    snum, tnum, vel = 4096, 10000, 1.69e8
    g3 = geo3 if geo3 is not None else synth.geometry(snum, tnum)
    if full_data is not None and full_data.shape == (snum, tnum):
        x64 = full_data.astype(np.float64)
    else:
        x64 = rng.standard_normal((snum, tnum))
    plan = KirchhoffPlan(ctx, np.float64, snum, tnum, g3['dist'], g3['travel_time'], vel, False, 'exact')
    d_in = _hip.DeviceArray.from_host(ctx, x64)
    d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float64)
    nstep = 5
    for _ in range(2):
        plan.prep(d_in, tnum, 0, tnum)
        plan.migrate(d_out, 0, tnum)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(nstep):
        plan.prep(d_in, tnum, 0, tnum)
        plan.migrate(d_out, 0, tnum)
    plan.sync()
    step_ms = (time.perf_counter() - t0) / nstep * 1e3
    kms = float(np.mean([plan.history_ms(b)[2] for b in range(nstep)]))
    pairs = plan.count_pairs(0, tnum)
    algo = pairs * 8 + snum * tnum * 8
    ach = algo / (kms * 1e-3) / 1e9
    rec = {"workload": "Kirchhoff diffraction sum, 10000 traces x 4096 samples, float64 data and arithmetic (what the "
                       "reference computes in), resident in HBM",
           "kernel": plan.kernel, "ms_per_step": step_ms, "kernel_ms": kms, "traces_per_s": tnum / (step_ms * 1e-3),
           "pairs": pairs,
           "roofline": {"bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBS,
                        "algorithmic_bytes_per_launch": algo}}
    if not no_cpu and full_data is not None and full_data.shape == (snum, tnum):
        from oracle import c_oracle
        cols = np.unique(np.linspace(0, tnum - 1, 24).round().astype(np.int32))
        want = c_oracle.kirchhoff(x64, g3['travel_time'], g3['dist'], vel, False, traces=cols)
        got = d_out.to_host()[:, cols]
        rec["parity_max_rel"] = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
        rec["parity_cols"] = int(len(cols))
        rec["parity_bar"] = 1e-12
        if not rec["parity_max_rel"] <= 1e-12:
            rec["error"] = "float64 output differs from the C oracle"
    plan.destroy()
    d_in.free()
    d_out.free()
    out["kirchhoff_f64_config3"] = rec

    # ---- config 3's size on a NON-UNIFORM profile (+-0.3 dx jitter, the recipe of golden K3): kirch_gen_kernel
    jit = np.random.default_rng(3).uniform(-0.3, 0.3, tnum)
    distj = (np.arange(tnum) + jit) * 1.0e-3
    x32 = full_data.astype(np.float32) if (full_data is not None and full_data.shape == (snum, tnum)) else \
        rng.standard_normal((snum, tnum)).astype(np.float32)
    plan = KirchhoffPlan(ctx, np.float32, snum, tnum, distj, g3['travel_time'], vel, False, 'auto')
    d_in = _hip.DeviceArray.from_host(ctx, x32)
    d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
    for _ in range(4):
        plan.prep(d_in, tnum, 0, tnum)
        plan.migrate(d_out, 0, tnum)
    plan.sync()
    kms = float(np.mean([plan.history_ms(b)[2] for b in range(3)]))
    rec = {"workload": "Kirchhoff diffraction sum, 10000 traces x 4096 samples, float32, trace positions jittered by "
                       "+-0.3 dx (non-uniform dist: mig_python.py:44 takes any), resident in HBM",
           "kernel": plan.kernel, "kernel_ms": kms, "traces_per_s": tnum / (kms * 1e-3),
           "note": "picks computed per pair from the positions; 0.77 s on the per-pair float64 kernel in rounds 1-3; "
                   "bound by vector issue (28 cycles per wave pair step), not by LDS or HBM"}
    if not no_cpu:
        from oracle import c_oracle
        cols = np.unique(np.linspace(0, tnum - 1, 8).round().astype(np.int32))
        want = c_oracle.kirchhoff(x32, g3['travel_time'], distj, vel, False, traces=cols)
        got = d_out.to_host()[:, cols].astype(np.float64)
        rec["parity_rel_l2"] = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))
        rec["parity_cols"] = int(len(cols))
        rec["parity_bar"] = PARITY_BAR
        if not rec["parity_rel_l2"] <= PARITY_BAR:
            rec["error"] = "general-geometry output differs from the C oracle"
    plan.destroy()
    d_in.free()
    d_out.free()
    out["kirchhoff_jittered_config3"] = rec
    # how often a timed call of these records ran well over its median (the intermittent device stall of profiles/r05_slow_call.txt:
    # the medians above do not show it, a one-shot user call would meet it)
    calls = [(k, v["device_ms_calls"]) for k, v in out.items() if isinstance(v, dict) and v.get("device_ms_calls")]
    slow = [(k, round(max(c), 2)) for k, c in calls if max(c) > 1.5 * float(np.median(c))]
    out["slow_calls"] = {"timed_calls": int(sum(len(c) for _, c in calls)), "over_1p5x_median": len(slow), "which": dict(slow)}
    return out


def first_call_child(kind):
    """`bench.py --first-call kirch|stolt|phsh`: a fresh process, as `impproc migrate file.mat` is -- import, context,
    the FIRST migration call of the process on a host array, then a second call for contrast.  One JSON line."""
    import contextlib
    import io
    t_imp = time.perf_counter()
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    import_ms = (time.perf_counter() - t_imp) * 1e3
    vel = 1.69e8
    if kind == 'kirch':
        snum, tnum = 4096, 10000
        geo = synth.geometry(snum, tnum)
        x = synth.diffractor_radargram(snum, tnum, vel=vel, dtype=np.float32, chunk=128, threads=host_threads())
        run = lambda d: d.migrate('kirch', vel=vel)
    elif kind == 'stolt':
        snum = tnum = 4096
        geo = synth.geometry(snum, tnum)
        x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(np.float32)
        run = lambda d: d.migrate('stolt', vel=1.68e8, htaper=100, vtaper=1000)
    else:
        snum = tnum = 8192
        geo = synth.geometry(snum, tnum)
        x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(np.float32)
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
        run = lambda d: d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
    t0 = time.perf_counter()
    _hip.load()
    _hip.context()
    ctx_ms = (time.perf_counter() - t0) * 1e3
    walls = []
    for _ in range(2):
        d = RadarData(None)
        d.data, d.snum, d.tnum = x, snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            run(d)
        walls.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"kind": kind, "import_ms": import_ms, "context_ms": ctx_ms, "first_call_ms": walls[0],
                      "second_call_ms": walls[1], "finite": bool(np.isfinite(d.data).all())}))


def first_call_records():
    """Run the first-call children (must happen before this process touches the GPU).  Per path two fresh processes:
    `cold` -- HOME, XDG_CACHE_HOME and rocFFT's user kernel database (ROCFFT_RTC_CACHE_PATH) in an empty temporary
    directory: what the first call on a machine that has never run the size costs; `warm_cache` -- once more against the
    database the first child left: what every later `impproc migrate` process pays (the library points rocFFT at a
    persistent user database by itself when the user has set none).  The record of a path is its cold child's, with the
    warm-cache child's beside it."""
    import subprocess
    import tempfile
    out = {}
    for kind in ('kirch', 'stolt', 'phsh'):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, HOME=tmp, XDG_CACHE_HOME=os.path.join(tmp, 'xdg'),
                       ROCFFT_RTC_CACHE_PATH=os.path.join(tmp, 'rocfft_kernel_cache.db'))
            recs = []
            for tag in ('cold', 'warm_cache'):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--first-call', kind], capture_output=True,
                                       text=True, timeout=240, env=env)
                    line = [l for l in r.stdout.splitlines() if l.startswith('{"kind"')]
                    recs.append(json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]})
                except Exception as exc:
                    recs.append({"error": "%s: %s" % (type(exc).__name__, exc)})
            out[kind] = dict(recs[0], kernel_cache="cold (empty HOME / ROCFFT_RTC_CACHE_PATH)",
                             warm_cache={k: v for k, v in recs[1].items() if k != 'kind'})
    return out


# ---------------------------------------------------------------------------------------------------------

# keys whose values are prose: explained once in profiles/bench_record_keys.md, not printed with every run (the driver keeps the
# last 8 KB of stdout: round 5's 14.7 KB line lost config 2, the one-shot calls and the first calls)
VALU_UNIT = "share of vector issue slots busy"
PROSE_KEYS = ('note', 'traffic_source', 'parity_vs', 'source', 'what', 'method')
SUBRECORD_DROPS = ('workload', 'parity_wavenumbers', 'steps_executed', 'rotate_accumulate_steps', 'algorithmic_flop', 'kernel_cache', 'import_ms',
                   'kind', 'finite', 'fused_floor_bytes', 'algorithmic_bytes', 'fetch_bytes_if_all_16B_per_lane', 'fetch_raw_over_write',
                   'valu_instructions', 'kernel_cycles', 'parity_cols', 'kernel_ms_over_hbm_floor', 'x_fp32_vector_peak')


SUBRECORD_DROPS_EXTRA = []


def compact_record(o, depth=0):
    """The record without its prose, floats to 5 significant digits; the keys stay."""
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in PROSE_KEYS:
                continue
            if (depth > 1 and k in SUBRECORD_DROPS) or (depth > 0 and k in SUBRECORD_DROPS_EXTRA):     # sub-records: constants of the workload (profiles/bench_record_keys.md)
                continue
            if depth > 1 and k == 'unit' and v == VALU_UNIT:          # (the unit of every "valu issue" roofline: profiles/bench_record_keys.md)
                continue
            if k == 'sample' and isinstance(v, str) and len(v) > (90 if depth < 2 else 44):
                v = v[:(87 if depth < 2 else 41)] + '...'
            out[k] = compact_record(v, depth + 1)
        return out
    if isinstance(o, (list, tuple)):
        return [compact_record(v, depth + 1) for v in o]
    if isinstance(o, float):
        return float('%.5g' % o) if np.isfinite(o) else None
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--tnum', type=int, default=10000)
    ap.add_argument('--snum', type=int, default=4096)
    ap.add_argument('--mode', default='fast', choices=['fast', 'exact', 'auto'])
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'],
                    help='f32 (default, BASELINE config 3); f64 with --mode exact runs the float64 parity path')
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='strong (default): the BASELINE radargram is fixed; weak: --tnum traces PER GPU')
    ap.add_argument('--exchange', default='auto', choices=['auto', 'halo', 'allgather'])
    ap.add_argument('--data', default='synthetic', choices=['synthetic', 'noise'])
    ap.add_argument('--no-cpu', action='store_true', help='skip the host-CPU legs (and the parity check that needs them)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes (roofline.traffic = null)')
    ap.add_argument('--no-paths', action='store_true', help='skip the config-2 / config-5 sub-records')
    ap.add_argument('--no-e2e', action='store_true', help='skip the PCIe-inclusive one-shot figure')
    ap.add_argument('--cpu-budget', type=float, default=15.0)
    ap.add_argument('--first-call', default=None, choices=['kirch', 'stolt', 'phsh'], help=argparse.SUPPRESS)
    ap.add_argument('--spot-oracle', nargs=2, default=None, metavar=('KIND', 'OUT'), help=argparse.SUPPRESS)
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--pmc-child-path', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--pmc-child-ps', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--full-line', action='store_true', help='print the record with its prose notes (default: numbers only, so that the line fits the driver\'s 8 KB tail; the keys are explained in profiles/bench_record_keys.md)')
    ap.add_argument('--data-child', default='zeros', choices=['zeros', 'synthetic'], help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.spot_oracle:
        spot_oracle_child(*args.spot_oracle)
        return
    if args.pmc_child_path:
        pmc_child_path(args.pmc_child_path, args.steps)
        return
    if args.pmc_child_ps:
        pmc_child_ps(args.pmc_child_ps)
        return
    if args.first_call:
        first_call_child(args.first_call)
        return

    # ---- `python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything touches the GPU
    # (the ranks are fresh child processes; this parent never loads the HIP library)
    if args.gpus > 1 and 'RANK' not in os.environ:
        from impdar_amd import parallel
        codes = parallel.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.exit(max(abs(c) for c in codes))

    # stdout must carry exactly ONE JSON line, but RCCL prints banners on fd 1 (in the C stdio buffer until
    # exit): point fd 1 at stderr for the whole run and keep the real stdout for the result line
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world

    # the FIRST call of a process (what `impproc migrate file.mat` is), measured in fresh child processes started
    # before this one touches the GPU
    first_calls = None
    if rank == 0 and world == 1 and not args.no_e2e and not args.pmc_child and not os.environ.get('IMPDAR_BENCH_FORCE_DIST'):
        t0 = time.time()
        first_calls = first_call_records()
        log('[bench] first-call children took %.1f s' % (time.time() - t0))
    # the CPU oracle of the 8192^2 phase-shift sub-records' spot wavenumbers: child processes beside everything below
    spot = None
    if rank == 0 and world == 1 and not args.no_paths and not args.no_cpu and not args.pmc_child:
        spot = start_spot_oracles()

    from impdar_amd import _hip, parallel, synth
    np_dtype = np.float32 if args.dtype == 'f32' else np.float64

    # IMPDAR_BENCH_FORCE_DIST=1 runs the whole multi-rank code path (rendezvous, unique id, RCCL communicator,
    # exchange) with a single rank, for 1-GPU boxes
    multi = world > 1 or os.environ.get('IMPDAR_BENCH_FORCE_DIST') == '1'
    rdv = parallel.Rendezvous(rank, world)

    _hip.load()
    ndev = _hip.device_count()
    if ndev <= 0:
        sys.exit('bench.py: no HIP device visible; the HIP path has no CPU fallback')
    ctx = _hip.context(local % ndev)
    if multi:
        parallel.init_communicator(ctx, rdv)
        assert 'torch' not in sys.modules

    snum, tnum, vel = args.snum, args.tnum * (world if args.scaling == 'weak' else 1), 1.69e8
    geo = synth.geometry(snum, tnum)
    sk = parallel.ShardedKirchhoff(ctx, snum, tnum, geo['dist'], geo['travel_time'], vel, rank, world, np_dtype,
                                   False, args.mode, args.exchange)
    plan = sk.engine.plan
    jlo, jhi, xlo, xhi, nloc = sk.jlo, sk.jhi, sk.xlo, sk.xhi, sk.nloc
    if args.exchange == 'halo' and world == 1:
        sk.xplan['mode'] = 'halo'                   # single-rank plumbing check: an empty grouped exchange

    t0 = time.time()
    full_data = None
    if args.pmc_child and args.data_child == 'zeros':
        local_data = np.zeros((snum, max(nloc, 1)), dtype=np_dtype)
    elif args.data == 'synthetic':
        # rank 0 builds the whole radargram when it needs it for the CPU legs / parity anyway
        need_full = rank == 0 and not args.no_cpu and not args.pmc_child
        lo, hi = (0, tnum) if need_full else (jlo, jhi)
        blk = synth.diffractor_radargram(snum, tnum, vel=vel, dtype=np_dtype, trace_lo=lo, trace_hi=hi, chunk=128,
                                         threads=host_threads())
        if need_full:
            full_data, local_data = blk, np.ascontiguousarray(blk[:, jlo:jhi])
        else:
            local_data = blk
    else:
        local_data = np.random.default_rng(rank).standard_normal((snum, max(nloc, 1))).astype(np_dtype)
    if rank == 0:
        log('[bench] rank0 input %s built in %.1f s (%d host threads)' % (local_data.shape, time.time() - t0, host_threads()))

    d_in = _hip.DeviceArray.from_host(ctx, local_data if local_data.size else np.zeros((snum, 1), np_dtype))
    d_out = _hip.DeviceArray(ctx, (snum, max(xhi - xlo, 1)), np_dtype)

    def step():
        sk.step(d_in, d_out, multi)

    def fence():
        plan.sync()
        rdv.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    # HIP-event durations recorded on the launch streams DURING the timed steps
    # (the plan keeps a 64-step ring of event pairs; nothing synchronised in the loop)
    hist = [plan.history_ms(back) for back in range(min(args.steps, 64))]
    prep_ms, gather_ms, kernel_ms = ([h[i] for h in hist] for i in range(3))
    elapsed = rdv.allreduce_max(elapsed)
    if args.pmc_child:
        plan.destroy()
        return

    # one radargram on an idle device: nothing to hide the exchange behind (the timed loop pipelines it under the
    # previous radargram's diffraction sum)
    fence()
    t1 = time.perf_counter()
    step()
    fence()
    single_ms = rdv.allreduce_max((time.perf_counter() - t1) * 1e3)
    s_prep, s_gather, s_mig = plan.history_ms(0)

    out_host = d_out.to_host()[:, :xhi - xlo]
    finite = bool(np.isfinite(out_host).all())

    # ---- what every rank did, gathered on rank 0: its block, the rows RCCL handed it, its own event times, four
    # columns of its timed output (rank 0 holds them to the C oracle below), and what RCCL itself says about the
    # communicator -- an N > 1 record then proves on its own that N ranks exchanged and migrated
    import ctypes as C
    rccl = None
    if multi:
        vals = [C.c_int(-1) for _ in range(4)]
        if _hip.load().impdar_comm_info(ctx, *[C.byref(v) for v in vals]) == 0:
            rccl = dict(ranks=vals[0].value, rank=vals[1].value, device=vals[2].value, version=vals[3].value)
    pcols = np.unique(np.linspace(xlo, max(xhi - 1, xlo), 4).round().astype(np.int64)) if xhi > xlo else np.zeros(0, np.int64)
    mine = dict(rank=rank, device=local % ndev, output_block=[int(xlo), int(xhi)], input_shard=[int(jlo), int(jhi)],
                rows_received=int(sk.xplan['rows_received'][rank]) if world > 1 else 0,
                prep_ms=float(np.mean(prep_ms)) if prep_ms else None,
                exchange_ms=float(np.mean(gather_ms)) if gather_ms else None,
                migrate_ms=float(np.mean(kernel_ms)) if kernel_ms else None,
                pairs=int(sk.pairs[rank]), rccl=rccl, finite=finite,
                cols=[int(c) for c in pcols],
                colvals=np.ascontiguousarray(out_host[:, pcols - xlo], dtype=np.float64) if len(pcols) else np.zeros((snum, 0)))
    everyone = rdv.allgather(mine)

    res = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = tnum * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float('nan')
        esz = np.dtype(np_dtype).itemsize
        pairs0 = sk.pairs[0]
        algo_bytes = pairs0 * esz + snum * (xhi - xlo) * esz       # one element per in-aperture pair + output
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
        kname = plan.kernel
        res = {
            "metric": "migrated traces/sec + achieved HBM GB/s, Kirchhoff 10000x4096 radargram",
            "value": value, "unit": "traces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": args.data,
            "config": {"workload": "Kirchhoff diffraction-sum migration, %d traces x %d samples, constant velocity "
                                   "1.69e8 m/s, dx 1 m, dt 10 ns (BASELINE config 3)" % (tnum, snum),
                       "kernel": plan.mode, "pairs_total": int(sum(sk.pairs)), "pairs_rank0": int(pairs0),
                       "output_block_rank0": [int(xlo), int(xhi)],
                       "parallelism": "output-trace blocks x%d, exchange of input-image rows: %s" % (world, sk.xplan['mode']),
                       "exchange": {"mode": sk.xplan['mode'],
                                    "rows_received_max": int(max(sk.xplan['rows_received'])) if world > 1 else 0,
                                    "rows_allgather": int(sk.xplan['rows_allgather'])},
                       "prep_ms": float(np.mean(prep_ms)) if prep_ms else None,
                       "exchange_ms": float(np.mean(gather_ms)) if gather_ms else None,
                       "single_radargram": {"wall_ms": single_ms, "prep_ms": s_prep, "exchange_ms": s_gather,
                                            "migrate_ms": s_mig,
                                            "note": "one radargram on an idle device: prep + exchange are exposed here; "
                                                    "the timed steps overlap them with the previous diffraction sum"},
                       "rccl": ({"ranks": everyone[0]['rccl']['ranks'], "version": everyone[0]['rccl']['version'],
                                 "ranks_seen_by_every_rank": [e['rccl']['ranks'] if e['rccl'] else None for e in everyone]}
                                if everyone[0]['rccl'] else None),
                       "ranks": [{k: e[k] for k in ('rank', 'device', 'output_block', 'input_shard', 'rows_received',
                                                    'prep_ms', 'exchange_ms', 'migrate_ms', 'pairs', 'finite')}
                                 for e in everyone],
                       "output_finite": bool(all(e['finite'] for e in everyone))},
            # the binding roof: every pair's sample is one 4-byte lane of a ds_read_b128, so the LDS read port
            # (256 B/clk/CU) bounds the diffraction sum; HBM only sees the compulsory image + output bytes
            "roofline": {"bound": "lds", "achieved": achieved, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / LDS_PEAK_GBS) if achieved else None, "traffic": None,
                         "kernel": kname, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
                         "peak_note": "256 B/clk/CU x 256 CUs x 2.4 GHz spec clock (MI355X_MICROARCH.md, LDS); the loop "
                                      "holds ~2.0-2.1 GHz under load",
                         "hbm_algorithmic": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "x_peak": (achieved / HBM_PEAK_GBS) if achieved else None,
                                             "note": "SURVEY 8(d) contract figure: algorithmic gather bytes over the HBM "
                                                     "peak. A multiple of the peak, not a fraction: the samples are "
                                                     "re-read from LDS, not from HBM"}},
        }
        if not finite:
            res["error"] = "non-finite values in the migrated output"

    # ---- CPU legs + in-run parity of the timed output (rank 0)
    if rank == 0 and not args.no_cpu and args.data == 'synthetic':
        t0 = time.time()
        if world == 1:
            rec, cols, want = cpu_baseline(full_data, geo, vel, tnum, args.cpu_budget)
            res["cpu_baseline"] = rec
            res["cpu_numpy_1core"] = cpu_numpy_1core(full_data, geo, vel, tnum)
        else:
            from oracle import c_oracle
            cols = np.unique(np.linspace(xlo, xhi - 1, 32).round().astype(np.int32))
            want = c_oracle.kirchhoff(full_data, geo['travel_time'], geo['dist'], vel, False, traces=cols)
            res["cpu_baseline"] = None
        got = out_host[:, cols - xlo].astype(np.float64)
        rel = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))
        res["parity_rel_l2"] = rel
        res["parity_cols"] = int(len(cols))
        res["parity_bar"] = PARITY_BAR if args.dtype == 'f32' else 1e-12
        if not rel <= res["parity_bar"]:
            res["error"] = "timed output differs from the oracle: rel L2 %.3e on %d columns" % (rel, len(cols))
        # every rank's own four columns against the C oracle
        from oracle import c_oracle as _co
        for e, rec in zip(everyone, res["config"]["ranks"]):
            if not e['cols']:
                rec["parity_rel_l2"] = None
                continue
            w = _co.kirchhoff(full_data, geo['travel_time'], geo['dist'], vel, False, traces=np.asarray(e['cols'], dtype=np.int32))
            r = float(np.linalg.norm(np.asarray(e['colvals']) - w) / max(np.linalg.norm(w), 1e-300))
            rec["parity_cols"] = e['cols']
            rec["parity_rel_l2"] = r
            if not r <= res["parity_bar"]:
                res["error"] = "rank %d: timed output differs from the oracle: rel L2 %.3e" % (e['rank'], r)
        log('[bench] cpu legs + parity took %.1f s (parity rel L2 %.2e on %d columns)' % (time.time() - t0, rel, len(cols)))
    elif rank == 0:
        res["cpu_baseline"] = None

    plan.destroy()
    d_in.free()
    d_out.free()

    if rank == 0 and world == 1 and not os.environ.get('IMPDAR_BENCH_FORCE_DIST'):
        # ---- PCIe-inclusive one-shot figure (never `value`): RadarData.migrate('kirch') on host arrays
        if not args.no_e2e and full_data is not None:
            import contextlib
            import io
            from impdar_amd.lib.RadarData import RadarData
            walls = []
            for _ in range(4):
                d = RadarData(None)
                d.data, d.snum, d.tnum = full_data, snum, tnum
                d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    d.migrate('kirch', vel=vel)
                walls.append(time.perf_counter() - t0)
            e2e = float(np.median(walls[1:]))
            # ... and on host arrays the runtime has not seen before (a radargram just read from a file): the first
            # upload from an address range also pins its pages.  Three distinct copies, all kept alive so that no
            # address range is handed out twice.
            fresh, keep = [], []
            for _ in range(3):
                keep.append(np.array(full_data))
                d = RadarData(None)
                d.data, d.snum, d.tnum = keep[-1], snum, tnum
                d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    d.migrate('kirch', vel=vel)
                fresh.append(time.perf_counter() - t0)
            del keep
            res["end_to_end"] = {"value": tnum / e2e, "unit": "traces/s", "wall_ms": e2e * 1e3,
                                 "wall_ms_fresh_array": float(np.median(fresh)) * 1e3,
                                 "note": "RadarData.migrate('kirch') on a host float32 array: plan + tables + H2D + prep + "
                                         "diffraction sum + D2H + widening to the float64 the reference returns; median of 3 "
                                         "calls on the same array (its pages stay registered with the runtime after the first "
                                         "upload); wall_ms_fresh_array: median of 3 calls on arrays uploaded for the first time; "
                                         "first_call_ms: the first call of a fresh process after its context exists "
                                         "(first_call has the three paths, with import and context creation beside them)"}
            # what the link alone allows: the (snum, tnum) float32 image up and down at the measured
            # 56.5 GB/s of this host's PCIe (profiles/r03_h2d_probe.txt: pageable 164 MB in 2.90 ms up, 2.93 ms down) around
            # the resident step.  sequential = up + step + down; pipelined = the first launch's share of the input (45 %:
            # its output block + the aperture), the step, the last output block (30 %) down.  The float64 widening on the
            # host (the reference returns float64) rides on the download threads.
            link_gbs = 56.5
            up_ms = snum * tnum * 4 / link_gbs / 1e6
            res["end_to_end"]["floor"] = {"link_GBps": link_gbs, "h2d_ms": up_ms, "d2h_ms": up_ms, "step_ms": res["ms_per_step"],
                                          "sequential_ms": 2 * up_ms + res["ms_per_step"],
                                          "pipelined_ms": 0.45 * up_ms + res["ms_per_step"] + 0.30 * up_ms,
                                          "wall_over_pipelined": e2e * 1e3 / (0.75 * up_ms + res["ms_per_step"]),
                                          "source": "profiles/r03_h2d_probe.txt"}
            if first_calls:
                res["end_to_end"]["first_call_ms"] = first_calls.get('kirch', {}).get('first_call_ms')
                res["end_to_end"]["first_call"] = first_calls
        # ---- HBM-side traffic of the dominant kernel, counted in child runs of this command
        if not args.no_pmc and kname in ('kirch_quad_kernel', 'kirch_dquad_kernel'):
            t0 = time.time()
            traffic, raw, note = pmc_traffic(args, kname)
            res["roofline"]["traffic"] = traffic
            res["roofline"]["traffic_source"] = note
            if traffic:
                compulsory = 2 * snum * tnum * esz
                res["roofline"]["fabric"] = {"bytes": traffic, "fetch_size_kib_raw": raw['FETCH_SIZE'],
                                             "write_size_kib_raw": raw['WRITE_SIZE'],
                                             "GBps": traffic / (k_ms * 1e-3) / 1e9,
                                             "frac_of_hbm": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "x_compulsory": traffic / compulsory,
                                             "note": "L2 <-> fabric bytes (Infinity-Cache hits included); compulsory = "
                                                     "image read once + output written once"}
            log('[bench] counter passes took %.1f s: %s' % (time.time() - t0, traffic))
        # ---- the other two single-GPU BASELINE configs, short driver-timed runs
        if not args.no_paths:
            t0 = time.time()
            try:
                res["paths"] = path_records(args.no_cpu, args.no_pmc, full_data, geo, spot)
            except Exception as exc:                      # a sub-record must not take the headline down
                res["paths"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            log('[bench] config-2 / config-5 sub-records took %.1f s' % (time.time() - t0))

    if rank == 0:
        line = json.dumps(res if args.full_line else compact_record(res), separators=(',', ':'))
        if not args.full_line:
            # (the driver keeps 8192 bytes: what goes first if the line still does not fit)
            for drop in ('device_ms_calls', 'warm_cache', 'floor', 'ranks', 'cpu_numpy_1core'):
                if len(line) <= 7800:
                    break
                SUBRECORD_DROPS_EXTRA.append(drop)
                line = json.dumps(compact_record(res), separators=(',', ':'))
        log('[bench] result line: %d bytes' % len(line))
        os.write(result_fd, (line + '\n').encode())
    rdv.barrier()
    rdv.close()
    if rank == 0 and res.get("error"):
        log('[bench] FAILED: ' + res["error"])
        sys.exit(1)


if __name__ == '__main__':
    main()
