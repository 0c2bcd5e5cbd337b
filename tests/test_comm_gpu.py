"""RCCL plumbing on one GPU: communicator bootstrap from a unique id, the
in-place all-gather of the image and the barrier (with one rank the
collectives are trivial, but every RCCL call of the multi-GPU path runs).
The N > 1 data movement itself is covered on CPU by test_parallel_gloo.py;
the driver's 8-GPU bench exercises it on hardware."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_and_allgather(hip):
    lib = hip.load()
    # a private context so the process-wide one stays communicator-free
    ctx = C.c_void_p()
    hip.check(lib.impdar_ctx_create(0, C.byref(ctx)), 'ctx')
    buf = C.create_string_buffer(hip.UNIQUE_ID_BYTES)
    hip.check(lib.impdar_comm_unique_id(buf), 'unique_id')
    assert any(b != 0 for b in buf.raw)
    hip.check(lib.impdar_comm_init(ctx, buf.raw, 0, 1), 'comm_init')
    assert lib.impdar_comm_rank(ctx) == 0 and lib.impdar_comm_size(ctx) == 1
    with pytest.raises(ValueError):
        hip.check(lib.impdar_comm_init(ctx, buf.raw, 0, 1), 'comm_init twice')
    hip.check(lib.impdar_comm_barrier(ctx), 'barrier')

    from impdar_amd import synth, _hip
    from impdar_amd.kirchhoff import KirchhoffPlan, migrate_resident
    snum, tnum = 512, 130
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=2).astype(np.float32)
    want, _, _ = migrate_resident(hip.context(), data, geo['dist'], geo['travel_time'], mode='fast')
    plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], mode='fast', nranks=1)
    d_in = _hip.DeviceArray.from_host(ctx, data)
    d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
    plan.prep(d_in, tnum, 0, tnum)
    plan.allgather()                       # ncclAllGather, in place, one rank
    plan.migrate(d_out, 0, tnum)
    plan.sync()
    got = d_out.to_host()
    prep_ms, gather_ms, mig_ms = plan.last_ms()
    assert gather_ms >= 0.0 and mig_ms > 0.0
    plan.destroy()
    d_in.free()
    d_out.free()
    lib.impdar_ctx_destroy(ctx)
    assert np.array_equal(got, want)


def test_plan_rank_mismatch_is_an_error(hip):
    from impdar_amd import synth
    from impdar_amd.kirchhoff import KirchhoffPlan
    geo = synth.geometry(64, 16)
    plan = KirchhoffPlan(hip.context(), np.float32, 64, 16, geo['dist'], geo['travel_time'], mode='fast', nranks=2)
    assert plan.tnum_pad == 16
    with pytest.raises(ValueError):
        plan.allgather()                   # context has no 2-rank communicator
    plan.destroy()


def test_halo_exchange_entry_point_single_rank(hip):
    """impdar_kirch_exchange on a 1-rank communicator: a grouped ncclSend/ncclRecv to itself moves image rows
    [a, b) onto rows [c, d) (the call path of the multi-GPU halo exchange; N > 1 movement: test_parallel_gloo.py).
    Checked through the result: with the received rows being a copy of the right rows, migrating from the
    patched image equals migrating a radargram whose columns were patched the same way."""
    lib = hip.load()
    ctx = C.c_void_p()
    hip.check(lib.impdar_ctx_create(0, C.byref(ctx)), 'ctx')
    buf = C.create_string_buffer(hip.UNIQUE_ID_BYTES)
    hip.check(lib.impdar_comm_unique_id(buf), 'unique_id')
    hip.check(lib.impdar_comm_init(ctx, buf.raw, 0, 1), 'comm_init')
    from impdar_amd import synth, _hip
    from impdar_amd.kirchhoff import KirchhoffPlan, migrate_resident
    snum, tnum = 512, 136
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=5).astype(np.float32)
    for mode in ('fast', 'exact'):
        patched = data.copy()
        patched[:, 96:120] = data[:, 16:40]
        want, _, _ = migrate_resident(hip.context(), patched, geo['dist'], geo['travel_time'], mode=mode)
        plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], mode=mode, nranks=1)
        d_in = _hip.DeviceArray.from_host(ctx, data)
        d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
        plan.prep(d_in, tnum, 0, tnum)
        plan.exchange([(0, 16, 40)], [(0, 96, 120)])
        plan.migrate(d_out, 0, tnum)
        plan.sync()
        got = d_out.to_host()
        _, x_ms, _ = plan.last_ms()
        assert x_ms >= 0.0
        with pytest.raises(ValueError):
            plan.exchange([(0, 3, 40)], [(0, 96, 133)])          # not whole 8-trace groups
        with pytest.raises(ValueError):
            plan.exchange([(1, 16, 40)], [(0, 96, 120)])         # peer outside the communicator
        plan.destroy()
        d_in.free()
        d_out.free()
        assert np.array_equal(got, want), mode
    lib.impdar_ctx_destroy(ctx)


def test_sharded_orchestration_single_rank(hip):
    """parallel.migrate_kirchhoff_sharded with one rank = the whole radargram through prep -> (no exchange) ->
    migrate of the product's multi-GPU entry point."""
    from impdar_amd import parallel, synth
    from impdar_amd.kirchhoff import migrate_resident
    snum, tnum = 256, 90
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=6).astype(np.float32)
    want, _, _ = migrate_resident(hip.context(), data, geo['dist'], geo['travel_time'], mode='auto')
    xlo, xhi, blk = parallel.migrate_kirchhoff_sharded(
        data, dict(snum=snum, tnum=tnum, dist=geo['dist'], travel_time=geo['travel_time']),
        rdv=parallel.Rendezvous(0, 1))
    assert (xlo, xhi) == (0, tnum) and np.array_equal(blk, want)


def test_rccl_is_the_rocm_install(hip):
    """The library is linked against /opt/rocm's RCCL; a torch wheel's older librccl under the same SONAME must
    not be what serves it (first mapped wins)."""
    hip.load()
    hip.require_system_rccl()
    paths = hip.mapped_rccl()
    assert paths and all('/opt/rocm' in p for p in paths), paths


def test_single_process_front_door_spawns_its_ranks(hip, tmp_path):
    """parallel.run_sharded / migrationKirchhoff(dat, ngpus=N): the radargram goes to shared memory, one worker
    process per GPU (impdar_amd._shard_worker: rendezvous, communicator, prep -> exchange -> migrate of its block)
    writes its output block into the shared result.  One GPU here, so one rank -- the N > 1 data movement is what
    tests/test_parallel_gloo.py covers -- but the whole process plumbing runs."""
    from impdar_amd import parallel, synth
    from impdar_amd.kirchhoff import migrate_resident
    snum, tnum = 300, 77
    geo = synth.geometry(snum, tnum)
    for dtype in (np.float32, np.float64):
        data = synth.noise_radargram(snum, tnum, seed=9).astype(dtype)
        want, _, _ = migrate_resident(hip.context(), data, geo['dist'], geo['travel_time'], mode='auto')
        got = parallel.run_sharded(data, geo['dist'], geo['travel_time'], ngpus=1)
        assert got.dtype == np.float64 and got.shape == (snum, tnum)
        assert np.array_equal(got, want.astype(np.float64))
    leftovers = [f for f in os.listdir('/dev/shm') if f.startswith('impdar_shard_')]
    assert not leftovers
