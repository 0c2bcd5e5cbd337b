"""The library's own batched power-of-two row transforms (csrc/own_fft.h), which carry the first Stolt / phase-shift call
of a process while rocFFT's plans are being made, against numpy.fft (the reference's transforms: mig_python.py:159, 202,
270, 282).  float32: relative L2 <= 2e-6; float64: max|diff| <= 1e-13 * max|ref| * log2(n)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('n', [32, 64, 128, 512, 1024, 2048, 4096, 8192, 16384])
def test_row_transforms_against_numpy(hip, n, dtype):
    from impdar_amd import _hip
    lib, ctx = hip.load(), hip.context()
    rng = np.random.default_rng(n)
    batch = 37 if n <= 4096 else 5
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    code = _hip.dtype_code(dtype)

    def run(mode, a, out_shape, out_dtype, scale=1.0, inplace=False):
        d_in = _hip.DeviceArray.from_host(ctx, np.ascontiguousarray(a).view(dtype).reshape(a.shape[0], -1))
        n_out = int(np.prod(out_shape)) * (2 if np.issubdtype(out_dtype, np.complexfloating) else 1)
        d_out = d_in if inplace else _hip.DeviceArray(ctx, (out_shape[0], n_out // out_shape[0]), dtype)
        _hip.check(lib.impdar_fft_rows_dev(ctx, mode, code, n, a.shape[0], d_in.ptr, d_out.ptr, float(scale)), 'impdar_fft_rows_dev')
        got = d_out.to_host().reshape(out_shape[0], -1).view(out_dtype).reshape(out_shape)
        d_in.free()
        if not inplace:
            d_out.free()
        return got

    def close(got, want):
        if dtype == np.float32:
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert err < 2e-6, err
        else:
            err = np.max(np.abs(got - want)) / np.max(np.abs(want))
            assert err < 1e-13 * np.log2(n), err

    if n <= 8192:
        z = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(cdt)
        close(run(0, z, (batch, n), cdt), np.fft.fft(z.astype(np.complex128), axis=1))
        close(run(1, z, (batch, n), cdt, scale=1.0 / n, inplace=True), np.fft.ifft(z.astype(np.complex128), axis=1))
        close(run(4, z, (batch, n), dtype, scale=1.0 / n), np.fft.ifft(z.astype(np.complex128), axis=1).real)      # Re only
    x = rng.standard_normal((batch, n)).astype(dtype)
    X = np.fft.rfft(x.astype(np.float64), axis=1)
    close(run(2, x, (batch, n // 2 + 1), cdt), X)
    Xc = X.astype(cdt)
    close(run(3, Xc, (batch, n), dtype, scale=1.0 / n), np.fft.irfft(Xc.astype(np.complex128), n=n, axis=1))


def test_row_transform_rejects_what_it_cannot_do(hip):
    from impdar_amd import _hip
    lib, ctx = hip.load(), hip.context()
    d = _hip.DeviceArray(ctx, (2, 64), np.float32)
    for mode, n in ((0, 48), (0, 16384), (2, 16), (5, 64)):
        assert lib.impdar_fft_rows_dev(ctx, mode, 0, n, 2, d.ptr, d.ptr, 1.0) != 0
    d.free()
