"""
GPU parity of the fused mixed-radix k <-> R fold (csrc/fold.hip) against numpy's FFT -- the same transform the reference calls
(system/fourier.py:160-177: FFTtoK = np.fft.fftn over the cell axes, FFTtoT = np.fft.ifftn, real part with the imaginary part
checked) -- on meshes that exercise every code path of the kernel: axis lengths 1..16 (fully unrolled DFTs up to 8, the runtime
loop above), column counts that are not a multiple of the 16-column tile, meshes large enough to shrink the tile (8 x 8 x 8:
8 columns, 12 x 12 x 8: 4), real and complex input, real and complex output, the |Im| flag word, and the dense DFT-GEMM the
library keeps for k subsets / axes longer than 16 / DMK_FOLD_FFT=0 (both paths must agree with each other too).
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MESHES = [(6, 6, 6), (4, 4, 4), (6, 1, 1), (1, 5, 1), (2, 3, 1), (3, 7, 2), (8, 8, 8), (12, 12, 8), (16, 9, 1), (13, 2, 11), (10, 15, 1),
          (1, 1, 1), (2, 1, 1), (14, 1, 1)]


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _fft(x, mesh, inverse):
    nk = int(np.prod(mesh))
    y = x.reshape(x.shape[0], *mesh, -1)
    y = np.fft.ifftn(y, axes=(1, 2, 3)) if inverse else np.fft.fftn(y, axes=(1, 2, 3))
    return y.reshape(x.shape[0], nk, -1)


@pytest.mark.parametrize("mesh", MESHES)
def test_fold_kernel_matches_fft(ctx, mesh):
    from libdmet_preview_amd.system import fourier
    nk = int(np.prod(mesh))
    rng = np.random.default_rng(nk + mesh[0])
    for ncol in (1, 7, 16, 37, 130):
        batch = 2 if ncol != 130 else 1
        # R -> k from real and from complex input
        xr = rng.standard_normal((batch, nk, ncol))
        got = fourier.fold_R2k_dev(ctx.to_device(xr), mesh, batch, ncol).get()
        ref = _fft(xr.astype(complex), mesh, False)
        assert np.abs(got - ref).max() <= 2e-13 * max(1.0, np.abs(ref).max()), (mesh, ncol, "R2k real")
        xc = xr + 1j * rng.standard_normal((batch, nk, ncol))
        got = fourier.fold_R2k_dev(ctx.to_device(xc, np.complex128), mesh, batch, ncol).get()
        ref = _fft(xc, mesh, False)
        assert np.abs(got - ref).max() <= 2e-13 * max(1.0, np.abs(ref).max()), (mesh, ncol, "R2k complex")
        # k -> R of the transform of a REAL field: real output, imaginary flag ~ 0
        d_k = ctx.to_device(_fft(xr.astype(complex), mesh, False), np.complex128)
        flag = ctx.zeros((1,), np.float64)
        back = fourier.fold_k2R_dev(d_k, mesh, batch, ncol, imag_max=flag).get()
        assert np.abs(back - xr).max() <= 2e-13 * max(1.0, np.abs(xr).max()), (mesh, ncol, "k2R real")
        assert flag.get()[0] <= 1e-12 * max(1.0, np.abs(xr).max())
        # ... and of a generic complex field: the flag word carries max |Im| of the (scaled) result
        d_c = ctx.to_device(xc, np.complex128)
        flag.zero_()
        back = fourier.fold_k2R_dev(d_c, mesh, batch, ncol, imag_max=flag).get()
        ref = _fft(xc, mesh, True)
        assert np.abs(back - ref.real).max() <= 2e-13 * max(1.0, np.abs(ref).max())
        assert abs(flag.get()[0] - np.abs(ref.imag).max()) <= 1e-12 * max(1.0, np.abs(ref).max())


def test_fold_kernel_and_dft_gemm_agree(ctx):
    """DMK_FOLD_FFT=0 selects the dense DFT-GEMM for full meshes too; a k subset and an axis longer than 16 always use it."""
    from libdmet_preview_amd.system import fourier
    rng = np.random.default_rng(3)
    mesh, ncol = (6, 6, 6), 50
    nk = 216
    x = rng.standard_normal((2, nk, ncol)) + 1j * rng.standard_normal((2, nk, ncol))
    d_x = ctx.to_device(x, np.complex128)
    fast_k = fourier.fold_R2k_dev(d_x, mesh, 2, ncol).get()
    fast_R = fourier.fold_k2R_dev(d_x, mesh, 2, ncol).get()
    os.environ["DMK_FOLD_FFT"] = "0"
    try:
        slow_k = fourier.fold_R2k_dev(d_x, mesh, 2, ncol).get()
        slow_R = fourier.fold_k2R_dev(d_x, mesh, 2, ncol).get()
    finally:
        del os.environ["DMK_FOLD_FFT"]
    assert np.abs(fast_k - slow_k).max() <= 1e-12 * np.abs(slow_k).max()
    assert np.abs(fast_R - slow_R).max() <= 1e-12 * np.abs(slow_R).max()
    # partial fold over a k subset (the multi-rank path): sum of the two halves = the full fold
    sub = [list(range(0, nk, 2)), list(range(1, nk, 2))]
    acc = np.zeros((2, nk, ncol))
    for ks in sub:
        d_part = ctx.to_device(np.ascontiguousarray(x[:, ks]), np.complex128)
        acc += fourier.fold_k2R_dev(d_part, mesh, 2, ncol, k_subset=ks).get()
    assert np.abs(acc - fast_R).max() <= 1e-12 * np.abs(fast_R).max()
    # an axis of 17 cells is outside the kernel: the GEMM path answers
    mesh17 = (17, 2, 1)
    y = rng.standard_normal((1, 34, 9))
    got = fourier.fold_R2k_dev(ctx.to_device(y), mesh17, 1, 9).get()
    assert np.abs(got - _fft(y.astype(complex), mesh17, False)).max() <= 1e-12 * 10
