"""Validation metrics: the CPU oracle against the reference-run golden (tests/golden/ref_swd.npz,
produced by the reference's own pgan_pytorch/metrics/swd.py), and the GPU implementations against the oracle."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as M


def test_oracle_reproduces_reference_swd_pieces(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, 'ref_swd.npz'))
    np.testing.assert_array_equal(M.pyr_down(z['x']), z['down'])
    np.testing.assert_array_equal(M.pyr_up(z['down'], 4.0), z['up_gain4'])
    pyr = M.generate_laplacian_pyramid(z['x'].copy(), 2)
    np.testing.assert_array_equal(pyr[1], z['pyr1'])
    # the 3-D tree's pyr_up has twice the gain of the 2-D-derived file the golden was run from (swd.py:70 vs 4.0)
    np.testing.assert_allclose(pyr[0], z['x'] - 2.0 * (z['x'] - z['pyr0_gain4']), rtol=0, atol=1e-5)
    np.random.seed(11)
    assert M.sliced_wasserstein(z['a'], z['b'], 3, 20) == float(z['swd_seed11'])


def test_oracle_skimage_definitions():
    rng = np.random.default_rng(3)
    a = rng.normal(size=(1, 1, 12, 16, 16))
    assert M.mean_squared_error(a, a) == 0.0
    b = a + 0.5
    assert abs(M.mean_squared_error(a, b) - 0.25) < 1e-12
    assert abs(M.peak_signal_noise_ratio(a, b, 4.0) - 10 * np.log10(16 / 0.25)) < 1e-9
    assert abs(M.normalized_root_mse(a, b) - 0.5 / (a.max() - a.min())) < 1e-12
    s = M.get_ssim(a, a)
    assert len(s) == 12 and all(abs(v - 1.0) < 1e-12 for v in s)       # batch of one: a 2-D SSIM per D slice
    assert all(v < 0.999 for v in M.get_ssim(a, a + rng.normal(size=a.shape) * 0.3))


@pytest.mark.gpu
def test_gpu_swd_matches_oracle():
    from saragan_amd.metrics import swd as G
    rng = np.random.default_rng(5)
    x = rng.normal(size=(4, 1, 8, 32, 32)).astype(np.float32)
    y = (rng.normal(size=(4, 1, 8, 32, 32)) * 1.2 + 0.1).astype(np.float32)
    np.testing.assert_allclose(G.pyr_down(x).cpu().numpy(), M.pyr_down(x), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(G.pyr_up(M.pyr_down(x)).cpu().numpy(), M.pyr_up(M.pyr_down(x)), rtol=1e-5, atol=1e-5)
    for a, b in zip(G.generate_laplacian_pyramid(x, 2), M.generate_laplacian_pyramid(x.copy(), 2)):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-5, atol=1e-5)
    np.random.seed(3)
    want = M.get_descriptors_for_minibatch(x, (2, 8, 8), 16)
    np.random.seed(3)
    got = G.get_descriptors_for_minibatch(x, (2, 8, 8), 16)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.random.seed(9)
    want = M.get_swd_for_volumes(x.copy(), y.copy(), nhoods_per_image=32, dir_repeats=2, dirs_per_repeat=64)   # (in place)
    np.random.seed(9)
    got = G.get_swd_for_volumes(x, y, nhoods_per_image=32, dir_repeats=2, dirs_per_repeat=64)
    assert len(got) == len(want) == 3
    np.testing.assert_allclose(got, want, rtol=1e-4)
    assert G.get_swd_for_volumes(x[..., :8], y[..., :8]) is None


@pytest.mark.gpu
def test_gpu_skim_metrics_match_oracle():
    from saragan_amd.metrics import skim_metrics as G
    rng = np.random.default_rng(8)
    a = (np.clip(rng.normal(size=(2, 1, 12, 24, 24)), -1, 2) * 1024).astype(np.int16)
    b = (np.clip(rng.normal(size=(2, 1, 12, 24, 24)), -1, 2) * 1024).astype(np.int16)
    assert abs(G.get_mean_squared_error(a, b) / M.mean_squared_error(a, b) - 1) < 1e-12
    assert abs(G.get_normalized_root_mse(a, b) / M.normalized_root_mse(a, b) - 1) < 1e-12
    assert abs(G.get_psnr(a, b) - M.peak_signal_noise_ratio(a, b, 3072)) < 1e-9
    np.testing.assert_allclose(G.get_ssim(a, b), M.get_ssim(a, b), rtol=1e-9, atol=1e-12)          # 3-D SSIM per volume
    np.testing.assert_allclose(G.get_ssim(a[:1], b[:1]), M.get_ssim(a[:1], b[:1]), rtol=1e-9, atol=1e-12)   # per slice


@pytest.mark.gpu
def test_gpu_metric_kernels_through_the_c_abi():
    """The pieces behind saragan_amd.metrics called directly (include/saragan_hip.h): the bitonic sort past one LDS chunk and
    with padding, both filter borders against scipy on extents shorter than the filter, channel normalisation, odd extents."""
    import ctypes as C
    import scipy.ndimage
    from saragan_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.default_rng(12)
    # rows of 10000 -> padded to 16384 = four chunks: local sort, global strides, local merges; ties and -0.0 / inf included
    n, rows = 10000, 6
    assert lib.sg_swd_padded_rows(n) == 16384 and lib.sg_swd_padded_rows(1) == 64 and lib.sg_swd_padded_rows(4096) == 4096
    vals = rng.normal(size=(rows, n)).astype(np.float32)
    vals[0, :50] = 0.25
    vals[1, 7] = -0.0
    vals[2, 9] = np.inf
    buf = torch.full((rows, 16384), float('inf'), dtype=torch.float32, device='cuda')
    buf[:, :n] = torch.as_tensor(vals, device='cuda')
    _lib.check(lib.sg_sort_rows(p(buf), rows, 16384, st))
    got = buf.cpu().numpy()
    np.testing.assert_array_equal(got[:, :n], np.sort(vals, axis=1))
    assert np.isinf(got[:, n:]).all()
    small = torch.as_tensor(rng.normal(size=(3, 64)).astype(np.float32), device='cuda')
    want = np.sort(small.cpu().numpy(), axis=1)
    _lib.check(lib.sg_sort_rows(p(small), 3, 64, st))
    np.testing.assert_array_equal(small.cpu().numpy(), want)
    assert lib.sg_sort_rows(p(small), 3, 48, st) == -1                         # not a power of two
    # projection + padding: a [n, f] . dirs [f, nd] with ragged n, f, nd
    n, f, nd = 333, 243, 70
    a = rng.normal(size=(n, f)).astype(np.float32)
    dirs = rng.normal(size=(f, nd)).astype(np.float32)
    npad = lib.sg_swd_padded_rows(n)
    pt = torch.empty((nd, npad), dtype=torch.float32, device='cuda')
    ad, dd = torch.as_tensor(a, device='cuda'), torch.as_tensor(dirs, device='cuda')
    _lib.check(lib.sg_swd_project(p(ad), p(dd), p(pt), n, f, nd, npad, st))
    got = pt.cpu().numpy()
    np.testing.assert_allclose(got[:, :n], (a.astype(np.float64) @ dirs.astype(np.float64)).T, rtol=1e-4, atol=1e-4)
    assert np.isinf(got[:, n:]).all()
    # distance of two sorted buffers
    pa = torch.as_tensor(np.sort(rng.normal(size=(5, 512)), axis=1).astype(np.float32), device='cuda')
    pb = torch.as_tensor(np.sort(rng.normal(size=(5, 512)), axis=1).astype(np.float32), device='cuda')
    out = torch.empty(6, dtype=torch.float64, device='cuda')
    _lib.check(lib.sg_swd_distance(p(pa), p(pb), p(out), 5, 300, 512, st))
    want = np.abs(pa.cpu().numpy()[:, :300] - pb.cpu().numpy()[:, :300]).astype(np.float64).mean()   # f32 differences, as np.abs(pa - pb)
    assert abs(float(out[0]) / want - 1) < 1e-12
    # filter: every mode / border against scipy, on an extent shorter than the filter too
    for nlen in (3, 9, 16):
        x = rng.normal(size=(2, nlen, 5))
        xd = torch.as_tensor(x, device='cuda')
        taps = rng.normal(size=11)
        tc = (C.c_double * 11)(*taps)
        for border, mode_name in ((0, 'mirror'), (1, 'reflect')):
            y = torch.empty_like(xd)
            _lib.check(lib.sg_filter_axis(p(xd), p(y), None, 2, nlen, 5, tc, 11, 0, border, 1.0, 1, st))
            want = scipy.ndimage.correlate1d(x, taps, axis=1, mode=mode_name)
            np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-12, atol=1e-12)
    # pyramid on odd extents (pyr_down keeps ceil(n/2) samples, as [::2] does)
    from saragan_amd.metrics import swd as G
    xo = rng.normal(size=(2, 2, 5, 9, 7)).astype(np.float32)
    np.testing.assert_allclose(G.pyr_down(xo).cpu().numpy(), M.pyr_down(xo), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(G.pyr_up(xo).cpu().numpy(), M.pyr_up(xo), rtol=1e-5, atol=1e-5)
    pyr = M.generate_laplacian_pyramid(rng.normal(size=(1, 1, 4, 16, 16)).astype(np.float32), 2)
    np.testing.assert_allclose(G.reconstruct_laplacian_pyramid([torch.as_tensor(l, device='cuda') for l in pyr]).cpu().numpy(),
                               M.pyr_up(pyr[1]) + pyr[0], rtol=1e-5, atol=1e-5)
    # two channels: finalize_descriptors normalises per channel; the whole distance against the oracle
    x2 = rng.normal(size=(3, 2, 8, 32, 32)).astype(np.float32)
    y2 = (rng.normal(size=(3, 2, 8, 32, 32)) * 0.7 - 0.2).astype(np.float32)
    np.random.seed(4)
    dw = M.finalize_descriptors(M.get_descriptors_for_minibatch(x2, (2, 8, 8), 8))
    np.random.seed(4)
    dg = G.finalize_descriptors(G.get_descriptors_for_minibatch(x2, (2, 8, 8), 8))
    np.testing.assert_allclose(dg.cpu().numpy(), dw, rtol=1e-4, atol=1e-5)
    np.random.seed(6)
    want = M.get_swd_for_volumes(x2.copy(), y2.copy(), nhoods_per_image=700, dir_repeats=1, dirs_per_repeat=40)
    np.random.seed(6)
    got = G.get_swd_for_volumes(x2, y2, nhoods_per_image=700, dir_repeats=1, dirs_per_repeat=40)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    with pytest.raises(RuntimeError):
        G.pyr_down(torch.zeros(1, 1, 4, 4, 4))                                   # CPU tensor: no fallback


@pytest.mark.gpu
def test_gpu_swd_at_the_benchmarked_volume_size():
    """get_swd_for_volumes with the reference's default arguments (3 x 9 x 9 neighbourhoods, 512 per volume, 8 x 512
    directions) on 32 x 128 x 128 volumes: four pyramid levels (128 .. 16), 1536 descriptors per level."""
    from saragan_amd.metrics import swd as G
    rng = np.random.default_rng(21)
    x = rng.normal(size=(3, 1, 32, 128, 128)).astype(np.float32)
    y = (0.8 * x + 0.6 * rng.normal(size=x.shape)).astype(np.float32) + 0.05
    np.random.seed(17)
    want = M.get_swd_for_volumes(x.copy(), y.copy())
    np.random.seed(17)
    got = G.get_swd_for_volumes(x, y)
    assert len(got) == len(want) == 5
    np.testing.assert_allclose(got, want, rtol=2e-4)
    from saragan_amd.metrics import skim_metrics as S
    xi, yi = (x * 300).astype(np.int16), (y * 300).astype(np.int16)
    np.testing.assert_allclose(S.get_ssim(xi[:2], yi[:2]), M.get_ssim(xi[:2], yi[:2]), rtol=1e-9, atol=1e-12)
    assert abs(S.get_psnr(xi, yi) - M.peak_signal_noise_ratio(xi, yi, 3072)) < 1e-9
