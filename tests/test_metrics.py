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
