"""SURFGAN_3D/metrics/skim_metrics.py:8-45 on the GPU.  The reference calls scikit-image (skimage.metrics); the same
published definitions are evaluated here in float64 on device tensors: mean_squared_error, normalized_root_mse
(min-max), peak_signal_noise_ratio and structural_similarity with gaussian_weights=True, multichannel=True (sigma 1.5,
truncate 3.5 -> 11 taps, scipy 'reflect' borders, sample covariance, K1 0.01, K2 0.03, borders of 5 cropped)."""
import math

import numpy as np
import torch


def _dev(x):
    if torch.is_tensor(x):
        return x.to(torch.float64)
    if not torch.cuda.is_available():
        raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
    return torch.as_tensor(np.asarray(x), device='cuda').to(torch.float64)


def get_mean_squared_error(real, fake):
    return float(((_dev(real) - _dev(fake)) ** 2).mean())


def get_normalized_root_mse(real, fake):
    r = _dev(real)
    return float(torch.sqrt(((r - _dev(fake)) ** 2).mean()) / (r.max() - r.min()))


def get_psnr(real, fake, data_range=3072):
    return float(10.0 * math.log10(data_range ** 2 / get_mean_squared_error(real, fake)))


def _gauss_reflect(x, dim, sigma=1.5, truncate=3.5):
    """scipy.ndimage.gaussian_filter1d(mode='reflect'): (d c b a | a b c d | d c b a)."""
    r = int(truncate * sigma + 0.5)
    k = torch.arange(-r, r + 1, device=x.device, dtype=torch.float64)
    wgt = torch.exp(-0.5 * (k / sigma) ** 2)
    wgt = wgt / wgt.sum()
    n = x.shape[dim]
    idx = torch.arange(-r, n + r, device=x.device)
    period = 2 * n
    idx = idx % period
    idx = torch.where(idx >= n, period - 1 - idx, idx)
    xp = x.index_select(dim, idx)
    out = None
    for j in range(2 * r + 1):
        sl = xp.narrow(dim, j, n) * wgt[j]
        out = sl if out is None else out + sl
    return out


def _ssim_channels_last(x, y, data_range):
    """x, y: [*spatial, C] float64 -> mean over channels of the cropped SSIM map."""
    nd = x.dim() - 1
    r = 5

    def filt(v):
        for d in range(nd):
            v = _gauss_reflect(v, d)
        return v
    NP = (2 * r + 1) ** nd
    cov_norm = NP / (NP - 1)
    ux, uy = filt(x), filt(y)
    vx = cov_norm * (filt(x * x) - ux * ux)
    vy = cov_norm * (filt(y * y) - uy * uy)
    vxy = cov_norm * (filt(x * y) - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    for d in range(nd):
        S = S.narrow(d, r, S.shape[d] - 2 * r)
    return float(S.reshape(-1, S.shape[-1]).mean(dim=0).mean())


def get_ssim(real, fake, data_range=3):
    """skim_metrics.py:20-45, quirk included: [N,C,D,H,W] is moved to channels-last and a batch of ONE is squeezed, so
    the per-item loop then runs over its D slices (2-D SSIM per slice); N > 1 gives one 3-D SSIM per volume."""
    real, fake = _dev(real).permute(0, 2, 3, 4, 1), _dev(fake).permute(0, 2, 3, 4, 1)
    if real.shape[0] == 1:
        real = real[0]
    if fake.shape[0] == 1:
        fake = fake[0]
    return [_ssim_channels_last(a, b, data_range) for a, b in zip(real, fake)]
