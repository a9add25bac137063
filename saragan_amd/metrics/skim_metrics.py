"""SURFGAN_3D/metrics/skim_metrics.py:8-45 on the GPU, on the library's own kernels (csrc/metrics.hip through the C ABI).
The reference calls scikit-image (skimage.metrics); the same published definitions are evaluated here in float64 on
device buffers: mean_squared_error (`sg_sqdiff_mean`), normalized_root_mse (min-max, `sg_minmax`),
peak_signal_noise_ratio and structural_similarity with gaussian_weights=True, multichannel=True (sigma 1.5,
truncate 3.5 -> 11 taps, scipy 'reflect' borders: `sg_filter_axis` per spatial axis over x, y, x*x, y*y, x*y from
`sg_ssim_products`; sample covariance, K1 0.01, K2 0.03, borders of 5 cropped: `sg_ssim_mean`).  torch only holds the
device buffers and converts the input dtype."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib


def _dev(x):
    if torch.is_tensor(x):
        if not x.is_cuda:
            raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
        return x.to(torch.float64)
    if not torch.cuda.is_available():
        raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
    return torch.as_tensor(np.asarray(x), device='cuda').to(torch.float64)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(device):
    nbytes = _lib.load().sg_metric_workspace()
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def get_mean_squared_error(real, fake):
    a, b = _dev(real).contiguous(), _dev(fake).contiguous()
    assert a.shape == b.shape
    ws, nbytes = _workspace(a.device)
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().sg_sqdiff_mean(_p(a), _p(b), _p(out), a.numel(), _p(ws), nbytes, _st()), 'sg_sqdiff_mean')
    return float(out[0])


def get_normalized_root_mse(real, fake):
    r = _dev(real).contiguous()
    ws, nbytes = _workspace(r.device)
    out = torch.empty(2, dtype=torch.float64, device=r.device)
    _lib.check(_lib.load().sg_minmax(_p(r), _p(out), r.numel(), _p(ws), nbytes, _st()), 'sg_minmax')
    lo, hi = out.tolist()
    return float(math.sqrt(get_mean_squared_error(r, fake)) / (hi - lo))


def get_psnr(real, fake, data_range=3072):
    return float(10.0 * math.log10(data_range ** 2 / get_mean_squared_error(real, fake)))


def _gauss_taps(sigma=1.5, truncate=3.5):
    """scipy.ndimage.gaussian_filter1d's kernel."""
    r = int(truncate * sigma + 0.5)
    k = np.arange(-r, r + 1, dtype=np.float64)
    wgt = np.exp(-0.5 / (sigma * sigma) * k ** 2)
    return wgt / wgt.sum(), r


def _gauss_filter(v, nd, taps_c, ntaps):
    """`nd` leading spatial axes of a channels-last f64 tensor, scipy 'reflect' borders (d c b a | a b c d | d c b a)."""
    lib = _lib.load()
    shape = list(v.shape)
    for d in range(nd):
        outer = int(np.prod(shape[:d], dtype=np.int64))
        inner = int(np.prod(shape[d + 1:], dtype=np.int64))
        y = torch.empty_like(v)
        _lib.check(lib.sg_filter_axis(_p(v), _p(y), None, outer, shape[d], inner, taps_c, ntaps, 0, 1, 1.0, 1, _st()),
                   'sg_filter_axis')
        v = y
    return v


def _ssim_channels_last(x, y, data_range):
    """x, y: [*spatial, C] float64 -> mean over channels of the cropped SSIM map."""
    lib = _lib.load()
    x, y = x.contiguous(), y.contiguous()
    nd = x.dim() - 1
    assert nd in (2, 3) and x.shape == y.shape
    taps, r = _gauss_taps()
    taps_c = (C.c_double * len(taps))(*taps)
    NP = (2 * r + 1) ** nd
    cov_norm = NP / (NP - 1)
    xx, yy, xy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.sg_ssim_products(_p(x), _p(y), _p(xx), _p(yy), _p(xy), x.numel(), _st()), 'sg_ssim_products')
    ux, uy, uxx, uyy, uxy = (_gauss_filter(v, nd, taps_c, len(taps)) for v in (x, y, xx, yy, xy))
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = [1] * (3 - nd) + list(x.shape[:nd])
    ws, nbytes = _workspace(x.device)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    _lib.check(lib.sg_ssim_mean(_p(ux), _p(uy), _p(uxx), _p(uyy), _p(uxy), _p(out), s[0], s[1], s[2], x.shape[-1],
                                r if nd == 3 else 0, r, cov_norm, C1, C2, _p(ws), nbytes, _st()), 'sg_ssim_mean')
    return float(out[0])


def get_ssim(real, fake, data_range=3):
    """skim_metrics.py:20-45, quirk included: [N,C,D,H,W] is moved to channels-last and a batch of ONE is squeezed, so
    the per-item loop then runs over its D slices (2-D SSIM per slice); N > 1 gives one 3-D SSIM per volume."""
    real, fake = _dev(real).permute(0, 2, 3, 4, 1), _dev(fake).permute(0, 2, 3, 4, 1)
    if real.shape[0] == 1:
        real = real[0]
    if fake.shape[0] == 1:
        fake = fake[0]
    return [_ssim_channels_last(a, b, data_range) for a, b in zip(real, fake)]
