"""CPU oracle for the download-free validation metrics.  TEST INFRASTRUCTURE ONLY.

* SWD: numpy / scipy restatement of SURFGAN_3D/metrics/swd.py:13-123 (Laplacian pyramid with the 5x5x5 binomial filter,
  random 3x9x9 neighbourhood descriptors, sorted random projections).  All randomness comes from numpy's global
  generator in the reference's call order, so a seeded run reproduces the reference's numbers.
  Pinned by tests/golden/ref_swd.npz: outputs of the reference's own pgan_pytorch/metrics/swd.py (pyr_down,
  sliced_wasserstein identical to the 3-D tree's; its pyr_up uses gain 4 where the 3-D tree uses 8, swd.py:70) run in
  the build container by oracle/make_golden.py.
* MSE / NRMSE / PSNR / SSIM: SURFGAN_3D/metrics/skim_metrics.py:8-45 calls scikit-image (skimage.metrics, not
  installed here and not vendored by the reference; API of scikit-image >= 0.16).  Restated from the published
  algorithms: mean_squared_error = mean((a-b)^2) in float64; normalized_root_mse(normalization='min-max') =
  sqrt(mse) / (max(a) - min(a)); peak_signal_noise_ratio = 10 log10(R^2 / mse); structural_similarity (Wang et al. 2004)
  with gaussian_weights=True: sigma 1.5, truncate 3.5 (11 taps), scipy.ndimage.gaussian_filter mode 'reflect',
  sample covariance (cov_norm = NP / (NP - 1), NP = 11^ndim), K1 0.01, K2 0.03, mean over the image cropped by 5 per
  side, channels averaged.  "parity unpinned" for this block (no runnable scikit-image here)."""
import numpy as np
import scipy.ndimage

_F1 = np.array([1, 4, 6, 4, 1], dtype=np.float32)
_G = _F1[:, None, None] * _F1[None, None, :] * _F1[None, :, None]
GAUSSIAN_FILTER = (_G / _G.sum()).reshape(5, 5, 5)


def get_descriptors_for_minibatch(minibatch, nhood_size, nhoods_per_image):
    """swd.py:13-26."""
    S = minibatch.shape
    assert len(S) == 5
    N = nhoods_per_image * S[0]
    D, H, W = nhood_size[0] // 2, nhood_size[1] // 2, nhood_size[2] // 2
    nhood, chan, d, x, y = np.ogrid[0:N, 0:S[1], -D:D + 1, -H:H + 1, -W:W + 1]
    img = nhood // nhoods_per_image
    d = d + np.random.randint(D, S[2] - D, size=(N, 1, 1, 1, 1))
    x = x + np.random.randint(W, S[4] - W, size=(N, 1, 1, 1, 1))
    y = y + np.random.randint(H, S[3] - H, size=(N, 1, 1, 1, 1))
    idx = (((img * S[1] + chan) * S[2] + d) * S[3] + y) * S[4] + x
    return minibatch.flat[idx]


def finalize_descriptors(desc):
    """swd.py:31-39."""
    if isinstance(desc, list):
        desc = np.concatenate(desc, axis=0)
    assert desc.ndim == 5
    if desc.shape[1] > 1:
        desc -= np.mean(desc, axis=(0, 2, 3, 4), keepdims=True)
        desc /= np.std(desc, axis=(0, 2, 3, 4), keepdims=True)
    return desc.reshape(desc.shape[0], -1)


def sliced_wasserstein(a, b, dir_repeats, dirs_per_repeat):
    """swd.py:44-58."""
    results = []
    for _ in range(dir_repeats):
        dirs = np.random.randn(a.shape[1], dirs_per_repeat)
        dirs /= np.sqrt(np.sum(np.square(dirs), axis=0, keepdims=True))
        dirs = dirs.astype(np.float32)
        pa = np.sort(np.matmul(a, dirs), axis=0)
        pb = np.sort(np.matmul(b, dirs), axis=0)
        results.append(np.mean(np.abs(pa - pb)))
    return np.mean(results)


def pyr_down(minibatch):
    """swd.py:63-66."""
    return scipy.ndimage.convolve(minibatch, GAUSSIAN_FILTER[np.newaxis, np.newaxis, ...], mode='mirror')[:, :, ::2, ::2, ::2]


def pyr_up(minibatch, gain=8.0):
    """swd.py:69-74."""
    S = minibatch.shape
    res = np.zeros((S[0], S[1], S[2] * 2, S[3] * 2, S[4] * 2), minibatch.dtype)
    res[:, :, ::2, ::2, ::2] = minibatch
    return scipy.ndimage.convolve(res, GAUSSIAN_FILTER[np.newaxis, np.newaxis, ...] * gain, mode='mirror')


def generate_laplacian_pyramid(minibatch, num_levels):
    """swd.py:77-82."""
    pyramid = [np.float32(minibatch)]
    for _ in range(1, num_levels):
        pyramid.append(pyr_down(pyramid[-1]))
        pyramid[-2] -= pyr_up(pyramid[-1])
    return pyramid


def get_swd_for_volumes(images1, images2, nhood_size=(2, 8, 8), nhoods_per_image=512, dir_repeats=8, dirs_per_repeat=512):
    """swd.py:92-123."""
    resolutions = []
    res = images1.shape[-1]
    while res >= 16:
        resolutions.append(res)
        res //= 2
    if not resolutions:
        return None
    dr = [get_descriptors_for_minibatch(lv, nhood_size, nhoods_per_image)
          for lv in generate_laplacian_pyramid(images1, len(resolutions))]
    df = [get_descriptors_for_minibatch(lv, nhood_size, nhoods_per_image)
          for lv in generate_laplacian_pyramid(images2, len(resolutions))]
    dr = [finalize_descriptors(d) for d in dr]
    df = [finalize_descriptors(d) for d in df]
    dist = [sliced_wasserstein(a, b, dir_repeats, dirs_per_repeat) for a, b in zip(dr, df)]
    return dist + [np.mean(dist)]


# ---- skimage.metrics restated ------------------------------------------------------------------------
def mean_squared_error(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.mean((a - b) ** 2))


def normalized_root_mse(a, b):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(mean_squared_error(a, b)) / (a.max() - a.min()))


def peak_signal_noise_ratio(a, b, data_range):
    return float(10 * np.log10(data_range ** 2 / mean_squared_error(a, b)))


def _ssim_single(x, y, data_range):
    sigma, truncate = 1.5, 3.5
    r = int(truncate * sigma + 0.5)
    win = 2 * r + 1
    x, y = x.astype(np.float64), y.astype(np.float64)
    filt = lambda v: scipy.ndimage.gaussian_filter(v, sigma, truncate=truncate, mode='reflect')
    NP = win ** x.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = filt(x), filt(y)
    uxx, uyy, uxy = filt(x * x), filt(y * y), filt(x * y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    sl = tuple(slice(r, s - r) for s in S.shape)
    return float(S[sl].mean())


def structural_similarity(im1, im2, data_range):
    """multichannel=True (last axis), gaussian_weights=True."""
    return float(np.mean([_ssim_single(im1[..., c], im2[..., c], data_range) for c in range(im1.shape[-1])]))


def get_ssim(real, fake, data_range=3):
    """skim_metrics.py:20-45: [N,C,D,H,W] -> channels last; N == 1 is squeezed, so the loop then runs over D slices."""
    real, fake = np.transpose(real, [0, 2, 3, 4, 1]), np.transpose(fake, [0, 2, 3, 4, 1])
    if real.shape[0] == 1:
        real = real[0, ...]
    if fake.shape[0] == 1:
        fake = fake[0, ...]
    return [structural_similarity(a, b, data_range) for a, b in zip(real, fake)]
