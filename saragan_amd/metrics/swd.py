"""SURFGAN_3D/metrics/swd.py:13-123 with the volumes resident on the GPU: Laplacian pyramid (separable binomial filter,
mirror borders), neighbourhood descriptors by one device gather, projections as one matrix product and a device sort per
repeat.  Random numbers (neighbourhood positions, projection directions) are drawn on the host from numpy's global
generator in the reference's call order and sizes, so a seeded run reproduces the reference's result up to float32
summation order.  Function names and defaults are the reference's."""
import numpy as np
import torch

_TAPS = (1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16)      # [1,4,6,4,1]/16 per axis = the 5x5x5 filter / 4096


def _dev(x, device=None):
    if torch.is_tensor(x):
        return x.to(torch.float32)
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
        device = 'cuda'
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=device)


def _binomial(x, dim, gain):
    """5-tap [1,4,6,4,1]/16 * gain along `dim` with scipy's 'mirror' border (d c b | a b c d | c b a)."""
    n = x.shape[dim]
    idx = torch.arange(-2, n + 2, device=x.device)
    idx = torch.where(idx < 0, -idx, idx)
    idx = torch.where(idx > n - 1, 2 * (n - 1) - idx, idx).clamp_(0, n - 1)
    xp = x.index_select(dim, idx)
    out = None
    for k, t in enumerate(_TAPS):
        sl = xp.narrow(dim, k, n) * (t * gain)
        out = sl if out is None else out + sl
    return out


def _gauss3(x, gain=1.0):
    for d in (2, 3, 4):
        x = _binomial(x, d, gain if d == 2 else 1.0)
    return x


def pyr_down(minibatch):
    """swd.py:63-66 (matches cv2.pyrDown per axis)."""
    x = _dev(minibatch)
    assert x.dim() == 5
    return _gauss3(x)[:, :, ::2, ::2, ::2].contiguous()


def pyr_up(minibatch):
    """swd.py:69-74: zero-insertion x2, filter * 8."""
    x = _dev(minibatch)
    assert x.dim() == 5
    n, c, d, h, w = x.shape
    res = torch.zeros((n, c, 2 * d, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    res[:, :, ::2, ::2, ::2] = x
    return _gauss3(res, 8.0)


def generate_laplacian_pyramid(minibatch, num_levels):
    """swd.py:77-82."""
    pyramid = [_dev(minibatch).clone()]
    for _ in range(1, num_levels):
        pyramid.append(pyr_down(pyramid[-1]))
        pyramid[-2] = pyramid[-2] - pyr_up(pyramid[-1])
    return pyramid


def reconstruct_laplacian_pyramid(pyramid):
    """swd.py:85-89."""
    minibatch = pyramid[-1]
    for level in pyramid[-2::-1]:
        minibatch = pyr_up(minibatch) + level
    return minibatch


def get_descriptors_for_minibatch(minibatch, nhood_size, nhoods_per_image):
    """swd.py:13-26: nhoods_per_image random (2D+1) x (2H+1) x (2W+1) neighbourhoods per volume."""
    x = _dev(minibatch)
    S = x.shape
    assert len(S) == 5
    N = nhoods_per_image * S[0]
    D, H, W = nhood_size[0] // 2, nhood_size[1] // 2, nhood_size[2] // 2
    d0 = np.random.randint(D, S[2] - D, size=(N, 1, 1, 1, 1))        # same order and shapes as the reference's draws
    x0 = np.random.randint(W, S[4] - W, size=(N, 1, 1, 1, 1))
    y0 = np.random.randint(H, S[3] - H, size=(N, 1, 1, 1, 1))
    dev = x.device
    t = lambda a: torch.as_tensor(a, device=dev, dtype=torch.int64)
    ar = lambda lo, hi, shape: torch.arange(lo, hi, device=dev, dtype=torch.int64).reshape(shape)
    img = ar(0, N, (N, 1, 1, 1, 1)) // nhoods_per_image
    chan = ar(0, S[1], (1, S[1], 1, 1, 1))
    d = ar(-D, D + 1, (1, 1, 2 * D + 1, 1, 1)) + t(d0)
    xx = ar(-H, H + 1, (1, 1, 1, 2 * H + 1, 1)) + t(x0)            # the reference's `x` grid runs over its 4th axis
    yy = ar(-W, W + 1, (1, 1, 1, 1, 2 * W + 1)) + t(y0)
    idx = (((img * S[1] + chan) * S[2] + d) * S[3] + yy) * S[4] + xx
    return x.reshape(-1)[idx]


def finalize_descriptors(desc):
    """swd.py:31-39."""
    if isinstance(desc, list):
        desc = torch.cat(desc, dim=0)
    assert desc.dim() == 5
    if desc.shape[1] > 1:
        desc = desc - desc.mean(dim=(0, 2, 3, 4), keepdim=True)
        desc = desc / desc.std(dim=(0, 2, 3, 4), keepdim=True, unbiased=False)
    return desc.reshape(desc.shape[0], -1)


def sliced_wasserstein(a, b, dir_repeats, dirs_per_repeat):
    """swd.py:44-58."""
    a, b = _dev(a), _dev(b)
    assert a.dim() == 2 and a.shape == b.shape
    results = []
    for _ in range(dir_repeats):
        dirs = np.random.randn(a.shape[1], dirs_per_repeat)
        dirs /= np.sqrt(np.sum(np.square(dirs), axis=0, keepdims=True))
        dirs = torch.as_tensor(dirs.astype(np.float32), device=a.device)
        pa = torch.sort(a @ dirs, dim=0).values
        pb = torch.sort(b @ dirs, dim=0).values
        results.append((pa - pb).abs().mean())
    return float(torch.stack(results).mean())


def get_swd_for_volumes(images1, images2, nhood_size=(2, 8, 8), nhoods_per_image=512, dir_repeats=8,
                        dirs_per_repeat=512):
    """swd.py:92-123: one distance per pyramid level (full resolution first) and their mean; None below 16 voxels."""
    resolutions = []
    res = images1.shape[-1]
    while res >= 16:
        resolutions.append(res)
        res //= 2
    if len(resolutions) == 0:
        print("No descriptors, probably resolution is too small. Returning None")
        return None
    dr = [get_descriptors_for_minibatch(lv, nhood_size, nhoods_per_image)
          for lv in generate_laplacian_pyramid(images1, len(resolutions))]
    df = [get_descriptors_for_minibatch(lv, nhood_size, nhoods_per_image)
          for lv in generate_laplacian_pyramid(images2, len(resolutions))]
    dr = [finalize_descriptors(d) for d in dr]
    df = [finalize_descriptors(d) for d in df]
    dist = [sliced_wasserstein(x, y, dir_repeats, dirs_per_repeat) for x, y in zip(dr, df)]
    return dist + [float(np.mean(dist))]
