"""SURFGAN_3D/metrics/swd.py:13-123 with the volumes resident on the GPU, on the library's own kernels
(csrc/metrics.hip through the C ABI): the Laplacian pyramid as separable 5-tap passes with the decimation / zero insertion
and the level subtraction fused (`sg_filter_axis`), neighbourhood descriptors by one gather (`sg_swd_gather`), channel
normalisation (`sg_desc_normalize`), projections on the random directions (`sg_swd_project`), a bitonic sort per direction
(`sg_sort_rows`) and the mean absolute difference (`sg_swd_distance`).  torch only holds the device buffers.  Random
numbers (neighbourhood positions, projection directions) are drawn on the host from numpy's global generator in the
reference's call order and sizes, so a seeded run reproduces the reference's result up to float32 summation order.
Function names and defaults are the reference's."""
import ctypes as C

import numpy as np
import torch

from .. import _lib

_TAPS = (1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16)      # [1,4,6,4,1]/16 per axis = the 5x5x5 filter / 4096
_TAPS_C = (C.c_double * 5)(*_TAPS)
_DOWN, _UP, _MIRROR = 1, 2, 0


def _dev(x, device=None):
    if torch.is_tensor(x):
        if not x.is_cuda:
            raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
        return x.to(torch.float32).contiguous()
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError('saragan_amd.metrics run on the GPU only (no CPU fallback)')
        device = 'cuda'
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=device).contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _filter_axis(x, dim, mode, alpha=1.0, add=None):
    """One axis of the separable binomial filter (mirror borders): decimating (`_DOWN`) or zero-inserting (`_UP`)."""
    shape = list(x.shape)
    n = shape[dim]
    outer = int(np.prod(shape[:dim], dtype=np.int64))
    inner = int(np.prod(shape[dim + 1:], dtype=np.int64))
    shape[dim] = (n + 1) // 2 if mode == _DOWN else 2 * n
    y = torch.empty(shape, dtype=x.dtype, device=x.device)
    if add is not None:
        assert add.shape == y.shape and add.is_contiguous() and add.dtype == x.dtype
    _lib.check(_lib.load().sg_filter_axis(_p(x), _p(y), _p(add), outer, n, inner, _TAPS_C, 5, mode, _MIRROR, alpha, 0, _st()),
               'sg_filter_axis')
    return y


def pyr_down(minibatch):
    """swd.py:63-66 (matches cv2.pyrDown per axis): the filter runs only where [::2] keeps a sample."""
    x = _dev(minibatch)
    assert x.dim() == 5
    for d in (4, 3, 2):
        x = _filter_axis(x, d, _DOWN)
    return x


def pyr_up(minibatch, _level=None):
    """swd.py:69-74: zero-insertion x2, filter * 8 (2 per axis).  `_level`: returns `_level - pyr_up(minibatch)` with the
    subtraction fused into the last pass (generate_laplacian_pyramid)."""
    x = _dev(minibatch)
    assert x.dim() == 5
    x = _filter_axis(x, 2, _UP, 2.0)
    x = _filter_axis(x, 3, _UP, 2.0)
    if _level is None:
        return _filter_axis(x, 4, _UP, 2.0)
    return _filter_axis(x, 4, _UP, -2.0, add=_level)


def generate_laplacian_pyramid(minibatch, num_levels):
    """swd.py:77-82."""
    pyramid = [_dev(minibatch).clone()]
    for _ in range(1, num_levels):
        pyramid.append(pyr_down(pyramid[-1]))
        pyramid[-2] = pyr_up(pyramid[-1], _level=pyramid[-2])
    return pyramid


def reconstruct_laplacian_pyramid(pyramid):
    """swd.py:85-89."""
    minibatch = _dev(pyramid[-1])
    for level in pyramid[-2::-1]:
        x = _filter_axis(minibatch, 2, _UP, 2.0)
        x = _filter_axis(x, 3, _UP, 2.0)
        minibatch = _filter_axis(x, 4, _UP, 2.0, add=_dev(level))
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
    t = lambda a: torch.as_tensor(a.reshape(-1).astype(np.int32), device=x.device)
    d0, x0, y0 = t(d0), t(x0), t(y0)
    out = torch.empty((N, S[1], 2 * D + 1, 2 * H + 1, 2 * W + 1), dtype=torch.float32, device=x.device)
    # the reference's `x` grid (2H+1 offsets, fourth axis) runs over the LAST volume axis, its `y` grid over the fourth
    _lib.check(_lib.load().sg_swd_gather(_p(x), _p(out), _p(d0), _p(y0), _p(x0), S[0], S[1], S[2], S[3], S[4],
                                         nhoods_per_image, D, H, W, _st()), 'sg_swd_gather')
    return out


def finalize_descriptors(desc):
    """swd.py:31-39."""
    if isinstance(desc, list):
        desc = torch.cat(desc, dim=0)
    assert desc.dim() == 5
    desc = desc.contiguous()
    if desc.shape[1] > 1:
        lib = _lib.load()
        c = desc.shape[1]
        nbytes = lib.sg_desc_normalize_workspace(c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=desc.device)
        inner = desc.shape[2] * desc.shape[3] * desc.shape[4]
        _lib.check(lib.sg_desc_normalize(_p(desc), desc.shape[0], c, inner, _p(ws), nbytes, _st()), 'sg_desc_normalize')
    return desc.reshape(desc.shape[0], -1)


def sliced_wasserstein(a, b, dir_repeats, dirs_per_repeat):
    """swd.py:44-58."""
    a, b = _dev(a), _dev(b)
    assert a.dim() == 2 and a.shape == b.shape
    lib = _lib.load()
    n, f = a.shape
    npad = lib.sg_swd_padded_rows(n)
    proj = torch.empty((2, dirs_per_repeat, npad), dtype=torch.float32, device=a.device)
    out = torch.empty(1 + dirs_per_repeat, dtype=torch.float64, device=a.device)
    results = []
    for _ in range(dir_repeats):
        dirs = np.random.randn(f, dirs_per_repeat)
        dirs /= np.sqrt(np.sum(np.square(dirs), axis=0, keepdims=True))
        dirs = torch.as_tensor(dirs.astype(np.float32), device=a.device)
        _lib.check(lib.sg_swd_project(_p(a), _p(dirs), _p(proj[0]), n, f, dirs_per_repeat, npad, _st()), 'sg_swd_project')
        _lib.check(lib.sg_swd_project(_p(b), _p(dirs), _p(proj[1]), n, f, dirs_per_repeat, npad, _st()), 'sg_swd_project')
        _lib.check(lib.sg_sort_rows(_p(proj), 2 * dirs_per_repeat, npad, _st()), 'sg_sort_rows')
        _lib.check(lib.sg_swd_distance(_p(proj[0]), _p(proj[1]), _p(out), dirs_per_repeat, n, npad, _st()), 'sg_swd_distance')
        results.append(out[0].clone())
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
