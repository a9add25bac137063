"""Mirror of SURFGAN_2D/networks/ops.py for the pgan path: conv2d (:99-102), to_rgb / from_rgb (:161-169),
upscale2d / downscale2d (:176-231), num_filters (:139-158), k (:25-29 of the 3-D file, same rule).  A 2-D image batch
[N,C,H,W] is a D == 1 volume: every op below runs the 3-D kernels on [N,C,1,H,W] (NDHWC with D = 1 is NHWC), with
(1,kh,kw) taps and up/down-sampling factors (1,2,2) (sg_upscale_nn / sg_downscale_sum).  Weights keep the 2-D tree's
HWIO shape [kh,kw,Cin,Cout]."""
import numpy as np
import torch

from .. import functional as F
from ..networks.ops import (ScaledWeight, _LazyConv, _consume, _val, act, apply_bias, calculate_gain, dense,  # noqa: F401
                            get_variable, get_weight, k, leaky_relu, lerp, materialize, minibatch_stddev_layer,
                            pixel_norm, variable_scope)
from ..varstore import compute_dtype


def as_volume(x):
    """[N,C,H,W] -> [N,C,1,H,W] (a view; channels-last storage is unchanged)."""
    return x.unsqueeze(2) if torch.is_tensor(x) and x.dim() == 4 else x


def as_image(x):
    x = materialize(x)
    return x.squeeze(2) if x.dim() == 5 else x


def conv2d(x, fmaps, kernel, activation, param=None, lrmul=1):
    """SURFGAN_2D/networks/ops.py:99-102: stride-1 SAME cross-correlation with the equalised-LR weight [kh,kw,Cin,fmaps]."""
    xin, in_info = _consume(as_volume(x), premask=True)
    xin = as_volume(xin)
    kh, kw = kernel
    w = get_weight([kh, kw, xin.shape[1], fmaps], activation, param=param, lrmul=lrmul)
    if xin.dtype != compute_dtype():
        xin, in_info = xin.to(compute_dtype()), None
    lz = _LazyConv(xin, w.var.unsqueeze(0), w.coef, False)
    lz.in_info = in_info
    return lz


def num_filters(phase, num_phases, base_dim=None, size=None):
    """SURFGAN_2D/networks/ops.py:139-158: the last `num_phases` entries of a 13-entry list."""
    lists = {
        'xxs': [64] * 8 + [32, 16, 8, 4, 2], 'xs': [128] * 8 + [64, 32, 16, 8, 4], 's': [256] * 8 + [128, 64, 32, 16, 8],
        'm': [512] * 8 + [256, 128, 64, 32, 16], 'l': [512] * 9 + [256, 128, 64, 32],
        'xl': [1024] * 9 + [512, 256, 128, 64], 'xxl': [2048] * 9 + [1024, 512, 256, 128]}
    if size not in lists:
        raise ValueError(f"Unknown size: {size}")
    return lists[size][-num_phases:][phase - 1]


def to_rgb(x, channels=3):
    """SURFGAN_2D/networks/ops.py:161-162."""
    return apply_bias(conv2d(x, channels, (1, 1), activation='linear'))


def from_rgb(x, filters_out, activation, param=None):
    """SURFGAN_2D/networks/ops.py:165-169."""
    x = conv2d(x, filters_out, (1, 1), activation, param)
    x = apply_bias(x)
    return act(x, activation, param=param)


def upscale2d(x, factor=2):
    """SURFGAN_2D/networks/ops.py:204-216 (nearest neighbour; gradient = 4 * avg_pool2d)."""
    if factor == 1:
        return x
    if factor != 2:
        raise NotImplementedError('only factor 2 is used by the pgan path')
    return F.upscale2x(as_volume(_val(as_volume(x))), 1.0, factors=(1, 2, 2))


def downscale2d(x, factor=2):
    """SURFGAN_2D/networks/ops.py:219-231 (2x2 mean; gradient = avg_unpool2d / 4)."""
    if factor == 1:
        return x
    if factor != 2:
        raise NotImplementedError('only factor 2 is used by the pgan path')
    xv, in_info = _consume(as_volume(x), premask=True)
    return F.downscale2x(as_volume(xv), 0.25, in_info, factors=(1, 2, 2))
