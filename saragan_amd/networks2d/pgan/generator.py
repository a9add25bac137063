"""Mirror of SURFGAN_2D/networks/pgan/generator.py (legacy signature: num_phases, base_dim, base_shape=[C,H0,W0], size):
generator_in :5-21, generator_block :24-43, generator :46-73.  Kernel per dimension = k(extent) (1 below 3, else 3)."""
import numpy as np

from ..ops import *  # noqa: F401,F403
from ..ops import (act, apply_bias, as_image, as_volume, conv2d, dense, k, lerp, materialize, num_filters, pixel_norm,
                   to_rgb, upscale2d, variable_scope)


def _stage(x, filters, activation, param):
    kernel = [k(s) for s in x.shape[-2:]]
    x = conv2d(x, filters, kernel, activation, param=param)
    x = apply_bias(x)
    x = act(x, activation, param=param)
    return pixel_norm(x)


def generator_in(x, filters, shape, activation, param=None):
    with variable_scope('dense'):
        x = dense(x, int(np.prod(shape)) * filters, activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
    x = materialize(x).reshape([-1, filters] + list(shape))
    with variable_scope('conv'):
        x = _stage(x, filters, activation, param)
    return x


def generator_block(x, filters_out, activation, param=None):
    with variable_scope('upsample'):
        x = upscale2d(x)
    with variable_scope('conv_1'):
        x = _stage(x, filters_out, activation, param)
    with variable_scope('conv_2'):
        x = _stage(x, filters_out, activation, param)
    return x


def generator(x, alpha, phase, num_phases, base_dim, base_shape, activation, param=None, size='medium', is_reuse=False):
    with variable_scope('generator', reuse=is_reuse):
        with variable_scope('generator_in'):
            x = generator_in(x, filters=base_dim, shape=base_shape[1:], activation=activation, param=param)
        x_upsample = None
        for i in range(2, phase + 1):
            if i == phase:
                with variable_scope(f'to_rgb_{phase - 1}'):
                    x_upsample = upscale2d(to_rgb(x, channels=base_shape[0]))
            filters_out = num_filters(i, num_phases, base_dim, size=size)
            with variable_scope(f'generator_block_{i}'):
                x = generator_block(x, filters_out, activation=activation, param=param)
        with variable_scope(f'to_rgb_{phase}'):
            x_out = to_rgb(x, channels=base_shape[0])
        if x_upsample is not None:
            x_out = lerp(as_volume(x_upsample), x_out, alpha)     # alpha * x_upsample + (1 - alpha) * x_out
        return as_image(x_out)
