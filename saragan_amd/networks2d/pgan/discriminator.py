"""Mirror of SURFGAN_2D/networks/pgan/discriminator.py (legacy signature): discriminator_block :4-18,
discriminator_out :21-38, discriminator :41-70."""
from ..ops import *  # noqa: F401,F403
from ..ops import (act, apply_bias, as_volume, conv2d, dense, downscale2d, from_rgb, k, lerp, materialize, num_filters,
                   variable_scope)


def _stage(x, filters, activation, param):
    kernel = [k(s) for s in x.shape[-2:]]
    x = conv2d(x, filters, kernel, activation, param=param)
    x = apply_bias(x)
    return act(x, activation, param=param)


def discriminator_block(x, filters_in, filters_out, activation, param=None):
    with variable_scope('conv_1'):
        x = _stage(x, filters_in, activation, param)
    with variable_scope('conv_2'):
        x = _stage(x, filters_out, activation, param)
    return downscale2d(x)


def discriminator_out(x, base_dim, latent_dim, filters_out, activation, param):
    with variable_scope('discriminator_out'):
        x = _stage(x, filters_out, activation, param)
        with variable_scope('dense_1'):
            x = dense(x, latent_dim, activation=activation, param=param)
            x = apply_bias(x)
            x = act(x, activation, param=param)
        with variable_scope('dense_2'):
            x = dense(x, 1, activation='linear')
            x = apply_bias(x)
        return x


def discriminator(x, alpha, phase, num_phases, base_dim, latent_dim, activation, param=None, is_reuse=False,
                  size='medium'):
    with variable_scope('discriminator', reuse=is_reuse):
        x = as_volume(x)
        x_downscale = x
        with variable_scope(f'from_rgb_{phase}'):
            filters_out = num_filters(phase, num_phases, base_dim, size=size)
            x = from_rgb(x, filters_out, activation, param=param)
        for i in reversed(range(2, phase + 1)):
            with variable_scope(f'discriminator_block_{i}'):
                filters_in = num_filters(i, num_phases, base_dim, size=size)
                filters_out = num_filters(i - 1, num_phases, base_dim, size=size)
                x = discriminator_block(x, filters_in, filters_out, activation, param=param)
            if i == phase:
                with variable_scope(f'from_rgb_{phase - 1}'):
                    fromrgb_prev = from_rgb(downscale2d(x_downscale), filters_out, activation, param=param)
                x = lerp(fromrgb_prev, x, alpha)       # alpha * fromrgb_prev + (1 - alpha) * x
        x = discriminator_out(x, base_dim, latent_dim, filters_out, activation, param)
        return materialize(x)
