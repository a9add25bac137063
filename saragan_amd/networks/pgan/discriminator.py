"""Mirror of SURFGAN_3D/networks/pgan/discriminator.py (same callables, argument order, variable scopes and
error behaviour); the ops are saragan_amd.networks.ops (gfx950 kernels)."""
from ..ops import *  # noqa: F401,F403
from ..specs import filters, kernels
from ..ops import act, apply_bias, conv3d, dense, downscale3d, from_rgb, lerp, materialize, variable_scope


def discriminator_block(x, activation, kernel_spec, filter_spec, i, param=None):
    """pgan/discriminator.py:25-45 (note the swapped kernel_spec layer indices of the reference)."""
    with variable_scope('conv_1'):
        kernel = kernels(kernel_spec, i - 1, 1)
        x = conv3d(x, filters(filter_spec, i - 1, 0), kernel, activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
    with variable_scope('conv_2'):
        kernel = kernels(kernel_spec, i - 1, 0)
        x = conv3d(x, filters(filter_spec, i - 2, 1), kernel, activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
    x = downscale3d(x)
    return x


def discriminator_out(x, latent_dim, activation, kernel_spec, filter_spec, param):
    """pgan/discriminator.py:48-68 (minibatch_stddev_layer stays disabled, :50)."""
    with variable_scope('discriminator_out'):
        kernel = kernels(kernel_spec, 0, 1)
        x = conv3d(x, filters(filter_spec, 0, 0), kernel, activation=activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
        with variable_scope('dense_1'):
            x = dense(x, latent_dim, activation=activation, param=param)
            x = apply_bias(x)
            x = act(x, activation, param=param)
        with variable_scope('dense_2'):
            x = dense(x, 1, activation='linear')
            x = apply_bias(x)
        return x


def discriminator(x, alpha, phase, latent_dim, activation, kernel_spec, filter_spec, param=None, is_reuse=False,
                  conditioning=None):
    """pgan/discriminator.py:71-108."""
    if conditioning is not None:
        raise NotImplementedError()
    with variable_scope('discriminator', reuse=is_reuse):
        x_downscale = x
        with variable_scope(f'from_rgb_{phase}'):
            x = from_rgb(x, filters(filter_spec, phase - 1, 1), activation, param=param)
        for i in reversed(range(2, phase + 1)):
            with variable_scope(f'discriminator_block_{i}'):
                x = discriminator_block(x, activation, kernel_spec, filter_spec, i=i, param=param)
            if i == phase:
                with variable_scope(f'from_rgb_{phase - 1}'):
                    fromrgb_prev = from_rgb(downscale3d(x_downscale),
                                            filters(filter_spec, phase - 2, 1), activation,
                                            param=param)
                x = lerp(fromrgb_prev, x, alpha)       # alpha * fromrgb_prev + (1 - alpha) * x
        x = discriminator_out(x, latent_dim, activation, kernel_spec, filter_spec, param)
        return materialize(x)
