"""Mirror of SURFGAN_3D/networks/pgandeep/discriminator.py (pgandeep/discriminator.py:25-131): N convolutions per
block.  The reference's index arithmetic is kept as it stands: every convolution of block i takes its kernel from
kernel_spec[i-1][1]; convolution j < N takes filter_spec[i-1][N-j-1] filters and the last one filter_spec[i-2][N-1];
from_rgb uses filter_spec[.][1]; discriminator_out runs N0 - 1 convolutions with kernel_spec[0][N0-j] and
filter_spec[0][N0-j-1] before the two dense layers."""
from ..ops import *  # noqa: F401,F403
from ..ops import act, apply_bias, conv3d, dense, downscale3d, from_rgb, lerp, materialize, variable_scope
from ..specs import filters, kernels


def _conv_stage(x, nfilters, kernel, activation, param):
    x = conv3d(x, nfilters, kernel, activation, param=param)
    x = apply_bias(x)
    return act(x, activation, param=param)


def discriminator_block(x, activation, kernel_spec, filter_spec, i, param=None):
    """pgandeep/discriminator.py:25-59."""
    num_layers = len(kernel_spec[i - 1])
    for layer_i in range(1, num_layers + 1):
        with variable_scope(f'conv_{layer_i}'):
            kernel = kernels(kernel_spec, i - 1, 1)
            if layer_i == num_layers:      # the last layer hands over to the previous phase's width
                nf = filters(filter_spec, i - 2, num_layers - 1)
            else:
                nf = filters(filter_spec, i - 1, num_layers - layer_i - 1)
            x = _conv_stage(x, nf, kernel, activation, param)
    return downscale3d(x)


def discriminator_out(x, latent_dim, activation, kernel_spec, filter_spec, param):
    """pgandeep/discriminator.py:62-94."""
    with variable_scope('discriminator_out'):
        num_layers = len(kernel_spec[0])
        for layer_i in range(1, num_layers):
            with variable_scope(f'conv_{layer_i}'):
                x = _conv_stage(x, filters(filter_spec, 0, num_layers - layer_i - 1),
                                kernels(kernel_spec, 0, num_layers - layer_i), activation, param)
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
    """pgandeep/discriminator.py:97-131."""
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
                    fromrgb_prev = from_rgb(downscale3d(x_downscale), filters(filter_spec, phase - 2, 1), activation,
                                            param=param)
                x = lerp(fromrgb_prev, x, alpha)       # alpha * fromrgb_prev + (1 - alpha) * x
        x = discriminator_out(x, latent_dim, activation, kernel_spec, filter_spec, param)
        return materialize(x)
